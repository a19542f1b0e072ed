/*
 * rvseg.h -- C ABI of librvseg.so: the MI355X (gfx950) per-pixel inference path
 *
 *     RGB-D frame -> feature_extractor -> libforest random-forest evaluation
 *                 -> DenseCRF mean-field (permutohedral lattice) -> softmax / argmax
 *
 * This is the drop-in boundary for the hot path of VisualComputingInstitute/
 * RovinaSemanticSegmentation.  The reference has no FFI: the path is reached by ordinary C++
 * calls from `class Segmenter` (include/segmenter.h:47-69).  Every entry point below names the
 * reference interface it replaces (paths relative to the reference tree).  Plain pointers and
 * sizes only; no C++ or torch types cross this boundary; nothing throws across it.
 *
 * Memory conventions
 *   - "host" entry points take host pointers and copy through HBM themselves.
 *   - "_device" entry points take device (HBM) pointers plus a hipStream_t passed as void*;
 *     they enqueue work on that stream and return without synchronising.
 *   - A ctx owns its device buffers, forest copy and LUTs; the caller owns every in/out buffer.
 *   - One ctx per (thread, device).  Calls on one ctx must be serialised by the caller, exactly
 *     as the reference serialises its libraries behind _frame_mtx (src/segmenter.cpp:336-435).
 *
 * Layouts (identical to the reference's)
 *   rgb        n x H x W x 3 uint8, channel order as delivered to FeatureExtractor::extract
 *              (RGB; the R/B swap inside CV_BGR2Lab is part of the trained feature definition,
 *              include/feature_extractor.h:129, src/test.cpp:130)
 *   depth_mm   n x H x W uint16, millimetres (include/feature_extractor.h:59-60)
 *   calib      21 floats: K^-1 (3x3 row-major), R (3x3 row-major), t (3): Calibration::
 *              _intrinsic_inverse, _extrinsic.linear(), _extrinsic.translation()
 *              (include/calibration.h:19-21; used at include/feature_extractor.h:223)
 *   posteriors per frame: layers concatenated, each [y][x][class] float32, offset of layer l =
 *              sum_{l'<l} H*W*C_l'  (src/segmenter.cpp:413-431; srv/SingleFrameSegmentation.srv
 *              label_distribution)
 *   labels     per frame: layers concatenated, each H x W int8
 *   CRF        unary / Q: N x C float32, class-contiguous per point (= column-major C x N
 *              Eigen::MatrixXf, third-party/densecrf/include/densecrf.h:56);
 *              features: N x d float32 (= column-major d x N)
 */
#ifndef RVSEG_H
#define RVSEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RVSEG_MAX_LAYERS 8

typedef struct rvseg_ctx rvseg_ctx;

typedef enum rvseg_status {
    RVSEG_OK = 0,
    RVSEG_ERR_INVALID_ARG = 1, /* bad pointer / size / parameter                                   */
    RVSEG_ERR_IO = 2,          /* libf::Exception("Could not open file.") (libforest io.h:118-121) */
    RVSEG_ERR_FORMAT = 3,      /* malformed or inconsistent forest.dat                             */
    RVSEG_ERR_NO_FOREST = 4,   /* empty forest: the reference asserts (classifier.cpp:168,189),
                                  compiled out in Release -> UB; here a clean error               */
    RVSEG_ERR_HIP = 5,         /* HIP runtime failure (message in rvseg_last_error)                */
    RVSEG_ERR_NO_DEVICE = 6,   /* no gfx950 device: the product never falls back to the CPU        */
    RVSEG_ERR_CAPACITY = 7,    /* lattice hash table / batch / ensemble capacity exceeded          */
    RVSEG_NOT_READY = 8        /* rvseg_poll_status without waiting: the work is still running      */
} rvseg_status;

/* Label rules found in the reference. */
typedef enum rvseg_label_mode {
    RVSEG_LABEL_EVAL = 0,     /* src/test.cpp:160-175: strict '>' from -1000, -1 when nothing wins  */
    RVSEG_LABEL_CRF = 1,      /* src/segmenter.cpp:646-657: strict '>' from 2.0/C else "Unknown"    */
    RVSEG_LABEL_NOCRF = 2,    /* src/segmenter.cpp:664-679: strict '>' from -1000, sum!=0 guard     */
    RVSEG_LABEL_ARGMAX = 3    /* DenseCRF::currentMap, densecrf.cpp:202-211: first maximum          */
} rvseg_label_mode;

/* Parameter block = the hot-path keys of resources/config.json. */
typedef struct rvseg_params {
    int32_t width, height;          /* camera size (Segmenter::_camera_w/_camera_h)                */
    int32_t stride;                 /* rf_prediction_stride, config.json:87                         */
    float depth_min, depth_max;     /* config.json:89-90 (metres)                                   */
    int32_t patch_size;             /* config.json:32                                               */
    int32_t patch_size_reduce;      /* config.json:34                                               */
    int32_t feature_color_patch, feature_depth, feature_height, feature_normal; /* config.json:41-44 */
    float fill_value;               /* low-res image init: 0 (segmenter.cpp:358-362) or -1000
                                       (test.cpp:143-147)                                           */
    int32_t use_dense_crf;          /* config.json:81 (per-frame CRF as composed by the north star) */
    float dcrf_xyz_kernel, dcrf_rgb_kernel, dcrf_kernel_weight; /* config.json:82-84 (multipliers)  */
    int32_t dcrf_iterations;        /* config.json:85                                               */
    int32_t multi_layer;            /* 1: multiClassLogPosterior (shared forest, segmenter.cpp:368),
                                       0: classLogPosterior (test.cpp:151)                          */
    int32_t label_mode;             /* rvseg_label_mode                                             */
    int32_t unknown_label[RVSEG_MAX_LAYERS]; /* Segmenter::_layer_unknown_label, segmenter.cpp:88-96 */
    int32_t max_batch;              /* frames processed per launch group (device buffers are sized
                                       for this many frames)                                        */
    int32_t device;                 /* HIP device ordinal                                           */
    int32_t lattice_capacity_log2;  /* hash-table slots per frame = 2^this; 0 = 2^12 (the Segmenter
                                       kernel gives ~300 vertices / frame on the synthetic scenes, up to
                                       ~2 200 on a real photo with a 1-10 m depth range), -1 = worst case
                                       2*N*(d+1).  Overflow is detected, never silent: the context then
                                       raises its capacity (x8 per step, up to the worst case) for all
                                       later work; host entry points redo the chunk themselves, the
                                       asynchronous _device entry point reports RVSEG_ERR_CAPACITY from
                                       rvseg_poll_status (and from the next call)                     */
} rvseg_params;

/* Fills *p with the defaults of resources/config.json. */
void rvseg_params_default(rvseg_params *p);

/* Replaces: Segmenter::Segmenter's model/feature set-up (src/segmenter.cpp:106-129) minus ROS.
 * Fails with RVSEG_ERR_NO_DEVICE when no GPU is present. */
rvseg_status rvseg_create(const rvseg_params *params, rvseg_ctx **out);
void rvseg_destroy(rvseg_ctx *ctx);
/* Message of the last failing call on ctx (or of the last failing rvseg_create if ctx == NULL). */
const char *rvseg_last_error(const rvseg_ctx *ctx);
const char *rvseg_status_string(rvseg_status s);
/* D of the feature vector, include/feature_extractor.h:46-51 (366 with the default config). */
int32_t rvseg_feature_length(const rvseg_ctx *ctx);

/* ---- forest: replaces libf::RandomForest::read (libforest classifier.cpp:222-235) ------------ */
rvseg_status rvseg_forest_load(rvseg_ctx *ctx, const char *path);
rvseg_status rvseg_forest_load_mem(rvseg_ctx *ctx, const void *buf, size_t size);
/* Host-only validation of a forest.dat image with the loader's own parser and limits (no context, no
 * GPU): format, child links, split features < feature_length (<= 0: not checked), at most 64 trees
 * (RVSEG_ERR_CAPACITY: libforest has no limit, the device evaluator does) and 64 classes.  err_out
 * (optional) receives the message.  The reference has no such check: a bad model "will result in
 * segfaults" (README.md:30). */
rvseg_status rvseg_forest_check(const void *buf, size_t size, int32_t feature_length, int32_t *n_trees,
                                int32_t *n_nodes_total, int32_t *max_depth, char *err_out, size_t err_cap);
/* Replaces libf::RandomForest::write (classifier.cpp:210-220; DecisionTree::write :144-152): the loaded
 * model in the reference's stream format, node for node -- a file read with rvseg_forest_load is
 * written back byte for byte.  _mem: *size_out receives the needed size; out may be NULL to query. */
rvseg_status rvseg_forest_write(const rvseg_ctx *ctx, const char *path);
rvseg_status rvseg_forest_write_mem(const rvseg_ctx *ctx, void *out, size_t out_cap, size_t *size_out);
/* The same writer without a context: parse `buf`, serialise it again (host only; used by the CPU tests
 * to pin the writer against the reference-written golden files). */
rvseg_status rvseg_forest_rewrite(const void *buf, size_t size, void *out, size_t out_cap, size_t *size_out);
/* ---- forest training on the GPU: replaces RandomForestLearner::learn + DecisionTreeLearner::learn +
 *      updateMultiHistograms + forest->write (libforest learning.cpp:410-1073, src/train.cpp:225-249) for the
 *      model family this path evaluates.  Parameters = the learner settings of src/train.cpp / config.json. */
typedef struct rvseg_train_params {
    int32_t num_trees;                /* config.json:37 (4)                                                   */
    int32_t max_depth;                /* config.json:38 (30): a node deeper than this is not split (learning.cpp:525) */
    int32_t min_split_examples;       /* config.json:39 (50)                                                  */
    int32_t min_child_split_examples; /* learning.h:116 (1)                                                   */
    int32_t num_features;             /* features tried per node; 0 = ceil(sqrt(D)) (autoconf, learning.cpp:367) */
    int32_t use_bootstrap;            /* 1 (train.cpp:226): N draws with replacement per tree                 */
    float smoothing;                  /* learning.h:117 (1): log((h + s) / (total + C s))                     */
    uint64_t seed;                    /* the reference seeds from std::random_device (learning.cpp:18) and is
                                         not reproducible; here equal seeds give equal bytes                  */
} rvseg_train_params;
void rvseg_train_params_default(rvseg_train_params *tp);
/* X: P x D host floats (one DataPoint per row), labels: P x n_layers class indices (DataStorage's multi labels),
 * class_counts per layer (at most 16 each).  Writes a forest.dat image (multi_histograms = the "shared" forest of
 * train.cpp:231; with one layer also `histograms`), loadable with rvseg_forest_load_mem and by the reference.
 * Search = the reference's: minimum of E(left) + E(right) (EfficientEntropyHistogram, fastlog2) over the thresholds at
 * midpoints of adjacent values at least 1e-6 apart (learning.cpp:578-592) -- exact for every feature: byte-valued
 * features through per-value histograms, the others through a sort per level; leaf histograms exactly as
 * updateMultiHistograms computes them (learning.cpp:960-1012); nodes numbered as the reference's depth-first stack
 * would.  Equal seeds give equal bytes, and the bytes equal the CPU oracle's depth-first learner (tests).
 * The trained model is also kept on the context: when out_cap is too small (RVSEG_ERR_INVALID_ARG, *size_out = needed
 * size) or forest_out is NULL, fetch it with rvseg_forest_train_result instead of training again. */
rvseg_status rvseg_forest_train(rvseg_ctx *ctx, const float *X, int32_t P, int32_t D, const int32_t *labels,
                                int32_t n_layers, const int32_t *class_counts, const rvseg_train_params *tp,
                                void *forest_out, size_t out_cap, size_t *size_out);
rvseg_status rvseg_forest_train_result(rvseg_ctx *ctx, void *forest_out, size_t out_cap, size_t *size_out);
/* Training straight from labelled key frames: replaces the extraction loop of src/train.cpp:115-147 as well.  Features
 * are extracted on the device (FeatureExtractor::extract, WITH_POSITIVE_LABEL branch: stride-grid points with valid
 * depth whose labels are all >= 0, include/feature_extractor.h:93-121) and packed there (Lab patch bytes as bytes,
 * depth / height / normal as floats); no P x D float matrix exists anywhere.
 *   rgb n x H x W x 3, depth_mm n x H x W, calib n x 21 (all host memory, the context's width / height / stride);
 *   labels n x L x H x W int8 (label_type = char, include/defines.h), < 0 = unlabelled
 *   augment != 0: the reference's augmentation -- every frame with colour offsets -20, 0, +20 (`color += a` on the
 *   8UC3 image, i.e. OpenCV's Scalar(a,0,0,0): channel 0 only, saturated) and each of those also flipped horizontally
 *   (colour, depth and labels, same calibration): six extractions per frame, in the reference's order
 *   n_examples_out (optional): training points extracted. */
rvseg_status rvseg_forest_train_frames(rvseg_ctx *ctx, int32_t n_frames, const uint8_t *rgb, const uint16_t *depth_mm,
                                       const float *calib, const int8_t *labels, int32_t n_layers,
                                       const int32_t *class_counts, int32_t augment, const rvseg_train_params *tp,
                                       void *forest_out, size_t out_cap, size_t *size_out, int32_t *n_examples_out);
/* n_layers / class_counts describe the active mode (multi_layer or single). */
rvseg_status rvseg_forest_info(const rvseg_ctx *ctx, int32_t *n_trees, int32_t *n_nodes_total,
                               int32_t *max_depth, int32_t *n_layers,
                               int32_t class_counts[RVSEG_MAX_LAYERS]);
/* Replaces: RandomForest::classLogPosterior / multiClassLogPosterior per DataPoint
 * (libforest classifier.cpp:166-208).  X: P x D host floats; out: P x sumC host floats. */
rvseg_status rvseg_forest_eval(rvseg_ctx *ctx, const float *X, int32_t P, int32_t D, float *out);

/* ---- features: replaces Features::FeatureExtractor::extract, NO_LABEL branch
 *      (include/feature_extractor.h:41-291).  Host buffers, one frame.  feat_out: capacity
 *      (H/stride+1)*(W/stride+1) x D floats; x_v / y_v same capacity.  Exists for parity tests:
 *      the production path never materialises features. */
rvseg_status rvseg_extract_features(rvseg_ctx *ctx, const uint8_t *rgb, const uint16_t *depth_mm,
                                    const float *calib, float *feat_out, int32_t *x_v, int32_t *y_v,
                                    int32_t *n_points);

/* ---- whole per-frame path: replaces the body of Segmenter::processFramesFromQueueInternalRF
 *      (src/segmenter.cpp:351-431) and, with use_dense_crf, the DenseCRF call shape of
 *      Segmenter::processMapFromQueue (src/segmenter.cpp:639-657) applied per frame.
 *      calib: n x 21 floats (HOST memory in both variants).  Any output pointer may be NULL.
 *        posteriors_out  n x sumC*H*W   RF log-posteriors (the node's `posteriors` vector)
 *        marginals_out   n x sumC*H*W   CRF marginals (only with use_dense_crf)
 *        labels_out      n x L*H*W      labels of the CRF marginals (or of the posteriors
 *                                       without CRF) under params.label_mode */
rvseg_status rvseg_segment_frames(rvseg_ctx *ctx, int32_t n_frames, const uint8_t *rgb,
                                  const uint16_t *depth_mm, const float *calib, float *posteriors_out,
                                  float *marginals_out, int8_t *labels_out);
/* Page-locks a caller buffer (hipHostRegister) / releases it.  rvseg_segment_frames recognises page-locked input and
 * output buffers (registered here, or allocated with hipHostMalloc) and lets the copy engines read / write them directly:
 * without it every output crosses the host memory twice (pinned staging, then the caller's pageable buffer), which
 * bounds a call that returns marginals -- 11 MB per frame, src/segmenter.cpp:413-434 -- at the host's memcpy rate.
 * Register once, reuse the buffers across calls (registration costs milliseconds). */
rvseg_status rvseg_host_register(void *p, size_t bytes);
rvseg_status rvseg_host_unregister(void *p);
rvseg_status rvseg_segment_frames_device(rvseg_ctx *ctx, int32_t n_frames, const uint8_t *d_rgb,
                                         const uint16_t *d_depth_mm, const float *calib,
                                         float *d_posteriors_out, float *d_marginals_out,
                                         int8_t *d_labels_out, void *hip_stream);

/* Status of the asynchronous work of the last rvseg_segment_frames_device call on this context (the
 * lattice build is the only stage that can fail on the device: hash-table overflow).  wait != 0 blocks
 * until that status is known (it does NOT wait for the outputs: synchronise the stream for those);
 * wait == 0 returns RVSEG_NOT_READY while the build is still running.  RVSEG_ERR_CAPACITY: the outputs
 * of that call are invalid (the kernels after an overflow exit without writing); the context has
 * already raised its capacity, so repeating the call succeeds.  The reference has no counterpart (its
 * hash table grows in place, permutohedral.cpp:59-79). */
rvseg_status rvseg_poll_status(rvseg_ctx *ctx, int32_t wait);

/* ---- CRF: replaces  DenseCRF crf(N,C); crf.setUnaryEnergy(U); crf.addPairwiseEnergy(feat,
 *      new PottsCompatibility(w)); Q = crf.inference(iters);  (src/segmenter.cpp:641-644;
 *      densecrf.cpp:54-60,85-91,115-131) with DIAG_KERNEL + NORMALIZE_SYMMETRIC defaults
 *      (densecrf.h:59).  n_kernels > 1 covers DenseCRF2D::addPairwiseGaussian/Bilateral
 *      (densecrf.cpp:61-81): kernel k has features[k] (N x ds[k]) and Potts weight ws[k].
 *      map_out (optional): labels under label_mode / unknown_label. */
rvseg_status rvseg_crf_infer(rvseg_ctx *ctx, int32_t N, int32_t C, int32_t d,
                             const float *unary_energy, const float *features, float potts_w,
                             int32_t iterations, float *Q_out, int8_t *map_out, int32_t label_mode,
                             int32_t unknown_label);
rvseg_status rvseg_crf_infer_multi(rvseg_ctx *ctx, int32_t N, int32_t C, int32_t n_kernels,
                                   const int32_t *ds, const float *const *features, const float *ws,
                                   const float *unary_energy, int32_t iterations, float *Q_out,
                                   int8_t *map_out, int32_t label_mode, int32_t unknown_label);

/* Host-side feature builders of DenseCRF2D (densecrf.cpp:61-81), for rvseg_crf_infer_multi:
 *   addPairwiseGaussian(sx, sy)                 f = (x / sx, y / sy)                            out: W*H x 2
 *   addPairwiseBilateral(sx, sy, sr, sg, sb, im) f = (x / sx, y / sy, r / sr, g / sg, b / sb)    out: W*H x 5
 * im: H x W x 3 uint8 in the channel order of the caller's image.  No context, no GPU. */
rvseg_status rvseg_crf_features_gaussian(int32_t W, int32_t H, float sx, float sy, float *out);
rvseg_status rvseg_crf_features_bilateral(int32_t W, int32_t H, float sx, float sy, float sr, float sg, float sb,
                                          const uint8_t *im, float *out);

/* ---- lattice introspection for parity tests: Permutohedral::init + compute
 *      (densecrf permutohedral.cpp:140-321,596-603).  offsets_out / bary_out: N x (d+1);
 *      keys_out: capacity M_cap x d int16; vertex numbering is arbitrary (results do not depend on
 *      it); M_out receives the number of lattice vertices. */
rvseg_status rvseg_lattice_build(rvseg_ctx *ctx, const float *features, int32_t N, int32_t d,
                                 int32_t *offsets_out, float *bary_out, int16_t *keys_out,
                                 int32_t keys_capacity, int32_t *M_out);
/* filter one value matrix (N x C) through the lattice last built on this ctx */
rvseg_status rvseg_lattice_filter(rvseg_ctx *ctx, const float *in, int32_t C, float *out);
/* blur neighbours of the lattice last built (permutohedral.cpp:296-318): n1_out / n2_out are
 * (d+1) x M vertex ids (-1 = absent), in this ctx's vertex numbering.  Optional: the vertex-major
 * entry order used by the ordered splat: csr_point (N*(d+1)), vstart / vend (M). */
rvseg_status rvseg_lattice_neighbours(rvseg_ctx *ctx, int32_t *n1_out, int32_t *n2_out, uint32_t *csr_point,
                                      uint32_t *vstart, uint32_t *vend);

/* Local-map fusion -- replaces the accumulation loop of Segmenter::processMapFromQueue,
 * src/segmenter.cpp:561-616:  unaries[l](c, index) += label_distribution[off_l + pixel*C_l + c]
 * for every image in call order, pixels in raster order (the order fixes the fp32 sums).
 *   index_images  n_images x H x W int32, the projector's IndexImage rows of one camera
 *                 (index_image.ptr<int>(y + i*_camera_h), :601): cloud point seen at the pixel, < 0 = none
 *   posteriors    n_images x (sum C_l * H * W) floats, each image in the layout rvseg_segment_frames
 *                 writes ([layer][y][x][class], segmenter.cpp:413-431)
 *   unaries_out   layers concatenated; layer l is cloud_size x C_l, point-major (== the C_l x cloud_size
 *                 column-major Eigen matrix of :563-567), starting at cloud_size * (C_0 + .. + C_{l-1})
 * H, W are the context's.  An index >= cloud_size is RVSEG_ERR_INVALID_ARG (the reference writes out of
 * bounds).  Feed -unaries_out[l] to rvseg_crf_infer (:642) or label it with the no-CRF rule (:660-681). */
rvseg_status rvseg_fuse_posteriors(rvseg_ctx *ctx, int32_t n_images, const int32_t *index_images,
                                   const float *posteriors, int32_t n_layers, const int32_t *class_counts,
                                   int32_t cloud_size, float *unaries_out);

/* ---- the same consumers with every buffer resident in HBM (device pointers, work enqueued on hip_stream,
 *      no synchronisation, no allocation per call once the context's buffers have grown): the local-map
 *      thread can take d_posteriors_out of rvseg_segment_frames_device as it is, instead of moving
 *      11-21 MB per frame over PCIe twice.  An index >= cloud_size is skipped like "no point" and reported
 *      by rvseg_poll_status (RVSEG_ERR_INVALID_ARG). */
rvseg_status rvseg_fuse_posteriors_device(rvseg_ctx *ctx, int32_t n_images, const int32_t *d_index_images,
                                          const float *d_posteriors, int32_t n_layers, const int32_t *class_counts,
                                          int32_t cloud_size, float *d_unaries_out, void *hip_stream);
/* pairwise = (x,y,z) * dcrf_xyz_kernel ++ (r,g,b) * dcrf_rgb_kernel per point, rgb in [0,1]
 * (src/segmenter.cpp:629-637).  d_features_out: N x 6. */
rvseg_status rvseg_cloud_features_device(rvseg_ctx *ctx, int32_t N, const float *d_xyz, const float *d_rgb,
                                         float *d_features_out, void *hip_stream);
/* DenseCRF call shape of src/segmenter.cpp:641-644 on device buffers, one Potts kernel.  unary_is_energy = 0:
 * d_unary holds the accumulated log-posteriors and the energy is their negative (crf.setUnaryEnergy(-unaries[l]),
 * :642) -- no negated copy is made.  d_Q_out or d_map_out may be NULL (not both).  The lattice build reads
 * its counters back once (one synchronisation of hip_stream) so that a hash overflow is retried here. */
rvseg_status rvseg_crf_infer_device(rvseg_ctx *ctx, int32_t N, int32_t C, int32_t d, const float *d_unary,
                                    int32_t unary_is_energy, const float *d_features, float potts_w,
                                    int32_t iterations, float *d_Q_out, int8_t *d_map_out, int32_t label_mode,
                                    int32_t unknown_label, void *hip_stream);
rvseg_status rvseg_label_values_device(rvseg_ctx *ctx, const float *d_values, int32_t N, int32_t C,
                                       int32_t label_mode, int32_t unknown_label, int8_t *d_labels_out,
                                       void *hip_stream);
/* Replaces the body of Segmenter::processMapFromQueue for one local map (src/segmenter.cpp:561-682) with
 * everything in HBM: fusion through the index images (:561-616), then per label layer of the loaded model
 * either the cloud DenseCRF + "> 2.0/C else Unknown" (params.use_dense_crf, :628-658; one lattice serves all
 * layers -- the reference builds the same one per layer) or the no-CRF rule (:660-681).
 *   d_labels_out   L x cloud_size int8 (result_labels[layer][i], :646-657)
 *   d_unaries_out  optional: the fused unaries, layers concatenated (else they stay in context memory) */
rvseg_status rvseg_process_map_device(rvseg_ctx *ctx, int32_t n_images, const int32_t *d_index_images,
                                      const float *d_posteriors, int32_t cloud_size, const float *d_cloud_xyz,
                                      const float *d_cloud_rgb, int8_t *d_labels_out, float *d_unaries_out,
                                      void *hip_stream);

/* One of the label rules above over a host matrix of N points x C classes (class-contiguous); the
 * no-CRF branch of processMapFromQueue applies RVSEG_LABEL_NOCRF to the fused unaries
 * (src/segmenter.cpp:660-681). */
rvseg_status rvseg_label_values(rvseg_ctx *ctx, const float *values, int32_t N, int32_t C, int32_t label_mode,
                                int32_t unknown_label, int8_t *labels_out);

/* ---- multi-GPU: the local-map gather over RCCL.  The reference is a single process; the build
 *      shards key frames over the GPUs of a node (one context per GPU, each in its own process or thread, no
 *      data-path collective) and sends every rank's fixed-size block -- int8 labels n x L x H x W, or fp32
 *      posteriors for the order-preserving fusion of src/segmenter.cpp:599-616 -- to the fusion rank: one
 *      direct transfer per peer over xGMI.  librccl.so is opened on first use.
 *        rvseg_comm_unique_id  rank 0 creates the 128-byte id; the host distributes it (any channel)
 *        rvseg_comm_init       collective over all ranks (ncclCommInitRank)
 *        rvseg_gather_frames   rank r's bytes_per_rank bytes land at d_recv + r * bytes_per_rank on `root`
 *                              (d_recv is ignored elsewhere); enqueued on hip_stream, not waited for */
#define RVSEG_COMM_ID_BYTES 128
rvseg_status rvseg_comm_unique_id(uint8_t id_out[RVSEG_COMM_ID_BYTES]);
rvseg_status rvseg_comm_init(rvseg_ctx *ctx, int32_t rank, int32_t world, const uint8_t id[RVSEG_COMM_ID_BYTES]);
void rvseg_comm_destroy(rvseg_ctx *ctx);
rvseg_status rvseg_gather_frames(rvseg_ctx *ctx, const void *d_local, size_t bytes_per_rank, void *d_recv,
                                 int32_t root, void *hip_stream);

/* ---- launch schedules.  The library picks the schedule of the ordered splat (the dominant kernel) and the stream
 *      overlaps from the shape of the work; this block overrides those choices for tests, profiling and tuning.
 *      Nothing on the call path reads the environment.  The reference has no counterpart (single thread). */
typedef struct rvseg_schedule {
    int32_t splat;               /* 0 = chosen from the chunk (default), 1 = list-major walk, 2 = resident bands
                                    (forced wherever its tables fit)                                              */
    int32_t resident_blocks;     /* resident bands: blocks per frame, 0 = CUs / frames (2..12)                    */
    int32_t resident_band;       /* wave-blocks of 256 points per band (default 16 = 4096 points)                 */
    int32_t resident_chunk;      /* entries per slot and tile: 64 or 128 (default 128)                            */
    int32_t resident_window;     /* pacing window in bands, -1 = no pacing (default)                              */
    int32_t resident_cap_tiles;  /* tile table per frame, 0 = N / 8 + 1024 (tests shrink it: planner fall-back)   */
    int32_t group_vertices;      /* list-major walk, C = 8 / 9: vertices per block, 0 = by the chunk, 6 or 7      */
    int32_t overlap_build;       /* 1 (default): lattice build on a side stream beside features + forest          */
    int32_t overlap_layers;      /* 1 (default): the label layers' mean fields on two streams                     */
    int32_t build_priority_high; /* 0 (default): the build stream has the lowest priority                         */
    int32_t trace;               /* 1: per-block trace of the resident splat (debugging); 2: + synchronous stderr marks */
    int32_t serial_chains;       /* 0 (default): the normaliser's ordered sums by exact wave scans (kernels_crf.hip:
                                    ordered_tile_sum); 1: one dependent addition per entry, like the reference loop.
                                    Results are bit-identical; 1 exists for tests and timing                        */
    int32_t csr_block;           /* points per wave-block of the counting sort: 0 = by the chunk (256 for <= 8 frames,
                                    else 1024), or 256 / 512 / 1024 / 2048 / 4096                                     */
} rvseg_schedule;
void rvseg_schedule_default(rvseg_schedule *s);
/* Applies to every later call on ctx (buffers of a schedule are allocated on first use). */
rvseg_status rvseg_set_schedule(rvseg_ctx *ctx, const rvseg_schedule *s);

/* What the last lattice build + mean field on this context ran with.  `planner_fallback`, `vertices` and
 * `longest_list` come from the device with the build's status: they are valid once rvseg_poll_status(ctx, 1) has returned (host entry points:
 * on return), -1 before.  A planner fall-back is not an error -- the same grid walks the lists the list-major way,
 * results are identical -- but it is slower, so it is reported here instead of staying silent. */
typedef struct rvseg_schedule_info {
    int32_t splat;               /* 0 = no lattice built yet, 1 = list-major walk, 2 = resident bands             */
    int32_t planner_fallback;    /* resident bands planned, but the planner gave up on this many frames           */
    int32_t csr_path;            /* 1 = counting sort, 2 = radix sort                                             */
    int32_t n_frames;            /* frames (1 for a cloud) and points per frame of that lattice                   */
    int32_t points_per_frame;
    int32_t vertices;            /* lattice vertices over all frames                                              */
    int32_t longest_list;        /* entries of the longest vertex list: the longest ordered chain of the splat    */
    int32_t resident_blocks, resident_band, resident_chunk;   /* the resident schedule's shape (0 when not used)  */
    int32_t capacity_log2;       /* hash slots per frame                                                          */
} rvseg_schedule_info;
rvseg_status rvseg_last_schedule(rvseg_ctx *ctx, rvseg_schedule_info *out);

/* ---- timing of the last segment_frames / crf_infer call, measured with HIP events on the
 *      stream the kernels ran on.  names_out receives a ';'-separated list of stage names,
 *      ms_out up to max_stages durations.  Returns the number of stages. */
int32_t rvseg_last_timing(const rvseg_ctx *ctx, char *names_out, size_t names_cap, float *ms_out,
                          int32_t max_stages);

#ifdef __cplusplus
}
#endif
#endif /* RVSEG_H */
