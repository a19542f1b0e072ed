/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the rvseg hot path.
 *
 * This is a plain-C restatement of the REFERENCE algorithm (VisualComputingInstitute/
 * RovinaSemanticSegmentation), written from the cited reference lines.  It is the checker the
 * HIP path is compared against; it is never linked, imported or called by the product library
 * (rovinasemanticsegmentation_amd/): only tests/, __graft_entry__.smoke() and the cpu_baseline leg
 * of bench.py use it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - forest model IO + evaluation (rows G-J of SURVEY.md 8a): PINNED bit-exactly against the
 *     reference's own classifier.cpp compiled in place (oracle/_ref/libforest_ref) through the
 *     committed vectors in tests/golden/forest_*.
 *   - permutohedral lattice / DenseCRF (rows N-W): PARITY UNPINNED.  permutohedral.cpp needs Eigen,
 *     which is not in this image, and no stand-in header may be written, so the reference code
 *     cannot be built here; the reference ships no test vectors for it.  The restatement follows
 *     the SSE branch line by line and is checked only by self-consistency properties and by
 *     hand-computed small cases.
 *   - feature extraction / up-sampling (rows A-F, K-M): PARITY UNPINNED at the third-party level
 *     (OpenCV cvtColor/resize/copyMakeBorder, PCL IntegralImageNormalEstimation, Eigen exp are not
 *     vendored and not installed).  Reference-owned rules (mask, stride grid, patch half size,
 *     feature layout, back-projection, -2 / acos(|nz|), fill value, packing) are restated from
 *     the cited lines and pinned by hand-computed known-answer tests.
 *
 * All citations are relative to /root/reference.
 */
#ifndef RVSEG_ORACLE_H
#define RVSEG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ forest (rows G-J) */
typedef struct orc_tree {
    int n_nodes;
    int *split_feature;  /* classifiers.h:191 */
    float *threshold;    /* classifiers.h:195 */
    int *left_child;     /* classifiers.h:200: 0 <=> leaf, right = left+1 */
    /* single-label histograms: hist_off[n]..hist_off[n+1] into hist */
    int *hist_off;
    float *hist;
    /* multi-layer histograms: for node n, layer l: mh_off[mh_node_off[n]+l] .. +1 into mh */
    int *mh_node_off; /* n_nodes+1 */
    int *mh_off;      /* total layers + 1 */
    float *mh;
} orc_tree;

typedef struct orc_forest {
    int n_trees;
    orc_tree *trees;
} orc_forest;

/* Parse the forest.dat stream written by RandomForest::write (classifier.cpp:210-220,144-152;
 * io.h:84-108).  Returns NULL on a malformed buffer. */
orc_forest *orc_forest_load_mem(const void *buf, size_t size);
orc_forest *orc_forest_load(const char *path);
void orc_forest_free(orc_forest *f);
/* number of classes of the single-label histograms (0 if none) and of each layer */
int orc_forest_single_classes(const orc_forest *f);
int orc_forest_layers(const orc_forest *f, int *class_counts, int max_layers);
/* DecisionTree::findLeafNode, classifier.cpp:97-117 */
int orc_tree_find_leaf(const orc_tree *t, const float *x);
/* RandomForest::classLogPosterior, classifier.cpp:166-184.  out has C floats. */
void orc_forest_class_log_posterior(const orc_forest *f, const float *x, float *out);
/* RandomForest::multiClassLogPosterior, classifier.cpp:187-208.  out = layers concatenated. */
void orc_forest_multi_class_log_posterior(const orc_forest *f, const float *x, float *out);
/* batch helper: X is P x D row-major, out is P x sumC */
void orc_forest_eval(const orc_forest *f, const float *X, int P, int D, int multi, float *out);

/* ------------------------------------------------------------------ features (rows A-F) */
typedef struct orc_params {
    int width, height, stride;         /* rf_prediction_stride, config.json:87 */
    float depth_min, depth_max;        /* config.json:89-90 */
    int patch_size, patch_size_reduce; /* config.json:32,34 */
    int feature_color_patch, feature_depth, feature_height, feature_normal; /* config.json:41-44 */
    float fill_value;                  /* 0 in the node (segmenter.cpp:358-362), -1000 in the tools */
    float dcrf_xyz_kernel, dcrf_rgb_kernel, dcrf_kernel_weight; /* config.json:82-84 */
    int dcrf_iterations;               /* config.json:85 */
} orc_params;

void orc_params_default(orc_params *p);
int orc_feature_length(const orc_params *p); /* feature_extractor.h:46-51 */

/* cvtColor(CV_BGR2Lab) on 8-bit data whose channel 0 is whatever the caller put there
 * (feature_extractor.h:129).  src/dst: H x W x 3 bytes. */
void orc_bgr2lab_u8(const uint8_t *src, uint8_t *dst, int n_pixels);
/* cv::resize(..., INTER_LINEAR) of an 8UC3 region (feature_extractor.h:142).  src addressed
 * through a reflect-border accessor on the W x H Lab image; roi top-left (x0,y0) in UNPADDED
 * coordinates (may be negative), side `size`; dst is r x r x 3 bytes. */
void orc_resize_patch_u8(const uint8_t *lab, int W, int H, int x0, int y0, int size, int r,
                         uint8_t *dst);
/* cv::resize(..., INTER_LINEAR) of a float image with C interleaved channels
 * (segmenter.cpp:380-382). */
void orc_resize_linear_f32(const float *src, int sw, int sh, int C, float *dst, int dw, int dh);

/* Back-projection (feature_extractor.h:200-232): calib = Kinv[9] row-major, R[9] row-major, t[3].
 * cloud: H*W*3 floats (x,y,z per pixel), NaN for invalid depth. */
void orc_cloud(const orc_params *p, const uint16_t *depth, const float *calib, float *cloud);
/* pcl::IntegralImageNormalEstimation, AVERAGE_3D_GRADIENT, depth change 0.02, smoothing 10
 * (feature_extractor.h:254-261).  nz_out: H*W floats holding normal_z, NaN where PCL yields NaN.
 * dist_out (optional): the distance map. */
void orc_normals_nz(int W, int H, const float *cloud, float *nz_out, float *dist_out);
/* acos(fabs(nz)) as evaluated by this build: fdlibm acos in double, rounded to float */
float orc_acos_f32(float x);
/* exp as evaluated by this build (densecrf.cpp:102): double range reduction + polynomial */
float orc_exp_f32(float x);

/* FeatureExtractor::extract, NO_LABEL branch (feature_extractor.h:41-291).
 * feat: capacity (H/stride+1)*(W/stride+1) x D.  Returns the number of points P. */
int orc_extract(const orc_params *p, const uint8_t *rgb, const uint16_t *depth, const float *calib,
                float *feat, int *x_v, int *y_v);

/* ------------------------------------------------------------------ RF driver (rows K-M) */
/* Per-frame RF inference as in Segmenter::processFramesFromQueueInternalRF
 * (segmenter.cpp:351-431): extract, per-point multiClassLogPosterior (or classLogPosterior when
 * multi==0), scatter into (H/s)x(W/s)xC_l images initialised to fill_value, cv::resize to WxH,
 * pack [layer][y][x][class].  posteriors: sumC*W*H floats. */
int orc_rf_frame(const orc_params *p, const orc_forest *f, int multi, const uint8_t *rgb,
                 const uint16_t *depth, const float *calib, float *posteriors);
/* Label rules.  mode 0: eval tools (test.cpp:160-175) strict '>' from
 * -1000, -1 if none; mode 1: CRF rule (segmenter.cpp:646-657) strict '>' from 2.0/C else unknown;
 * mode 2: no-CRF rule (segmenter.cpp:664-679) strict '>' from -1000 with sum!=0 guard else unknown;
 * mode 3: DenseCRF::currentMap (densecrf.cpp:202-211) first maximum. */
void orc_labels(const float *values, int N, int C, int mode, int unknown_label, int8_t *labels);

/* local-map fusion, src/segmenter.cpp:561-616: unaries (layers concatenated, layer l = cloud_size x C_l
 * point-major, zero-initialised) += posteriors through the index images, images in order, raster order */
void orc_fuse_posteriors(int n_images, int W, int H, const int32_t *index_images, const float *posteriors, int n_layers,
                         const int *class_counts, int cloud_size, float *unaries);

/* ------------------------------------------------------------------ lattice + CRF (rows N-W) */
typedef struct orc_lattice {
    int N, d, M;
    int *offset;        /* (N+16)*(d+1), permutohedral.cpp:154 */
    float *barycentric; /* (N+16)*(d+1) */
    float *rank;        /* unused by compute, kept for tests */
    int *blur_n1, *blur_n2; /* (d+1)*M each, permutohedral.cpp:296-318 */
    short *keys;        /* M*d: key of vertex i (HashTable::getKey) */
} orc_lattice;

/* Permutohedral::init, SSE branch (permutohedral.cpp:140-321).  feature: N x d, point-major
 * (= column-major d x N MatrixXf). */
orc_lattice *orc_lattice_init(const float *feature, int N, int d);
void orc_lattice_free(orc_lattice *l);
/* Permutohedral::seqCompute (permutohedral.cpp:476-527) */
void orc_lattice_compute_seq(const orc_lattice *l, float *out, const float *in, int value_size,
                             int reverse);
/* Permutohedral::sseCompute (permutohedral.cpp:529-589), scalar restatement with the same
 * operation order (mul then add, no FMA) */
void orc_lattice_compute_sse(const orc_lattice *l, float *out, const float *in, int value_size,
                             int reverse);
/* Permutohedral::compute dispatch (permutohedral.cpp:596-603) */
void orc_lattice_compute(const orc_lattice *l, float *out, const float *in, int value_size,
                         int reverse);
/* DenseKernel::initLattice normaliser, NORMALIZE_SYMMETRIC (pairwise.cpp:40-62). norm: N floats */
void orc_kernel_norm(const orc_lattice *l, float *norm);
/* expAndNormalize (densecrf.cpp:98-106): in/out N x C point-major */
void orc_exp_and_normalize(float *out, const float *in, int N, int C);
/* DenseCRF::inference with one Potts kernel, symmetric normalisation (densecrf.cpp:115-131,
 * pairwise.cpp:63-80,173-178, labelcompatibility.cpp:46-48).  unary_energy, Q: N x C; feature N x d. */
void orc_crf_inference(int N, int C, int d, const float *unary_energy, const float *feature,
                       float potts_w, int iterations, float *Q);
/* Multi-kernel variant for DenseCRF2D examples (densecrf.cpp:61-81): kernel k has feature block
 * features[k] (N x ds[k]) and Potts weight ws[k]. */
void orc_crf_inference_multi(int N, int C, int n_kernels, const int *ds, const float *const *features,
                             const float *ws, const float *unary_energy, int iterations, float *Q);

/* Per-frame CRF features as composed by the north star:
 * f = (x,y,z)*xyz_kernel, (r,g,b)/255*rgb_kernel (segmenter.cpp:629-637); invalid-depth pixels use
 * xyz = 0.  feat: N x 6. */
void orc_frame_crf_features(const orc_params *p, const uint8_t *rgb, const float *cloud, float *feat);

/* Whole hot path for one frame: RF posteriors -> per-layer 5-iteration DenseCRF -> marginals and
 * labels.  marginals: sumC*W*H (layout row M), labels: L*W*H int8 (label mode as given). */
int orc_segment_frame(const orc_params *p, const orc_forest *f, int multi, const uint8_t *rgb,
                      const uint16_t *depth, const float *calib, float *posteriors, float *marginals,
                      int8_t *labels, int label_mode, const int *unknown_labels);

/* ------------------------------------------------------------------ forest learner */
/* DecisionTreeLearner::learn (multi-layer branch, learning.cpp:410-662) + updateMultiHistograms (:960-1012) +
 * RandomForest::write, depth-first with sorts; see rvseg_oracle_train.c for the build-owned definitions (random
 * choices keyed by (seed, tree, node path), objective from the class counts).  X: P x D row-major, labels: P x L.
 * *forest_out receives a malloc'ed forest.dat image (orc_free).  Returns 0 on success. */
int orc_forest_train(const float *X, int P, int D, const int32_t *labels, int n_layers, const int32_t *class_counts,
                     int num_trees, int max_depth, int min_split_examples, int min_child_split_examples, int num_features,
                     int use_bootstrap, float smoothing, uint64_t seed, void **forest_out, size_t *size_out);
void orc_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
