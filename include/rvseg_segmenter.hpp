// rvseg_segmenter.hpp -- C++ host-side mirror of the reference's `class Segmenter` for the hot
// path, over the C ABI of librvseg.so (include/rvseg.h).  Header only; no ROS, OpenCV or Eigen.
//
// It keeps the reference's names, argument meaning and error behaviour for this path so that
// src/segmenter.cpp can be re-pointed at it:
//
//   reference (include/segmenter.h, src/segmenter.cpp)        here
//   ---------------------------------------------------------------------------------------------
//   Segmenter::Segmenter(..., config_file, ...)   :38-129     Segmenter(const Config&)
//       throws std::runtime_error                 :65,198          throws std::runtime_error
//       loads forest.dat, silently continues if missing :106-115   throws (RVSEG_ERR_IO)
//   processFramesFromQueueInternalRF()            :323-443    processFrames(): one call per
//       posteriors vector [layer][y][x][class]    :413-431        dequeued batch, same layout
//   processMapFromQueue() CRF branch              :628-658    processCloud(): DenseCRF per layer
//       label = max marginal > 2.0/C else Unknown :646-657        same rule (RVSEG_LABEL_CRF)
//   processMapFromQueue() no-CRF branch           :660-681    labelCloud(): RVSEG_LABEL_NOCRF rule
//   processMapFromQueue() accumulation loop       :561-616    fusePosteriors(), processMap()
//   srvSegmentationInformation()                  :776-791    srvSegmentationInformation()
//   _cloud_results + srvStoredSemanticsIds()      :711-729    storeMapResult(), srvStoredSemanticsIds()
//   srvGetLocalMapSegmentation()                  :731-774    srvGetLocalMapSegmentation(): false for an unknown
//                                                                 layer name or map id, layers concatenated
//   /tmp/cloud<ID>_rgb.cld, _layer_<l>.cld dumps  :684-706    dumpClouds()
//   DenseCRF2D::addPairwiseGaussian/Bilateral     densecrf.cpp:61-81   DenseCRF2D (rvseg_crf_features_*)
//
// Thread rule as in the reference: one thread drives one Segmenter (the RF worker owns the frame
// context, the fusion thread the cloud context); create one object per thread and GPU.
#ifndef RVSEG_SEGMENTER_HPP
#define RVSEG_SEGMENTER_HPP

#include <cstdint>
#include <cstdio>
#include <fstream>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "rvseg.h"

namespace rvseg {

// One entry of config.json "color_codings" (resources/config.json:50-79) with label >= 0
struct LabelClass {
    std::string name;
    uint8_t color[3];
};

struct Layer {
    std::string name;                 // "material", "object"
    std::vector<LabelClass> classes;  // ordered by label id (segmenter.cpp:73-98)
    int unknown_label = 0;            // index of the class named "Unknown" (segmenter.cpp:88-96)
};

// The hot-path keys of resources/config.json plus the camera size that the
// node infers from the intrinsics (segmenter.cpp:194-199).
struct Config {
    int width = 640, height = 480;
    std::string forest_file_name;     // config.json:48 ("forest_file_name", resolved against root_dir)
    std::vector<Layer> layers;
    int rf_prediction_stride = 2;
    float depth_min = 0.5f, depth_max = 15.0f;
    int patch_size = 77, patch_size_reduce = 11;
    bool feature_color_patch = true, feature_depth = true, feature_height = true, feature_normal = true;
    bool use_dense_crf = false;
    float dcrf_xyz_kernel = 0.5f, dcrf_rgb_kernel = 4.0f, dcrf_kernel_weight = 10.0f;
    int dcrf_iterations = 10;
    int max_batch = 8;
    int device = 0;
};

// srv/SegmentationInformationSrv.srv response (segmenter.cpp:776-791)
struct SegmentationInformation {
    std::vector<std::string> layer_names;
    std::vector<uint32_t> class_counts;
    std::vector<std::string> class_names;   // flattened
    std::vector<uint8_t> class_colors;      // flattened RGB
};

// srv/IdsSrv.srv response
struct IdsSrvResponse {
    std::vector<int32_t> local_map_ids;
};

// srv/LocalMapSegmentationSrv.srv
struct LocalMapSegmentationRequest {
    int32_t local_map_id = 0;
    std::vector<std::string> segmentation_layers;
};
struct LocalMapSegmentationResponse {
    int32_t local_map_id = 0;
    std::vector<uint8_t> point_labels;   // requested layers concatenated, each cloud_size long (segmenter.cpp:757-768)
};

// One point of the cloud as the debug dumps write it.  fps_mapper's Cloud::write is not in the reference tree
// (external dependency), so the record below is this build's own: size_t count, then per point
// 9 little-endian floats (x, y, z, nx, ny, nz, r, g, b), rgb in [0, 1].
struct CloudPoint {
    float xyz[3];
    float normal[3];
    float rgb[3];
};

// The node's result store and the two services that read it (segmenter.cpp:711-774), free of any device
// state: (map id, result_labels[layer][point]) pairs in arrival order behind one mutex.
class LocalMapStore {
public:
    explicit LocalMapStore(std::vector<std::string> layer_names = {}) : layer_names_(std::move(layer_names)) {}

    // "Save data for the service based on the map id" (segmenter.cpp:711-713)
    void store(int32_t local_map_id, const std::vector<std::vector<unsigned char>>& result_labels) {
        std::lock_guard<std::mutex> g(mtx_);
        results_.emplace_back(local_map_id, result_labels);
    }

    bool srvStoredSemanticsIds(IdsSrvResponse& resp) const {   // segmenter.cpp:722-729
        std::lock_guard<std::mutex> g(mtx_);
        for (const auto& m : results_) resp.local_map_ids.push_back(m.first);
        return true;
    }

    bool srvGetLocalMapSegmentation(const LocalMapSegmentationRequest& req, LocalMapSegmentationResponse& resp) const {   // :731-774
        std::vector<int> layer_indices;
        for (const std::string& l : req.segmentation_layers)
            for (size_t i = 0; i < layer_names_.size(); i++)
                if (l == layer_names_[i]) { layer_indices.push_back((int)i); break; }
        if (req.segmentation_layers.size() != layer_indices.size()) return false;   // an unknown layer name (:744-746)
        std::lock_guard<std::mutex> g(mtx_);
        for (const auto& m : results_) {
            if (m.first != req.local_map_id) continue;
            const std::vector<std::vector<unsigned char>>& result_labels = m.second;
            resp.local_map_id = m.first;
            const size_t point_count = result_labels.empty() ? 0 : result_labels[0].size();
            resp.point_labels.reserve(point_count * layer_indices.size());
            for (int l : layer_indices)
                resp.point_labels.insert(resp.point_labels.end(), result_labels[(size_t)l].begin(), result_labels[(size_t)l].end());
            return true;   // the first stored result of that id, like the reference's linear search
        }
        return false;      // an unknown map id (:773)
    }

private:
    std::vector<std::string> layer_names_;
    mutable std::mutex mtx_;
    std::vector<std::pair<int32_t, std::vector<std::vector<unsigned char>>>> results_;
};

// The debug dumps of processMapFromQueue (segmenter.cpp:684-706): <dir>/cloud<ID>_rgb.cld with the cloud's own
// colours, then per layer <dir>/cloud<ID>_layer_<l>.cld with every point painted in its class colour and
// near-zero normals replaced by (0,0,1) (kept for the following layers, as in the reference, which edits the
// cloud in place).  The reference's directory is fixed to /tmp.
inline void dump_clouds(int32_t local_map_id, std::vector<CloudPoint> cloud, const std::vector<std::vector<unsigned char>>& result_labels,
                        const std::vector<Layer>& layers, const std::string& dir = "/tmp") {
    auto write = [&](const std::string& path) {
        std::ofstream os(path, std::ios::binary);
        if (!os.is_open()) throw std::runtime_error("Could not open file. (" + path + ")");
        const size_t n = cloud.size();
        os.write(reinterpret_cast<const char*>(&n), sizeof(n));
        os.write(reinterpret_cast<const char*>(cloud.data()), (std::streamsize)(n * sizeof(CloudPoint)));
    };
    const std::string base = dir + "/cloud" + std::to_string(local_map_id);
    write(base + "_rgb.cld");
    for (size_t layer = 0; layer < result_labels.size() && layer < layers.size(); layer++) {
        for (size_t i = 0; i < cloud.size() && i < result_labels[layer].size(); i++) {
            CloudPoint& pt = cloud[i];
            const float n2 = pt.normal[0] * pt.normal[0] + pt.normal[1] * pt.normal[1] + pt.normal[2] * pt.normal[2];
            if (n2 < 0.1f) { pt.normal[0] = 0.f; pt.normal[1] = 0.f; pt.normal[2] = 1.f; }
            const LabelClass& c = layers[layer].classes.at(result_labels[layer][i]);
            for (int k = 0; k < 3; k++) pt.rgb[k] = static_cast<float>(c.color[k]) / 255.0f;
        }
        write(base + "_layer_" + std::to_string(layer) + ".cld");
    }
}

class Segmenter {
public:
    explicit Segmenter(const Config& conf) : conf_(conf), store_(layer_names_of(conf)) {
        rvseg_params p;
        rvseg_params_default(&p);
        p.width = conf.width; p.height = conf.height;
        p.stride = conf.rf_prediction_stride;
        p.depth_min = conf.depth_min; p.depth_max = conf.depth_max;
        p.patch_size = conf.patch_size; p.patch_size_reduce = conf.patch_size_reduce;
        p.feature_color_patch = conf.feature_color_patch; p.feature_depth = conf.feature_depth;
        p.feature_height = conf.feature_height; p.feature_normal = conf.feature_normal;
        p.fill_value = 0.0f;            // the node zero-fills its low-res images (segmenter.cpp:358-362)
        p.use_dense_crf = 0;            // the node runs the CRF on the fused cloud, not per frame
        p.dcrf_xyz_kernel = conf.dcrf_xyz_kernel; p.dcrf_rgb_kernel = conf.dcrf_rgb_kernel;
        p.dcrf_kernel_weight = conf.dcrf_kernel_weight; p.dcrf_iterations = conf.dcrf_iterations;
        p.multi_layer = 1;              // shared forest, multiClassLogPosterior (segmenter.cpp:368)
        p.label_mode = RVSEG_LABEL_NOCRF;
        if (conf.layers.size() > RVSEG_MAX_LAYERS) throw std::runtime_error("too many label layers");
        for (size_t l = 0; l < conf.layers.size(); l++) p.unknown_label[l] = conf.layers[l].unknown_label;
        p.max_batch = conf.max_batch;
        p.device = conf.device;
        if (rvseg_create(&p, &ctx_) != RVSEG_OK) throw std::runtime_error(std::string("rvseg_create: ") + rvseg_last_error(nullptr));
        if (rvseg_forest_load(ctx_, conf.forest_file_name.c_str()) != RVSEG_OK) {
            const std::string msg = rvseg_last_error(ctx_);
            rvseg_destroy(ctx_);
            ctx_ = nullptr;
            throw std::runtime_error("forest: " + msg);
        }
        int32_t n_layers = 0, cc[RVSEG_MAX_LAYERS];
        rvseg_forest_info(ctx_, nullptr, nullptr, nullptr, &n_layers, cc);
        if ((size_t)n_layers != conf.layers.size()) fail("model / config mismatch: layer count");   // README.md:30
        total_labels_ = 0;
        for (int l = 0; l < n_layers; l++) {
            if ((size_t)cc[l] != conf.layers[l].classes.size()) fail("model / config mismatch: class count of layer " + conf.layers[l].name);
            total_labels_ += cc[l];
        }
    }
    ~Segmenter() {
        if (map_ctx_) rvseg_destroy(map_ctx_);
        if (ctx_) rvseg_destroy(ctx_);
    }
    Segmenter(const Segmenter&) = delete;
    Segmenter& operator=(const Segmenter&) = delete;

    // Body of processFramesFromQueueInternalRF for n dequeued frames: color = rgb8 (n x H x W x 3),
    // depth = 16UC1 millimetres, calib = n x 21 floats (K^-1, R, t).  Returns one `posteriors`
    // vector per frame, layout [layer][y][x][class] (segmenter.cpp:413-431).
    std::vector<std::vector<float>> processFrames(int n, const uint8_t* color, const uint16_t* depth, const float* calib) {
        const size_t per = (size_t)total_labels_ * conf_.width * conf_.height;
        std::vector<float> flat((size_t)n * per);
        check(rvseg_segment_frames(ctx_, n, color, depth, calib, flat.data(), nullptr, nullptr));
        std::vector<std::vector<float>> out((size_t)n);
        for (int i = 0; i < n; i++) out[i].assign(flat.begin() + (size_t)i * per, flat.begin() + (size_t)(i + 1) * per);
        return out;
    }

    // CRF branch of processMapFromQueue for one layer: `unaries` is the accumulated posterior
    // matrix C x cloud_size (Eigen column-major == cloud_size x C point-major), `pairwise` the
    // 6 x cloud_size feature matrix of segmenter.cpp:629-637.  Returns result_labels[l].
    std::vector<unsigned char> processCloud(size_t layer, size_t cloud_size, const float* unaries, const float* pairwise) {
        const int C = (int)conf_.layers.at(layer).classes.size();
        std::vector<float> energy(cloud_size * (size_t)C), Q(cloud_size * (size_t)C);
        for (size_t i = 0; i < energy.size(); i++) energy[i] = -unaries[i];   // crf.setUnaryEnergy(-unaries[l]), :642
        std::vector<int8_t> map(cloud_size);
        check(rvseg_crf_infer(ctx_, (int32_t)cloud_size, C, 6, energy.data(), pairwise, conf_.dcrf_kernel_weight,
                              conf_.dcrf_iterations, Q.data(), map.data(), RVSEG_LABEL_CRF, conf_.layers[layer].unknown_label));
        return std::vector<unsigned char>(map.begin(), map.end());
    }

    // no-CRF branch (segmenter.cpp:660-681): strict '>' from -1000 with the sum != 0 guard
    std::vector<unsigned char> labelCloud(size_t layer, size_t cloud_size, const float* unaries) {
        const int C = (int)conf_.layers.at(layer).classes.size();
        std::vector<int8_t> map(cloud_size);
        check(rvseg_label_values(ctx_, unaries, (int32_t)cloud_size, C, RVSEG_LABEL_NOCRF, conf_.layers[layer].unknown_label, map.data()));
        return std::vector<unsigned char>(map.begin(), map.end());
    }

    // Accumulation loop of processMapFromQueue (segmenter.cpp:561-616): index_images holds one
    // H x W IndexImage per (map node, camera) that has a segmentation, posteriors the matching
    // label distributions (processFrames output) in the same order.  Returns unaries[layer], each
    // C_l x cloud_size column-major like the Eigen matrices of :563-567.
    std::vector<std::vector<float>> fusePosteriors(int n_images, const int32_t* index_images, const float* posteriors, size_t cloud_size) {
        std::vector<int32_t> cc;
        for (const Layer& l : conf_.layers) cc.push_back((int32_t)l.classes.size());
        std::vector<float> flat(cloud_size * (size_t)total_labels_);
        check(rvseg_fuse_posteriors(ctx_, n_images, index_images, posteriors, (int32_t)cc.size(), cc.data(), (int32_t)cloud_size, flat.data()));
        std::vector<std::vector<float>> out(cc.size());
        size_t off = 0;
        for (size_t l = 0; l < cc.size(); l++) {
            out[l].assign(flat.begin() + off, flat.begin() + off + cloud_size * (size_t)cc[l]);
            off += cloud_size * (size_t)cc[l];
        }
        return out;
    }

    // processMapFromQueue for one local map (segmenter.cpp:561-682): fusion, then per layer the cloud
    // CRF (:628-658) or the no-CRF rule (:660-681).  cloud_xyz / cloud_rgb: cloud_size x 3, rgb in [0,1].
    std::vector<std::vector<unsigned char>> processMap(int n_images, const int32_t* index_images, const float* posteriors,
                                                       size_t cloud_size, const float* cloud_xyz, const float* cloud_rgb) {
        const std::vector<std::vector<float>> unaries = fusePosteriors(n_images, index_images, posteriors, cloud_size);
        std::vector<std::vector<unsigned char>> result_labels(unaries.size());
        std::vector<float> pairwise;
        if (conf_.use_dense_crf) {
            pairwise.resize(cloud_size * 6);
            for (size_t i = 0; i < cloud_size; i++) {            // segmenter.cpp:629-637
                for (int k = 0; k < 3; k++) pairwise[i * 6 + k] = cloud_xyz[i * 3 + k] * conf_.dcrf_xyz_kernel;
                for (int k = 0; k < 3; k++) pairwise[i * 6 + 3 + k] = cloud_rgb[i * 3 + k] * conf_.dcrf_rgb_kernel;
            }
        }
        for (size_t l = 0; l < unaries.size(); l++)
            result_labels[l] = conf_.use_dense_crf ? processCloud(l, cloud_size, unaries[l].data(), pairwise.data())
                                                   : labelCloud(l, cloud_size, unaries[l].data());
        return result_labels;
    }

    // processMapFromQueue with every buffer in HBM (device pointers; work is enqueued on hip_stream and
    // not waited for): d_posteriors is what rvseg_segment_frames_device wrote, d_labels_out receives
    // L x cloud_size labels.  Nothing crosses PCIe.
    void processMapDevice(int n_images, const int32_t* d_index_images, const float* d_posteriors, size_t cloud_size,
                          const float* d_cloud_xyz, const float* d_cloud_rgb, int8_t* d_labels_out, void* hip_stream) {
        check(rvseg_process_map_device(map_ctx(), n_images, d_index_images, d_posteriors, (int32_t)cloud_size, d_cloud_xyz, d_cloud_rgb,
                                       d_labels_out, nullptr, hip_stream));
    }

    // "Save data for the service based on the map id" (segmenter.cpp:711-713)
    void storeMapResult(int32_t local_map_id, const std::vector<std::vector<unsigned char>>& result_labels) { store_.store(local_map_id, result_labels); }

    // processMap + storeMapResult: what processMapFromQueue does for one dequeued local map
    std::vector<std::vector<unsigned char>> processMap(int32_t local_map_id, int n_images, const int32_t* index_images,
                                                       const float* posteriors, size_t cloud_size, const float* cloud_xyz,
                                                       const float* cloud_rgb) {
        std::vector<std::vector<unsigned char>> r = processMap(n_images, index_images, posteriors, cloud_size, cloud_xyz, cloud_rgb);
        storeMapResult(local_map_id, r);
        return r;
    }

    bool srvStoredSemanticsIds(IdsSrvResponse& resp) const { return store_.srvStoredSemanticsIds(resp); }
    bool srvGetLocalMapSegmentation(const LocalMapSegmentationRequest& req, LocalMapSegmentationResponse& resp) const {
        return store_.srvGetLocalMapSegmentation(req, resp);
    }
    void dumpClouds(int32_t local_map_id, const std::vector<CloudPoint>& cloud, const std::vector<std::vector<unsigned char>>& result_labels,
                    const std::string& dir = "/tmp") const {
        dump_clouds(local_map_id, cloud, result_labels, conf_.layers, dir);
    }

    // Device-resident twin of processFrames: rgb / depth / posteriors are device pointers, work is enqueued on
    // hip_stream (rvseg_segment_frames_device); calib stays a host array.
    void processFramesDevice(int n, const uint8_t* d_color, const uint16_t* d_depth, const float* calib, float* d_posteriors_out,
                             void* hip_stream) {
        check(rvseg_segment_frames_device(ctx_, n, d_color, d_depth, calib, d_posteriors_out, nullptr, nullptr, hip_stream));
    }

    bool srvSegmentationInformation(SegmentationInformation& resp) const {
        resp = SegmentationInformation();
        for (const Layer& l : conf_.layers) {
            resp.layer_names.push_back(l.name);
            resp.class_counts.push_back((uint32_t)l.classes.size());
            for (const LabelClass& c : l.classes) {
                resp.class_names.push_back(c.name);
                resp.class_colors.insert(resp.class_colors.end(), c.color, c.color + 3);
            }
        }
        return true;
    }

    unsigned totalLabels() const { return total_labels_; }
    rvseg_ctx* context() { return ctx_; }
    // The fusion thread's context: same model and parameters, CRF switch as configured (the frame context keeps
    // use_dense_crf = 0 because the node runs the CRF on the fused cloud only).  Created on first use.
    rvseg_ctx* map_ctx() {
        if (map_ctx_) return map_ctx_;
        rvseg_params p;
        rvseg_params_default(&p);
        p.width = conf_.width; p.height = conf_.height;
        p.use_dense_crf = conf_.use_dense_crf ? 1 : 0;
        p.dcrf_xyz_kernel = conf_.dcrf_xyz_kernel; p.dcrf_rgb_kernel = conf_.dcrf_rgb_kernel;
        p.dcrf_kernel_weight = conf_.dcrf_kernel_weight; p.dcrf_iterations = conf_.dcrf_iterations;
        p.patch_size = conf_.patch_size; p.patch_size_reduce = conf_.patch_size_reduce;
        p.feature_color_patch = conf_.feature_color_patch; p.feature_depth = conf_.feature_depth;
        p.feature_height = conf_.feature_height; p.feature_normal = conf_.feature_normal;
        p.multi_layer = 1;
        for (size_t l = 0; l < conf_.layers.size(); l++) p.unknown_label[l] = conf_.layers[l].unknown_label;
        p.device = conf_.device;
        if (rvseg_create(&p, &map_ctx_) != RVSEG_OK) throw std::runtime_error(std::string("rvseg_create: ") + rvseg_last_error(nullptr));
        if (rvseg_forest_load(map_ctx_, conf_.forest_file_name.c_str()) != RVSEG_OK) {
            const std::string msg = rvseg_last_error(map_ctx_);
            rvseg_destroy(map_ctx_);
            map_ctx_ = nullptr;
            throw std::runtime_error("forest: " + msg);
        }
        return map_ctx_;
    }

private:
    void check(rvseg_status st) const {
        if (st != RVSEG_OK) throw std::runtime_error(std::string(rvseg_status_string(st)) + ": " + rvseg_last_error(ctx_));
    }
    [[noreturn]] void fail(const std::string& msg) {
        rvseg_destroy(ctx_);
        ctx_ = nullptr;
        throw std::runtime_error(msg);
    }
    static std::vector<std::string> layer_names_of(const Config& c) {
        std::vector<std::string> n;
        for (const Layer& l : c.layers) n.push_back(l.name);
        return n;
    }
    Config conf_;
    rvseg_ctx* ctx_ = nullptr;
    rvseg_ctx* map_ctx_ = nullptr;
    unsigned total_labels_ = 0;
    LocalMapStore store_;   // _cloud_results / _cloud_mtx (segmenter.h:94-108)
};

// DenseCRF2D as examples/dense_inference.cpp:83-107 drives it: the two image kernels, then map().
class DenseCRF2D {
public:
    DenseCRF2D(rvseg_ctx* ctx, int W, int H, int M) : ctx_(ctx), W_(W), H_(H), M_(M) {}
    void setUnaryEnergy(const float* unary /* N x M */) { unary_.assign(unary, unary + (size_t)W_ * H_ * M_); }
    void addPairwiseGaussian(float sx, float sy, float potts_w) {                                   // densecrf.cpp:61-69
        feats_.emplace_back((size_t)W_ * H_ * 2);
        if (rvseg_crf_features_gaussian(W_, H_, sx, sy, feats_.back().data()) != RVSEG_OK) throw std::runtime_error("bad arguments");
        ds_.push_back(2); ws_.push_back(potts_w);
    }
    void addPairwiseBilateral(float sx, float sy, float sr, float sg, float sb, const unsigned char* im, float potts_w) {   // :70-81
        feats_.emplace_back((size_t)W_ * H_ * 5);
        if (rvseg_crf_features_bilateral(W_, H_, sx, sy, sr, sg, sb, im, feats_.back().data()) != RVSEG_OK) throw std::runtime_error("bad arguments");
        ds_.push_back(5); ws_.push_back(potts_w);
    }
    // DenseCRF::inference (densecrf.cpp:115-131); map_out (optional) = DenseCRF::map (:132-137)
    std::vector<float> inference(int n_iterations, std::vector<int8_t>* map_out = nullptr) {
        const size_t N = (size_t)W_ * H_;
        if (unary_.empty()) unary_.assign(N * M_, 0.f);
        std::vector<float> Q(N * M_);
        std::vector<const float*> fp;
        for (const auto& f : feats_) fp.push_back(f.data());
        if (map_out) map_out->resize(N);
        const rvseg_status st = rvseg_crf_infer_multi(ctx_, (int32_t)N, M_, (int32_t)fp.size(), ds_.data(), fp.data(), ws_.data(), unary_.data(),
                                                      n_iterations, Q.data(), map_out ? map_out->data() : nullptr, RVSEG_LABEL_ARGMAX, 0);
        if (st != RVSEG_OK) throw std::runtime_error(std::string(rvseg_status_string(st)) + ": " + rvseg_last_error(ctx_));
        return Q;
    }
private:
    rvseg_ctx* ctx_;
    int W_, H_, M_;
    std::vector<float> unary_;
    std::vector<std::vector<float>> feats_;
    std::vector<int32_t> ds_;
    std::vector<float> ws_;
};

}  // namespace rvseg
#endif
