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
//
// Thread rule as in the reference: one thread drives one Segmenter (the RF worker owns the frame
// context, the fusion thread the cloud context); create one object per thread and GPU.
#ifndef RVSEG_SEGMENTER_HPP
#define RVSEG_SEGMENTER_HPP

#include <cstdint>
#include <stdexcept>
#include <string>
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

class Segmenter {
public:
    explicit Segmenter(const Config& conf) : conf_(conf) {
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
    ~Segmenter() { if (ctx_) rvseg_destroy(ctx_); }
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

private:
    void check(rvseg_status st) const {
        if (st != RVSEG_OK) throw std::runtime_error(std::string(rvseg_status_string(st)) + ": " + rvseg_last_error(ctx_));
    }
    [[noreturn]] void fail(const std::string& msg) {
        rvseg_destroy(ctx_);
        ctx_ = nullptr;
        throw std::runtime_error(msg);
    }
    Config conf_;
    rvseg_ctx* ctx_ = nullptr;
    unsigned total_labels_ = 0;
};

}  // namespace rvseg
#endif
