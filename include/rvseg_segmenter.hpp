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
//   onNewNode(): (seq, depth, colour) per camera  :245-293    enqueueFrame(camera, seq, colour, depth)
//       pushed onto _image_queues[camera]         :283
//   processFramesFromQueueInternalRF()            :323-443    processFramesFromQueueInternalRF(): the worker's loop
//       pops ONE frame per iteration              :340-346        body; pops up to max_batch frames over all cameras into
//       pushes (seq, posteriors) per camera       :434            ONE rvseg_segment_frames call, pushes per camera in order
//       posteriors vector [layer][y][x][class]    :413-431    processFrames(): the body for an explicit batch, same layout
//   onNewLocalMap(): push onto _local_map_queue   :300-304    onNewLocalMap(LocalMap)
//   processMapFromQueue(): wait for the newest    :518-719    processMapFromQueue(): false while the map has to be postponed
//       result of every camera, drop skipped      :527-553        (:527-553), else drops skipped results, matches by seq
//       results, match by seq, fuse, label, store :589-616        (:589-597), fuses, labels, stores under the map id
//   the two detached worker threads               :227-232    start() / stop(): the same two loops, joined on stop
//   (no counterpart: single process)                          commInit() / gatherLabels(): the local-map label gather over
//                                                                 RCCL when key frames are sharded over several GPUs
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

#include <array>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <fstream>
#include <mutex>
#include <thread>
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

// One fps_mapper::MultiImageMapNode of a local map as the fusion loop needs it (segmenter.cpp:571-621).  The projector is
// external to the reference (fps_mapper::MultiProjector::project, :578), so the caller hands over its result.
struct LocalMapNode {
    std::vector<int> subimage_seqs;        // m_multi->subimageSeqs(): the depth sequence number of every camera's sub-image
    std::vector<int32_t> index_image;      // IndexImage, cameras stacked row-wise: (n_cameras * H) x W, < 0 = no cloud point (:601-604)
};
// One fps_mapper::LocalMap: its id, nodes and cloud (xyz in the map frame, rgb in [0, 1]; :560, :629-637)
struct LocalMap {
    int32_t id = 0;
    std::vector<LocalMapNode> nodes;
    size_t cloud_size = 0;
    std::vector<float> cloud_xyz, cloud_rgb;   // cloud_size x 3 each (only read with use_dense_crf)
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
        stop();
        if (map_ctx_) rvseg_destroy(map_ctx_);
        if (ctx_) rvseg_destroy(ctx_);
    }

    // ---- the queue layer (segmenter.h:94-108): _image_queues / _result_queues under _frame_mtx, _local_map_queue under
    //      _cloud_processing_mtx ---------------------------------------------------------------------------------------
    // initializeProjector's camera table (segmenter.cpp:161-205): one calibration (K^-1, R, t: 21 floats) per camera, in the
    // mapper's camera order; sizes the queues (:207-225).
    void setCameras(int n_cameras, const float* calib) {
        std::lock_guard<std::mutex> g(frame_mtx_);
        camera_calib_.assign(calib, calib + (size_t)n_cameras * 21);
        image_queues_.assign((size_t)n_cameras, {});
        result_queues_.assign((size_t)n_cameras, {});
    }
    int cameraCount() const { return (int)image_queues_.size(); }

    // onNewNode for one camera's sub-image (segmenter.cpp:271-284): the frames are copied (the reference holds cv::Mat
    // references) and pushed with their depth sequence number.
    void enqueueFrame(int camera, int seq, const uint8_t* color, const uint16_t* depth) {
        const size_t npix = (size_t)conf_.width * conf_.height;
        QueuedFrame f;
        f.seq = seq;
        f.color.assign(color, color + npix * 3);
        f.depth.assign(depth, depth + npix);
        std::lock_guard<std::mutex> g(frame_mtx_);
        if (camera < 0 || (size_t)camera >= image_queues_.size()) throw std::runtime_error("Found a frame for an unknown camera!");   // :203
        image_queues_[(size_t)camera].push_back(std::move(f));
    }

    // One iteration of the RF worker (segmenter.cpp:334-443).  The reference pops ONE frame per iteration and holds
    // _frame_mtx while it extracts and classifies; here up to max_batch queued frames -- cameras visited round robin like
    // the reference's `for i < max` loop, every camera in FIFO order -- go through ONE rvseg_segment_frames call (a
    // chunk of 64 frames runs at 12x the per-frame rate of single-frame calls on MI355X), the mutex is held only to pop
    // and to push.  Returns the number of frames processed (0: nothing queued; the reference sleeps 1 ms then, :438-439).
    int processFramesFromQueueInternalRF() {
        std::vector<QueuedFrame> batch;
        std::vector<int> cams;
        {
            std::lock_guard<std::mutex> g(frame_mtx_);
            bool any = true;
            while (any && (int)batch.size() < conf_.max_batch) {
                any = false;
                for (size_t i = 0; i < image_queues_.size() && (int)batch.size() < conf_.max_batch; i++) {
                    if (image_queues_[i].empty()) continue;
                    batch.push_back(std::move(image_queues_[i].front()));   // :340-341
                    image_queues_[i].pop_front();
                    cams.push_back((int)i);
                    any = true;
                }
            }
        }
        if (batch.empty()) return 0;
        const size_t npix = (size_t)conf_.width * conf_.height;
        const int n = (int)batch.size();
        std::vector<uint8_t> color((size_t)n * npix * 3);
        std::vector<uint16_t> depth((size_t)n * npix);
        std::vector<float> calib((size_t)n * 21);
        for (int k = 0; k < n; k++) {
            std::memcpy(color.data() + (size_t)k * npix * 3, batch[(size_t)k].color.data(), npix * 3);
            std::memcpy(depth.data() + (size_t)k * npix, batch[(size_t)k].depth.data(), npix * 2);
            std::memcpy(calib.data() + (size_t)k * 21, camera_calib_.data() + (size_t)cams[(size_t)k] * 21, 21 * sizeof(float));   // :349
        }
        std::vector<std::vector<float>> post = processFrames(n, color.data(), depth.data(), calib.data());
        std::lock_guard<std::mutex> g(frame_mtx_);
        for (int k = 0; k < n; k++)   // per camera the pop order is the push order: sequence numbers stay ascending (:434)
            result_queues_[(size_t)cams[(size_t)k]].emplace_back(batch[(size_t)k].seq, std::move(post[(size_t)k]));
        return n;
    }

    // read access for tests / monitoring: (seq, posteriors) pairs waiting for the fusion thread
    size_t resultCount(int camera) const { std::lock_guard<std::mutex> g(frame_mtx_); return result_queues_.at((size_t)camera).size(); }
    std::pair<int, std::vector<float>> resultAt(int camera, size_t k) const { std::lock_guard<std::mutex> g(frame_mtx_); return result_queues_.at((size_t)camera).at(k); }

    void onNewLocalMap(LocalMap lmap) {   // segmenter.cpp:300-304
        std::lock_guard<std::mutex> g(cloud_processing_mtx_);
        local_map_queue_.push_back(std::move(lmap));
    }

    // One iteration of the fusion worker (segmenter.cpp:521-719).  false: no map queued, or the front map has to be
    // postponed because some camera's newest result is older than the map's last sub-image (:527-553; an empty result
    // queue counts as "not there yet" -- the reference reads .back() of an empty deque).  true: the front map was
    // processed: per node and camera results older than the wanted sequence number are dropped (:589-592), an exact
    // match is fused (:594-616), a missing one is skipped with a message (:618-621); then CRF / no-CRF labelling and
    // the store under the map id (:628-713).
    bool processMapFromQueue() {
        LocalMap lmap;
        {
            std::lock_guard<std::mutex> g(cloud_processing_mtx_);
            if (local_map_queue_.empty()) return false;
            const LocalMap& front = local_map_queue_.front();
            std::vector<int> last_ids;
            for (const LocalMapNode& nd : front.nodes) last_ids = nd.subimage_seqs;   // :531-538: the LAST node's ids
            {
                std::lock_guard<std::mutex> f(frame_mtx_);
                for (size_t i = 0; i < last_ids.size() && i < result_queues_.size(); i++)
                    if (result_queues_[i].empty() || result_queues_[i].back().first < last_ids[i]) return false;   // :541-546
            }
            lmap = std::move(local_map_queue_.front());
            local_map_queue_.pop_front();   // :556
        }
        const size_t npix = (size_t)conf_.width * conf_.height;
        const size_t per = (size_t)total_labels_ * npix;
        std::vector<int32_t> index_images;
        std::vector<float> posteriors;
        int n_images = 0;
        for (const LocalMapNode& nd : lmap.nodes) {
            for (size_t i = 0; i < nd.subimage_seqs.size() && i < result_queues_.size(); i++) {
                std::lock_guard<std::mutex> f(frame_mtx_);
                std::deque<std::pair<int, std::vector<float>>>& q = result_queues_[i];
                while (!q.empty() && q.front().first < nd.subimage_seqs[i]) q.pop_front();   // "Drop skipped maps", :589-592
                if (!q.empty() && q.front().first == nd.subimage_seqs[i]) {
                    if (nd.index_image.size() < (i + 1) * npix) throw std::runtime_error("index image smaller than n_cameras * H x W");
                    index_images.insert(index_images.end(), nd.index_image.begin() + (std::ptrdiff_t)(i * npix),
                                        nd.index_image.begin() + (std::ptrdiff_t)((i + 1) * npix));   // rows y + i*_camera_h, :601
                    posteriors.insert(posteriors.end(), q.front().second.begin(), q.front().second.end());
                    q.pop_front();
                    n_images++;
                } else {
                    std::fprintf(stderr, "Couldn't find a semantic map for key frame: %d\n", nd.subimage_seqs[i]);   // :618-621
                }
            }
        }
        if (posteriors.size() != (size_t)n_images * per) throw std::runtime_error("result queue entry of the wrong size");
        processMap(lmap.id, n_images, index_images.data(), posteriors.data(), lmap.cloud_size, lmap.cloud_xyz.data(), lmap.cloud_rgb.data());
        return true;
    }

    // The two worker threads of initializeProjector (segmenter.cpp:227-232), joinable instead of detached.  Each loop
    // sleeps 1 ms when it has nothing to do (:438-439, :549-551).  An exception of a worker ends that worker and is
    // re-thrown by stop().
    void start() {
        if (running_.exchange(true)) return;
        rf_thread_ = std::thread([this] { worker([this] { return processFramesFromQueueInternalRF() > 0; }, rf_error_); });
        map_thread_ = std::thread([this] { worker([this] { return processMapFromQueue(); }, map_error_); });
    }
    void stop() {
        if (!running_.exchange(false)) return;
        if (rf_thread_.joinable()) rf_thread_.join();
        if (map_thread_.joinable()) map_thread_.join();
        std::string err = rf_error_.empty() ? map_error_ : rf_error_;
        rf_error_.clear(); map_error_.clear();
        if (!err.empty()) throw std::runtime_error(err);
    }

    // ---- key frames sharded over several GPUs (one Segmenter per GPU, in its own process or thread): the local-map label
    //      gather to the fusion rank over RCCL / xGMI (rvseg_comm_*, include/rvseg.h) ------------------------------------------
    static std::array<uint8_t, RVSEG_COMM_ID_BYTES> commUniqueId() {   // rank 0 creates it, the host distributes it
        std::array<uint8_t, RVSEG_COMM_ID_BYTES> id{};
        if (rvseg_comm_unique_id(id.data()) != RVSEG_OK) throw std::runtime_error("rvseg_comm_unique_id failed (librccl.so missing?)");
        return id;
    }
    void commInit(int rank, int world, const std::array<uint8_t, RVSEG_COMM_ID_BYTES>& id) { check(rvseg_comm_init(ctx_, rank, world, id.data())); }
    // every rank's bytes_per_rank bytes of device memory (int8 labels n x L x H x W, or fp32 posteriors for the
    // order-preserving fusion) land at d_recv + rank * bytes_per_rank on `root`; enqueued on hip_stream
    void gatherLabels(const void* d_local, size_t bytes_per_rank, void* d_recv, int root, void* hip_stream) {
        check(rvseg_gather_frames(ctx_, d_local, bytes_per_rank, d_recv, root, hip_stream));
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
        rvseg_ctx* mc = map_ctx();   // the fusion thread's context: the RF worker may be inside ctx_ at this moment
        check(rvseg_crf_infer(mc, (int32_t)cloud_size, C, 6, energy.data(), pairwise, conf_.dcrf_kernel_weight,
                              conf_.dcrf_iterations, Q.data(), map.data(), RVSEG_LABEL_CRF, conf_.layers[layer].unknown_label), mc);
        return std::vector<unsigned char>(map.begin(), map.end());
    }

    // no-CRF branch (segmenter.cpp:660-681): strict '>' from -1000 with the sum != 0 guard
    std::vector<unsigned char> labelCloud(size_t layer, size_t cloud_size, const float* unaries) {
        const int C = (int)conf_.layers.at(layer).classes.size();
        std::vector<int8_t> map(cloud_size);
        rvseg_ctx* mc = map_ctx();
        check(rvseg_label_values(mc, unaries, (int32_t)cloud_size, C, RVSEG_LABEL_NOCRF, conf_.layers[layer].unknown_label, map.data()), mc);
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
        rvseg_ctx* mc = map_ctx();
        check(rvseg_fuse_posteriors(mc, n_images, index_images, posteriors, (int32_t)cc.size(), cc.data(), (int32_t)cloud_size, flat.data()), mc);
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
        rvseg_ctx* mc = map_ctx();
        check(rvseg_process_map_device(mc, n_images, d_index_images, d_posteriors, (int32_t)cloud_size, d_cloud_xyz, d_cloud_rgb,
                                       d_labels_out, nullptr, hip_stream), mc);
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
    // use_dense_crf = 0 because the node runs the CRF on the fused cloud only).  Created on first use; everything the
    // fusion thread calls (fusePosteriors, processCloud, labelCloud, processMap*) runs on it, everything the RF worker
    // calls (processFrames*) on the frame context -- one context per thread, as include/rvseg.h requires.
    rvseg_ctx* map_ctx() {
        std::lock_guard<std::mutex> g(map_ctx_mtx_);
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
    struct QueuedFrame {
        int seq = 0;
        std::vector<uint8_t> color;
        std::vector<uint16_t> depth;
    };
    template <class Step>
    void worker(Step step, std::string& error) {
        try {
            while (running_.load()) {
                if (!step()) std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
        } catch (const std::exception& e) {
            error = e.what();
        }
    }
    void check(rvseg_status st, const rvseg_ctx* c = nullptr) const {
        if (st != RVSEG_OK) throw std::runtime_error(std::string(rvseg_status_string(st)) + ": " + rvseg_last_error(c ? c : ctx_));
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
    // queue layer
    mutable std::mutex frame_mtx_;            // _frame_mtx: image and result queues
    std::mutex cloud_processing_mtx_;         // _cloud_processing_mtx: the local-map queue
    std::vector<float> camera_calib_;         // 21 floats per camera
    std::vector<std::deque<QueuedFrame>> image_queues_;                               // _image_queues
    std::vector<std::deque<std::pair<int, std::vector<float>>>> result_queues_;      // _result_queues
    std::deque<LocalMap> local_map_queue_;                                            // _local_map_queue
    std::mutex map_ctx_mtx_;
    std::atomic<bool> running_{false};
    std::thread rf_thread_, map_thread_;
    std::string rf_error_, map_error_;
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
