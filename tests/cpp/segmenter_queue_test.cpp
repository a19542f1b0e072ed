// The queue layer of the C++ Segmenter facade (include/rvseg_segmenter.hpp) driven like the reference's node drives
// its own (src/segmenter.cpp:245-304, 334-346, 434, 518-621): two cameras, 20 key frames enqueued out of order between
// the cameras, the RF worker's loop body draining them in batches, a local map whose nodes skip some key frames, a map
// that has to be postponed, the two worker threads, and the RCCL label gather at world size 1.
// Everything the batched queue path produces is compared with per-frame calls of the same library, bit for bit.
// usage: segmenter_queue_test <forest.dat> <rgb.u8> <depth.u16>
#include <cstdio>
#include <cstring>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "rvseg_segmenter.hpp"

static std::vector<uint8_t> slurp(const char* path) {
    FILE* f = std::fopen(path, "rb");
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> b((size_t)n);
    if (std::fread(b.data(), 1, b.size(), f) != b.size()) throw std::runtime_error("short read");
    std::fclose(f);
    return b;
}

#define REQUIRE(cond) do { if (!(cond)) { std::fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #cond); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage\n"); return 2; }
    try {
        rvseg::Config conf;
        conf.width = 160; conf.height = 120;
        conf.forest_file_name = argv[1];
        conf.max_batch = 8;
        const char* names[2] = {"material", "object"};
        const int counts[2] = {8, 9};
        for (int l = 0; l < 2; l++) {
            rvseg::Layer layer;
            layer.name = names[l];
            for (int c = 0; c < counts[l]; c++) layer.classes.push_back({"class" + std::to_string(c), {(uint8_t)c, (uint8_t)(2 * c), (uint8_t)(3 * c)}});
            layer.unknown_label = counts[l] - 1;
            conf.layers.push_back(layer);
        }
        const int W = conf.width, H = conf.height;
        const size_t N = (size_t)W * H;
        std::vector<uint8_t> rgb0 = slurp(argv[2]);
        std::vector<uint8_t> draw = slurp(argv[3]);
        REQUIRE(rgb0.size() == N * 3 && draw.size() == N * 2);
        const uint16_t* depth0 = reinterpret_cast<const uint16_t*>(draw.data());
        // 20 different key frames derived from the one on disk: colours rolled, depth shifted
        const int n_cam = 2, per_cam = 10;
        std::vector<std::vector<uint8_t>> colors;
        std::vector<std::vector<uint16_t>> depths;
        for (int k = 0; k < n_cam * per_cam; k++) {
            std::vector<uint8_t> c(N * 3);
            std::vector<uint16_t> d(N);
            for (size_t i = 0; i < N; i++) {
                const size_t j = (i + (size_t)k * 37) % N;
                for (int ch = 0; ch < 3; ch++) c[i * 3 + ch] = (uint8_t)(rgb0[j * 3 + ch] + 5 * k);
                d[i] = depth0[j] ? (uint16_t)(depth0[j] + 13 * k) : 0;
            }
            colors.push_back(c);
            depths.push_back(d);
        }
        const float fx = 525.f * W / 640.f;
        float calib[2 * 21] = {1 / fx, 0, -(W / 2.f) / fx, 0, 1 / fx, -(H / 2.f) / fx, 0, 0, 1, 0, 0, 1, -1, 0, 0, 0, -1, 0, 0, 0, 0.6f};
        std::memcpy(calib + 21, calib, 21 * sizeof(float));
        calib[21 + 18] = 0.25f; calib[21 + 20] = 0.9f;   // the second camera sits elsewhere

        rvseg::Segmenter seg(conf);
        seg.setCameras(n_cam, calib);
        // per-frame reference results: one processFrames call per key frame
        std::vector<std::vector<float>> want;
        for (int k = 0; k < n_cam * per_cam; k++) {
            const int cam = k / per_cam;
            want.push_back(seg.processFrames(1, colors[(size_t)k].data(), depths[(size_t)k].data(), calib + cam * 21)[0]);
        }
        REQUIRE(std::memcmp(want[0].data(), want[1].data(), want[0].size() * 4) != 0);
        // ---- enqueue: camera 1 runs ahead of camera 0, then camera 0 catches up (order inside a camera is ascending) ----
        auto seq_of = [&](int cam, int i) { return 100 * (cam + 1) + i; };
        for (int i = 0; i < 6; i++) seg.enqueueFrame(1, seq_of(1, i), colors[(size_t)(per_cam + i)].data(), depths[(size_t)(per_cam + i)].data());
        for (int i = 0; i < 3; i++) seg.enqueueFrame(0, seq_of(0, i), colors[(size_t)i].data(), depths[(size_t)i].data());
        int done = seg.processFramesFromQueueInternalRF();     // 9 queued, max_batch 8: one call takes 8
        REQUIRE(done == 8);
        for (int i = 3; i < per_cam; i++) seg.enqueueFrame(0, seq_of(0, i), colors[(size_t)i].data(), depths[(size_t)i].data());
        for (int i = 6; i < per_cam; i++) seg.enqueueFrame(1, seq_of(1, i), colors[(size_t)(per_cam + i)].data(), depths[(size_t)(per_cam + i)].data());
        int total = done;
        while ((done = seg.processFramesFromQueueInternalRF()) > 0) total += done;
        REQUIRE(total == n_cam * per_cam);
        bool threw = false;
        try { seg.enqueueFrame(2, 1, colors[0].data(), depths[0].data()); } catch (const std::runtime_error&) { threw = true; }
        REQUIRE(threw);
        // every camera's result queue: ascending sequence numbers, posteriors identical to the per-frame calls
        for (int cam = 0; cam < n_cam; cam++) {
            REQUIRE(seg.resultCount(cam) == (size_t)per_cam);
            for (int i = 0; i < per_cam; i++) {
                const std::pair<int, std::vector<float>> r = seg.resultAt(cam, (size_t)i);
                REQUIRE(r.first == seq_of(cam, i));
                const std::vector<float>& w = want[(size_t)(cam * per_cam + i)];
                REQUIRE(r.second.size() == w.size() && std::memcmp(r.second.data(), w.data(), w.size() * 4) == 0);
            }
        }
        // ---- a local map of two nodes that skip key frames (results 100..102 and 200..203 are dropped, :589-592) ----
        auto index_image = [&](int salt) {
            std::vector<int32_t> idx((size_t)n_cam * N);
            for (size_t i = 0; i < idx.size(); i++) idx[i] = ((i * 7 + (size_t)salt) % 5 == 0) ? -1 : (int32_t)((i * 3 + (size_t)salt) % N);
            return idx;
        };
        rvseg::LocalMap m1;
        m1.id = 11; m1.cloud_size = N;
        m1.nodes.resize(2);
        m1.nodes[0].subimage_seqs = {seq_of(0, 3), seq_of(1, 4)};
        m1.nodes[0].index_image = index_image(1);
        m1.nodes[1].subimage_seqs = {seq_of(0, 5), seq_of(1, 7)};
        m1.nodes[1].index_image = index_image(2);
        // a second map that wants a key frame nobody has enqueued yet: it has to wait
        rvseg::LocalMap m2 = m1;
        m2.id = 12;
        m2.nodes.resize(1);
        m2.nodes[0].subimage_seqs = {seq_of(0, 9), seq_of(1, 12)};
        seg.onNewLocalMap(m1);
        seg.onNewLocalMap(m2);
        REQUIRE(seg.processMapFromQueue());          // map 11
        REQUIRE(!seg.processMapFromQueue());         // map 12: camera 1's newest result is 209 < 212 -> postponed (:541-553)
        rvseg::IdsSrvResponse ids;
        REQUIRE(seg.srvStoredSemanticsIds(ids) && ids.local_map_ids == std::vector<int32_t>{11});
        // the same fusion by hand: images in (node, camera) order with the per-frame posteriors
        std::vector<int32_t> idx4;
        std::vector<float> post4;
        const int picks[4][2] = {{0, 3}, {1, 4}, {0, 5}, {1, 7}};
        for (int q = 0; q < 4; q++) {
            const std::vector<int32_t>& im = m1.nodes[(size_t)(q / 2)].index_image;
            idx4.insert(idx4.end(), im.begin() + (std::ptrdiff_t)((size_t)picks[q][0] * N), im.begin() + (std::ptrdiff_t)((size_t)(picks[q][0] + 1) * N));
            const std::vector<float>& w = want[(size_t)(picks[q][0] * per_cam + picks[q][1])];
            post4.insert(post4.end(), w.begin(), w.end());
        }
        const std::vector<std::vector<unsigned char>> by_hand = seg.processMap(4, idx4.data(), post4.data(), N, nullptr, nullptr);
        rvseg::LocalMapSegmentationRequest req;
        rvseg::LocalMapSegmentationResponse resp;
        req.local_map_id = 11;
        req.segmentation_layers = {"material", "object"};
        REQUIRE(seg.srvGetLocalMapSegmentation(req, resp) && resp.point_labels.size() == 2 * N);
        REQUIRE(std::memcmp(resp.point_labels.data(), by_hand[0].data(), N) == 0 && std::memcmp(resp.point_labels.data() + N, by_hand[1].data(), N) == 0);
        // results older than the consumed ones are gone, newer ones wait: camera 0 keeps 106..109, camera 1 208..209
        REQUIRE(seg.resultCount(0) == 4 && seg.resultAt(0, 0).first == seq_of(0, 6));
        REQUIRE(seg.resultCount(1) == 2 && seg.resultAt(1, 0).first == seq_of(1, 8));
        // ---- the worker threads: the postponed map completes once its key frames arrive ----
        seg.start();
        for (int i = 10; i < 13; i++) seg.enqueueFrame(1, seq_of(1, i), colors[(size_t)(i % per_cam)].data(), depths[(size_t)(i % per_cam)].data());
        bool stored = false;
        for (int spin = 0; spin < 5000 && !stored; spin++) {
            rvseg::IdsSrvResponse r;
            seg.srvStoredSemanticsIds(r);
            stored = r.local_map_ids.size() == 2;
            if (!stored) std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
        seg.stop();
        REQUIRE(stored);
        // ---- the label gather of a sharded local map, world size 1 on this box ----
        seg.commInit(0, 1, rvseg::Segmenter::commUniqueId());
        int8_t *d_a = nullptr, *d_b = nullptr;
        REQUIRE(hipMalloc((void**)&d_a, N) == hipSuccess && hipMalloc((void**)&d_b, N) == hipSuccess);
        REQUIRE(hipMemcpy(d_a, by_hand[0].data(), N, hipMemcpyHostToDevice) == hipSuccess);
        REQUIRE(hipMemset(d_b, 0x7f, N) == hipSuccess);
        seg.gatherLabels(d_a, N, d_b, 0, nullptr);
        REQUIRE(hipDeviceSynchronize() == hipSuccess);
        std::vector<unsigned char> back(N);
        REQUIRE(hipMemcpy(back.data(), d_b, N, hipMemcpyDeviceToHost) == hipSuccess);
        REQUIRE(back == by_hand[0]);
        (void)hipFree(d_a); (void)hipFree(d_b);
        std::printf("queue ok\n");
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
}
