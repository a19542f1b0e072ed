// Exercises the C++ Segmenter facade (include/rvseg_segmenter.hpp) the way src/segmenter.cpp
// drives the reference objects: construct from a config, run the per-frame RF body, fuse the
// posteriors of the frame into a "cloud" (here: the frame's own pixels) and label it with and
// without the dense CRF.  Prints the results in a small text format that the Python test compares
// with the oracle.  usage: segmenter_facade_test <forest.dat> <rgb.u8> <depth.u16> <out.bin>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

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

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage\n"); return 2; }
    try {
        // a missing model must throw, like libf::read's "Could not open file." (io.h:118-121)
        bool threw = false;
        try {
            rvseg::Config bad;
            bad.forest_file_name = "/nonexistent/forest.dat";
            rvseg::Segmenter s(bad);
        } catch (const std::runtime_error& e) {
            threw = std::strstr(e.what(), "Could not open file") != nullptr;
        }
        if (!threw) { std::fprintf(stderr, "missing model did not throw\n"); return 1; }

        rvseg::Config conf;
        conf.width = 160; conf.height = 120;
        conf.forest_file_name = argv[1];
        conf.dcrf_iterations = 3;
        const char* names[2] = {"material", "object"};
        const int counts[2] = {8, 9};
        for (int l = 0; l < 2; l++) {
            rvseg::Layer layer;
            layer.name = names[l];
            for (int c = 0; c < counts[l]; c++) layer.classes.push_back({"class" + std::to_string(c), {(uint8_t)c, (uint8_t)(2 * c), (uint8_t)(3 * c)}});
            layer.unknown_label = counts[l] - 1;
            conf.layers.push_back(layer);
        }
        rvseg::Segmenter seg(conf);
        rvseg::SegmentationInformation info;
        seg.srvSegmentationInformation(info);
        if (info.layer_names.size() != 2 || info.class_counts[1] != 9 || info.class_names.size() != 17 || info.class_colors.size() != 51) return 1;

        const size_t N = (size_t)conf.width * conf.height;
        std::vector<uint8_t> rgb = slurp(argv[2]);
        std::vector<uint8_t> draw = slurp(argv[3]);
        if (rgb.size() != N * 3 || draw.size() != N * 2) { std::fprintf(stderr, "bad input sizes\n"); return 1; }
        const float fx = 525.f * conf.width / 640.f;
        const float calib[21] = {1 / fx, 0, -(conf.width / 2.f) / fx, 0, 1 / fx, -(conf.height / 2.f) / fx, 0, 0, 1,
                                 0, 0, 1, -1, 0, 0, 0, -1, 0, 0, 0, 0.6f};
        auto post = seg.processFrames(1, rgb.data(), reinterpret_cast<const uint16_t*>(draw.data()), calib);
        if (post.size() != 1 || post[0].size() != 17 * N) return 1;

        // "cloud" = the frame's pixels; pairwise features as segmenter.cpp:629-637 would fill them
        std::vector<float> pairwise(N * 6);
        for (size_t i = 0; i < N; i++) {
            pairwise[i * 6 + 0] = (float)(i % conf.width) * 0.01f * conf.dcrf_xyz_kernel;
            pairwise[i * 6 + 1] = (float)(i / conf.width) * 0.01f * conf.dcrf_xyz_kernel;
            pairwise[i * 6 + 2] = 1.0f * conf.dcrf_xyz_kernel;
            for (int c = 0; c < 3; c++) pairwise[i * 6 + 3 + c] = (rgb[i * 3 + c] / 255.0f) * conf.dcrf_rgb_kernel;
        }
        FILE* out = std::fopen(argv[4], "wb");
        std::fwrite(post[0].data(), 4, post[0].size(), out);
        std::fwrite(pairwise.data(), 4, pairwise.size(), out);
        size_t off = 0;
        for (size_t l = 0; l < 2; l++) {
            std::vector<unsigned char> crf = seg.processCloud(l, N, post[0].data() + off, pairwise.data());
            std::vector<unsigned char> plain = seg.labelCloud(l, N, post[0].data() + off);
            std::fwrite(crf.data(), 1, crf.size(), out);
            std::fwrite(plain.data(), 1, plain.size(), out);
            off += N * counts[l];
        }
        // local-map fusion (segmenter.cpp:561-616): the frame is seen twice, once pixel == point and once
        // mirrored on every third pixel; then the no-CRF labelling of the fused cloud (:660-681)
        std::vector<int32_t> index_images(2 * N);
        for (size_t i = 0; i < N; i++) {
            index_images[i] = (int32_t)i;
            index_images[N + i] = i % 3 == 0 ? (int32_t)(N - 1 - i) : -1;
        }
        std::vector<float> two(2 * post[0].size());
        std::memcpy(two.data(), post[0].data(), post[0].size() * 4);
        std::memcpy(two.data() + post[0].size(), post[0].data(), post[0].size() * 4);
        auto unaries = seg.fusePosteriors(2, index_images.data(), two.data(), N);
        auto fused_labels = seg.processMap(2, index_images.data(), two.data(), N, nullptr, nullptr);   // conf.use_dense_crf == false
        // the same map again under an id: kept for the services (segmenter.cpp:711-774)
        auto stored = seg.processMap(5, 2, index_images.data(), two.data(), N, nullptr, nullptr);
        rvseg::IdsSrvResponse ids;
        rvseg::LocalMapSegmentationRequest req;
        rvseg::LocalMapSegmentationResponse resp;
        req.local_map_id = 5;
        req.segmentation_layers = {"object", "material"};
        if (!seg.srvStoredSemanticsIds(ids) || ids.local_map_ids != std::vector<int32_t>{5}) return 1;
        if (!seg.srvGetLocalMapSegmentation(req, resp) || resp.local_map_id != 5 || resp.point_labels.size() != 2 * N) return 1;
        if (std::memcmp(resp.point_labels.data(), fused_labels[1].data(), N) != 0 || std::memcmp(resp.point_labels.data() + N, fused_labels[0].data(), N) != 0) return 1;
        if (stored != fused_labels) return 1;
        req.local_map_id = 6;
        if (seg.srvGetLocalMapSegmentation(req, resp)) return 1;
        for (size_t l = 0; l < 2; l++) std::fwrite(unaries[l].data(), 4, unaries[l].size(), out);
        for (size_t l = 0; l < 2; l++) std::fwrite(fused_labels[l].data(), 1, fused_labels[l].size(), out);
        std::fclose(out);
        std::printf("facade ok\n");
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
}
