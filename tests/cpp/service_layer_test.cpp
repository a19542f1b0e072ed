// Host-only test of the service / wire layer of include/rvseg_segmenter.hpp (no GPU, no librvseg call
// is reached): LocalMapStore = _cloud_results + srvStoredSemanticsIds + srvGetLocalMapSegmentation
// (src/segmenter.cpp:711-774) and the debug cloud dumps (:684-706).  Exit code 0 = all checks passed.
#include <cstdio>
#include <cstring>

#include "rvseg_segmenter.hpp"

#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    rvseg::LocalMapStore store({"material", "object"});
    rvseg::IdsSrvResponse ids;
    REQUIRE(store.srvStoredSemanticsIds(ids) && ids.local_map_ids.empty());
    store.store(7, {{1, 2, 3}, {4, 5, 6}});
    store.store(-3, {{0, 0}, {8, 8}});
    store.store(7, {{9, 9, 9}, {9, 9, 9}});     // a second result under an id already present
    REQUIRE(store.srvStoredSemanticsIds(ids));
    REQUIRE((ids.local_map_ids == std::vector<int32_t>{7, -3, 7}));   // arrival order, duplicates kept (:722-729)

    rvseg::LocalMapSegmentationRequest req;
    rvseg::LocalMapSegmentationResponse resp;
    req.local_map_id = 7;
    req.segmentation_layers = {"object", "material"};
    REQUIRE(store.srvGetLocalMapSegmentation(req, resp));
    REQUIRE(resp.local_map_id == 7);
    REQUIRE((resp.point_labels == std::vector<uint8_t>{4, 5, 6, 1, 2, 3}));   // requested order, first stored result of the id
    resp = rvseg::LocalMapSegmentationResponse();
    req.segmentation_layers = {"material", "material"};                       // a layer may be asked for twice
    REQUIRE(store.srvGetLocalMapSegmentation(req, resp) && (resp.point_labels == std::vector<uint8_t>{1, 2, 3, 1, 2, 3}));
    resp = rvseg::LocalMapSegmentationResponse();
    req.segmentation_layers = {};
    REQUIRE(store.srvGetLocalMapSegmentation(req, resp) && resp.point_labels.empty());
    req.segmentation_layers = {"material", "texture"};                        // unknown layer -> false (:744-746)
    REQUIRE(!store.srvGetLocalMapSegmentation(req, resp));
    req.segmentation_layers = {"material"};
    req.local_map_id = 8;                                                     // unknown id -> false (:773)
    REQUIRE(!store.srvGetLocalMapSegmentation(req, resp));

    // cloud dumps
    std::vector<rvseg::Layer> layers(2);
    layers[0].name = "material"; layers[1].name = "object";
    for (int c = 0; c < 3; c++) {
        layers[0].classes.push_back({"m" + std::to_string(c), {(uint8_t)(10 * c), (uint8_t)(20 * c), (uint8_t)(30 * c)}});
        layers[1].classes.push_back({"o" + std::to_string(c), {(uint8_t)(255 - c), (uint8_t)c, 128}});
    }
    std::vector<rvseg::CloudPoint> cloud(3);
    for (int i = 0; i < 3; i++) {
        for (int k = 0; k < 3; k++) { cloud[i].xyz[k] = (float)(i + k); cloud[i].rgb[k] = 0.25f * (float)k; cloud[i].normal[k] = 0.f; }
    }
    cloud[1].normal[0] = 1.f;   // a proper normal stays; the zero ones become (0,0,1)
    rvseg::dump_clouds(42, cloud, {{0, 1, 2}, {2, 2, 0}}, layers, dir);
    auto slurp = [&](const std::string& name, std::vector<rvseg::CloudPoint>& out) {
        std::ifstream is(dir + "/" + name, std::ios::binary);
        if (!is.is_open()) return false;
        size_t n = 0;
        is.read(reinterpret_cast<char*>(&n), sizeof(n));
        out.resize(n);
        is.read(reinterpret_cast<char*>(out.data()), (std::streamsize)(n * sizeof(rvseg::CloudPoint)));
        return (bool)is;
    };
    std::vector<rvseg::CloudPoint> a, b, c;
    REQUIRE(slurp("cloud42_rgb.cld", a) && slurp("cloud42_layer_0.cld", b) && slurp("cloud42_layer_1.cld", c));
    REQUIRE(a.size() == 3 && b.size() == 3 && c.size() == 3);
    REQUIRE(std::memcmp(a.data(), cloud.data(), 3 * sizeof(rvseg::CloudPoint)) == 0);          // untouched colours and normals
    REQUIRE(b[0].normal[2] == 1.f && b[1].normal[0] == 1.f && b[1].normal[2] == 0.f);
    REQUIRE(b[2].rgb[0] == 20.f / 255.0f && b[2].rgb[1] == 40.f / 255.0f && b[2].rgb[2] == 60.f / 255.0f);
    REQUIRE(c[0].rgb[0] == 253.f / 255.0f && c[2].rgb[0] == 1.0f && c[2].rgb[2] == 128.f / 255.0f);
    REQUIRE(c[0].xyz[1] == 1.f);
    std::printf("service layer ok\n");
    return 0;
}
