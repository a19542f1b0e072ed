// C-ABI entry points that do not belong to the frame / CRF pipelines: context life cycle, forest
// loading, the unit-parity forest evaluation, timing read-back.  See include/rvseg.h for the
// reference interface each one replaces.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <mutex>

#include "rvseg_internal.h"

namespace rvseg {

static std::mutex g_create_err_mtx;
static std::string g_create_err;

bool hip_ok(rvseg_ctx* ctx, hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    std::string msg = std::string(what) + ": " + hipGetErrorString(e);
    if (ctx) ctx->err = msg;
    else { std::lock_guard<std::mutex> g(g_create_err_mtx); g_create_err = msg; }
    return false;
}

static thread_local std::string t_launch_err;

void launch_check(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess && t_launch_err.empty()) t_launch_err = std::string("kernel launch failed: ") + what + ": " + hipGetErrorString(e);
}

rvseg_status launch_error_take(rvseg_ctx* ctx) {
    launch_check("(runtime)");
    if (t_launch_err.empty()) return RVSEG_OK;
    if (ctx) ctx->err = t_launch_err;
    t_launch_err.clear();
    return RVSEG_ERR_HIP;
}

rvseg_status dev_alloc(rvseg_ctx* ctx, DevBuf& b, size_t bytes) {
    dev_free(b);
    if (bytes == 0) bytes = 16;
    RV_HIP(ctx, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return RVSEG_OK;
}

void dev_free(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}

rvseg_status dev_reserve(rvseg_ctx* ctx, DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return RVSEG_OK;
    return dev_alloc(ctx, b, bytes);
}

// ---- Lab tables: OpenCV 2.4 RGB2Lab_b constants (imgproc/color.cpp), see DESIGN.md ------------
static void build_lab_tables(uint16_t gamma[256], uint16_t cbrt_tab[3072], int coeffs[9]) {
    auto sat16 = [](long v) { return (uint16_t)(v < 0 ? 0 : (v > 65535 ? 65535 : v)); };
    for (int i = 0; i < 256; i++) {
        float x = i * (1.f / 255.f);
        float g = x <= 0.04045f ? x * (1.f / 12.92f) : (float)std::pow((double)(x + 0.055) * (1. / 1.055), 2.4);
        gamma[i] = sat16(std::lrintf(255.f * 8 * g));
    }
    for (int i = 0; i < 3072; i++) {
        float x = i * (1.f / (255.f * 8));
        float v = x < 0.008856f ? x * 7.787f + 0.13793103448275862f : std::cbrt(x);
        cbrt_tab[i] = sat16(std::lrintf(32768.f * v));
    }
    static const float xyz[9] = {0.412453f, 0.357580f, 0.180423f, 0.212671f, 0.715160f,
                                 0.072169f, 0.019334f, 0.119193f, 0.950227f};
    static const float white[3] = {0.950456f, 1.f, 1.088754f};
    const float scale[3] = {4096.f / white[0], 4096.f, 4096.f / white[2]};
    for (int i = 0; i < 3; i++) {  // blueIdx = 0 (CV_BGR2Lab): the "R" coefficient meets channel 2
        coeffs[i * 3 + 2] = (int)std::lrint((double)(xyz[i * 3] * scale[i]));
        coeffs[i * 3 + 1] = (int)std::lrint((double)(xyz[i * 3 + 1] * scale[i]));
        coeffs[i * 3 + 0] = (int)std::lrint((double)(xyz[i * 3 + 2] * scale[i]));
    }
}

static int feature_length_of(const rvseg_params& p) {  // feature_extractor.h:46-51
    int n = 0;
    if (p.feature_color_patch) n += p.patch_size_reduce * p.patch_size_reduce * 3;
    if (p.feature_depth) n += 1;
    if (p.feature_height) n += 1;
    if (p.feature_normal) n += 1;
    return n;
}

static rvseg_status upload_forest(rvseg_ctx* ctx) {
    const ForestModel& m = ctx->host_forest;
    DeviceForest& f = ctx->forest;
    const bool multi = ctx->params.multi_layer != 0;
    if (multi && m.layer_classes.empty()) {
        ctx->err = "params.multi_layer is set but the forest carries no multi-layer histograms";
        return RVSEG_ERR_FORMAT;
    }
    if (!multi && m.single_classes == 0) {
        ctx->err = "params.multi_layer is 0 but the forest carries no single-label histograms";
        return RVSEG_ERR_FORMAT;
    }
    f.n_trees = m.n_trees;
    f.max_depth = m.max_depth;
    f.n_nodes = (int)m.nodes.size();
    f.n_leaves = m.n_leaves;
    if (multi) {
        if (m.layer_classes.size() > RVSEG_MAX_LAYERS) { ctx->err = "too many label layers"; return RVSEG_ERR_FORMAT; }
        f.n_layers = (int)m.layer_classes.size();
        f.sum_classes = 0;
        for (int l = 0; l < f.n_layers; l++) { f.class_counts[l] = m.layer_classes[l]; f.sum_classes += m.layer_classes[l]; }
    } else {
        f.n_layers = 1;
        f.class_counts[0] = m.single_classes;
        f.sum_classes = m.single_classes;
    }
    if (f.sum_classes > kMaxClasses) { ctx->err = "more than 64 classes over all layers is not supported"; return RVSEG_ERR_FORMAT; }
    if (m.n_trees > kMaxTrees) { ctx->err = "more than 64 trees are not supported"; return RVSEG_ERR_CAPACITY; }   // parse_forest refuses these already
    const std::vector<float>& hist = multi ? m.multi_hist : m.single_hist;
    rvseg_status st;
    if ((st = dev_alloc(ctx, f.nodes, m.nodes.size() * sizeof(DeviceNode))) != RVSEG_OK) return st;
    if ((st = dev_alloc(ctx, f.roots, m.roots.size() * sizeof(int32_t))) != RVSEG_OK) return st;
    if ((st = dev_alloc(ctx, f.hist, hist.size() * sizeof(float))) != RVSEG_OK) return st;
    {
        // inner nodes do not use `leaf_row`: park the patch coordinates of the tested feature there
        // (feature f of the r x r x 3 patch = cell (dy, dx), channel c; feature_extractor.h:160-167), so the
        // frame kernel needs no divisions per visited node
        std::vector<DeviceNode> nodes = m.nodes;
        const int r = ctx->params.patch_size_reduce;
        const int n_patch = ctx->params.feature_color_patch ? r * r * 3 : 0;
        for (DeviceNode& dn : nodes) {
            if (dn.left == 0) continue;
            dn.leaf_row = 0;
            if (dn.feature < n_patch) {
                const int k = dn.feature / 3, c = dn.feature - 3 * k;
                const int dy = k / r, dx = k - dy * r;
                dn.leaf_row = (c << 16) | (dy << 8) | dx;
            }
        }
        RV_HIP(ctx, hipMemcpy(f.nodes.p, nodes.data(), nodes.size() * sizeof(DeviceNode), hipMemcpyHostToDevice));
        // The frame kernel's 8-byte node: {threshold bits | leaf row, kind << 30 | channel << 28 | dy << 24 | dx << 20 | left}:
        // a node's two children (adjacent in the array) arrive with ONE 16-byte load.  kind: 0 patch value, 1 depth,
        // 2 height, 3 normal.  Possible while node indices fit 20 bits and patch cells 4 bits (r <= 16); larger models
        // keep the 16-byte nodes.
        dev_free(f.nodes8);
        if (nodes.size() < (1u << 20) && r <= 16) {
            const rvseg_params& pp = ctx->params;
            int pos = n_patch;
            const int pos_depth = pp.feature_depth ? pos++ : -1;
            const int pos_height = pp.feature_height ? pos++ : -1;
            const int pos_normal = pp.feature_normal ? pos++ : -1;
            std::vector<uint32_t> n8(nodes.size() * 2);
            for (size_t i = 0; i < nodes.size(); i++) {
                const DeviceNode& dn = nodes[i];
                if (dn.left == 0) { n8[2 * i] = (uint32_t)m.nodes[i].leaf_row; n8[2 * i + 1] = 0u; continue; }
                uint32_t thr_bits;
                std::memcpy(&thr_bits, &dn.threshold, 4);
                uint32_t kind = 0, cell = 0;
                if (dn.feature < n_patch) {
                    const int c = dn.leaf_row >> 16, dy = (dn.leaf_row >> 8) & 255, dx = dn.leaf_row & 255;
                    cell = ((uint32_t)c << 8) | ((uint32_t)dy << 4) | (uint32_t)dx;
                } else {
                    kind = dn.feature == pos_depth ? 1u : (dn.feature == pos_height ? 2u : (dn.feature == pos_normal ? 3u : 1u));
                }
                n8[2 * i] = thr_bits;
                n8[2 * i + 1] = (kind << 30) | (cell << 20) | (uint32_t)dn.left;
            }
            if ((st = dev_alloc(ctx, f.nodes8, n8.size() * 4 + 16)) != RVSEG_OK) return st;
            RV_HIP(ctx, hipMemcpy(f.nodes8.p, n8.data(), n8.size() * 4, hipMemcpyHostToDevice));
        }
    }
    RV_HIP(ctx, hipMemcpy(f.roots.p, m.roots.data(), m.roots.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    RV_HIP(ctx, hipMemcpy(f.hist.p, hist.data(), hist.size() * sizeof(float), hipMemcpyHostToDevice));
    ctx->forest_loaded = true;
    return RVSEG_OK;
}

static rvseg_status parse_error_status(const std::string& err) {
    if (err == "forest has no trees") return RVSEG_ERR_NO_FOREST;
    if (err.find("at most") != std::string::npos && err.find("trees") != std::string::npos) return RVSEG_ERR_CAPACITY;
    return RVSEG_ERR_FORMAT;
}

static rvseg_status copy_out(const std::vector<uint8_t>& bytes, void* out, size_t out_cap, size_t* size_out) {
    if (size_out) *size_out = bytes.size();
    if (!out) return RVSEG_OK;                       // size query
    if (out_cap < bytes.size()) return RVSEG_ERR_INVALID_ARG;
    std::memcpy(out, bytes.data(), bytes.size());
    return RVSEG_OK;
}

}  // namespace rvseg

using namespace rvseg;

extern "C" {

void rvseg_params_default(rvseg_params* p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->width = 640; p->height = 480;
    p->stride = 2;                                   // config.json:87
    p->depth_min = 0.5f; p->depth_max = 15.0f;       // config.json:89-90
    p->patch_size = 77; p->patch_size_reduce = 11;   // config.json:32,34
    p->feature_color_patch = p->feature_depth = p->feature_height = p->feature_normal = 1;
    p->fill_value = 0.0f;                            // segmenter.cpp:358-362
    p->use_dense_crf = 0;                            // config.json:81
    p->dcrf_xyz_kernel = 0.5f; p->dcrf_rgb_kernel = 4.0f; p->dcrf_kernel_weight = 10.0f;
    p->dcrf_iterations = 10;                         // config.json:85
    p->multi_layer = 1;                              // segmenter.cpp:368
    p->label_mode = RVSEG_LABEL_ARGMAX;
    p->unknown_label[0] = 7; p->unknown_label[1] = 8;  // "Unknown" of material / object, config.json:59,73
    p->max_batch = 8;
    p->device = 0;
    p->lattice_capacity_log2 = 0;
}

void rvseg_schedule_default(rvseg_schedule* s) {
    if (!s) return;
    std::memset(s, 0, sizeof(*s));
    s->resident_band = 16;
    s->resident_chunk = 128;
    s->resident_window = -1;
    s->overlap_build = 1;
    s->overlap_layers = 1;
}

rvseg_status rvseg_set_schedule(rvseg_ctx* ctx, const rvseg_schedule* s) {
    if (!ctx) return RVSEG_ERR_INVALID_ARG;
    if (!s || s->splat < 0 || s->splat > 2 || s->resident_blocks < 0 || s->resident_blocks > 16 || s->resident_band < 1 ||
        (s->resident_chunk != 64 && s->resident_chunk != 128) || s->resident_cap_tiles < 0 ||
        (s->group_vertices != 0 && s->group_vertices != 6 && s->group_vertices != 7) ||
        (s->csr_block != 0 && s->csr_block != 256 && s->csr_block != 512 && s->csr_block != 1024 && s->csr_block != 2048 && s->csr_block != 4096)) {
        ctx->err = "bad schedule";
        return RVSEG_ERR_INVALID_ARG;
    }
    ctx->sched = *s;
    return RVSEG_OK;
}

const char* rvseg_status_string(rvseg_status s) {
    switch (s) {
        case RVSEG_OK: return "ok";
        case RVSEG_ERR_INVALID_ARG: return "invalid argument";
        case RVSEG_ERR_IO: return "could not open file";
        case RVSEG_ERR_FORMAT: return "malformed forest";
        case RVSEG_ERR_NO_FOREST: return "no forest loaded";
        case RVSEG_ERR_HIP: return "HIP runtime error";
        case RVSEG_ERR_NO_DEVICE: return "no HIP device";
        case RVSEG_ERR_CAPACITY: return "capacity exceeded";
        case RVSEG_NOT_READY: return "not ready";
    }
    return "unknown";
}

const char* rvseg_last_error(const rvseg_ctx* ctx) {
    if (ctx) return ctx->err.c_str();
    std::lock_guard<std::mutex> g(g_create_err_mtx);
    static thread_local std::string copy;
    copy = g_create_err;
    return copy.c_str();
}

rvseg_status rvseg_create(const rvseg_params* params, rvseg_ctx** out) {
    auto fail = [](rvseg_status st, const std::string& msg) {
        std::lock_guard<std::mutex> g(g_create_err_mtx);
        g_create_err = msg;
        return st;
    };
    if (!params || !out) return fail(RVSEG_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    const rvseg_params& p = *params;
    if (p.width < 4 || p.height < 4 || p.width > 16384 || p.height > 16384) return fail(RVSEG_ERR_INVALID_ARG, "bad image size");
    if (p.stride < 1 || p.stride > 64) return fail(RVSEG_ERR_INVALID_ARG, "bad stride");
    if (!(p.depth_min > 0.f) || !(p.depth_max >= p.depth_min)) return fail(RVSEG_ERR_INVALID_ARG, "bad depth range");
    // RT_MAXR (rvseg_kernels.h): the resize tables hold 16 cells per axis
    if (p.patch_size_reduce < 1 || p.patch_size_reduce > 16 || p.patch_size < 1) return fail(RVSEG_ERR_INVALID_ARG, "bad patch size (patch_size_reduce must be in [1,16])");
    if (p.lattice_capacity_log2 > 30) return fail(RVSEG_ERR_INVALID_ARG, "bad lattice_capacity_log2");
    // The reflected border is patch_size wide (feature_extractor.h:37,130): the largest ROI half
    // size int(patch_size / (2*depth_min)) must fit into it, and a single reflection must do.
    if (p.feature_color_patch) {
        int half_max = (int)(p.patch_size / (2.0 * p.depth_min));
        if (half_max > p.patch_size || p.patch_size > p.width || p.patch_size > p.height)
            return fail(RVSEG_ERR_INVALID_ARG, "patch ROI would leave the reflected border (depth_min too small or image too small)");
    }
    if (p.max_batch < 1 || p.max_batch > 4096) return fail(RVSEG_ERR_INVALID_ARG, "bad max_batch");
    if (p.label_mode < 0 || p.label_mode > 3) return fail(RVSEG_ERR_INVALID_ARG, "bad label_mode");
    if (p.dcrf_iterations < 0) return fail(RVSEG_ERR_INVALID_ARG, "bad dcrf_iterations");

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(RVSEG_ERR_NO_DEVICE, "no HIP device available: librvseg has no CPU fallback");
    if (p.device < 0 || p.device >= ndev) return fail(RVSEG_ERR_INVALID_ARG, "device ordinal out of range");
    if (!hip_ok(nullptr, hipSetDevice(p.device), "hipSetDevice")) return RVSEG_ERR_HIP;

    rvseg_ctx* ctx = new rvseg_ctx();
    ctx->params = p;
    rvseg_schedule_default(&ctx->sched);
    ctx->feature_length = feature_length_of(p);
    if (!hip_ok(nullptr, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking), "hipStreamCreate")) {
        delete ctx;
        return RVSEG_ERR_HIP;
    }
    uint16_t gamma[256], cbrt_tab[3072];
    build_lab_tables(gamma, cbrt_tab, ctx->lab.coeffs);
    rvseg_status st;
    if ((st = dev_alloc(ctx, ctx->lab.gamma, sizeof(gamma))) != RVSEG_OK ||
        (st = dev_alloc(ctx, ctx->lab.cbrt, sizeof(cbrt_tab))) != RVSEG_OK ||
        !hip_ok(ctx, hipMemcpy(ctx->lab.gamma.p, gamma, sizeof(gamma), hipMemcpyHostToDevice), "upload gamma") ||
        !hip_ok(ctx, hipMemcpy(ctx->lab.cbrt.p, cbrt_tab, sizeof(cbrt_tab), hipMemcpyHostToDevice), "upload cbrt")) {
        fail(RVSEG_ERR_HIP, ctx->err);
        rvseg_destroy(ctx);
        return RVSEG_ERR_HIP;
    }
    *out = ctx;
    return RVSEG_OK;
}

void rvseg_pipeline_destroy(rvseg_ctx* ctx);  // rvseg_pipeline.hip
void rvseg_comm_destroy(rvseg_ctx* ctx);      // rvseg_comm.cpp

void rvseg_destroy(rvseg_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->params.device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    rvseg_comm_destroy(ctx);
    rvseg_pipeline_destroy(ctx);
    dev_free(ctx->forest.nodes);
    dev_free(ctx->forest.nodes8);
    dev_free(ctx->forest.roots);
    dev_free(ctx->forest.hist);
    dev_free(ctx->lab.gamma);
    dev_free(ctx->lab.cbrt);
    for (auto& b : ctx->pool) dev_free(b);
    for (auto ev : ctx->timer.events) (void)hipEventDestroy(ev);
    if (ctx->timer.side0) (void)hipEventDestroy(ctx->timer.side0);
    if (ctx->timer.side1) (void)hipEventDestroy(ctx->timer.side1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int32_t rvseg_feature_length(const rvseg_ctx* ctx) { return ctx ? ctx->feature_length : 0; }

rvseg_status rvseg_forest_load_mem(rvseg_ctx* ctx, const void* buf, size_t size) {
    if (!ctx) return RVSEG_ERR_INVALID_ARG;
    if (!buf) { ctx->err = "null forest buffer"; return RVSEG_ERR_INVALID_ARG; }
    RV_HIP(ctx, hipSetDevice(ctx->params.device));
    ForestModel m;
    std::string err;
    if (!parse_forest(buf, size, ctx->feature_length, m, err)) {
        ctx->err = err;
        return parse_error_status(err);
    }
    // the device copy of the previous model may still be read by work the caller enqueued on its own
    // stream (rvseg_segment_frames_device returns without synchronising)
    RV_HIP(ctx, hipDeviceSynchronize());
    ctx->host_forest = std::move(m);
    ctx->forest_loaded = false;
    return upload_forest(ctx);
}

rvseg_status rvseg_forest_check(const void* buf, size_t size, int32_t feature_length, int32_t* n_trees,
                                int32_t* n_nodes_total, int32_t* max_depth, char* err_out, size_t err_cap) {
    ForestModel m;
    std::string err;
    rvseg_status st = RVSEG_OK;
    if (!parse_forest(buf, size, feature_length > 0 ? feature_length : 0x7fffffff, m, err)) {
        st = parse_error_status(err);
    } else {
        int sum_multi = 0;
        for (int c : m.layer_classes) sum_multi += c;
        if (m.single_classes > kMaxClasses || sum_multi > kMaxClasses || m.layer_classes.size() > RVSEG_MAX_LAYERS) {
            err = "more than 64 classes or 8 layers are not supported";
            st = RVSEG_ERR_FORMAT;
        }
    }
    if (err_out && err_cap) std::snprintf(err_out, err_cap, "%s", err.c_str());
    if (st != RVSEG_OK) return st;
    if (n_trees) *n_trees = m.n_trees;
    if (n_nodes_total) *n_nodes_total = (int32_t)m.nodes.size();
    if (max_depth) *max_depth = m.max_depth;
    return RVSEG_OK;
}

rvseg_status rvseg_forest_rewrite(const void* buf, size_t size, void* out, size_t out_cap, size_t* size_out) {
    ForestModel m;
    std::string err;
    if (!parse_forest(buf, size, 0x7fffffff, m, err)) return parse_error_status(err);
    return copy_out(serialize_forest(m), out, out_cap, size_out);
}

rvseg_status rvseg_forest_write_mem(const rvseg_ctx* ctx, void* out, size_t out_cap, size_t* size_out) {
    if (!ctx) return RVSEG_ERR_INVALID_ARG;
    if (!ctx->forest_loaded) return RVSEG_ERR_NO_FOREST;
    return copy_out(serialize_forest(ctx->host_forest), out, out_cap, size_out);
}

rvseg_status rvseg_forest_write(const rvseg_ctx* ctx, const char* path) {
    if (!ctx || !path) return RVSEG_ERR_INVALID_ARG;
    if (!ctx->forest_loaded) return RVSEG_ERR_NO_FOREST;
    const std::vector<uint8_t> bytes = serialize_forest(ctx->host_forest);
    std::ofstream os(path, std::ios::binary);
    if (!os.is_open()) return RVSEG_ERR_IO;          // libf::Exception("Could not open file."), io.h:118-121
    os.write(reinterpret_cast<const char*>(bytes.data()), (std::streamsize)bytes.size());
    return os.good() ? RVSEG_OK : RVSEG_ERR_IO;
}

rvseg_status rvseg_forest_load(rvseg_ctx* ctx, const char* path) {
    if (!ctx) return RVSEG_ERR_INVALID_ARG;
    if (!path) { ctx->err = "null path"; return RVSEG_ERR_INVALID_ARG; }
    std::ifstream is(path, std::ios::binary);
    if (!is.is_open()) { ctx->err = std::string("Could not open file. (") + path + ")"; return RVSEG_ERR_IO; }
    std::vector<char> data((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
    return rvseg_forest_load_mem(ctx, data.data(), data.size());
}

rvseg_status rvseg_forest_info(const rvseg_ctx* ctx, int32_t* n_trees, int32_t* n_nodes_total,
                               int32_t* max_depth, int32_t* n_layers,
                               int32_t class_counts[RVSEG_MAX_LAYERS]) {
    if (!ctx) return RVSEG_ERR_INVALID_ARG;
    if (!ctx->forest_loaded) return RVSEG_ERR_NO_FOREST;
    if (n_trees) *n_trees = ctx->forest.n_trees;
    if (n_nodes_total) *n_nodes_total = ctx->forest.n_nodes;
    if (max_depth) *max_depth = ctx->forest.max_depth;
    if (n_layers) *n_layers = ctx->forest.n_layers;
    if (class_counts)
        for (int l = 0; l < RVSEG_MAX_LAYERS; l++) class_counts[l] = l < ctx->forest.n_layers ? ctx->forest.class_counts[l] : 0;
    return RVSEG_OK;
}

rvseg_status rvseg_forest_eval(rvseg_ctx* ctx, const float* X, int32_t P, int32_t D, float* out) {
    if (!ctx) return RVSEG_ERR_INVALID_ARG;
    if (!ctx->forest_loaded) { ctx->err = "no forest loaded"; return RVSEG_ERR_NO_FOREST; }
    if (P < 0 || !out || (!X && P > 0)) { ctx->err = "bad arguments"; return RVSEG_ERR_INVALID_ARG; }
    if (D != ctx->feature_length) { ctx->err = "D does not match the configured feature length"; return RVSEG_ERR_INVALID_ARG; }
    if (P == 0) return RVSEG_OK;
    RV_HIP(ctx, hipSetDevice(ctx->params.device));
    DevBuf dX, dO;
    rvseg_status st;
    const size_t S = (size_t)ctx->forest.sum_classes;
    if ((st = dev_alloc(ctx, dX, (size_t)P * D * sizeof(float))) != RVSEG_OK) return st;
    if ((st = dev_alloc(ctx, dO, (size_t)P * S * sizeof(float))) != RVSEG_OK) { dev_free(dX); return st; }
    rvseg_status rc = RVSEG_OK;
    do {
        if (!hip_ok(ctx, hipMemcpyAsync(dX.p, X, (size_t)P * D * sizeof(float), hipMemcpyHostToDevice, ctx->stream), "H2D X")) { rc = RVSEG_ERR_HIP; break; }
        launch_forest_eval(ctx->forest, dX.as<float>(), P, D, dO.as<float>(), ctx->stream);
        if (launch_error_take(ctx) != RVSEG_OK) { rc = RVSEG_ERR_HIP; break; }
        if (!hip_ok(ctx, hipMemcpyAsync(out, dO.p, (size_t)P * S * sizeof(float), hipMemcpyDeviceToHost, ctx->stream), "D2H out")) { rc = RVSEG_ERR_HIP; break; }
        if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "sync")) { rc = RVSEG_ERR_HIP; break; }
    } while (0);
    dev_free(dX);
    dev_free(dO);
    return rc;
}

// DenseCRF2D::addPairwiseGaussian / addPairwiseBilateral (densecrf.cpp:61-81): the feature matrices they
// build, point-major (N x d == the column-major d x N Eigen matrix).  int / float and uchar / float
// divisions in fp32, exactly the reference's expressions.
rvseg_status rvseg_crf_features_gaussian(int32_t W, int32_t H, float sx, float sy, float* out) {
    if (W <= 0 || H <= 0 || !out) return RVSEG_ERR_INVALID_ARG;
    for (int j = 0; j < H; j++)
        for (int i = 0; i < W; i++) {
            float* f = out + ((size_t)j * W + i) * 2;
            f[0] = i / sx;
            f[1] = j / sy;
        }
    return RVSEG_OK;
}

rvseg_status rvseg_crf_features_bilateral(int32_t W, int32_t H, float sx, float sy, float sr, float sg, float sb,
                                          const uint8_t* im, float* out) {
    if (W <= 0 || H <= 0 || !im || !out) return RVSEG_ERR_INVALID_ARG;
    for (int j = 0; j < H; j++)
        for (int i = 0; i < W; i++) {
            float* f = out + ((size_t)j * W + i) * 5;
            const uint8_t* px = im + ((size_t)i + (size_t)j * W) * 3;
            f[0] = i / sx;
            f[1] = j / sy;
            f[2] = px[0] / sr;
            f[3] = px[1] / sg;
            f[4] = px[2] / sb;
        }
    return RVSEG_OK;
}

int32_t rvseg_last_timing(const rvseg_ctx* ctx, char* names_out, size_t names_cap, float* ms_out, int32_t max_stages) {
    if (!ctx) return 0;
    const auto& t = ctx->timer;
    // events[i] starts stage names[i]; the next event ends it.  Same-named stages accumulate.
    std::vector<std::string> names;
    std::vector<float> ms;
    if (t.used >= 2) {
        (void)hipEventSynchronize(t.events[t.used - 1]);
        for (size_t i = 0; i + 1 < t.used; i++) {
            if (t.names[i] == "end") continue;
            float v = 0.f;
            if (hipEventElapsedTime(&v, t.events[i], t.events[i + 1]) != hipSuccess) v = 0.f;
            size_t k = 0;
            for (; k < names.size(); k++) if (names[k] == t.names[i]) break;
            if (k == names.size()) { names.push_back(t.names[i]); ms.push_back(0.f); }
            ms[k] += v;
        }
    }
    if (t.side_used && t.side0 && t.side1) {   // the overlapped stage has its own pair of events
        (void)hipEventSynchronize(t.side1);
        float v = 0.f;
        if (hipEventElapsedTime(&v, t.side0, t.side1) != hipSuccess) v = 0.f;
        names.push_back(t.side_name);
        ms.push_back(v);
    }
    std::string joined;
    for (size_t i = 0; i < names.size(); i++) { if (i) joined += ';'; joined += names[i]; }
    if (names_out && names_cap) std::snprintf(names_out, names_cap, "%s", joined.c_str());
    const int n = (int)ms.size();
    for (int i = 0; i < n && i < max_stages; i++) if (ms_out) ms_out[i] = ms[i];
    return n;
}

}  // extern "C"
