// Frame pipeline: host orchestration of the per-frame hot path over batches of key frames
// (replaces the body of Segmenter::processFramesFromQueueInternalRF, src/segmenter.cpp:351-431,
// and -- with use_dense_crf -- the DenseCRF call shape of src/segmenter.cpp:639-657 per frame).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "rvseg_internal.h"
#include "rvseg_kernels.h"
#include "rvseg_pipeline.h"

namespace rvseg {

// ---- cv::resize coefficient rule (OpenCV 2.4 imgwarp.cpp), identical to the oracle's statement
static void resize_coeffs(int ssize, int dsize, bool clamp_weights, std::vector<int>& ofs,
                          std::vector<float>& w0, std::vector<float>& w1) {
    ofs.resize(dsize); w0.resize(dsize); w1.resize(dsize);
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= s;
        if (clamp_weights) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
        ofs[d] = s; w0[d] = 1.f - f; w1[d] = f;
    }
}

static rvseg_status upload(rvseg_ctx* ctx, DevBuf& b, const void* src, size_t bytes) {
    rvseg_status st = dev_alloc(ctx, b, bytes);
    if (st != RVSEG_OK) return st;
    RV_HIP(ctx, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return RVSEG_OK;
}

rvseg_status pipeline_init(rvseg_ctx* ctx) {
    if (ctx->impl && !reinterpret_cast<Pipeline*>(ctx->impl)->bare) return RVSEG_OK;
    const rvseg_params& p = ctx->params;
    if (p.width % p.stride != 0 || p.height % p.stride != 0) {
        // the reference would scatter outside its low-res image (segmenter.cpp:357,370)
        ctx->err = "width and height must be multiples of rf_prediction_stride";
        return RVSEG_ERR_INVALID_ARG;
    }
    if (p.patch_size_reduce > 16) { ctx->err = "patch_size_reduce > 16 is not supported"; return RVSEG_ERR_INVALID_ARG; }
    Pipeline* im = ctx->impl ? reinterpret_cast<Pipeline*>(ctx->impl) : new Pipeline();
    ctx->impl = reinterpret_cast<rvseg_ctx::Impl*>(im);
    im->bare = false;
    FrameGeom& g = im->geom;
    g.W = p.width; g.H = p.height; g.stride = p.stride;
    g.lw = p.width / p.stride; g.lh = p.height / p.stride;
    g.depth_min = p.depth_min; g.depth_max = p.depth_max;
    g.dmin_mm = (float)(p.depth_min * 1000.0); g.dmax_mm = (float)(p.depth_max * 1000.0);  // feature_extractor.h:43-44
    g.patch_size = p.patch_size; g.r = p.patch_size_reduce;
    g.n_patch = p.feature_color_patch ? g.r * g.r * 3 : 0;
    int pos = g.n_patch;
    g.pos_depth = p.feature_depth ? pos++ : -1;
    g.pos_height = p.feature_height ? pos++ : -1;
    g.pos_normal = p.feature_normal ? pos++ : -1;
    g.D = pos;
    g.fill = p.fill_value;
    g.rt_rows = 0;

    rvseg_status st;
    // 8-bit patch resize tables, one row per ROI half size
    if (p.feature_color_patch) {
        const int max_half = (int)(p.patch_size / (2.0 * p.depth_min));
        std::vector<ResizeRow> rows((size_t)max_half + 1);
        std::vector<int> ofs; std::vector<float> w0, w1;
        for (int half = 0; half <= max_half; half++) {
            const int size = 2 * half + 1;
            ResizeRow& rr = rows[half];
            std::memset(&rr, 0, sizeof(rr));
            resize_coeffs(size, g.r, true, ofs, w0, w1);
            for (int d = 0; d < g.r; d++) {
                rr.x[d].ofs = (int16_t)ofs[d];
                rr.x[d].w0 = (int16_t)std::lrintf(w0[d] * 2048.f);  // saturate_cast<short>: round half to even
                rr.x[d].w1 = (int16_t)std::lrintf(w1[d] * 2048.f);
                rr.x[d].ofs1 = (int16_t)(ofs[d] + 1 < size ? ofs[d] + 1 : ofs[d]);   // second tap, kept inside the ROI
            }
            resize_coeffs(size, g.r, false, ofs, w0, w1);
            for (int d = 0; d < g.r; d++) {
                // rows are clipped to the ROI (the weights are not): both clipped taps are part of the record
                const int s0 = ofs[d] < 0 ? 0 : (ofs[d] >= size ? size - 1 : ofs[d]);
                const int s1 = ofs[d] + 1 < 0 ? 0 : (ofs[d] + 1 >= size ? size - 1 : ofs[d] + 1);
                rr.y[d].ofs = (int16_t)s0;
                rr.y[d].w0 = (int16_t)std::lrintf(w0[d] * 2048.f);
                rr.y[d].w1 = (int16_t)std::lrintf(w1[d] * 2048.f);
                rr.y[d].ofs1 = (int16_t)s1;
            }
        }
        if ((st = upload(ctx, im->resize_rows, rows.data(), rows.size() * sizeof(ResizeRow))) != RVSEG_OK) return st;
        g.rt_rows = (int)rows.size();
    }
    // float up-sampling tables
    {
        std::vector<int> ofs; std::vector<float> w0, w1;
        resize_coeffs(g.lw, g.W, true, ofs, w0, w1);
        if ((st = upload(ctx, im->up.xofs, ofs.data(), ofs.size() * 4)) != RVSEG_OK) return st;
        if ((st = upload(ctx, im->up.ax0, w0.data(), w0.size() * 4)) != RVSEG_OK) return st;
        if ((st = upload(ctx, im->up.ax1, w1.data(), w1.size() * 4)) != RVSEG_OK) return st;
        resize_coeffs(g.lh, g.H, false, ofs, w0, w1);
        if ((st = upload(ctx, im->up.yofs, ofs.data(), ofs.size() * 4)) != RVSEG_OK) return st;
        if ((st = upload(ctx, im->up.ay0, w0.data(), w0.size() * 4)) != RVSEG_OK) return st;
        if ((st = upload(ctx, im->up.ay1, w1.data(), w1.size() * 4)) != RVSEG_OK) return st;
    }
    return RVSEG_OK;
}

static void pipeline_free(Pipeline* im) {
    DevBuf* all[] = {&im->resize_rows, &im->up.xofs, &im->up.ax0, &im->up.ax1, &im->up.yofs, &im->up.ay0, &im->up.ay1,
                     &im->calibA, &im->lab, &im->lab2, &im->cloud, &im->rect, &im->nfeat, &im->low, &im->post, &im->marg,
                     &im->labels, &im->in_rgb, &im->in_depth, &im->dump, &im->valid, &im->change};
    for (DevBuf* b : all) dev_free(*b);
    for (int i = 0; i < Pipeline::CALIB_RING; i++) {
        if (im->calib_ev[i]) { (void)hipEventSynchronize(im->calib_ev[i]); (void)hipEventDestroy(im->calib_ev[i]); }
        if (im->h_calibA[i]) (void)hipHostFree(im->h_calibA[i]);
    }
    {
        HostStage& hs = im->stage;
        for (int i = 0; i < HostStage::SLOTS; i++) {
            void* hp[] = {hs.h_rgb[i], hs.h_depth[i], hs.h_post[i], hs.h_marg[i], hs.h_lab[i]};
            for (void* p : hp) if (p) (void)hipHostFree(p);
            DevBuf* db[] = {&hs.d_rgb[i], &hs.d_depth[i], &hs.d_post[i], &hs.d_marg[i], &hs.d_lab[i]};
            for (DevBuf* b : db) dev_free(*b);
            hipEvent_t ev[] = {hs.ev_in[i], hs.ev_done[i], hs.ev_out[i]};
            for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
        }
        if (hs.s_in) (void)hipStreamDestroy(hs.s_in);
        if (hs.s_out) (void)hipStreamDestroy(hs.s_out);
    }
    crf_state_free(im);
    fusion_state_free(im);
    if (im->side) (void)hipStreamDestroy(im->side);
    if (im->ev_fork) (void)hipEventDestroy(im->ev_fork);
    if (im->ev_entry) (void)hipEventDestroy(im->ev_entry);
    if (im->ev_join) (void)hipEventDestroy(im->ev_join);
}

// A = R*Kinv, Eigen fixed 3x3 product accumulated left to right (feature_extractor.h:223)
static void calib_to_A(const float* calib, float* out12) {
    const float *Kinv = calib, *R = calib + 9, *t = calib + 18;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            out12[i * 3 + j] = (R[i * 3 + 0] * Kinv[0 * 3 + j] + R[i * 3 + 1] * Kinv[1 * 3 + j]) + R[i * 3 + 2] * Kinv[2 * 3 + j];
    out12[9] = t[0]; out12[10] = t[1]; out12[11] = t[2];
}

rvseg_status upload_calib(rvseg_ctx* ctx, Pipeline* im, const float* calib, int n, hipStream_t s) {
    const int slot = im->calib_next;
    im->calib_next = (slot + 1) % Pipeline::CALIB_RING;
    if (!im->calib_ev[slot]) RV_HIP(ctx, hipEventCreateWithFlags(&im->calib_ev[slot], hipEventDisableTiming));
    // the copy that last read this slot must have run before the host rewrites (or frees) it
    if (im->calib_ev_live[slot]) { RV_HIP(ctx, hipEventSynchronize(im->calib_ev[slot])); im->calib_ev_live[slot] = false; }
    const size_t bytes = (size_t)n * 12 * sizeof(float);
    if (bytes > im->h_calibA_bytes[slot]) {
        if (im->h_calibA[slot]) (void)hipHostFree(im->h_calibA[slot]);
        im->h_calibA[slot] = nullptr;
        im->h_calibA_bytes[slot] = 0;
        RV_HIP(ctx, hipHostMalloc((void**)&im->h_calibA[slot], bytes, hipHostMallocDefault));
        im->h_calibA_bytes[slot] = bytes;
    }
    float* h = im->h_calibA[slot];
    for (int i = 0; i < n; i++) calib_to_A(calib + (size_t)i * 21, h + (size_t)i * 12);
    rvseg_status st = dev_reserve(ctx, im->calibA, bytes);   // (a reallocation frees with hipFree, which waits for the device)
    if (st != RVSEG_OK) return st;
    RV_HIP(ctx, hipMemcpyAsync(im->calibA.p, h, bytes, hipMemcpyHostToDevice, s));
    RV_HIP(ctx, hipEventRecord(im->calib_ev[slot], s));
    im->calib_ev_live[slot] = true;
    return RVSEG_OK;
}

// ---- stage timing ----------------------------------------------------------------------------
void timer_reset(rvseg_ctx* ctx) {
    ctx->timer.names.clear();
    ctx->timer.ms.clear();
    ctx->timer.used = 0;
    ctx->timer.side_used = false;
}

void timer_mark(rvseg_ctx* ctx, const char* name, hipStream_t s) {
    StageTimer& t = ctx->timer;
    if (t.used >= t.events.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        t.events.push_back(e);
    }
    (void)hipEventRecord(t.events[t.used++], s);
    t.names.push_back(name);
}

// ---- the frame path for one chunk of at most max_batch frames ----------------------------------
static rvseg_status run_chunk(rvseg_ctx* ctx, Pipeline* im, int n, const uint8_t* d_rgb, const uint16_t* d_depth,
                              const float* d_calibA, float* d_post, float* d_marg, int8_t* d_labels, hipStream_t s) {
    const FrameGeom& g = im->geom;
    const rvseg_params& p = ctx->params;
    const DeviceForest& f = ctx->forest;
    const size_t npix = (size_t)g.W * g.H;
    rvseg_status st;
    const bool need_cloud = p.feature_height || p.feature_normal || p.use_dense_crf;
    if (p.feature_color_patch && (st = dev_reserve(ctx, im->lab, npix * 4 * n)) != RVSEG_OK) return st;
    const bool use_lab2 = p.feature_color_patch && rf_frames_wants_lab2(f);
    if (use_lab2 && (st = dev_reserve(ctx, im->lab2, npix * 8 * n + 16)) != RVSEG_OK) return st;
    if (need_cloud && (st = dev_reserve(ctx, im->cloud, npix * 16 * n)) != RVSEG_OK) return st;
    if (p.feature_normal) {
        if ((st = dev_reserve(ctx, im->rect, npix * n)) != RVSEG_OK) return st;
        if ((st = dev_reserve(ctx, im->change, npix * n)) != RVSEG_OK) return st;
        if ((st = dev_reserve(ctx, im->nfeat, (size_t)g.lw * g.lh * 4 * n)) != RVSEG_OK) return st;
    }
    if ((st = dev_reserve(ctx, im->low, (size_t)g.lw * g.lh * f.sum_classes * 4 * n)) != RVSEG_OK) return st;
    float* post = d_post;
    if (!post) {
        if ((st = dev_reserve(ctx, im->post, npix * f.sum_classes * 4 * n)) != RVSEG_OK) return st;
        post = im->post.as<float>();
    }
    const bool fork_build = p.use_dense_crf && ctx->sched.overlap_build;
    if (fork_build) {
        // the build stream first: table and list heads of the new lattice are cleared while prep_kernel runs.  The
        // lattice's previous user (the last chunk's mean field, a cloud CRF on this context) ran on the caller's
        // stream: ev_entry, recorded there before anything of this chunk, orders the memsets behind it.
        if (!im->side) {
            int prio_lo = 0, prio_hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
            // priority of the build stream.  Round 1 (build = the longer branch) measured the highest priority ahead,
            // 13.85 vs 13.98 ms per step; since the feature branch is the longer one (5.2 vs 3.6 ms) the lowest is, by
            // a little: 13.31 / 13.33 vs 13.37 / 13.42 ms (rvseg_schedule.build_priority_high restores the old choice;
            // it is read when the stream is created, i.e. before the first frame call of the context)
            RV_HIP(ctx, hipStreamCreateWithPriority(&im->side, hipStreamNonBlocking, ctx->sched.build_priority_high ? prio_hi : prio_lo));
            RV_HIP(ctx, hipEventCreateWithFlags(&im->ev_fork, hipEventDisableTiming));
            RV_HIP(ctx, hipEventCreateWithFlags(&im->ev_join, hipEventDisableTiming));
            RV_HIP(ctx, hipEventCreateWithFlags(&im->ev_entry, hipEventDisableTiming));
            RV_HIP(ctx, hipEventCreate(&ctx->timer.side0));
            RV_HIP(ctx, hipEventCreate(&ctx->timer.side1));
        }
        RV_HIP(ctx, hipEventRecord(im->ev_entry, s));
        RV_HIP(ctx, hipStreamWaitEvent(im->side, im->ev_entry, 0));
        if ((st = crf_frames_build_begin(ctx, im, n, im->side)) != RVSEG_OK) return st;
    }
    timer_mark(ctx, "prep", s);
    launch_prep(g, ctx->lab, d_rgb, d_depth, d_calibA, p.feature_color_patch ? im->lab.as<uint32_t>() : nullptr,
                need_cloud ? im->cloud.as<float4>() : nullptr, p.feature_normal ? im->change.as<uint8_t>() : nullptr, n, s,
                use_lab2 ? im->lab2.as<uint2>() : nullptr);
    bool forked = false;
    if (fork_build) {
        // fork: the lattice build runs on the side stream while this stream extracts features and walks the forest
        RV_HIP(ctx, hipEventRecord(im->ev_fork, s));
        RV_HIP(ctx, hipStreamWaitEvent(im->side, im->ev_fork, 0));
        RV_HIP(ctx, hipEventRecord(ctx->timer.side0, im->side));
        if ((st = crf_frames_build(ctx, im, n, d_rgb, im->side)) != RVSEG_OK) { (void)hipStreamSynchronize(im->side); return st; }
        RV_HIP(ctx, hipEventRecord(ctx->timer.side1, im->side));
        RV_HIP(ctx, hipEventRecord(im->ev_join, im->side));
        ctx->timer.side_name = "lattice_build";
        ctx->timer.side_used = true;
        forked = true;
    }
    if (p.feature_normal) {
        timer_mark(ctx, "window_map", s);
        launch_window_map(g, im->cloud.as<float4>(), im->change.as<uint8_t>(), im->rect.as<uint8_t>(), n, s);
        timer_mark(ctx, "normal_feature", s);
        launch_normal_feature(g, im->cloud.as<float4>(), im->rect.as<uint8_t>(), im->nfeat.as<float>(), n, s);
    }
    timer_mark(ctx, "rf_frames", s);
    launch_rf_frames(g, f, im->resize_rows.as<ResizeRow>(), im->lab.as<uint32_t>(), d_depth, im->cloud.as<float4>(),
                     im->nfeat.as<float>(), im->low.as<float>(), nullptr, nullptr, n, s, use_lab2 ? im->lab2.as<uint2>() : nullptr);
    timer_mark(ctx, "upsample_pack", s);
    launch_upsample_pack(g, f, im->up, im->low.as<float>(), post, n, s);
    if (p.use_dense_crf) {
        if (forked) {
            RV_HIP(ctx, hipStreamWaitEvent(s, im->ev_join, 0));   // join
        } else {
            timer_mark(ctx, "lattice_build", s);
            if ((st = crf_frames_build(ctx, im, n, d_rgb, s)) != RVSEG_OK) return st;
        }
        st = crf_frames_infer(ctx, im, n, post, d_marg, d_labels, s);
        if (st != RVSEG_OK) return st;
    } else if (d_labels) {
        timer_mark(ctx, "labels", s);
        launch_labels_frames(post, n, (int)npix, f, p.label_mode, p.unknown_label, d_labels, s);
    }
    timer_mark(ctx, "end", s);
    RV_LAUNCH_OK(ctx);
    return RVSEG_OK;
}

}  // namespace rvseg

using namespace rvseg;

extern "C" {

void rvseg_pipeline_destroy(rvseg_ctx* ctx) {
    if (!ctx || !ctx->impl) return;
    Pipeline* im = reinterpret_cast<Pipeline*>(ctx->impl);
    pipeline_free(im);
    delete im;
    ctx->impl = nullptr;
}

rvseg_status rvseg_segment_frames_device(rvseg_ctx* ctx, int32_t n_frames, const uint8_t* d_rgb,
                                         const uint16_t* d_depth_mm, const float* calib, float* d_posteriors_out,
                                         float* d_marginals_out, int8_t* d_labels_out, void* hip_stream) {
    if (!ctx) return RVSEG_ERR_INVALID_ARG;
    if (!ctx->forest_loaded) { ctx->err = "no forest loaded"; return RVSEG_ERR_NO_FOREST; }
    if (n_frames < 0 || (n_frames > 0 && (!d_rgb || !d_depth_mm || !calib))) { ctx->err = "bad arguments"; return RVSEG_ERR_INVALID_ARG; }
    if (n_frames == 0) return RVSEG_OK;
    RV_HIP(ctx, hipSetDevice(ctx->params.device));
    rvseg_status st = pipeline_init(ctx);
    if (st != RVSEG_OK) return st;
    Pipeline* im = reinterpret_cast<Pipeline*>(ctx->impl);
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
    timer_reset(ctx);
    if ((st = upload_calib(ctx, im, calib, n_frames, s)) != RVSEG_OK) return st;
    const FrameGeom& g = im->geom;
    const size_t npix = (size_t)g.W * g.H;
    const size_t S = (size_t)ctx->forest.sum_classes, L = (size_t)ctx->forest.n_layers;
    for (int start = 0; start < n_frames; start += ctx->params.max_batch) {
        const int n = std::min(ctx->params.max_batch, n_frames - start);
        st = run_chunk(ctx, im, n, d_rgb + (size_t)start * npix * 3, d_depth_mm + (size_t)start * npix,
                       im->calibA.as<float>() + (size_t)start * 12,
                       d_posteriors_out ? d_posteriors_out + (size_t)start * npix * S : nullptr,
                       d_marginals_out ? d_marginals_out + (size_t)start * npix * S : nullptr,
                       d_labels_out ? d_labels_out + (size_t)start * npix * L : nullptr, s);
        if (st != RVSEG_OK) return st;
    }
    return RVSEG_OK;
}

// ---- host-buffer entry: pinned staging ring, copies of neighbouring chunks under the compute ----
namespace {

// pageable <-> pinned copies; large ones are split over a few host threads (one memcpy stream saturates
// well below the memory bandwidth of the host)
void host_copy(void* dst, const void* src, size_t bytes) {
    const size_t min_part = (size_t)4 << 20;
    unsigned nt = (unsigned)std::min<size_t>(8, bytes / min_part);
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw && nt > hw) nt = hw;
    if (nt <= 1) { std::memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    const size_t part = (bytes / nt + 4095) & ~(size_t)4095;
    for (unsigned t = 0; t < nt; t++) {
        const size_t off = (size_t)t * part;
        if (off >= bytes) break;
        const size_t len = std::min(part, bytes - off);
        th.emplace_back([=] { std::memcpy((char*)dst + off, (const char*)src + off, len); });
    }
    for (auto& x : th) x.join();
}

rvseg_status pinned_reserve(rvseg_ctx* ctx, void*& p, size_t& cap, size_t bytes) {
    if (cap >= bytes && p) return RVSEG_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    RV_HIP(ctx, hipHostMalloc(&p, bytes, hipHostMallocDefault));
    cap = bytes;
    return RVSEG_OK;
}

// is p page-locked host memory (hipHostMalloc / hipHostRegister, e.g. through rvseg_host_register)?  Then the DMA engines
// reach it directly and the pinned staging copy is skipped.
bool is_pinned_host(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // pageable memory is unknown to the runtime
    return at.type == hipMemoryTypeHost;
}

rvseg_status stage_init(rvseg_ctx* ctx, HostStage& hs) {
    if (hs.ready) return RVSEG_OK;
    RV_HIP(ctx, hipStreamCreateWithFlags(&hs.s_in, hipStreamNonBlocking));
    RV_HIP(ctx, hipStreamCreateWithFlags(&hs.s_out, hipStreamNonBlocking));
    for (int i = 0; i < HostStage::SLOTS; i++) {
        RV_HIP(ctx, hipEventCreateWithFlags(&hs.ev_in[i], hipEventDisableTiming));
        RV_HIP(ctx, hipEventCreateWithFlags(&hs.ev_done[i], hipEventDisableTiming));
        RV_HIP(ctx, hipEventCreateWithFlags(&hs.ev_out[i], hipEventDisableTiming));
    }
    hs.ready = true;
    return RVSEG_OK;
}

}  // namespace

rvseg_status rvseg_segment_frames(rvseg_ctx* ctx, int32_t n_frames, const uint8_t* rgb, const uint16_t* depth_mm,
                                  const float* calib, float* posteriors_out, float* marginals_out, int8_t* labels_out) {
    if (!ctx) return RVSEG_ERR_INVALID_ARG;
    if (!ctx->forest_loaded) { ctx->err = "no forest loaded"; return RVSEG_ERR_NO_FOREST; }
    if (n_frames < 0 || (n_frames > 0 && (!rgb || !depth_mm || !calib))) { ctx->err = "bad arguments"; return RVSEG_ERR_INVALID_ARG; }
    if (n_frames == 0) return RVSEG_OK;
    RV_HIP(ctx, hipSetDevice(ctx->params.device));
    rvseg_status st = pipeline_init(ctx);
    if (st != RVSEG_OK) return st;
    Pipeline* im = reinterpret_cast<Pipeline*>(ctx->impl);
    HostStage& hs = im->stage;
    if ((st = stage_init(ctx, hs)) != RVSEG_OK) return st;
    const FrameGeom& g = im->geom;
    const size_t npix = (size_t)g.W * g.H;
    const size_t S = (size_t)ctx->forest.sum_classes, L = (size_t)ctx->forest.n_layers;
    const bool want_marg = marginals_out && ctx->params.use_dense_crf;
    // page-locked caller buffers are read / written by the copy engines directly (no staging copy on the host):
    // the 11 MB of marginals per frame otherwise cross the host memory twice, which bounds the call at ~10 GB/s
    const bool in_pinned = is_pinned_host(rgb) && is_pinned_host(depth_mm);
    const bool post_pinned = is_pinned_host(posteriors_out), marg_pinned = want_marg && is_pinned_host(marginals_out),
               lab_pinned = is_pinned_host(labels_out);
    hipStream_t s = ctx->stream;
    // chunk size: at most max_batch, and small enough that a call has a few chunks to overlap
    const int chunk = std::max(1, std::min(ctx->params.max_batch, std::max(8, (n_frames + 3) / 4)));
    const int n_chunks = (n_frames + chunk - 1) / chunk;
    auto chunk_n = [&](int c) { return std::min(chunk, n_frames - c * chunk); };

    // copies the outputs of chunk c from its pinned slot into the caller's buffers (after its D2H has run)
    auto retire = [&](int c) -> rvseg_status {
        const int slot = c % HostStage::SLOTS, n = chunk_n(c);
        const size_t start = (size_t)c * chunk;
        RV_HIP(ctx, hipEventSynchronize(hs.ev_out[slot]));
        if (posteriors_out && !post_pinned) host_copy(posteriors_out + start * npix * S, hs.h_post[slot], npix * S * 4 * n);
        if (want_marg && !marg_pinned) host_copy(marginals_out + start * npix * S, hs.h_marg[slot], npix * S * 4 * n);
        if (labels_out && !lab_pinned) host_copy(labels_out + start * npix * L, hs.h_lab[slot], npix * L * n);
        return RVSEG_OK;
    };
    auto drain = [&]() { (void)hipStreamSynchronize(hs.s_in); (void)hipStreamSynchronize(s); (void)hipStreamSynchronize(hs.s_out); };

    int retries = 0;
    for (int c = 0; c <= n_chunks; c++) {
        if (c == n_chunks) {
            // all chunks are enqueued: the status of the last build is the only one nobody has looked at yet
            RV_HIP(ctx, hipStreamSynchronize(s));
            st = ctx->params.use_dense_crf ? crf_frames_status(ctx, im, true) : RVSEG_OK;
            if (st == RVSEG_ERR_CAPACITY && retries++ < 16) { drain(); c = n_chunks - 2; continue; }   // redo the last chunk
            if (st != RVSEG_OK) { drain(); return st; }
            if ((st = retire(n_chunks - 1)) != RVSEG_OK) { drain(); return st; }
            break;
        }
        const int slot = c % HostStage::SLOTS, n = chunk_n(c);
        const size_t start = (size_t)c * chunk;
        // staging + device buffers of this slot (grow only; the slot's previous chunk c - 2 has been retired)
        if ((!in_pinned && ((st = pinned_reserve(ctx, hs.h_rgb[slot], hs.c_rgb[slot], npix * 3 * n)) != RVSEG_OK ||
                            (st = pinned_reserve(ctx, hs.h_depth[slot], hs.c_depth[slot], npix * 2 * n)) != RVSEG_OK)) ||
            (st = dev_reserve(ctx, hs.d_rgb[slot], npix * 3 * n)) != RVSEG_OK ||
            (st = dev_reserve(ctx, hs.d_depth[slot], npix * 2 * n)) != RVSEG_OK) { drain(); return st; }
        const bool need_post_dev = posteriors_out != nullptr;
        if (need_post_dev && ((!post_pinned && (st = pinned_reserve(ctx, hs.h_post[slot], hs.c_post[slot], npix * S * 4 * n)) != RVSEG_OK) ||
                              (st = dev_reserve(ctx, hs.d_post[slot], npix * S * 4 * n)) != RVSEG_OK)) { drain(); return st; }
        if (want_marg && ((!marg_pinned && (st = pinned_reserve(ctx, hs.h_marg[slot], hs.c_marg[slot], npix * S * 4 * n)) != RVSEG_OK) ||
                          (st = dev_reserve(ctx, hs.d_marg[slot], npix * S * 4 * n)) != RVSEG_OK)) { drain(); return st; }
        if (labels_out && ((!lab_pinned && (st = pinned_reserve(ctx, hs.h_lab[slot], hs.c_lab[slot], npix * L * n)) != RVSEG_OK) ||
                           (st = dev_reserve(ctx, hs.d_lab[slot], npix * L * n)) != RVSEG_OK)) { drain(); return st; }
        // 1. caller's pageable buffers -> pinned (host threads; the GPU is busy with chunk c - 1 meanwhile)
        const void* src_rgb = rgb + start * npix * 3;
        const void* src_depth = depth_mm + start * npix;
        if (!in_pinned) {
            host_copy(hs.h_rgb[slot], src_rgb, npix * 3 * n);
            host_copy(hs.h_depth[slot], src_depth, npix * 2 * n);
            src_rgb = hs.h_rgb[slot];
            src_depth = hs.h_depth[slot];
        }
        // 2. H2D on the input stream, after the compute of chunk c - 2 (the last reader of these device buffers)
        if (c >= HostStage::SLOTS) RV_HIP(ctx, hipStreamWaitEvent(hs.s_in, hs.ev_done[slot], 0));
        RV_HIP(ctx, hipMemcpyAsync(hs.d_rgb[slot].p, src_rgb, npix * 3 * n, hipMemcpyHostToDevice, hs.s_in));
        RV_HIP(ctx, hipMemcpyAsync(hs.d_depth[slot].p, src_depth, npix * 2 * n, hipMemcpyHostToDevice, hs.s_in));
        RV_HIP(ctx, hipEventRecord(hs.ev_in[slot], hs.s_in));
        // 3. compute: after its inputs arrived and after the D2H of chunk c - 2 released the output buffers
        RV_HIP(ctx, hipStreamWaitEvent(s, hs.ev_in[slot], 0));
        if (c >= HostStage::SLOTS) RV_HIP(ctx, hipStreamWaitEvent(s, hs.ev_out[slot], 0));
        timer_reset(ctx);
        if ((st = upload_calib(ctx, im, calib + start * 21, n, s)) != RVSEG_OK) { drain(); return st; }
        st = run_chunk(ctx, im, n, hs.d_rgb[slot].as<uint8_t>(), hs.d_depth[slot].as<uint16_t>(), im->calibA.as<float>(),
                       need_post_dev ? hs.d_post[slot].as<float>() : nullptr, want_marg ? hs.d_marg[slot].as<float>() : nullptr,
                       labels_out ? hs.d_lab[slot].as<int8_t>() : nullptr, s);
        if (st == RVSEG_ERR_CAPACITY && retries++ < 16) {
            // the lattice build of chunk c - 1 overflowed its hash table (its status is read at the start of this
            // chunk's build): nothing of chunk c has been enqueued past the feature branch.  The context has raised
            // its capacity; take the two chunks again.
            drain();
            c = std::max(0, c - 1) - 1;
            continue;
        }
        if (st != RVSEG_OK) { drain(); return st; }
        RV_HIP(ctx, hipEventRecord(hs.ev_done[slot], s));
        // 4. D2H on the output stream
        RV_HIP(ctx, hipStreamWaitEvent(hs.s_out, hs.ev_done[slot], 0));
        if (posteriors_out) RV_HIP(ctx, hipMemcpyAsync(post_pinned ? (void*)(posteriors_out + start * npix * S) : hs.h_post[slot], hs.d_post[slot].p,
                                                       npix * S * 4 * n, hipMemcpyDeviceToHost, hs.s_out));
        if (want_marg) RV_HIP(ctx, hipMemcpyAsync(marg_pinned ? (void*)(marginals_out + start * npix * S) : hs.h_marg[slot], hs.d_marg[slot].p,
                                                  npix * S * 4 * n, hipMemcpyDeviceToHost, hs.s_out));
        if (labels_out) RV_HIP(ctx, hipMemcpyAsync(lab_pinned ? (void*)(labels_out + start * npix * L) : hs.h_lab[slot], hs.d_lab[slot].p,
                                                   npix * L * n, hipMemcpyDeviceToHost, hs.s_out));
        RV_HIP(ctx, hipEventRecord(hs.ev_out[slot], hs.s_out));
        // 5. hand chunk c - 1 to the caller while chunk c runs (its build status was checked by run_chunk above)
        if (c >= 1 && (st = retire(c - 1)) != RVSEG_OK) { drain(); return st; }
    }
    return RVSEG_OK;
}

rvseg_status rvseg_host_register(void* p, size_t bytes) {
    if (!p || !bytes) return RVSEG_ERR_INVALID_ARG;
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); return e == hipErrorHostMemoryAlreadyRegistered ? RVSEG_OK : RVSEG_ERR_HIP; }
    return RVSEG_OK;
}

rvseg_status rvseg_host_unregister(void* p) {
    if (!p) return RVSEG_ERR_INVALID_ARG;
    if (hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); return RVSEG_ERR_HIP; }
    return RVSEG_OK;
}

rvseg_status rvseg_poll_status(rvseg_ctx* ctx, int32_t wait) {
    if (!ctx) return RVSEG_ERR_INVALID_ARG;
    if (!ctx->impl) return RVSEG_OK;
    RV_HIP(ctx, hipSetDevice(ctx->params.device));
    Pipeline* im = reinterpret_cast<Pipeline*>(ctx->impl);
    const rvseg_status a = crf_frames_status(ctx, im, wait != 0);
    if (a != RVSEG_OK) return a;
    return fusion_status(ctx, im, wait != 0);
}

rvseg_status rvseg_extract_features(rvseg_ctx* ctx, const uint8_t* rgb, const uint16_t* depth_mm, const float* calib,
                                    float* feat_out, int32_t* x_v, int32_t* y_v, int32_t* n_points) {
    if (!ctx) return RVSEG_ERR_INVALID_ARG;
    if (!rgb || !depth_mm || !calib || !feat_out || !x_v || !y_v || !n_points) { ctx->err = "null argument"; return RVSEG_ERR_INVALID_ARG; }
    RV_HIP(ctx, hipSetDevice(ctx->params.device));
    rvseg_status st = pipeline_init(ctx);
    if (st != RVSEG_OK) return st;
    Pipeline* im = reinterpret_cast<Pipeline*>(ctx->impl);
    const FrameGeom& g = im->geom;
    const rvseg_params& p = ctx->params;
    const size_t npix = (size_t)g.W * g.H;
    const int P = g.lw * g.lh;
    hipStream_t s = ctx->stream;
    if ((st = dev_reserve(ctx, im->in_rgb, npix * 3)) != RVSEG_OK) return st;
    if ((st = dev_reserve(ctx, im->in_depth, npix * 2)) != RVSEG_OK) return st;
    if ((st = dev_reserve(ctx, im->lab, npix * 4)) != RVSEG_OK) return st;
    if ((st = dev_reserve(ctx, im->cloud, npix * 16)) != RVSEG_OK) return st;
    if ((st = dev_reserve(ctx, im->rect, npix)) != RVSEG_OK) return st;
    if ((st = dev_reserve(ctx, im->change, npix)) != RVSEG_OK) return st;
    if ((st = dev_reserve(ctx, im->nfeat, (size_t)P * 4)) != RVSEG_OK) return st;
    if ((st = dev_reserve(ctx, im->dump, (size_t)P * g.D * 4)) != RVSEG_OK) return st;
    if ((st = dev_reserve(ctx, im->valid, (size_t)P)) != RVSEG_OK) return st;
    RV_HIP(ctx, hipMemcpyAsync(im->in_rgb.p, rgb, npix * 3, hipMemcpyHostToDevice, s));
    RV_HIP(ctx, hipMemcpyAsync(im->in_depth.p, depth_mm, npix * 2, hipMemcpyHostToDevice, s));
    if ((st = upload_calib(ctx, im, calib, 1, s)) != RVSEG_OK) return st;
    launch_prep(g, ctx->lab, im->in_rgb.as<uint8_t>(), im->in_depth.as<uint16_t>(), im->calibA.as<float>(),
                im->lab.as<uint32_t>(), im->cloud.as<float4>(), p.feature_normal ? im->change.as<uint8_t>() : nullptr, 1, s);
    if (p.feature_normal) {
        launch_window_map(g, im->cloud.as<float4>(), im->change.as<uint8_t>(), im->rect.as<uint8_t>(), 1, s);
        launch_normal_feature(g, im->cloud.as<float4>(), im->rect.as<uint8_t>(), im->nfeat.as<float>(), 1, s);
    }
    // the dump variant never touches the forest; a context without a model can still extract
    DeviceForest none = ctx->forest;
    launch_rf_frames(g, none, im->resize_rows.as<ResizeRow>(), im->lab.as<uint32_t>(), im->in_depth.as<uint16_t>(),
                     im->cloud.as<float4>(), im->nfeat.as<float>(), nullptr, im->dump.as<float>(), im->valid.as<uint8_t>(), 1, s);
    RV_LAUNCH_OK(ctx);
    std::vector<float> dump((size_t)P * g.D);
    std::vector<uint8_t> valid((size_t)P);
    RV_HIP(ctx, hipMemcpyAsync(dump.data(), im->dump.p, dump.size() * 4, hipMemcpyDeviceToHost, s));
    RV_HIP(ctx, hipMemcpyAsync(valid.data(), im->valid.p, valid.size(), hipMemcpyDeviceToHost, s));
    RV_HIP(ctx, hipStreamSynchronize(s));
    // compact in the order of the reference's stride-grid scan (feature_extractor.h:56-71)
    int n = 0;
    for (int ly = 0; ly < g.lh; ly++)
        for (int lx = 0; lx < g.lw; lx++) {
            const size_t pi = (size_t)ly * g.lw + lx;
            if (!valid[pi]) continue;
            std::memcpy(feat_out + (size_t)n * g.D, dump.data() + pi * g.D, (size_t)g.D * 4);
            x_v[n] = lx * g.stride;
            y_v[n] = ly * g.stride;
            n++;
        }
    *n_points = n;
    return RVSEG_OK;
}

}  // extern "C"
