// Local-map fusion (src/segmenter.cpp:561-616 of the reference): the per-frame label distributions
// are added into per-point unaries of the local map's cloud through the projector's index images,
//     unaries[l](c, index) += label_distribution[off_l + pixel * C_l + c]          (:599-611)
// image after image, pixels in raster order.  fp32 addition is not associative, so the build keeps
// exactly that order: all (image, pixel) hits are sorted by point index with a STABLE radix sort --
// equal points keep their (image, pixel) order -- and one thread per (point, class) then walks its
// point's hits front to back.  No atomics, no float reassociation.
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "rvseg_internal.h"
#include "rvseg_kernels.h"
#include "rvseg_pipeline.h"

namespace rvseg {

__global__ void __launch_bounds__(256)
fusion_keys_kernel(const int32_t* __restrict__ index_images, unsigned n_hits, int cloud_size, unsigned* __restrict__ keys,
                   unsigned* __restrict__ vals, int* __restrict__ bad) {
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_hits) return;
    const int idx = index_images[e];
    if (idx >= cloud_size) *bad = 1;   // the reference would write out of bounds here (segmenter.cpp:607)
    keys[e] = (idx >= 0 && idx < cloud_size) ? (unsigned)idx : (unsigned)cloud_size;   // "no point" sorts to the end
    vals[e] = e;
}

// first / one-past-last sorted position of every point that was hit
__global__ void __launch_bounds__(256)
fusion_runs_kernel(const unsigned* __restrict__ keys_sorted, unsigned n_hits, int cloud_size, unsigned* __restrict__ start,
                   unsigned* __restrict__ end) {
    const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_hits) return;
    const unsigned key = keys_sorted[k];
    if (key >= (unsigned)cloud_size) return;
    if (k == 0 || keys_sorted[k - 1] != key) start[key] = k;
    if (k == n_hits - 1 || keys_sorted[k + 1] != key) end[key] = k + 1;
}

struct FusionLayers {
    int n_layers;
    int C[RVSEG_MAX_LAYERS];
    int prefix[RVSEG_MAX_LAYERS + 1];   // classes before layer l
};

// one thread per (point, class over all layers); posteriors: image x [layer][pixel][class]
__global__ void __launch_bounds__(256)
fusion_gather_kernel(FusionLayers fl, int cloud_size, unsigned pixels, const unsigned* __restrict__ vals_sorted,
                     const unsigned* __restrict__ start, const unsigned* __restrict__ end,
                     const float* __restrict__ posteriors, float* __restrict__ unaries) {
    const int S = fl.prefix[fl.n_layers];
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)cloud_size * S) return;
    const int point = (int)(gid / S), sc = (int)(gid - (long long)point * S);
    int l = 0;
    while (l + 1 < fl.n_layers && sc >= fl.prefix[l + 1]) l++;
    const int C = fl.C[l], c = sc - fl.prefix[l];
    float acc = 0.0f;   // MatrixXf::Constant(label_count, cloud_size, 0.0), segmenter.cpp:566
    const unsigned k0 = start[point], k1 = end[point];
    for (unsigned k = k0; k < k1; k++) {
        const unsigned e = vals_sorted[k];
        const unsigned image = e / pixels, pix = e - image * pixels;
        const float v = posteriors[(size_t)image * pixels * S + (size_t)pixels * fl.prefix[l] + (size_t)pix * C + c];
        acc += v;
    }
    unaries[(size_t)cloud_size * fl.prefix[l] + (size_t)point * C + c] = acc;
}

// xyz * dcrf_xyz_kernel, rgb * dcrf_rgb_kernel per cloud point (src/segmenter.cpp:629-637); rgb in [0, 1]
__global__ void __launch_bounds__(256)
cloud_features_kernel(const float* __restrict__ xyz, const float* __restrict__ rgb, float kx, float kc, float* __restrict__ feat, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float* f = feat + i * 6;
#pragma unroll
    for (int k = 0; k < 3; k++) { f[k] = xyz[i * 3 + k] * kx; f[3 + k] = rgb[i * 3 + k] * kc; }
}

// buffers of the fusion, owned by the context (no allocation per call once they have grown)
struct FusionState {
    DevBuf kin, kout, vin, vout, temp, start, end, bad;
    DevBuf idx, post, un;          // staging of the host entry point
    DevBuf map_un, map_feat, map_q, map_lab;   // intermediates of rvseg_process_map_device
    int* h_bad = nullptr;          // pinned copy of the "index beyond cloud_size" flag
    hipEvent_t bad_ev = nullptr;
    bool bad_pending = false;
};

static rvseg_status fusion_state(rvseg_ctx* ctx, FusionState** out) {
    if (!ctx->impl) {
        Pipeline* im = new Pipeline();
        ctx->impl = reinterpret_cast<rvseg_ctx::Impl*>(im);
        im->bare = true;
    }
    Pipeline* im = reinterpret_cast<Pipeline*>(ctx->impl);
    if (!im->fusion) {
        FusionState* fs = new FusionState();
        if (!hip_ok(ctx, hipHostMalloc((void**)&fs->h_bad, sizeof(int), hipHostMallocDefault), "hipHostMalloc(fusion flag)") ||
            !hip_ok(ctx, hipEventCreateWithFlags(&fs->bad_ev, hipEventDisableTiming), "hipEventCreate(fusion flag)")) {
            if (fs->h_bad) (void)hipHostFree(fs->h_bad);
            delete fs;
            return RVSEG_ERR_HIP;
        }
        *fs->h_bad = 0;
        im->fusion = fs;
    }
    *out = im->fusion;
    return RVSEG_OK;
}

void fusion_state_free(Pipeline* im) {
    FusionState* fs = im->fusion;
    if (!fs) return;
    DevBuf* all[] = {&fs->kin, &fs->kout, &fs->vin, &fs->vout, &fs->temp, &fs->start, &fs->end, &fs->bad, &fs->idx, &fs->post, &fs->un,
                     &fs->map_un, &fs->map_feat, &fs->map_q, &fs->map_lab};
    for (DevBuf* b : all) dev_free(*b);
    if (fs->h_bad) (void)hipHostFree(fs->h_bad);
    if (fs->bad_ev) (void)hipEventDestroy(fs->bad_ev);
    delete fs;
    im->fusion = nullptr;
}

rvseg_status fusion_status(rvseg_ctx* ctx, Pipeline* im, bool wait) {
    FusionState* fs = im->fusion;
    if (!fs || !fs->bad_pending) return RVSEG_OK;
    if (wait) {
        RV_HIP(ctx, hipEventSynchronize(fs->bad_ev));
    } else {
        const hipError_t e = hipEventQuery(fs->bad_ev);
        if (e == hipErrorNotReady) return RVSEG_NOT_READY;
        RV_HIP(ctx, e);
    }
    fs->bad_pending = false;
    if (*fs->h_bad) {
        ctx->err = "index image refers to a point beyond cloud_size";
        return RVSEG_ERR_INVALID_ARG;
    }
    return RVSEG_OK;
}

static rvseg_status fusion_layers(rvseg_ctx* ctx, int32_t n_layers, const int32_t* class_counts, FusionLayers& fl) {
    if (n_layers < 1 || n_layers > RVSEG_MAX_LAYERS || !class_counts) { ctx->err = "bad arguments"; return RVSEG_ERR_INVALID_ARG; }
    fl = FusionLayers{};
    fl.n_layers = n_layers;
    for (int l = 0; l < n_layers; l++) {
        if (class_counts[l] < 1 || class_counts[l] > 64) { ctx->err = "bad class count"; return RVSEG_ERR_INVALID_ARG; }
        fl.C[l] = class_counts[l];
        fl.prefix[l + 1] = fl.prefix[l] + class_counts[l];
    }
    return RVSEG_OK;
}

// everything on the device, enqueued on `s`: keys -> stable sort by point -> runs -> ordered gather
static rvseg_status fuse_device(rvseg_ctx* ctx, FusionState* fs, const FusionLayers& fl, int32_t n_images, const int32_t* d_index_images,
                                const float* d_posteriors, int32_t cloud_size, float* d_unaries, hipStream_t s) {
    const int S = fl.prefix[fl.n_layers];
    const size_t pixels = (size_t)ctx->params.width * ctx->params.height;
    const unsigned long long hits = (unsigned long long)n_images * pixels;
    if (hits >= 0xFFFFFFFFull) { ctx->err = "too many index-image pixels for one call"; return RVSEG_ERR_INVALID_ARG; }
    if (cloud_size == 0) return RVSEG_OK;
    if (hits == 0) {
        RV_HIP(ctx, hipMemsetAsync(d_unaries, 0, (size_t)cloud_size * S * sizeof(float), s));
        return RVSEG_OK;
    }
    const unsigned n_hits = (unsigned)hits;
    int key_bits = 1;
    while ((1ull << key_bits) <= (unsigned long long)cloud_size) key_bits++;   // keys run 0 .. cloud_size
    size_t temp_bytes = 0;
    {
        unsigned* nul = nullptr;
        (void)rocprim::radix_sort_pairs(nullptr, temp_bytes, nul, nul, nul, nul, (size_t)n_hits, 0, (unsigned)key_bits, (hipStream_t)0);
    }
    rvseg_status st;
    DevBuf* bufs[] = {&fs->kin, &fs->kout, &fs->vin, &fs->vout, &fs->temp, &fs->start, &fs->end, &fs->bad};
    const size_t sizes[] = {(size_t)n_hits * 4, (size_t)n_hits * 4, (size_t)n_hits * 4, (size_t)n_hits * 4, temp_bytes ? temp_bytes : 4,
                            (size_t)cloud_size * 4, (size_t)cloud_size * 4, 4};
    for (size_t i = 0; i < sizeof(bufs) / sizeof(bufs[0]); i++)
        if ((st = dev_reserve(ctx, *bufs[i], sizes[i])) != RVSEG_OK) return st;
    RV_HIP(ctx, hipMemsetAsync(fs->start.p, 0, (size_t)cloud_size * 4, s));
    RV_HIP(ctx, hipMemsetAsync(fs->end.p, 0, (size_t)cloud_size * 4, s));
    RV_HIP(ctx, hipMemsetAsync(fs->bad.p, 0, 4, s));
    fusion_keys_kernel<<<dim3((n_hits + 255) / 256), dim3(256), 0, s>>>(d_index_images, n_hits, cloud_size, fs->kin.as<unsigned>(),
                                                                       fs->vin.as<unsigned>(), fs->bad.as<int>());
    size_t tb = temp_bytes;
    RV_HIP(ctx, rocprim::radix_sort_pairs(fs->temp.p, tb, fs->kin.as<unsigned>(), fs->kout.as<unsigned>(), fs->vin.as<unsigned>(),
                                          fs->vout.as<unsigned>(), (size_t)n_hits, 0, (unsigned)key_bits, s));
    fusion_runs_kernel<<<dim3((n_hits + 255) / 256), dim3(256), 0, s>>>(fs->kout.as<unsigned>(), n_hits, cloud_size,
                                                                       fs->start.as<unsigned>(), fs->end.as<unsigned>());
    const long long threads = (long long)cloud_size * S;
    fusion_gather_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s>>>(
        fl, cloud_size, (unsigned)pixels, fs->vout.as<unsigned>(), fs->start.as<unsigned>(), fs->end.as<unsigned>(), d_posteriors, d_unaries);
    RV_LAUNCH_OK(ctx);
    RV_HIP(ctx, hipMemcpyAsync(fs->h_bad, fs->bad.p, 4, hipMemcpyDeviceToHost, s));
    RV_HIP(ctx, hipEventRecord(fs->bad_ev, s));
    fs->bad_pending = true;
    return RVSEG_OK;
}

}  // namespace rvseg

using namespace rvseg;

extern "C" rvseg_status rvseg_fuse_posteriors_device(rvseg_ctx* ctx, int32_t n_images, const int32_t* d_index_images,
                                                     const float* d_posteriors, int32_t n_layers, const int32_t* class_counts,
                                                     int32_t cloud_size, float* d_unaries_out, void* hip_stream) {
    if (!ctx) return RVSEG_ERR_INVALID_ARG;
    if (n_images < 0 || cloud_size < 0 || !d_unaries_out || (n_images > 0 && (!d_index_images || !d_posteriors))) {
        ctx->err = "bad arguments";
        return RVSEG_ERR_INVALID_ARG;
    }
    FusionLayers fl;
    rvseg_status st = fusion_layers(ctx, n_layers, class_counts, fl);
    if (st != RVSEG_OK) return st;
    RV_HIP(ctx, hipSetDevice(ctx->params.device));
    FusionState* fs;
    if ((st = fusion_state(ctx, &fs)) != RVSEG_OK) return st;
    Pipeline* im = reinterpret_cast<Pipeline*>(ctx->impl);
    if ((st = fusion_status(ctx, im, true)) != RVSEG_OK) return st;   // an unpolled failure of the previous fusion
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
    timer_reset(ctx);
    timer_mark(ctx, "fusion", s);
    st = fuse_device(ctx, fs, fl, n_images, d_index_images, d_posteriors, cloud_size, d_unaries_out, s);
    timer_mark(ctx, "end", s);
    return st;
}

extern "C" rvseg_status rvseg_fuse_posteriors(rvseg_ctx* ctx, int32_t n_images, const int32_t* index_images,
                                              const float* posteriors, int32_t n_layers, const int32_t* class_counts,
                                              int32_t cloud_size, float* unaries_out) {
    if (!ctx) return RVSEG_ERR_INVALID_ARG;
    if (n_images < 0 || cloud_size < 0 || !unaries_out || (n_images > 0 && (!index_images || !posteriors))) {
        ctx->err = "bad arguments";
        return RVSEG_ERR_INVALID_ARG;
    }
    FusionLayers fl;
    rvseg_status st = fusion_layers(ctx, n_layers, class_counts, fl);
    if (st != RVSEG_OK) return st;
    const int S = fl.prefix[n_layers];
    const size_t pixels = (size_t)ctx->params.width * ctx->params.height;
    const size_t n_hits = (size_t)n_images * pixels;
    if (cloud_size == 0) return RVSEG_OK;
    if (n_hits == 0) {
        std::memset(unaries_out, 0, (size_t)cloud_size * S * sizeof(float));
        return RVSEG_OK;
    }
    RV_HIP(ctx, hipSetDevice(ctx->params.device));
    FusionState* fs;
    if ((st = fusion_state(ctx, &fs)) != RVSEG_OK) return st;
    Pipeline* im = reinterpret_cast<Pipeline*>(ctx->impl);
    hipStream_t s = ctx->stream;
    if ((st = dev_reserve(ctx, fs->idx, n_hits * 4)) != RVSEG_OK) return st;
    if ((st = dev_reserve(ctx, fs->post, n_hits * S * 4)) != RVSEG_OK) return st;
    if ((st = dev_reserve(ctx, fs->un, (size_t)cloud_size * S * 4)) != RVSEG_OK) return st;
    RV_HIP(ctx, hipMemcpyAsync(fs->idx.p, index_images, n_hits * 4, hipMemcpyHostToDevice, s));
    RV_HIP(ctx, hipMemcpyAsync(fs->post.p, posteriors, n_hits * S * 4, hipMemcpyHostToDevice, s));
    (void)fusion_status(ctx, im, true);
    if ((st = fuse_device(ctx, fs, fl, n_images, fs->idx.as<int32_t>(), fs->post.as<float>(), cloud_size, fs->un.as<float>(), s)) != RVSEG_OK) return st;
    RV_HIP(ctx, hipMemcpyAsync(unaries_out, fs->un.p, (size_t)cloud_size * S * 4, hipMemcpyDeviceToHost, s));
    RV_HIP(ctx, hipStreamSynchronize(s));
    return fusion_status(ctx, im, true);
}

extern "C" rvseg_status rvseg_cloud_features_device(rvseg_ctx* ctx, int32_t N, const float* d_xyz, const float* d_rgb,
                                                    float* d_features_out, void* hip_stream) {
    if (!ctx) return RVSEG_ERR_INVALID_ARG;
    if (N < 0 || (N > 0 && (!d_xyz || !d_rgb || !d_features_out))) { ctx->err = "bad arguments"; return RVSEG_ERR_INVALID_ARG; }
    if (N == 0) return RVSEG_OK;
    RV_HIP(ctx, hipSetDevice(ctx->params.device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
    cloud_features_kernel<<<dim3((unsigned)(((long long)N + 255) / 256)), dim3(256), 0, s>>>(d_xyz, d_rgb, ctx->params.dcrf_xyz_kernel,
                                                                                             ctx->params.dcrf_rgb_kernel, d_features_out, N);
    RV_LAUNCH_OK(ctx);
    return RVSEG_OK;
}

extern "C" rvseg_status rvseg_label_values_device(rvseg_ctx* ctx, const float* d_values, int32_t N, int32_t C, int32_t label_mode,
                                                  int32_t unknown_label, int8_t* d_labels_out, void* hip_stream) {
    if (!ctx) return RVSEG_ERR_INVALID_ARG;
    if (N < 0 || C < 1 || C > 64 || label_mode < 0 || label_mode > 3 || (N > 0 && (!d_values || !d_labels_out))) {
        ctx->err = "bad arguments";
        return RVSEG_ERR_INVALID_ARG;
    }
    if (N == 0) return RVSEG_OK;
    RV_HIP(ctx, hipSetDevice(ctx->params.device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
    launch_labels(d_values, (size_t)N, C, label_mode, unknown_label, d_labels_out, s);
    RV_LAUNCH_OK(ctx);
    return RVSEG_OK;
}

// processMapFromQueue for one local map with every buffer in HBM (src/segmenter.cpp:561-682): fusion of
// the frames' label distributions, then per layer the cloud DenseCRF + thresholded argmax (:628-658)
// or the no-CRF rule (:660-681).  Intermediates (unaries, 6-D features, marginals) belong to the context.
extern "C" rvseg_status rvseg_process_map_device(rvseg_ctx* ctx, int32_t n_images, const int32_t* d_index_images,
                                                 const float* d_posteriors, int32_t cloud_size, const float* d_cloud_xyz,
                                                 const float* d_cloud_rgb, int8_t* d_labels_out, float* d_unaries_out, void* hip_stream) {
    if (!ctx) return RVSEG_ERR_INVALID_ARG;
    if (!ctx->forest_loaded) { ctx->err = "no forest loaded (the layer / class layout comes from the model)"; return RVSEG_ERR_NO_FOREST; }
    if (n_images < 0 || cloud_size < 0 || !d_labels_out || (n_images > 0 && (!d_index_images || !d_posteriors)) ||
        (ctx->params.use_dense_crf && cloud_size > 0 && (!d_cloud_xyz || !d_cloud_rgb))) {
        ctx->err = "bad arguments";
        return RVSEG_ERR_INVALID_ARG;
    }
    if (cloud_size == 0) return RVSEG_OK;
    const DeviceForest& f = ctx->forest;
    FusionLayers fl;
    rvseg_status st = fusion_layers(ctx, f.n_layers, f.class_counts, fl);
    if (st != RVSEG_OK) return st;
    RV_HIP(ctx, hipSetDevice(ctx->params.device));
    FusionState* fs;
    if ((st = fusion_state(ctx, &fs)) != RVSEG_OK) return st;
    Pipeline* im = reinterpret_cast<Pipeline*>(ctx->impl);
    if ((st = fusion_status(ctx, im, true)) != RVSEG_OK) return st;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
    const int S = f.sum_classes;
    float* un = d_unaries_out;
    if (!un) {
        if ((st = dev_reserve(ctx, fs->map_un, (size_t)cloud_size * S * 4)) != RVSEG_OK) return st;
        un = fs->map_un.as<float>();
    }
    timer_reset(ctx);
    timer_mark(ctx, "fusion", s);
    if ((st = fuse_device(ctx, fs, fl, n_images, d_index_images, d_posteriors, cloud_size, un, s)) != RVSEG_OK) return st;
    const rvseg_params& p = ctx->params;
    if (p.use_dense_crf) {
        if ((st = dev_reserve(ctx, fs->map_feat, (size_t)cloud_size * 6 * 4)) != RVSEG_OK) return st;
        timer_mark(ctx, "cloud_features", s);
        cloud_features_kernel<<<dim3((unsigned)(((long long)cloud_size + 255) / 256)), dim3(256), 0, s>>>(
            d_cloud_xyz, d_cloud_rgb, p.dcrf_xyz_kernel, p.dcrf_rgb_kernel, fs->map_feat.as<float>(), cloud_size);
        RV_LAUNCH_OK(ctx);
        // one lattice serves every layer (the reference builds an identical one per layer, :639-644)
        if ((st = crf_cloud_layers(ctx, cloud_size, f.n_layers, f.class_counts, un, fs->map_feat.as<float>(), p.dcrf_kernel_weight,
                                   p.dcrf_iterations, RVSEG_LABEL_CRF, p.unknown_label, d_labels_out, s)) != RVSEG_OK) return st;
    } else {
        timer_mark(ctx, "labels", s);
        for (int l = 0; l < f.n_layers; l++)
            launch_labels(un + (size_t)cloud_size * fl.prefix[l], (size_t)cloud_size, f.class_counts[l], RVSEG_LABEL_NOCRF, p.unknown_label[l],
                          d_labels_out + (size_t)l * cloud_size, s);
        RV_LAUNCH_OK(ctx);
    }
    timer_mark(ctx, "end", s);
    return RVSEG_OK;
}

// The label rules on host matrices (the no-CRF branch of processMapFromQueue labels the fused
// unaries directly, src/segmenter.cpp:660-681).
extern "C" rvseg_status rvseg_label_values(rvseg_ctx* ctx, const float* values, int32_t N, int32_t C, int32_t label_mode,
                                           int32_t unknown_label, int8_t* labels_out) {
    if (!ctx) return RVSEG_ERR_INVALID_ARG;
    if (N < 0 || C < 1 || C > 64 || label_mode < 0 || label_mode > 3 || !labels_out || (N > 0 && !values)) {
        ctx->err = "bad arguments";
        return RVSEG_ERR_INVALID_ARG;
    }
    if (N == 0) return RVSEG_OK;
    RV_HIP(ctx, hipSetDevice(ctx->params.device));
    hipStream_t s = ctx->stream;
    DevBuf d_v, d_l;
    rvseg_status rc = dev_alloc(ctx, d_v, (size_t)N * C * 4);
    if (rc == RVSEG_OK) rc = dev_alloc(ctx, d_l, (size_t)N);
    do {
        if (rc != RVSEG_OK) break;
#define RV_TRY(call) if (!hip_ok(ctx, (call), #call)) { rc = RVSEG_ERR_HIP; break; }
        RV_TRY(hipMemcpyAsync(d_v.p, values, (size_t)N * C * 4, hipMemcpyHostToDevice, s));
        launch_labels(d_v.as<float>(), (size_t)N, C, label_mode, unknown_label, d_l.as<int8_t>(), s);
        if ((rc = launch_error_take(ctx)) != RVSEG_OK) break;
        RV_TRY(hipMemcpyAsync(labels_out, d_l.p, (size_t)N, hipMemcpyDeviceToHost, s));
        RV_TRY(hipStreamSynchronize(s));
#undef RV_TRY
    } while (0);
    dev_free(d_v);
    dev_free(d_l);
    return rc;
}
