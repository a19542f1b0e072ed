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

}  // namespace rvseg

using namespace rvseg;

extern "C" rvseg_status rvseg_fuse_posteriors(rvseg_ctx* ctx, int32_t n_images, const int32_t* index_images,
                                              const float* posteriors, int32_t n_layers, const int32_t* class_counts,
                                              int32_t cloud_size, float* unaries_out) {
    if (!ctx) return RVSEG_ERR_INVALID_ARG;
    if (n_images < 0 || cloud_size < 0 || n_layers < 1 || n_layers > RVSEG_MAX_LAYERS || !class_counts || !unaries_out ||
        (n_images > 0 && (!index_images || !posteriors))) {
        ctx->err = "bad arguments";
        return RVSEG_ERR_INVALID_ARG;
    }
    FusionLayers fl{};
    fl.n_layers = n_layers;
    for (int l = 0; l < n_layers; l++) {
        if (class_counts[l] < 1 || class_counts[l] > 64) { ctx->err = "bad class count"; return RVSEG_ERR_INVALID_ARG; }
        fl.C[l] = class_counts[l];
        fl.prefix[l + 1] = fl.prefix[l] + class_counts[l];
    }
    const int S = fl.prefix[n_layers];
    const size_t pixels = (size_t)ctx->params.width * ctx->params.height;
    const unsigned long long hits = (unsigned long long)n_images * pixels;
    if (hits >= 0xFFFFFFFFull) { ctx->err = "too many index-image pixels for one call"; return RVSEG_ERR_INVALID_ARG; }
    if (cloud_size == 0) return RVSEG_OK;
    if (hits == 0) {
        std::memset(unaries_out, 0, (size_t)cloud_size * S * sizeof(float));
        return RVSEG_OK;
    }
    RV_HIP(ctx, hipSetDevice(ctx->params.device));
    hipStream_t s = ctx->stream;
    const unsigned n_hits = (unsigned)hits;
    int key_bits = 1;
    while ((1ull << key_bits) <= (unsigned long long)cloud_size) key_bits++;   // keys run 0 .. cloud_size
    size_t temp_bytes = 0;
    {
        unsigned* nul = nullptr;
        (void)rocprim::radix_sort_pairs(nullptr, temp_bytes, nul, nul, nul, nul, (size_t)n_hits, 0, (unsigned)key_bits, (hipStream_t)0);
    }
    DevBuf d_idx, d_post, d_kin, d_kout, d_vin, d_vout, d_temp, d_start, d_end, d_un, d_bad;
    DevBuf* all[] = {&d_idx, &d_post, &d_kin, &d_kout, &d_vin, &d_vout, &d_temp, &d_start, &d_end, &d_un, &d_bad};
    const size_t sizes[] = {(size_t)n_hits * 4, (size_t)n_hits * S * 4, (size_t)n_hits * 4, (size_t)n_hits * 4, (size_t)n_hits * 4,
                            (size_t)n_hits * 4, temp_bytes ? temp_bytes : 4, (size_t)cloud_size * 4, (size_t)cloud_size * 4,
                            (size_t)cloud_size * S * 4, 4};
    rvseg_status rc = RVSEG_OK;
    for (size_t i = 0; i < sizeof(all) / sizeof(all[0]) && rc == RVSEG_OK; i++) rc = dev_alloc(ctx, *all[i], sizes[i]);
    int bad = 0;
    do {
        if (rc != RVSEG_OK) break;
#define RV_TRY(call) if (!hip_ok(ctx, (call), #call)) { rc = RVSEG_ERR_HIP; break; }
        RV_TRY(hipMemcpyAsync(d_idx.p, index_images, (size_t)n_hits * 4, hipMemcpyHostToDevice, s));
        RV_TRY(hipMemcpyAsync(d_post.p, posteriors, (size_t)n_hits * S * 4, hipMemcpyHostToDevice, s));
        RV_TRY(hipMemsetAsync(d_start.p, 0, (size_t)cloud_size * 4, s));
        RV_TRY(hipMemsetAsync(d_end.p, 0, (size_t)cloud_size * 4, s));
        RV_TRY(hipMemsetAsync(d_bad.p, 0, 4, s));
        fusion_keys_kernel<<<dim3((n_hits + 255) / 256), dim3(256), 0, s>>>(d_idx.as<int32_t>(), n_hits, cloud_size, d_kin.as<unsigned>(),
                                                                           d_vin.as<unsigned>(), d_bad.as<int>());
        size_t tb = temp_bytes;
        RV_TRY(rocprim::radix_sort_pairs(d_temp.p, tb, d_kin.as<unsigned>(), d_kout.as<unsigned>(), d_vin.as<unsigned>(),
                                         d_vout.as<unsigned>(), (size_t)n_hits, 0, (unsigned)key_bits, s));
        fusion_runs_kernel<<<dim3((n_hits + 255) / 256), dim3(256), 0, s>>>(d_kout.as<unsigned>(), n_hits, cloud_size,
                                                                           d_start.as<unsigned>(), d_end.as<unsigned>());
        const long long threads = (long long)cloud_size * S;
        fusion_gather_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s>>>(
            fl, cloud_size, (unsigned)pixels, d_vout.as<unsigned>(), d_start.as<unsigned>(), d_end.as<unsigned>(), d_post.as<float>(),
            d_un.as<float>());
        RV_TRY(hipGetLastError());
        RV_TRY(hipMemcpyAsync(unaries_out, d_un.p, (size_t)cloud_size * S * 4, hipMemcpyDeviceToHost, s));
        RV_TRY(hipMemcpyAsync(&bad, d_bad.p, 4, hipMemcpyDeviceToHost, s));
        RV_TRY(hipStreamSynchronize(s));
#undef RV_TRY
    } while (0);
    for (DevBuf* b : all) dev_free(*b);
    if (rc == RVSEG_OK && bad) { ctx->err = "index image refers to a point beyond cloud_size"; return RVSEG_ERR_INVALID_ARG; }
    return rc;
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
        RV_TRY(hipGetLastError());
        RV_TRY(hipMemcpyAsync(labels_out, d_l.p, (size_t)N, hipMemcpyDeviceToHost, s));
        RV_TRY(hipStreamSynchronize(s));
#undef RV_TRY
    } while (0);
    dev_free(d_v);
    dev_free(d_l);
    return rc;
}
