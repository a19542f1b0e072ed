// fetch_calibration -- what does rocprofv3's FETCH_SIZE mean for the access shapes of this repo?
//
// MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half the bytes of a 16-B-per-lane
// streaming read and is "uncalibrated" for other shapes.  The splat (kernels_crf.hip,
// splat_group_kernel) reads an 8-byte-per-lane coalesced stream (CSR pairs) and gathers one 36-byte
// row per lane as dwordx4 + dwordx4 + dword.  This program runs kernels with exactly those shapes over
// buffers far larger than the 256 MiB Infinity Cache, on byte counts known in advance, and prints per
// kernel: useful bytes, bytes at 64-B and at 128-B granularity of the lines touched, and the time.
// Run once plain (times) and once under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (counter);
// profiles/scripts/fetch_calibration.py joins the two into profiles/rNN_fetch_calibration.json.
//
// build: hipcc -O3 --offload-arch=gfx950 -o profiles/scripts/fetch_calibration profiles/scripts/fetch_calibration.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// ---- streaming reads, W bytes per lane (coalesced) -------------------------------------------------
__global__ void __launch_bounds__(256) cal_stream16(const float4* __restrict__ src, size_t n, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}
__global__ void __launch_bounds__(256) cal_stream8(const uint2* __restrict__ src, size_t n, float* __restrict__ sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint2 v = src[i];
        acc += v.x ^ v.y;
    }
    if (acc == 0x12345u) sink[0] = (float)acc;
}
__global__ void __launch_bounds__(256) cal_stream4(const unsigned* __restrict__ src, size_t n, float* __restrict__ sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += src[i];
    if (acc == 0x12345u) sink[0] = (float)acc;
}

// ---- the splat's row gather: one 36-byte row (9 floats, 4-byte aligned) per lane as x4 + x4 + x1.
//      idx[i] = row of lane i (64 consecutive i form one wave-instruction) ---------------------------
__global__ void __launch_bounds__(256) cal_gather36(const float* __restrict__ rows, const unsigned* __restrict__ idx, size_t n,
                                                    float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float* p = rows + (size_t)idx[i] * 9;
        const f32x4_u a = *reinterpret_cast<const f32x4_u*>(p);
        const f32x4_u b = *reinterpret_cast<const f32x4_u*>(p + 4);
        const float c = p[8];
        acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// ---- the same gather from rows padded to 48 bytes (12 floats, 16-byte aligned): three aligned x4 per row ----------
__global__ void __launch_bounds__(256) cal_gather48(const float* __restrict__ rows, const unsigned* __restrict__ idx, size_t n,
                                                    float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4* p = reinterpret_cast<const float4*>(rows + (size_t)idx[i] * 12);
        const float4 a = p[0], b = p[1], c = p[2];
        acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// ---- streaming 36-byte rows, one row per lane (lane i reads bytes [36 i, 36 i + 36) as x4 + x4 + x1): the access shape
//      of the row-per-thread streaming kernels (mf_update, softmax_unary, upsample) ---------------------------------
__global__ void __launch_bounds__(256) cal_rows36_per_lane(const float* __restrict__ rows, size_t n_rows, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_rows; i += (size_t)gridDim.x * 256) {
        const float* p = rows + i * 9;
        const f32x4_u a = *reinterpret_cast<const f32x4_u*>(p);
        const f32x4_u b = *reinterpret_cast<const f32x4_u*>(p + 4);
        const float c = p[8];
        acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c;
    }
    if (acc == 12345.678f) sink[0] = acc;
}
// the same rows fetched as 16 bytes per lane, coalesced (a wave's 64 rows are 2304 contiguous bytes = 144 x 16 B: three
// wave-loads of which the third is three-quarters empty), parked in LDS and read back row-wise by the lane that owns them
__global__ void __launch_bounds__(256) cal_rows36_via_lds(const float* __restrict__ rows, size_t n_rows, float* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) float stage[4][64 * 9 + 12];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    float* st = stage[wave];
    for (size_t r0 = ((size_t)blockIdx.x * 4 + wave) * 64; r0 + 64 <= n_rows; r0 += (size_t)gridDim.x * 256) {
        const float4* src = reinterpret_cast<const float4*>(rows + r0 * 9);   // 36 * 64 * k bytes: 16-byte aligned when r0 is a multiple of 4
        const float4 v0 = src[lane], v1 = src[64 + lane];
        float4 v2 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < 16) v2 = src[128 + lane];
        reinterpret_cast<float4*>(st)[lane] = v0;
        reinterpret_cast<float4*>(st)[64 + lane] = v1;
        if (lane < 16) reinterpret_cast<float4*>(st)[128 + lane] = v2;
        __builtin_amdgcn_wave_barrier();
        const float* p = st + lane * 9;   // stride 9 words: conflict-free
#pragma unroll
        for (int k = 0; k < 9; k++) acc += p[k];
        __builtin_amdgcn_wave_barrier();
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// ---- K bytes per lane from the start of a random 128-byte-aligned line (granularity probe) ----------
template <int K>
__global__ void __launch_bounds__(256) cal_gather_line(const float* __restrict__ base, const unsigned* __restrict__ idx, size_t n,
                                                       float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float* p = base + (size_t)idx[i] * 32;   // 128-byte lines
        if (K == 4) acc += p[0];
        else {
#pragma unroll
            for (int k = 0; k < K / 16; k++) {
                const float4 v = *reinterpret_cast<const float4*>(p + 4 * k);
                acc += v.x + v.y + v.z + v.w;
            }
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rng() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    void start() { CK(hipEventRecord(a, 0)); }
    float stop() { CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }
};

// bytes of the 64-B sectors / 128-B lines touched by the rows of ONE wave-instruction group of 64 lanes
static void granule_bytes(const std::vector<unsigned>& idx, double& b64, double& b128) {
    b64 = b128 = 0;
    std::vector<uint64_t> s64, s128;
    for (size_t w = 0; w + 64 <= idx.size(); w += 64) {
        s64.clear(); s128.clear();
        for (int l = 0; l < 64; l++) {
            const uint64_t o = (uint64_t)idx[w + l] * 36;
            for (uint64_t q = o / 64; q <= (o + 35) / 64; q++) s64.push_back(q);
            for (uint64_t q = o / 128; q <= (o + 35) / 128; q++) s128.push_back(q);
        }
        std::sort(s64.begin(), s64.end()); s64.erase(std::unique(s64.begin(), s64.end()), s64.end());
        std::sort(s128.begin(), s128.end()); s128.erase(std::unique(s128.begin(), s128.end()), s128.end());
        b64 += 64.0 * s64.size(); b128 += 128.0 * s128.size();
    }
}

int main() {
    const size_t BYTES = (size_t)1 << 30;             // 1 GiB: four times the Infinity Cache
    const size_t n_rows = BYTES / 36;
    float* buf; float* sink; unsigned* d_idx;
    CK(hipMalloc(&buf, BYTES + 256));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, BYTES + 256));
    const size_t NG = (size_t)24 << 20;               // gathers per launch
    CK(hipMalloc(&d_idx, NG * 4));
    std::vector<unsigned> idx(NG);
    Timer t;
    const dim3 grid(256 * 16), block(256);
    auto report = [&](const char* name, double useful, double b64, double b128, float ms) {
        std::printf("{\"kernel\": \"%s\", \"useful_bytes\": %.0f, \"bytes_64B_granules\": %.0f, \"bytes_128B_lines\": %.0f, \"ms\": %.4f, "
                    "\"useful_GBps\": %.1f}\n", name, useful, b64, b128, ms, useful / ms / 1e6);
    };
    for (int rep = 0; rep < 2; rep++) {   // second repetition is the one to read (first touches pages)
        t.start(); cal_stream16<<<grid, block>>>(reinterpret_cast<const float4*>(buf), BYTES / 16, sink); report("cal_stream16", (double)BYTES, (double)BYTES, (double)BYTES, t.stop());
        t.start(); cal_stream8<<<grid, block>>>(reinterpret_cast<const uint2*>(buf), BYTES / 8, sink); report("cal_stream8", (double)BYTES, (double)BYTES, (double)BYTES, t.stop());
        t.start(); cal_stream4<<<grid, block>>>(reinterpret_cast<const unsigned*>(buf), BYTES / 4, sink); report("cal_stream4", (double)BYTES, (double)BYTES, (double)BYTES, t.stop());
    }
    for (int rep = 0; rep < 2; rep++) {
        t.start(); cal_rows36_per_lane<<<grid, block>>>(buf, n_rows, sink); report("cal_rows36_per_lane", 36.0 * n_rows, (double)BYTES, (double)BYTES, t.stop());
        t.start(); cal_rows36_via_lds<<<grid, block>>>(buf, n_rows, sink); report("cal_rows36_via_lds", 36.0 * n_rows, (double)BYTES, (double)BYTES, t.stop());
    }
    // (a) every lane its own random row: no coalescing at all
    for (size_t i = 0; i < NG; i++) idx[i] = (unsigned)(rng() % n_rows);
    {
        double b64, b128; granule_bytes(idx, b64, b128);
        CK(hipMemcpy(d_idx, idx.data(), NG * 4, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; rep++) { t.start(); cal_gather36<<<grid, block>>>(buf, d_idx, NG, sink); report("cal_gather36_random", 36.0 * NG, b64, b128, t.stop()); }
    }
    // (b) the splat's shape: a vertex's list is made of runs of consecutive pixels; runs of 16 rows
    for (size_t i = 0; i < NG; i += 16) { const unsigned r0 = (unsigned)(rng() % (n_rows - 16)); for (int k = 0; k < 16; k++) idx[i + k] = r0 + k; }
    {
        double b64, b128; granule_bytes(idx, b64, b128);
        CK(hipMemcpy(d_idx, idx.data(), NG * 4, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; rep++) { t.start(); cal_gather36<<<grid, block>>>(buf, d_idx, NG, sink); report("cal_gather36_runs16", 36.0 * NG, b64, b128, t.stop()); }
    }
    {   // (b') the same runs of 16 rows from 48-byte rows (indices scaled down so that they stay inside the buffer)
        std::vector<unsigned> idx48(NG);
        const size_t n_rows48 = BYTES / 48;
        for (size_t i = 0; i < NG; i += 16) { const unsigned r0 = (unsigned)(rng() % (n_rows48 - 16)); for (int k = 0; k < 16; k++) idx48[i + k] = r0 + k; }
        CK(hipMemcpy(d_idx, idx48.data(), NG * 4, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; rep++) { t.start(); cal_gather48<<<grid, block>>>(buf, d_idx, NG, sink); report("cal_gather48_runs16", 36.0 * NG, 48.0 * NG, 48.0 * NG, t.stop()); }
        for (size_t i = 0; i < NG; i += 16) { const unsigned r0 = (unsigned)(rng() % (n_rows - 16)); for (int k = 0; k < 16; k++) idx[i + k] = r0 + k; }
        CK(hipMemcpy(d_idx, idx.data(), NG * 4, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; rep++) { t.start(); cal_gather36<<<grid, block>>>(buf, d_idx, NG, sink); report("cal_gather36_runs16_again", 36.0 * NG, 0, 0, t.stop()); }
    }
    // (c) whole tiles of 64 consecutive rows
    for (size_t i = 0; i < NG; i += 64) { const unsigned r0 = (unsigned)(rng() % (n_rows - 64)); for (int k = 0; k < 64; k++) idx[i + k] = r0 + k; }
    {
        double b64, b128; granule_bytes(idx, b64, b128);
        CK(hipMemcpy(d_idx, idx.data(), NG * 4, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; rep++) { t.start(); cal_gather36<<<grid, block>>>(buf, d_idx, NG, sink); report("cal_gather36_runs64", 36.0 * NG, b64, b128, t.stop()); }
    }
    // (d) granularity probe: K bytes from the start of a random 128-B line, every lane a different line
    const size_t n_lines = BYTES / 128;
    for (size_t i = 0; i < NG; i++) idx[i] = (unsigned)(rng() % n_lines);
    CK(hipMemcpy(d_idx, idx.data(), NG * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; rep++) {
        t.start(); cal_gather_line<4><<<grid, block>>>(buf, d_idx, NG, sink); report("cal_gather_line<4>", 4.0 * NG, 64.0 * NG, 128.0 * NG, t.stop());
        t.start(); cal_gather_line<16><<<grid, block>>>(buf, d_idx, NG, sink); report("cal_gather_line<16>", 16.0 * NG, 64.0 * NG, 128.0 * NG, t.stop());
        t.start(); cal_gather_line<64><<<grid, block>>>(buf, d_idx, NG, sink); report("cal_gather_line<64>", 64.0 * NG, 64.0 * NG, 128.0 * NG, t.stop());
        t.start(); cal_gather_line<128><<<grid, block>>>(buf, d_idx, NG, sink); report("cal_gather_line<128>", 128.0 * NG, 128.0 * NG, 128.0 * NG, t.stop());
    }
    CK(hipDeviceSynchronize());
    return 0;
}
