// ta_shapes -- what does a row gather cost the texture addresser when the data is cache resident?
//
// The resident splat (kernels_crf.hip) is bound by its vector-memory instructions, not by bytes (PMC: TA busy 77 %).
// Its row gather is one 36-byte row per lane as dwordx4 + dwordx4 + dword.  This program times that shape and
// alternatives on a table small enough to stay in L2 (so that HBM is out of the picture), with the splat's occupancy
// (7 gathering waves per CU, 8 gathers in flight per wave) and its index pattern (runs of consecutive rows):
//   rows36      x4 + x4 + x1 from 36-byte rows                      (what the splat does)
//   rows36_8    x4 + x4 only                                         (what the ninth class costs)
//   rows32_p1   x4 + x4 from 32-byte-aligned rows + x1 from a separate plane of ninth classes
//   rows48_3l   three lanes per row, one aligned x4 each, from rows padded to 48 bytes
// build: hipcc -O3 --offload-arch=gfx950 -o profiles/scripts/ta_shapes profiles/scripts/ta_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int DEPTH = 8;   // gathers in flight per wave

template <int SHAPE>
__global__ void __launch_bounds__(448) gather_kernel(const float* tab, const float* plane, const unsigned* __restrict__ idx, int iters, int n_idx,
                                                     float* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 7 + (threadIdx.x >> 6));
    float acc = 0.f;
    unsigned base = (unsigned)wave * 64u * 131u;
    for (int it = 0; it < iters; it += DEPTH) {
        f32x4_u a[DEPTH], b[DEPTH];
        float c[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; k++) {
            const unsigned row = idx[(base + (unsigned)(it + k) * 64u + (unsigned)lane) % (unsigned)n_idx];
            if (SHAPE == 0 || SHAPE == 1) {
                const float* p = tab + (size_t)row * 9;
                a[k] = *reinterpret_cast<const f32x4_u*>(p);
                b[k] = *reinterpret_cast<const f32x4_u*>(p + 4);
                c[k] = SHAPE == 0 ? p[8] : 0.f;
            } else if (SHAPE == 2) {
                const float4* p = reinterpret_cast<const float4*>(tab + (size_t)row * 8);
                const float4 u = p[0], v = p[1];
                a[k] = f32x4_u{u.x, u.y, u.z, u.w}; b[k] = f32x4_u{v.x, v.y, v.z, v.w};
                c[k] = plane[row];
            } else {
                // three lanes per row: lane l of instruction q handles row index (64 q + l) / 3 of the group's 64 rows
                // (emulated: the 64 rows of a group are idx[...]; lanes read rows (l + 64 q) / 3, piece (l + 64 q) % 3)
                f32x4_u t[3];
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    const int e = (lane + 64 * q) / 3, piece = (lane + 64 * q) % 3;
                    const unsigned r2 = __shfl(row, e, 64);
                    const float4 u = *reinterpret_cast<const float4*>(tab + (size_t)r2 * 12 + 4 * piece);
                    t[q] = f32x4_u{u.x, u.y, u.z, u.w};
                }
                a[k] = t[0]; b[k] = t[1]; c[k] = t[2].x;
            }
        }
#pragma unroll
        for (int k = 0; k < DEPTH; k++) acc += a[k].x + a[k].w + b[k].y + c[k];
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main() {
    const int n_rows = 1 << 16;              // 64 K rows: 2.3 MB of 36-byte rows (L2 resident)
    const int n_idx = 1 << 20;
    std::vector<unsigned> idx(n_idx);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rng = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int i = 0; i < n_idx;) {            // runs of 4..40 consecutive rows, like a vertex list's pixel runs
        const int run = 4 + (int)(rng() % 37);
        unsigned r = (unsigned)(rng() % (n_rows - 64));
        for (int k = 0; k < run && i < n_idx; k++, i++) idx[i] = r + k;
    }
    float *tab, *plane, *sink;
    unsigned* d_idx;
    CK(hipMalloc(&tab, (size_t)n_rows * 12 * 4 + 256));
    CK(hipMalloc(&plane, (size_t)n_rows * 4));
    CK(hipMalloc(&sink, 16));
    CK(hipMalloc(&d_idx, (size_t)n_idx * 4));
    CK(hipMemset(tab, 0, (size_t)n_rows * 12 * 4 + 256));
    CK(hipMemset(plane, 0, (size_t)n_rows * 4));
    CK(hipMemcpy(d_idx, idx.data(), (size_t)n_idx * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4096, blocks = 256;
    const char* names[4] = {"rows36", "rows36_8", "rows32_p1", "rows48_3l"};
    for (int rep = 0; rep < 2; rep++)
        for (int shape = 0; shape < 4; shape++) {
            CK(hipEventRecord(e0, 0));
            if (shape == 0) gather_kernel<0><<<blocks, 448>>>(tab, plane, d_idx, iters, n_idx, sink);
            if (shape == 1) gather_kernel<1><<<blocks, 448>>>(tab, plane, d_idx, iters, n_idx, sink);
            if (shape == 2) gather_kernel<2><<<blocks, 448>>>(tab, plane, d_idx, iters, n_idx, sink);
            if (shape == 3) gather_kernel<3><<<blocks, 448>>>(tab, plane, d_idx, iters, n_idx, sink);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            // per CU: 7 waves x iters gathers of 64 rows
            const double ns_per_gather = ms * 1e6 / (7.0 * iters);
            if (rep) std::printf("{\"shape\": \"%s\", \"ms\": %.3f, \"ns_per_64row_gather_per_cu\": %.2f}\n", names[shape], ms, ns_per_gather);
        }
    return 0;
}
