// Deterministic elementary functions for the device.  Each is a fixed sequence of IEEE double
// +,-,*,/,sqrt,rint, compiled with -ffp-contract=off, so that it rounds identically on gfx950 and
// on the x86 host that runs the CPU oracle (which states the same sequences independently in
// oracle/rvseg_oracle.c).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace rvseg {

// fdlibm acos for x in [0,1], narrowed to float.  Replaces libm acos at
// include/feature_extractor.h:283 of the reference.
__device__ __forceinline__ float acos_f32_dev(float xf) {
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17,
                 pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                 pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                 pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                 qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    double x = (double)xf;
    if (x != x) return xf;
    if (x >= 1.0) return 0.0f;
    if (x < 0.0) x = 0.0;
    if (x < 0.5) {
        const double z = x * x;
        const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double r = p / q;
        return (float)(pio2_hi - (x - (pio2_lo - x * r)));
    }
    const double z = (1.0 - x) * 0.5;
    const double s = sqrt(z);
    const double df = __longlong_as_double(__double_as_longlong(s) & 0xFFFFFFFF00000000ll);
    const double c = (z - df * df) / (s + df);
    const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const double r = p / q;
    const double w = r * s + c;
    return (float)(2.0 * (df + w));
}

// exp for expAndNormalize (third-party/densecrf/src/densecrf.cpp:102: `b.array().exp()`).  Eigen is not in the tree and its
// version is unpinned, so the definition is build-owned (parity unpinned); it restates what Eigen 3's float packet
// path (pexp<Packet4f>, Cephes expf) evaluates on an SSE build without FMA -- the reference's build -- in fp32:
//   clamp to +-88.3762626647949; fx = floor(x * log2(e) + 0.5); x -= fx * 0.693359375; x -= fx * -2.12194440e-4;
//   degree-5 polynomial in x (Horner, separately rounded multiply and add), y = poly * x^2 + x + 1; result = y * 2^fx.
// Same sequence as oracle/rvseg_oracle.c:orc_exp_f32, operation for operation (the library is compiled with
// -ffp-contract=off).  Round 1-2 used a double-precision Taylor chain here: 13 v_fma_f64 at half rate per class made
// the fused update VALU bound.
__device__ __forceinline__ float exp_f32_dev(float xf) {
    float x = xf > 88.3762626647949f ? 88.3762626647949f : xf;
    x = x < -88.3762626647949f ? -88.3762626647949f : x;
    float fx = x * 1.44269504088896341f;
    fx = fx + 0.5f;
    const float t = (float)(int)fx;            // cvttps_epi32 + cvtepi32_ps: toward zero
    fx = t > fx ? t - 1.0f : t;                // -> floor
    const float hi = fx * 0.693359375f;
    const float lo = fx * -2.12194440e-4f;
    x = x - hi;
    x = x - lo;
    const float z = x * x;
    float y = 1.9875691500E-4f;
    y = y * x; y = y + 1.3981999507E-3f;
    y = y * x; y = y + 8.3334519073E-3f;
    y = y * x; y = y + 4.1665795894E-2f;
    y = y * x; y = y + 1.6666665459E-1f;
    y = y * x; y = y + 5.0000001201E-1f;
    y = y * z; y = y + x;
    y = y + 1.0f;
    const float scale = __int_as_float(((int)fx + 127) << 23);
    const float r = y * scale;
    return xf != xf ? xf : r;
}

// Rows of C floats / d+1 ints are only 4-byte aligned; these vector types let the compiler fetch
// them with dwordx4 / dwordx2 instructions (global memory tolerates dword-aligned wide accesses),
// i.e. ceil(C/4) vector-memory requests per lane instead of C.
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(4)));
typedef int i32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef int i32x2_u __attribute__((ext_vector_type(2), aligned(4)));

template <int N, typename T>
__device__ __forceinline__ void load_row(const T* __restrict__ p, T* __restrict__ out) {
    typedef T v4 __attribute__((ext_vector_type(4), aligned(4)));
    typedef T v2 __attribute__((ext_vector_type(2), aligned(4)));
    int i = 0;
#pragma unroll
    for (; i + 4 <= N; i += 4) {
        const v4 t = *reinterpret_cast<const v4*>(p + i);
        out[i] = t.x; out[i + 1] = t.y; out[i + 2] = t.z; out[i + 3] = t.w;
    }
    if (i + 2 <= N) {
        const v2 t = *reinterpret_cast<const v2*>(p + i);
        out[i] = t.x; out[i + 1] = t.y;
        i += 2;
    }
    if (i < N) out[i] = p[i];
}

template <int N, typename T>
__device__ __forceinline__ void store_row(T* __restrict__ p, const T* __restrict__ in) {
    typedef T v4 __attribute__((ext_vector_type(4), aligned(4)));
    typedef T v2 __attribute__((ext_vector_type(2), aligned(4)));
    int i = 0;
#pragma unroll
    for (; i + 4 <= N; i += 4) {
        v4 t; t.x = in[i]; t.y = in[i + 1]; t.z = in[i + 2]; t.w = in[i + 3];
        *reinterpret_cast<v4*>(p + i) = t;
    }
    if (i + 2 <= N) {
        v2 t; t.x = in[i]; t.y = in[i + 1];
        *reinterpret_cast<v2*>(p + i) = t;
        i += 2;
    }
    if (i < N) p[i] = in[i];
}

// The four label rules (rvseg_label_mode in include/rvseg.h) over one point's C class values:
//   0 eval tool   src/test.cpp:160-175       strict '>' from -1000, -1 when nothing wins
//   1 CRF         src/segmenter.cpp:646-657  strict '>' from 2.0/C, else the layer's "Unknown"
//   2 no CRF      src/segmenter.cpp:664-679  strict '>' from -1000, "Unknown" unless the sum is non-zero
//   3 arg max     densecrf.cpp:202-211       first maximum
__device__ __forceinline__ int label_rule(const float* v, int C, int mode, int unknown) {
    int best;
    float mx;
    if (mode == 0) {
        best = -1; mx = -1000.f;
        for (int c = 0; c < C; c++) { const float x = v[c]; if (x > mx) { mx = x; best = c; } }
    } else if (mode == 1) {
        best = unknown; mx = (float)(2.0 / (double)C);
        for (int c = 0; c < C; c++) { const float x = v[c]; if (x > mx) { mx = x; best = c; } }
    } else if (mode == 2) {
        best = unknown; mx = -1000.f;
        float sum = 0.f;
        for (int c = 0; c < C; c++) { const float x = v[c]; sum += x; if (x > mx) { mx = x; best = c; } }
        if (!(sum != 0.0f)) best = unknown;
    } else {
        best = 0; mx = v[0];
        for (int c = 1; c < C; c++) { const float x = v[c]; if (x > mx) { mx = x; best = c; } }
    }
    return best;
}

__device__ __forceinline__ bool finite_f(float v) { return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u; }

}  // namespace rvseg
