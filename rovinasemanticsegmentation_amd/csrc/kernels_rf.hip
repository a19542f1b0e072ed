// Random-forest traversal kernels for gfx950.
//
// Reference semantics (third-party/libforest/src/classifier.cpp):
//   findLeafNode  :97-117   node = x[feature] < threshold ? left : left + 1   (strict '<', fp32)
//   classLogPosterior / multiClassLogPosterior :166-208
//                           copy tree 0's leaf histogram, then += trees 1..T-1 IN ORDER (fp32)
// The node array is the breadth-first re-layout built by forest_model.cpp: 16-byte nodes, one
// dwordx4 load per visited node, the hot top levels of every tree packed at the front of each
// tree's block so that they stay in L1/L2.
#include <cstdlib>

#include "device_math.h"
#include "rvseg_internal.h"
#include "rvseg_kernels.h"

namespace rvseg {

// ---------------------------------------------------------------------------------------------
// forest_eval: P points with materialised features (the unit-parity entry point
// rvseg_forest_eval).  One lane per (point, tree) so that the T dependent walks of a point
// overlap; lanes of a point then add their leaf rows in tree order.
// ---------------------------------------------------------------------------------------------
template <int LOG2_TPP>  // lanes per point = 2^LOG2_TPP >= n_trees (trees beyond are looped)
__global__ void __launch_bounds__(256)
forest_eval_kernel(const DeviceNode* __restrict__ nodes, const int32_t* __restrict__ roots,
                   const float* __restrict__ hist, int n_trees, int S,
                   const float* __restrict__ X, int P, int D, float* __restrict__ out) {
    constexpr int TPP = 1 << LOG2_TPP;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int point = gid >> LOG2_TPP;
    const int sub = gid & (TPP - 1);
    const bool active = point < P;
    const float* x = X + (size_t)(active ? point : 0) * D;
    // every lane walks trees sub, sub+TPP, ... and remembers the leaf rows in a small local list
    // (n_trees <= 64 is enforced by the launcher)
    int leaf_rows[64 / TPP > 0 ? 64 / TPP : 1];
    int n_mine = 0;
    for (int t = sub; t < n_trees; t += TPP) {
        int node = roots[t];
        if (active) {
            const int4* np = reinterpret_cast<const int4*>(nodes);
            int4 nd = np[node];
            while (nd.z != 0) {
                const float v = x[nd.x];
                node = (v < __int_as_float(nd.y)) ? nd.z : nd.z + 1;
                nd = np[node];
            }
            leaf_rows[n_mine] = nd.w;
        } else {
            leaf_rows[n_mine] = 0;
        }
        n_mine++;
    }
    // ordered accumulation: class c of the point is summed by lane (c % TPP) over trees 0..T-1
    // (tree t's leaf row lives in lane t % TPP, slot t / TPP)
    const int lane = threadIdx.x & 63;
    const int lane_base = lane & ~(TPP - 1);
    for (int c0 = 0; c0 < S; c0 += TPP) {
        const int c = c0 + sub;
        float acc = 0.f;
        for (int t = 0; t < n_trees; t++) {
            const int slot = t >> LOG2_TPP;
            // all lanes take part in the shuffle; slot is uniform across the wave
            int row = 0;
#pragma unroll
            for (int k = 0; k < (64 / TPP > 0 ? 64 / TPP : 1); k++)
                if (k == slot) row = leaf_rows[k < n_mine ? k : 0];
            row = __shfl(row, lane_base + (t & (TPP - 1)), 64);
            if (c < S) {
                const float h = hist[(size_t)row * S + c];
                acc = (t == 0) ? h : acc + h;
            }
        }
        if (active && c < S) out[(size_t)point * S + c] = acc;
    }
}

void launch_forest_eval(const DeviceForest& f, const float* d_X, int P, int D, float* d_out,
                        hipStream_t s) {
    if (P <= 0) return;
    const DeviceNode* nodes = f.nodes.as<DeviceNode>();
    const int32_t* roots = f.roots.as<int32_t>();
    const float* hist = f.hist.as<float>();
    const int block = 256;
    if (f.n_trees <= 4) {
        const long threads = (long)P * 4;
        forest_eval_kernel<2><<<dim3((unsigned)((threads + block - 1) / block)), dim3(block), 0, s>>>(
            nodes, roots, hist, f.n_trees, f.sum_classes, d_X, P, D, d_out);
    } else {
        const long threads = (long)P * 16;
        forest_eval_kernel<4><<<dim3((unsigned)((threads + block - 1) / block)), dim3(block), 0, s>>>(
            nodes, roots, hist, f.n_trees, f.sum_classes, d_X, P, D, d_out);
    }
    RV_LAUNCHED("forest_eval_kernel");
}

}  // namespace rvseg

// =============================================================================================
// Frame path: fused per-point feature vector + forest traversal over the stride grid
// (replaces FeatureExtractor::extract + the per-point multiClassLogPosterior loop,
// include/feature_extractor.h:125-197 and src/segmenter.cpp:351-376).
//
// A wavefront owns 16 consecutive sample points.  Their 363-byte Lab patch vectors are built
// directly in LDS (the reference materialises P x 366 floats on the heap; here features never
// touch HBM), then the 64 lanes become (point, tree) pairs and walk the breadth-first node array,
// fetching one 16-byte node per level from HBM/L2 and the tested feature byte from LDS.
// =============================================================================================
namespace rvseg {

constexpr int PPW = 16;              // points per wave
constexpr int WAVES_PER_BLOCK = 4;

struct ClassMap {
    int S, n_layers;
    int layer_base[64];  // offset (in floats) of the class's layer block inside a frame's low-res buffer
    int layer_C[64];     // class count of that layer
    int cl[64];          // class index inside the layer
};

__device__ __forceinline__ int reflect_idx(int p, int len) {  // BORDER_REFLECT: fedcba|abcdefgh|hgfedcb
    return p < 0 ? -p - 1 : (p >= len ? 2 * len - p - 1 : p);
}

template <bool DUMP>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK)
rf_frames_kernel(FrameGeom g, ClassMap cm, const DeviceNode* __restrict__ nodes, const int32_t* __restrict__ roots,
                 const float* __restrict__ hist, int n_trees, const ResizeRow* __restrict__ rt,
                 const uint32_t* __restrict__ lab_all, const uint16_t* __restrict__ depth_all,
                 const float4* __restrict__ cloud_all, const float* __restrict__ nfeat_all,
                 float* __restrict__ low_all, float* __restrict__ dump_all, uint8_t* __restrict__ valid_all,
                 int n_points_total, int fb_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // per wave: PPW x fb_stride bytes of patch features, then PPW x 8 ints of per-point state
    unsigned char* fb = smem + (size_t)wave * (PPW * fb_stride + PPW * 32);
    int* pst = reinterpret_cast<int*>(fb + PPW * fb_stride);  // [PPW][8]: valid,half,x,y,depth_f,height_f,normal_f,frame
    const int per_frame = g.lw * g.lh;
    const int base = (blockIdx.x * WAVES_PER_BLOCK + wave) * PPW;
    const int W = g.W, H = g.H;

    // ---- phase A: per-point state (mask rule feature_extractor.h:60; half size :139-140)
    if (lane < PPW) {
        const int pid = base + lane;
        int valid = 0, half = 0, x = 0, y = 0, frame = 0;
        float depth_m = 0.f, height = 0.f, nrm = 0.f;
        if (pid < n_points_total) {
            frame = pid / per_frame;
            const int p = pid - frame * per_frame;
            const int ly = p / g.lw, lx = p - ly * g.lw;
            y = ly * g.stride; x = lx * g.stride;
            const size_t pix = (size_t)frame * W * H + (size_t)y * W + x;
            const uint16_t d = depth_all[pix];
            const float dv = (float)d;
            valid = (dv >= g.dmin_mm && dv <= g.dmax_mm) ? 1 : 0;
            depth_m = dv / 1000.0f;
            if (valid) {
                half = (int)((double)g.patch_size / (2.0 * (double)depth_m));
                if (g.pos_height >= 0) height = cloud_all[pix].z;
                if (g.pos_normal >= 0) nrm = nfeat_all[pid];
            }
        }
        int* s = pst + lane * 8;
        s[0] = valid; s[1] = half; s[2] = x; s[3] = y;
        s[4] = __float_as_int(depth_m); s[5] = __float_as_int(height); s[6] = __float_as_int(nrm); s[7] = frame;
    }
    __builtin_amdgcn_wave_barrier();
    __syncthreads();

    // ---- phase B: Lab patch, cv::resize(ROI -> r x r) in 11-bit fixed point (feature_extractor.h:142)
    if (g.n_patch > 0) {
        const int rr = g.r * g.r;
        for (int k0 = 0; k0 < rr; k0 += 64) {
            const int k = k0 + lane;
            const int dy = k / g.r, dx = k - dy * g.r;
            for (int pt = 0; pt < PPW; pt++) {
                const int* s = pst + pt * 8;
                if (!s[0] || k >= rr) continue;
                const int half = s[1], size = 2 * half + 1;
                const int x0 = s[2] - half, y0 = s[3] - half;
                // one 8-byte record per axis: {source index, weight of it, weight of the next}
                const uint2 xr = *reinterpret_cast<const uint2*>(&rt[half].x[dx]);
                const uint2 yr = *reinterpret_cast<const uint2*>(&rt[half].y[dy]);
                const int sx0 = (int)(short)(xr.x & 0xffffu), ia0 = (int)(short)(xr.x >> 16), ia1 = (int)(short)(xr.y & 0xffffu);
                const int ib0 = (int)(short)(yr.x >> 16), ib1 = (int)(short)(yr.y & 0xffffu);
                const int sx1 = (int)(short)(xr.y >> 16);
                const int sy0 = (int)(short)(yr.x & 0xffffu), sy1 = (int)(short)(yr.y >> 16);
                (void)size;
                const int rx0 = reflect_idx(x0 + sx0, W), rx1 = reflect_idx(x0 + sx1, W);
                const int ry0 = reflect_idx(y0 + sy0, H), ry1 = reflect_idx(y0 + sy1, H);
                const uint32_t* lab = lab_all + (size_t)s[7] * W * H;
                // the two taps of a row are always at most one pixel apart (also across the mirrored
                // border): fetch them with one 8-byte load and pick
                int xb = rx0 < rx1 ? rx0 : rx1;
                xb = xb < W - 1 ? xb : W - 2;
                typedef uint32_t u32x2_u __attribute__((ext_vector_type(2), aligned(4)));
                const u32x2_u q0 = *reinterpret_cast<const u32x2_u*>(lab + (size_t)ry0 * W + xb);
                const u32x2_u q1 = *reinterpret_cast<const u32x2_u*>(lab + (size_t)ry1 * W + xb);
                const uint32_t p00 = rx0 == xb ? q0.x : q0.y, p01 = rx1 == xb ? q0.x : q0.y;
                const uint32_t p10 = rx0 == xb ? q1.x : q1.y, p11 = rx1 == xb ? q1.x : q1.y;
                unsigned char* dst = fb + pt * fb_stride + k * 3;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const int sh = 8 * c;
                    const int r0 = (int)((p00 >> sh) & 255u) * ia0 + (int)((p01 >> sh) & 255u) * ia1;
                    const int r1 = (int)((p10 >> sh) & 255u) * ia0 + (int)((p11 >> sh) & 255u) * ia1;
                    int v = (((ib0 * (r0 >> 4)) >> 16) + ((ib1 * (r1 >> 4)) >> 16) + 2) >> 2;
                    v = v < 0 ? 0 : (v > 255 ? 255 : v);
                    dst[c] = (unsigned char)v;
                }
            }
        }
    }
    __syncthreads();

    if (DUMP) {
        for (int pt = 0; pt < PPW; pt++) {
            const int pid = base + pt;
            if (pid >= n_points_total) break;
            const int* s = pst + pt * 8;
            if (lane == 0) valid_all[pid] = (uint8_t)s[0];
            float* d = dump_all + (size_t)pid * g.D;
            for (int k = lane; k < g.n_patch; k += 64) d[k] = s[0] ? (float)fb[pt * fb_stride + k] : 0.f;
            if (lane == 0) {
                if (g.pos_depth >= 0) d[g.pos_depth] = __int_as_float(s[4]);
                if (g.pos_height >= 0) d[g.pos_height] = __int_as_float(s[5]);
                if (g.pos_normal >= 0) d[g.pos_normal] = __int_as_float(s[6]);
            }
        }
        return;
    }

    // ---- phase C: lanes = (point, tree); findLeafNode (classifier.cpp:97-117)
    const int pt = lane >> 2, sub = lane & 3;
    const int* s = pst + pt * 8;
    const bool valid = s[0] != 0;
    const unsigned char* myfb = fb + pt * fb_stride;
    int leaf_rows[16];
    int n_mine = 0;
    const int4* np = reinterpret_cast<const int4*>(nodes);
    for (int t = sub; t < n_trees; t += 4) {
        int row = 0;
        if (valid) {
            int4 nd = np[roots[t]];
            while (nd.z != 0) {
                const int f = nd.x;
                float v;
                if (f < g.n_patch) v = (float)myfb[f];
                else v = __int_as_float(f == g.pos_depth ? s[4] : (f == g.pos_height ? s[5] : s[6]));
                nd = np[(v < __int_as_float(nd.y)) ? nd.z : nd.z + 1];
            }
            row = nd.w;
        }
#pragma unroll
        for (int q = 0; q < 16; q++) if (q == n_mine) leaf_rows[q] = row;
        n_mine++;
    }

    // ---- phase D: tree-order accumulation (classifier.cpp:193-206) and scatter into the low-res
    //      image at (y/stride, x/stride) (segmenter.cpp:369-375); invalid cells get the fill value
    const int lane_base = lane & ~3;
    const int pid = base + pt;
    const int frame = s[7];
    const int p = pid - frame * per_frame;
    for (int c0 = 0; c0 < cm.S; c0 += 4) {
        const int c = c0 + sub;
        float acc = 0.f;
        for (int t = 0; t < n_trees; t++) {
            const int slot = t >> 2;
            int row = 0;
#pragma unroll
            for (int q = 0; q < 16; q++) if (q == slot) row = leaf_rows[q];
            row = __shfl(row, lane_base + (t & 3), 64);
            if (c < cm.S && valid) {
                const float h = hist[(size_t)row * cm.S + c];
                acc = (t == 0) ? h : acc + h;
            }
        }
        if (pid < n_points_total && c < cm.S) {
            float* low = low_all + (size_t)frame * per_frame * cm.S;
            low[cm.layer_base[c] + (size_t)p * cm.layer_C[c] + cm.cl[c]] = valid ? acc : g.fill;
        }
    }
}

// =============================================================================================
// Frame path, on-demand features (the default).  PMC on the kernel above: vector-ALU bound, and 85 %
// of it is phase B, which resizes all 121 x 3 patch values of every point although a walk through
// T trees of depth <= 31 tests only ~20 per tree.  Here the lanes are (point, tree) pairs from the
// start and a patch value is computed when a node asks for it: one channel of one cv::resize cell
// = 2 LDS weight records + two 8-byte tap loads + ~25 integer ops -- the same arithmetic as phase B.
// A node's two children are adjacent, so both are fetched together with the taps of the node's
// own test: still one memory round trip per level.
// COMPACT (round 3; PMC of the 16-byte-node version: texture addresser 92 % busy, 4 vector-memory instructions per
// tree level): 8-byte nodes, so that BOTH children of a node arrive with one 16-byte load, and the row-pair Lab image
// `lab2` ({lab(y, x), lab(y + 1, x)} per pixel), so that the 2 x 2 tap quad of a patch value is one 16-byte load as
// well: two vector-memory instructions per level instead of four.  The arithmetic is unchanged.
// =============================================================================================
template <bool COMPACT>
__global__ void __launch_bounds__(256)
rf_frames_lazy_kernel(FrameGeom g, ClassMap cm, const DeviceNode* __restrict__ nodes, const int32_t* __restrict__ roots,
                      const float* __restrict__ hist, int n_trees, const ResizeRow* __restrict__ rt, int rt_in_lds,
                      const uint32_t* __restrict__ lab_all, const uint16_t* __restrict__ depth_all,
                      const float4* __restrict__ cloud_all, const float* __restrict__ nfeat_all,
                      float* __restrict__ low_all, int n_points_total, const uint32_t* __restrict__ nodes8,
                      const uint32_t* __restrict__ lab2_all) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const ResizeRow* rtl = rt;
    if (rt_in_lds) {
        // weight records of every ROI size, 256 B per row
        uint4* dst = reinterpret_cast<uint4*>(smem);
        const uint4* src = reinterpret_cast<const uint4*>(rt);
        for (int i = threadIdx.x; i < g.rt_rows * (int)(sizeof(ResizeRow) / 16); i += 256) dst[i] = src[i];
        __syncthreads();
        rtl = reinterpret_cast<const ResizeRow*>(smem);
    }
    const int lane = threadIdx.x & 63;
    const int per_frame = g.lw * g.lh;
    const int base = (blockIdx.x * 4 + (threadIdx.x >> 6)) * PPW;
    const int W = g.W, H = g.H;
    const int pt = lane >> 2, sub = lane & 3;
    const int pid = base + pt;

    // per-point state (mask rule feature_extractor.h:60; half size :139-140), once per lane
    bool valid = false;
    int half = 0, x = 0, y = 0, frame = 0;
    float depth_m = 0.f, height = 0.f, nrm = 0.f;
    if (pid < n_points_total) {
        frame = pid / per_frame;
        const int p = pid - frame * per_frame;
        const int ly = p / g.lw, lx = p - ly * g.lw;
        y = ly * g.stride; x = lx * g.stride;
        const size_t pix = (size_t)frame * W * H + (size_t)y * W + x;
        const float dv = (float)depth_all[pix];
        valid = dv >= g.dmin_mm && dv <= g.dmax_mm;
        depth_m = dv / 1000.0f;
        if (valid) {
            half = (int)((double)g.patch_size / (2.0 * (double)depth_m));
            if (g.pos_height >= 0) height = cloud_all[pix].z;
            if (g.pos_normal >= 0) nrm = nfeat_all[pid];
        }
    }
    const int size = 2 * half + 1, x0 = x - half, y0 = y - half;
    const uint32_t* lab = lab_all + (size_t)frame * W * H;
    const ResizeRow* myrt = rtl + (valid ? half : 0);
    typedef uint32_t u32x2_u __attribute__((ext_vector_type(2), aligned(4)));

    // one value of the r x r x 3 patch vector: cv::resize(ROI -> r x r) in 11-bit fixed point
    // (feature_extractor.h:142), exactly phase B of the kernel above for cell k, channel c
    auto patch_value = [&](int packed) -> float {   // packed = channel << 16 | dy << 8 | dx (upload_forest)
        const int c = packed >> 16, dy = (packed >> 8) & 255, dx = packed & 255;
        const uint2 xr = *reinterpret_cast<const uint2*>(&myrt->x[dx]);
        const uint2 yr = *reinterpret_cast<const uint2*>(&myrt->y[dy]);
        const int sx0 = (int)(short)(xr.x & 0xffffu), ia0 = (int)(short)(xr.x >> 16), ia1 = (int)(short)(xr.y & 0xffffu);
        const int ib0 = (int)(short)(yr.x >> 16), ib1 = (int)(short)(yr.y & 0xffffu);
        const int sx1 = (int)(short)(xr.y >> 16);
        const int sy0 = (int)(short)(yr.x & 0xffffu), sy1 = (int)(short)(yr.y >> 16);
        const int rx0 = reflect_idx(x0 + sx0, W), rx1 = reflect_idx(x0 + sx1, W);
        const int ry0 = reflect_idx(y0 + sy0, H), ry1 = reflect_idx(y0 + sy1, H);
        int xb = rx0 < rx1 ? rx0 : rx1;
        xb = xb < W - 1 ? xb : W - 2;
        const u32x2_u q0 = *reinterpret_cast<const u32x2_u*>(lab + (size_t)ry0 * W + xb);
        const u32x2_u q1 = *reinterpret_cast<const u32x2_u*>(lab + (size_t)ry1 * W + xb);
        const uint32_t p00 = rx0 == xb ? q0.x : q0.y, p01 = rx1 == xb ? q0.x : q0.y;
        const uint32_t p10 = rx0 == xb ? q1.x : q1.y, p11 = rx1 == xb ? q1.x : q1.y;
        const int sh = 8 * c;
        const int r0 = (int)((p00 >> sh) & 255u) * ia0 + (int)((p01 >> sh) & 255u) * ia1;
        const int r1 = (int)((p10 >> sh) & 255u) * ia0 + (int)((p11 >> sh) & 255u) * ia1;
        int v = (((ib0 * (r0 >> 4)) >> 16) + ((ib1 * (r1 >> 4)) >> 16) + 2) >> 2;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        return (float)v;
    };
    // The same value when the ROI of every point of the wave lies strictly inside the image (no mirrored border: all
    // but a ~25-pixel frame of the sample points at the bench's depths): the taps are at (y0 + sy, x0 + sx) directly and
    // the second column is the first one's right neighbour or itself.  PMC: this kernel is VALU bound (64 % of the
    // issue cycles, 83 vector instructions per tree level); the four reflections and the pair selection were ~30 of them.
    const int lab_base = y0 * W + x0;
    auto patch_value_inside = [&](int packed) -> float {
        const int c = packed >> 16, dy = (packed >> 8) & 255, dx = packed & 255;
        const uint2 xr = *reinterpret_cast<const uint2*>(&myrt->x[dx]);
        const uint2 yr = *reinterpret_cast<const uint2*>(&myrt->y[dy]);
        const int sx0 = (int)(short)(xr.x & 0xffffu), ia0 = (int)(short)(xr.x >> 16), ia1 = (int)(short)(xr.y & 0xffffu);
        const int ib0 = (int)(short)(yr.x >> 16), ib1 = (int)(short)(yr.y & 0xffffu);
        const bool two = (int)(short)(xr.y >> 16) != sx0;
        const int sy0 = (int)(short)(yr.x & 0xffffu), sy1 = (int)(short)(yr.y >> 16);
        const u32x2_u q0 = *reinterpret_cast<const u32x2_u*>(lab + (lab_base + sy0 * W + sx0));
        const u32x2_u q1 = *reinterpret_cast<const u32x2_u*>(lab + (lab_base + sy1 * W + sx0));
        const int sh = 8 * c;
        const int a00 = (int)((q0.x >> sh) & 255u), a01 = (int)(((two ? q0.y : q0.x) >> sh) & 255u);
        const int a10 = (int)((q1.x >> sh) & 255u), a11 = (int)(((two ? q1.y : q1.x) >> sh) & 255u);
        const int r0 = a00 * ia0 + a01 * ia1;
        const int r1 = a10 * ia0 + a11 * ia1;
        int v = (((ib0 * (r0 >> 4)) >> 16) + ((ib1 * (r1 >> 4)) >> 16) + 2) >> 2;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        return (float)v;
    };
    // the same value from the row-pair image: one 16-byte load {top(x), bottom(x), top(x + 1), bottom(x + 1)}; a record's
    // second row / column is the first one or its neighbour (pipeline_init: ofs1 = ofs or ofs + 1)
    typedef uint32_t u32x4_p __attribute__((ext_vector_type(4), aligned(8)));
    const uint32_t* lab2 = COMPACT ? lab2_all + 2 * (size_t)frame * W * H : nullptr;
    auto patch_value_inside2 = [&](unsigned cell) -> float {   // cell = channel << 8 | dy << 4 | dx
        const int c = (int)(cell >> 8) & 3, dy = (int)(cell >> 4) & 15, dx = (int)cell & 15;
        const uint2 xr = *reinterpret_cast<const uint2*>(&myrt->x[dx]);
        const uint2 yr = *reinterpret_cast<const uint2*>(&myrt->y[dy]);
        const int sx0 = (int)(short)(xr.x & 0xffffu), ia0 = (int)(short)(xr.x >> 16), ia1 = (int)(short)(xr.y & 0xffffu);
        const int ib0 = (int)(short)(yr.x >> 16), ib1 = (int)(short)(yr.y & 0xffffu);
        const bool two = (int)(short)(xr.y >> 16) != sx0;
        const int sy0 = (int)(short)(yr.x & 0xffffu);
        const bool below = (int)(short)(yr.y >> 16) != sy0;
        const u32x4_p q = *reinterpret_cast<const u32x4_p*>(lab2 + 2 * (lab_base + sy0 * W + sx0));
        const int sh = 8 * c;
        const uint32_t t0 = q.x, b0 = below ? q.y : q.x, t1 = two ? q.z : q.x, b1 = below ? (two ? q.w : q.y) : (two ? q.z : q.x);
        const int a00 = (int)((t0 >> sh) & 255u), a01 = (int)((t1 >> sh) & 255u);
        const int a10 = (int)((b0 >> sh) & 255u), a11 = (int)((b1 >> sh) & 255u);
        const int r0 = a00 * ia0 + a01 * ia1;
        const int r1 = a10 * ia0 + a11 * ia1;
        int v = (((ib0 * (r0 >> 4)) >> 16) + ((ib1 * (r1 >> 4)) >> 16) + 2) >> 2;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        return (float)v;
    };
    // strictly inside: the 8-byte tap load at column x0 + sx0 <= x0 + size - 1 must not be the image's last column
    const bool roi_inside = x0 >= 0 && y0 >= 0 && x0 + size < W && y0 + size <= H;
    const bool wave_inside = __ballot(valid && !roi_inside) == 0ull;

    // lanes = (point, tree); findLeafNode (classifier.cpp:97-117)
    int leaf_rows[16];
    int n_mine = 0;
    const int4* np = reinterpret_cast<const int4*>(nodes);
    for (int t = sub; t < n_trees; t += 4) {
        int row = 0;
        if (valid && COMPACT) {
            const uint2* n8 = reinterpret_cast<const uint2*>(nodes8);
            uint2 nd = n8[roots[t]];
            while ((nd.y & 0xFFFFFu) != 0u) {
                // both children in one load, in flight while the node's own test is evaluated
                const u32x4_p ch = *reinterpret_cast<const u32x4_p*>(nodes8 + 2 * (size_t)(nd.y & 0xFFFFFu));
                const unsigned kind = nd.y >> 30, cell = (nd.y >> 20) & 0x3FFu;
                float v;
                if (kind == 0u) v = wave_inside ? patch_value_inside2(cell) : patch_value((int)(((cell >> 8) & 3u) << 16 | ((cell >> 4) & 15u) << 8 | (cell & 15u)));
                else v = kind == 1u ? depth_m : (kind == 2u ? height : nrm);
                const bool go_left = v < __uint_as_float(nd.x);
                nd = go_left ? make_uint2(ch.x, ch.y) : make_uint2(ch.z, ch.w);
            }
            row = (int)nd.x;
        } else if (valid) {
            int4 nd = np[roots[t]];
            while (nd.z != 0) {
                // both children travel while the node's own test is evaluated
                const int4 c0 = np[nd.z], c1 = np[nd.z + 1];
                const int f = nd.x;
                float v;
                if (f < g.n_patch) v = wave_inside ? patch_value_inside(nd.w) : patch_value(nd.w);
                else v = f == g.pos_depth ? depth_m : (f == g.pos_height ? height : nrm);
                nd = (v < __int_as_float(nd.y)) ? c0 : c1;
            }
            row = nd.w;
        }
#pragma unroll
        for (int q = 0; q < 16; q++) if (q == n_mine) leaf_rows[q] = row;
        n_mine++;
    }

    // tree-order accumulation (classifier.cpp:193-206) and scatter into the low-res image at
    // (y/stride, x/stride) (segmenter.cpp:369-375); invalid cells get the fill value
    const int lane_base = lane & ~3;
    const int p = pid - frame * per_frame;
    for (int c0 = 0; c0 < cm.S; c0 += 4) {
        const int c = c0 + sub;
        const int cc = c < cm.S ? c : cm.S - 1;
        float acc = 0.f;
        // four leaf rows at a time: the loads are unconditional (invalid points read row 0) and go out
        // together; the adds stay in tree order
        for (int t0 = 0; t0 < n_trees; t0 += 4) {
            float h[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int t = t0 + u < n_trees ? t0 + u : n_trees - 1;
                const int slot = t >> 2;
                int row = 0;
#pragma unroll
                for (int q = 0; q < 16; q++) if (q == slot) row = leaf_rows[q];
                row = __shfl(row, lane_base + (t & 3), 64);
                h[u] = hist[(size_t)row * cm.S + cc];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (t0 + u < n_trees) acc = (t0 + u == 0) ? h[u] : acc + h[u];
            }
        }
        if (pid < n_points_total && c < cm.S) {
            float* low = low_all + (size_t)frame * per_frame * cm.S;
            low[cm.layer_base[c] + (size_t)p * cm.layer_C[c] + cm.cl[c]] = valid ? acc : g.fill;
        }
    }
}

static bool n_trees_ok(int n_trees) { return n_trees >= 1 && n_trees <= 64; }   // leaf_rows[16] x 4 lanes per point

bool rf_frames_wants_lab2(const DeviceForest& f) { return f.nodes8.p != nullptr && n_trees_ok(f.n_trees); }

void launch_rf_frames(const FrameGeom& g, const DeviceForest& f, const ResizeRow* d_rt, const uint32_t* d_lab,
                      const uint16_t* d_depth, const float4* d_cloud, const float* d_nfeat, float* d_low,
                      float* d_dump, uint8_t* d_valid, int n, hipStream_t s, const uint2* d_lab2) {
    ClassMap cm{};
    cm.S = f.sum_classes;
    cm.n_layers = f.n_layers;
    const int per_frame = g.lw * g.lh;
    int c = 0, prefix = 0;
    for (int l = 0; l < f.n_layers; l++) {
        for (int k = 0; k < f.class_counts[l]; k++, c++) {
            cm.layer_base[c] = per_frame * prefix;
            cm.layer_C[c] = f.class_counts[l];
            cm.cl[c] = k;
        }
        prefix += f.class_counts[l];
    }
    const int total = per_frame * n;
    const int fb_stride = ((g.n_patch + 3) / 4) * 4 + 4;
    const size_t smem = (size_t)WAVES_PER_BLOCK * (PPW * fb_stride + PPW * 32);
    const int pts_per_block = PPW * WAVES_PER_BLOCK;
    const dim3 grid((unsigned)((total + pts_per_block - 1) / pts_per_block)), block(64 * WAVES_PER_BLOCK);
    if (!d_dump && n_trees_ok(f.n_trees)) {
        const size_t rt_bytes = (size_t)g.rt_rows * sizeof(ResizeRow);
        const int in_lds = g.n_patch > 0 && rt_bytes <= 40 * 1024 ? 1 : 0;   // else the records come through L1
        if (d_lab2 && rf_frames_wants_lab2(f))
            rf_frames_lazy_kernel<true><<<grid, dim3(256), in_lds ? rt_bytes : 0, s>>>(
                g, cm, f.nodes.as<DeviceNode>(), f.roots.as<int32_t>(), f.hist.as<float>(), f.n_trees, d_rt, in_lds, d_lab, d_depth,
                d_cloud, d_nfeat, d_low, total, f.nodes8.as<uint32_t>(), reinterpret_cast<const uint32_t*>(d_lab2));
        else
            rf_frames_lazy_kernel<false><<<grid, dim3(256), in_lds ? rt_bytes : 0, s>>>(
                g, cm, f.nodes.as<DeviceNode>(), f.roots.as<int32_t>(), f.hist.as<float>(), f.n_trees, d_rt, in_lds, d_lab, d_depth,
                d_cloud, d_nfeat, d_low, total, nullptr, nullptr);
        RV_LAUNCHED("rf_frames_lazy_kernel");
        return;
    }
    if (d_dump)
        rf_frames_kernel<true><<<grid, block, smem, s>>>(g, cm, f.nodes.as<DeviceNode>(), f.roots.as<int32_t>(), f.hist.as<float>(),
                                                         f.n_trees, d_rt, d_lab, d_depth, d_cloud, d_nfeat, d_low, d_dump, d_valid, total, fb_stride);
    else
        rf_frames_kernel<false><<<grid, block, smem, s>>>(g, cm, f.nodes.as<DeviceNode>(), f.roots.as<int32_t>(), f.hist.as<float>(),
                                                          f.n_trees, d_rt, d_lab, d_depth, d_cloud, d_nfeat, d_low, d_dump, d_valid, total, fb_stride);
    RV_LAUNCHED("rf_frames_kernel");
}

// =============================================================================================
// cv::resize(result, Size(W,H)) INTER_LINEAR on CV_32FC(C) + pack (segmenter.cpp:380-431).
// One thread per output pixel, all C classes of the layer in registers: four low-res rows in
// (wide dword-aligned loads), one full-res row out.
// =============================================================================================
template <int C>
__global__ void __launch_bounds__(256)
upsample_pack_kernel(int W, int H, int lw, int lh, size_t low_frame_stride, size_t low_layer_off,
                     size_t post_frame_stride, size_t post_layer_off, const int* __restrict__ xofs,
                     const float* __restrict__ ax0, const float* __restrict__ ax1, const int* __restrict__ yofs,
                     const float* __restrict__ ay0, const float* __restrict__ ay1, const float* __restrict__ low_all,
                     float* __restrict__ post_all, size_t total_pixels) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total_pixels) return;
    const size_t per_frame = (size_t)W * H;
    const int frame = (int)(gid / per_frame);
    const int pix = (int)(gid - (size_t)frame * per_frame);
    const int y = pix / W, x = pix - y * W;
    const float* low = low_all + (size_t)frame * low_frame_stride + low_layer_off;
    int sy0 = yofs[y], sy1 = sy0 + 1;
    sy0 = sy0 < 0 ? 0 : (sy0 >= lh ? lh - 1 : sy0);
    sy1 = sy1 < 0 ? 0 : (sy1 >= lh ? lh - 1 : sy1);
    const int sx0 = xofs[x];
    const bool tail = sx0 + 1 >= lw;   // dx >= xmax: D = S[sx]*ONE
    const int sx1 = tail ? sx0 : sx0 + 1;
    float s00[C], s01[C], s10[C], s11[C], o[C];
    load_row<C>(low + ((size_t)sy0 * lw + sx0) * C, s00);
    load_row<C>(low + ((size_t)sy0 * lw + sx1) * C, s01);
    load_row<C>(low + ((size_t)sy1 * lw + sx0) * C, s10);
    load_row<C>(low + ((size_t)sy1 * lw + sx1) * C, s11);
    const float a0 = ax0[x], a1 = ax1[x], b0 = ay0[y], b1 = ay1[y];
#pragma unroll
    for (int c = 0; c < C; c++) {
        float h0, h1;
        if (tail) { h0 = s00[c] * 1.f; h1 = s10[c] * 1.f; }
        else { h0 = s00[c] * a0 + s01[c] * a1; h1 = s10[c] * a0 + s11[c] * a1; }
        o[c] = h0 * b0 + h1 * b1;
    }
    store_row<C>(post_all + (size_t)frame * post_frame_stride + post_layer_off + (size_t)pix * C, o);
}

// The same through LDS: a block owns UP_TX x UP_TY output pixels; the low-resolution rows they interpolate between
// ((UP_TX / scale + 2) x (UP_TY / scale + 2) points of C floats) come in once, coalesced, instead of four 36-byte rows
// per pixel through the texture path (PMC of the kernel above: TA 93 % busy at 2.2 TB/s of useful bytes).
constexpr int UP_TX = 64, UP_TY = 4;
template <int C>
__global__ void __launch_bounds__(UP_TX * UP_TY)
upsample_pack_tiled_kernel(int W, int H, int lw, int lh, size_t low_frame_stride, size_t low_layer_off,
                           size_t post_frame_stride, size_t post_layer_off, const int* __restrict__ xofs,
                           const float* __restrict__ ax0, const float* __restrict__ ax1, const int* __restrict__ yofs,
                           const float* __restrict__ ay0, const float* __restrict__ ay1, const float* __restrict__ low_all,
                           float* __restrict__ post_all, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) float up_tile[];   // [rows][cols][C]
    const int tile = blockIdx.x % (tiles_x * tiles_y), frame = blockIdx.x / (tiles_x * tiles_y);
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x_first = tx * UP_TX, y_first = ty * UP_TY;
    auto clampy = [&](int v) { return v < 0 ? 0 : (v >= lh ? lh - 1 : v); };
    // source window of the tile (the offset tables are non-decreasing)
    const int sxa = xofs[x_first];
    int sxb = xofs[x_first + UP_TX - 1] + 1;
    sxb = sxb >= lw ? lw - 1 : sxb;
    const int sya = clampy(yofs[y_first]), syb = clampy(yofs[y_first + UP_TY - 1] + 1);
    const int ncols = sxb - sxa + 1, nrows = syb - sya + 1;
    const float* low = low_all + (size_t)frame * low_frame_stride + low_layer_off;
    const int row_floats = ncols * C;
    for (int idx = threadIdx.x; idx < nrows * row_floats; idx += UP_TX * UP_TY) {
        const int r = idx / row_floats, e = idx - r * row_floats;
        up_tile[idx] = low[((size_t)(sya + r) * lw + sxa) * C + e];
    }
    __syncthreads();
    const int x = x_first + (int)(threadIdx.x % UP_TX), y = y_first + (int)(threadIdx.x / UP_TX);
    const int sy0 = clampy(yofs[y]), sy1 = clampy(yofs[y] + 1);
    const int sx0 = xofs[x];
    const bool tail = sx0 + 1 >= lw;   // dx >= xmax: D = S[sx]*ONE
    const int sx1 = tail ? sx0 : sx0 + 1;
    const float* r0 = up_tile + (size_t)(sy0 - sya) * row_floats;
    const float* r1 = up_tile + (size_t)(sy1 - sya) * row_floats;
    float o[C];
    const float a0 = ax0[x], a1 = ax1[x], b0 = ay0[y], b1 = ay1[y];
#pragma unroll
    for (int c = 0; c < C; c++) {
        const float s00 = r0[(sx0 - sxa) * C + c], s01 = r0[(sx1 - sxa) * C + c];
        const float s10 = r1[(sx0 - sxa) * C + c], s11 = r1[(sx1 - sxa) * C + c];
        float h0, h1;
        if (tail) { h0 = s00 * 1.f; h1 = s10 * 1.f; }
        else { h0 = s00 * a0 + s01 * a1; h1 = s10 * a0 + s11 * a1; }
        o[c] = h0 * b0 + h1 * b1;
    }
    // out through LDS as well: a tile row is UP_TX * C contiguous floats in memory, written 16 bytes per lane
    __syncthreads();
#pragma unroll
    for (int c = 0; c < C; c++) up_tile[threadIdx.x * C + c] = o[c];
    __syncthreads();
    static_assert((UP_TX * C) % 4 == 0, "float4 rows");
    constexpr int ROW4 = UP_TX * C / 4;
    float* post = post_all + (size_t)frame * post_frame_stride + post_layer_off;
    for (int idx = threadIdx.x; idx < UP_TY * ROW4; idx += UP_TX * UP_TY) {
        const int r = idx / ROW4, e4 = idx - r * ROW4;
        *reinterpret_cast<float4*>(post + ((size_t)(y_first + r) * W + x_first) * C + 4 * e4) =
            *reinterpret_cast<const float4*>(up_tile + (size_t)r * UP_TX * C + 4 * e4);
    }
}

// generic class count: one thread per output float
__global__ void __launch_bounds__(256)
upsample_pack_generic_kernel(int W, int H, int lw, int lh, int C, size_t low_frame_stride, size_t low_layer_off,
                     size_t post_frame_stride, size_t post_layer_off, const int* __restrict__ xofs,
                     const float* __restrict__ ax0, const float* __restrict__ ax1, const int* __restrict__ yofs,
                     const float* __restrict__ ay0, const float* __restrict__ ay1, const float* __restrict__ low_all,
                     float* __restrict__ post_all, size_t total) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const size_t per_frame = (size_t)W * H * C;
    const int frame = (int)(gid / per_frame);
    const size_t e = gid - (size_t)frame * per_frame;
    const int c = (int)(e % C);
    const int pix = (int)(e / C);
    const int y = pix / W, x = pix - y * W;
    const float* low = low_all + (size_t)frame * low_frame_stride + low_layer_off;
    int sy0 = yofs[y], sy1 = sy0 + 1;
    sy0 = sy0 < 0 ? 0 : (sy0 >= lh ? lh - 1 : sy0);
    sy1 = sy1 < 0 ? 0 : (sy1 >= lh ? lh - 1 : sy1);
    const int sx0 = xofs[x];
    const float* S0 = low + (size_t)sy0 * lw * C;
    const float* S1 = low + (size_t)sy1 * lw * C;
    float h0, h1;
    if (sx0 + 1 >= lw) {  // dx >= xmax: D = S[sx]*ONE
        h0 = S0[(size_t)sx0 * C + c] * 1.f;
        h1 = S1[(size_t)sx0 * C + c] * 1.f;
    } else {
        const float a0 = ax0[x], a1 = ax1[x];
        h0 = S0[(size_t)sx0 * C + c] * a0 + S0[(size_t)(sx0 + 1) * C + c] * a1;
        h1 = S1[(size_t)sx0 * C + c] * a0 + S1[(size_t)(sx0 + 1) * C + c] * a1;
    }
    post_all[(size_t)frame * post_frame_stride + post_layer_off + e] = h0 * ay0[y] + h1 * ay1[y];
}

void launch_upsample_pack(const FrameGeom& g, const DeviceForest& f, const UpsampleTables& t,
                          const float* d_low, float* d_post, int n, hipStream_t s) {
    const size_t low_frame = (size_t)g.lw * g.lh * f.sum_classes;
    const size_t post_frame = (size_t)g.W * g.H * f.sum_classes;
    int prefix = 0;
    for (int l = 0; l < f.n_layers; l++) {
        const int C = f.class_counts[l];
        const size_t pixels = (size_t)g.W * g.H * n;
        const dim3 pgrid((unsigned)((pixels + 255) / 256)), block(256);
#define RV_UP(CC)                                                                                                        \
    upsample_pack_kernel<CC><<<pgrid, block, 0, s>>>(g.W, g.H, g.lw, g.lh, low_frame, (size_t)g.lw * g.lh * prefix, post_frame, \
                                                     (size_t)g.W * g.H * prefix, t.xofs.as<int>(), t.ax0.as<float>(),          \
                                                     t.ax1.as<float>(), t.yofs.as<int>(), t.ay0.as<float>(), t.ay1.as<float>(), \
                                                     d_low, d_post, pixels)
        // the tiled kernel: up-sampling (lw <= W, lh <= H) of 8 / 9 classes on images its tiles divide
        if ((C == 9 || C == 8) && g.W % UP_TX == 0 && g.H % UP_TY == 0 && g.lw <= g.W && g.lh <= g.H) {
            const int tiles_x = g.W / UP_TX, tiles_y = g.H / UP_TY;
            const size_t lds = (size_t)(UP_TX + 2) * (UP_TY + 2) * C * sizeof(float);   // >= UP_TX * UP_TY * C floats of output
            const dim3 tgrid((unsigned)(tiles_x * tiles_y * n));
#define RV_UPT(CC)                                                                                                       \
    upsample_pack_tiled_kernel<CC><<<tgrid, dim3(UP_TX * UP_TY), lds, s>>>(g.W, g.H, g.lw, g.lh, low_frame, (size_t)g.lw * g.lh * prefix, \
        post_frame, (size_t)g.W * g.H * prefix, t.xofs.as<int>(), t.ax0.as<float>(), t.ax1.as<float>(), t.yofs.as<int>(),     \
        t.ay0.as<float>(), t.ay1.as<float>(), d_low, d_post, tiles_x, tiles_y)
            if (C == 9) RV_UPT(9); else RV_UPT(8);
#undef RV_UPT
            prefix += C;
            continue;
        }
        switch (C) {
            case 2: RV_UP(2); break; case 3: RV_UP(3); break; case 4: RV_UP(4); break; case 5: RV_UP(5); break;
            case 6: RV_UP(6); break; case 7: RV_UP(7); break; case 8: RV_UP(8); break; case 9: RV_UP(9); break;
            case 10: RV_UP(10); break; case 12: RV_UP(12); break; case 16: RV_UP(16); break;
            default: {
                const size_t total = pixels * C;
                upsample_pack_generic_kernel<<<dim3((unsigned)((total + 255) / 256)), block, 0, s>>>(
                    g.W, g.H, g.lw, g.lh, C, low_frame, (size_t)g.lw * g.lh * prefix, post_frame, (size_t)g.W * g.H * prefix,
                    t.xofs.as<int>(), t.ax0.as<float>(), t.ax1.as<float>(), t.yofs.as<int>(), t.ay0.as<float>(), t.ay1.as<float>(),
                    d_low, d_post, total);
            }
        }
#undef RV_UP
        prefix += C;
    }
    RV_LAUNCHED("upsample_pack_kernel");
}

// =============================================================================================
// label rules, one thread per point
// =============================================================================================
__global__ void __launch_bounds__(256)
labels_kernel(const float* __restrict__ values, size_t n_points, int C, int mode, int unknown, int8_t* __restrict__ labels) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_points) return;
    const float* v = values + i * C;
    const int best = label_rule(v, C, mode, unknown);
    labels[i] = (int8_t)best;
}

// labels of every frame and layer of a chunk: values in the posterior layout (frame stride
// sumC*N, layer blocks of N*C_l), labels as n x L x N
__global__ void __launch_bounds__(256)
labels_frames_kernel(const float* __restrict__ values, int n_frames, int N, int C, size_t frame_stride, size_t layer_off,
                     int n_layers, int layer, int mode, int unknown, int8_t* __restrict__ labels) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)n_frames * N) return;
    const size_t frame = gid / (size_t)N, i = gid - frame * (size_t)N;
    const float* v = values + frame * frame_stride + layer_off + i * C;
    const int best = label_rule(v, C, mode, unknown);
    labels[(frame * n_layers + layer) * (size_t)N + i] = (int8_t)best;
}

void launch_labels_frames(const float* d_values, int n_frames, int N, const DeviceForest& f, int mode, const int* unknown,
                          int8_t* d_labels, hipStream_t s) {
    const size_t total = (size_t)n_frames * N;
    int prefix = 0;
    for (int l = 0; l < f.n_layers; l++) {
        labels_frames_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(
            d_values, n_frames, N, f.class_counts[l], (size_t)N * f.sum_classes, (size_t)N * prefix, f.n_layers, l, mode,
            unknown[l], d_labels);
        prefix += f.class_counts[l];
    }
    RV_LAUNCHED("labels_frames_kernel");
}

void launch_labels(const float* d_values, size_t n_points, int C, int mode, int unknown, int8_t* d_labels, hipStream_t s) {
    if (n_points == 0) return;
    labels_kernel<<<dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, s>>>(d_values, n_points, C, mode, unknown, d_labels);
    RV_LAUNCHED("labels_kernel");
}

}  // namespace rvseg
