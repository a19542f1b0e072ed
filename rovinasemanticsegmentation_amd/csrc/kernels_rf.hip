// Random-forest traversal kernels for gfx950.
//
// Reference semantics (third-party/libforest/src/classifier.cpp):
//   findLeafNode  :97-117   node = x[feature] < threshold ? left : left + 1   (strict '<', fp32)
//   classLogPosterior / multiClassLogPosterior :166-208
//                           copy tree 0's leaf histogram, then += trees 1..T-1 IN ORDER (fp32)
// The node array is the breadth-first re-layout built by forest_model.cpp: 16-byte nodes, one
// dwordx4 load per visited node, the hot top levels of every tree packed at the front of each
// tree's block so that they stay in L1/L2.
#include "rvseg_internal.h"

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
}

}  // namespace rvseg
