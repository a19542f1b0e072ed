// Forest training on the GPU: replaces, for the forests this path evaluates,
//   RandomForestLearner::learn                 third-party/libforest/src/learning.cpp:1031-1073
//   DecisionTreeLearner::learn (multi-layer)   learning.cpp:410-662   (single layer: :663-915, same search)
//   updateMultiHistograms / updateHistograms   learning.cpp:918-1012
//   forest->write                              src/train.cpp:244-249
//   the extraction + augmentation loop         src/train.cpp:115-147  (rvseg_forest_train_frames)
//
// What the reference does per tree: bootstrap N examples; depth-first over an explicit stack, per node pick a
// random label layer, stop if mass < minSplitExamples / pure / depth > maxDepth, otherwise try numFeatures random
// features: sort the node's examples by the feature and take the threshold (midpoint of two adjacent values that
// differ by at least 1e-6) that minimises E(left) + E(right); split unless a child would have fewer than
// minChildSplitExamples.  Finally the leaf histograms are recomputed from ALL examples, each adding the inverted
// class frequency of its label, and stored as log((h + s) / (total + C*s)).
//
// MI355X design: level by level instead of depth-first -- the tree a given set of random choices produces does not
// depend on the visiting order once those choices are keyed by the node's PATH (below) -- and per level
//   * byte-valued features (the 363 colour-patch features; any feature whose values are integers in [0, 255]):
//     one pass adds every bootstrap example into its node's (feature, value, class) histogram; one block per
//     (node, feature) prefix-sums the 256 values and evaluates every cut between two occupied values: exactly the
//     reference's candidates;
//   * other features (depth, height, normal): the (node, feature, value) triples of the level are sorted once
//     (64-bit radix sort), one wave per (node, feature) segment walks its values in ascending order with running class
//     counts (wave scans) and evaluates every cut between two values at least 1e-6 apart: the reference's candidates
//     again (learning.cpp:578-585), with no binning;
//   * the host applies the stop rules in the reference's order, appends children, and one pass routes every
//     example with the evaluator's own rule `x[f] < threshold`.
// Nodes are renumbered at the end in the order the reference's stack would have created them (children appended when
// the parent is popped, right child popped first: learning.cpp:646-655), so the file equals the depth-first learner's.
//
// Oracle: oracle/rvseg_oracle_train.c restates the reference learner depth-first with sorts; tests compare forest.dat
// byte for byte.  Both sides implement the same build-owned definitions, written down in that file's header: random
// choices from a counter-based generator keyed by (seed, tree, node path); the objective evaluated from the class counts
// by the expression of initEntropies (learning.cpp:279-293) with fastlog2 (fastlog.h:47-58), in float, classes in
// ascending order; bootstrap duplicates as multiplicities; the adjacent-floats threshold guard.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "forest_model.h"
#include "rvseg_internal.h"
#include "rvseg_kernels.h"
#include "rvseg_pipeline.h"

namespace rvseg {
namespace {

constexpr int TR_BINS = 256;
constexpr int TR_CMAX = 16;     // classes per layer the trainer handles (the reference's layers have 8 and 9)

// ---- the shared random source (oracle/rvseg_oracle_train.c, definition 1) -----------------------------------------
inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ inline uint64_t mix64_hd(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint64_t draw64(uint64_t key, uint64_t i) { return mix64(key ^ mix64(i + 0x632BE59BD9B4E019ull)); }

// order-preserving map float -> uint (sort keys)
__host__ __device__ __forceinline__ unsigned f2ord(float v) {
    unsigned u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(v);
#else
    std::memcpy(&u, &v, 4);
#endif
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f_dev(unsigned o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// fastlog2 (fastlog.h:47-58) and ENTROPY(p) = -(p) * fastlog2(p) (learning.cpp:13), in the oracle's operation order
__device__ __forceinline__ float fastlog2_dev(float x) {
    const unsigned vi = __float_as_uint(x);
    const float mx = __uint_as_float((vi & 0x007FFFFFu) | 0x3f000000u);
    float y = (float)vi;
    y = y * 1.1920928955078125e-7f;
    const float a = 1.498030302f * mx;
    const float den = 0.3520887068f + mx;
    const float b = 1.72587999f / den;
    float r = y - 124.22551499f;
    r = r - a;
    r = r - b;
    return r;
}
__device__ __forceinline__ float entropy_term(float p) { return (-p) * fastlog2_dev(p); }

// initEntropies (learning.cpp:279-293) from integer counts: -ENTROPY(mass) + sum of ENTROPY(count) over the classes
// with a non-zero count, ascending
__device__ __forceinline__ float hist_entropy(const unsigned (&cnt)[TR_CMAX]) {
    unsigned mass = 0;
#pragma unroll
    for (int c = 0; c < TR_CMAX; c++) mass += cnt[c];
    float total = -entropy_term((float)mass);
#pragma unroll
    for (int c = 0; c < TR_CMAX; c++) {
        if (cnt[c] == 0) continue;
        total += entropy_term((float)cnt[c]);
    }
    return total;
}

// ---- the training set on the device: feature-major, bytes for byte-valued features, floats for the others ---------
struct TrainSet {
    int P = 0, D = 0, L = 0;
    size_t stride = 0;                 // elements per feature row (>= P)
    uint8_t* Xb = nullptr;             // [D][stride]
    float* Xf = nullptr;               // [n_nb][stride]
    int* lab = nullptr;                // [L][stride] class index per layer
    int* d_nb_index = nullptr;         // [D]: row of the feature in Xf, -1 for a byte feature
    std::vector<int> nb_index;
    int n_nb = 0;
    std::vector<int> class_counts;
};

struct DevArena {   // frees on scope exit
    std::vector<void*> ptrs;
    ~DevArena() { for (void* p : ptrs) (void)hipFree(p); }
    template <class T> T* alloc(rvseg_ctx* ctx, size_t n, bool* ok) {
        void* p = nullptr;
        if (!hip_ok(ctx, hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)), "hipMalloc(train)")) { *ok = false; return nullptr; }
        ptrs.push_back(p);
        return static_cast<T*>(p);
    }
};

// grow-only device buffer that lives across levels and trees (no allocation per level)
template <class T>
struct GrowBuf {
    T* p = nullptr;
    size_t cap = 0;
    ~GrowBuf() { if (p) (void)hipFree(p); }
    bool reserve(rvseg_ctx* ctx, size_t n) {
        if (n <= cap) return true;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = std::max<size_t>(n, 1024);
        want += want / 2;
        if (!hip_ok(ctx, hipMalloc((void**)&p, want * sizeof(T)), "hipMalloc(train level)")) return false;
        cap = want;
        return true;
    }
};

// ---- data set construction from a host matrix ----------------------------------------------------------------------
// per feature: is every value an integer in [0, 255]?
__global__ void __launch_bounds__(256)
train_feature_stats_kernel(const float* __restrict__ X, int P, int D, int* __restrict__ not_byte, int* __restrict__ not_finite) {
    const int f = blockIdx.x;
    int bad = 0, inf = 0;
    for (int i = threadIdx.x; i < P; i += 256) {
        const float v = X[(size_t)i * D + f];
        if (!(v >= 0.f && v <= 255.f && v == floorf(v))) bad = 1;
        if (!(fabsf(v) <= 3.4e38f)) inf = 1;
    }
    if (bad) not_byte[f] = 1;
    if (inf) not_finite[0] = 1;
}

__global__ void __launch_bounds__(256)
train_pack_kernel(const float* __restrict__ X, int P, int D, size_t stride, const int* __restrict__ nb_index, uint8_t* __restrict__ Xb,
                  float* __restrict__ Xf) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)P * D) return;
    const int f = (int)(gid / P), i = (int)(gid - (long long)f * P);
    const float v = X[(size_t)i * D + f];
    const int nb = nb_index[f];
    if (nb < 0) Xb[(size_t)f * stride + i] = (uint8_t)(int)v;
    else Xf[(size_t)nb * stride + i] = v;
}

__global__ void __launch_bounds__(256)
train_class_count_kernel(int P, int L, size_t stride, const int* __restrict__ lab, unsigned* __restrict__ cnt) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)P * L) return;
    const int l = (int)(gid / P), i = (int)(gid - (long long)l * P);
    atomicAdd(&cnt[l * TR_CMAX + lab[(size_t)l * stride + i]], 1u);
}

// ---- per tree ------------------------------------------------------------------------------------------------------
// bootstrap: P draws with replacement as multiplicities (DataStorage::bootstrapmulti, data.cpp:325-349)
__global__ void __launch_bounds__(256)
train_bootstrap_kernel(int P, unsigned long long kt, unsigned* __restrict__ w) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= P) return;
    const unsigned long long d = mix64_hd(kt ^ mix64_hd(0x100000000ull + (unsigned long long)n + 0x632BE59BD9B4E019ull));
    atomicAdd(&w[d % (unsigned long long)P], 1u);
}

// hist[slot][k][value][class] += multiplicity, byte features of the slot only
__global__ void __launch_bounds__(256)
train_hist_kernel(int P, int K, size_t stride, const int* __restrict__ node_of, const int* __restrict__ slot_of, const unsigned* __restrict__ w,
                  const int* __restrict__ slot_layer, const int* __restrict__ slot_feat, const int* __restrict__ nb_index,
                  const int* __restrict__ lab, const uint8_t* __restrict__ Xb, unsigned* __restrict__ hist) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const unsigned wi = w[i];
    if (!wi) return;
    const int slot = slot_of[node_of[i]];
    if (slot < 0) return;
    const int c = lab[(size_t)slot_layer[slot] * stride + i];
    for (int k = 0; k < K; k++) {
        const int f = slot_feat[slot * K + k];
        if (nb_index[f] >= 0) continue;
        const int b = Xb[(size_t)f * stride + i];
        atomicAdd(&hist[(((size_t)slot * K + k) * TR_BINS + b) * TR_CMAX + c], wi);
    }
}

// class totals of every slot (the node's histogram, learning.cpp:508-516)
__global__ void __launch_bounds__(256)
train_slot_totals_kernel(int P, size_t stride, const int* __restrict__ node_of, const int* __restrict__ slot_of, const unsigned* __restrict__ w,
                         const int* __restrict__ slot_layer, const int* __restrict__ lab, unsigned* __restrict__ totals) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const unsigned wi = w[i];
    if (!wi) return;
    const int slot = slot_of[node_of[i]];
    if (slot < 0) return;
    atomicAdd(&totals[slot * TR_CMAX + lab[(size_t)slot_layer[slot] * stride + i]], wi);
}

struct CutResult {
    float objective;           // E(left) + E(right) of the best cut; 1e35 when the feature offers no cut
    float left_value, right_value;   // the two adjacent values the cut lies between
    unsigned left_mass, right_mass;
    int valid;
};

// byte features: one block per (slot, feature), one thread per value
__global__ void __launch_bounds__(TR_BINS)
train_best_cut_kernel(int K, const unsigned* __restrict__ hist, const int* __restrict__ slot_feat, const int* __restrict__ nb_index,
                      CutResult* __restrict__ out) {
    __shared__ unsigned h[TR_CMAX][TR_BINS + 1];   // inclusive prefix over the values, per class
    __shared__ unsigned occ[TR_BINS];
    __shared__ float best_obj[TR_BINS];
    __shared__ int best_bin[TR_BINS];
    if (nb_index[slot_feat[blockIdx.x]] >= 0) return;   // a float feature: the sorted path writes this record
    const int b = threadIdx.x;
    const unsigned* src = hist + ((size_t)blockIdx.x * TR_BINS + b) * TR_CMAX;
    unsigned row = 0;
#pragma unroll
    for (int c = 0; c < TR_CMAX; c++) { const unsigned v = src[c]; h[c][b] = v; row += v; }
    occ[b] = row;
    __syncthreads();
    for (int off = 1; off < TR_BINS; off <<= 1) {   // inclusive scan over the 256 values, all classes at once
        unsigned add[TR_CMAX];
#pragma unroll
        for (int c = 0; c < TR_CMAX; c++) add[c] = b >= off ? h[c][b - off] : 0u;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < TR_CMAX; c++) h[c][b] += add[c];
        __syncthreads();
    }
    int nb = -1;   // next occupied value above b
    if (row) for (int q = b + 1; q < TR_BINS; q++) if (occ[q]) { nb = q; break; }
    float obj = 1e35f;
    unsigned lm = 0, rm = 0;
    if (nb >= 0) {
        unsigned l[TR_CMAX], r[TR_CMAX];
#pragma unroll
        for (int c = 0; c < TR_CMAX; c++) { l[c] = h[c][b]; r[c] = h[c][TR_BINS - 1] - l[c]; lm += l[c]; rm += r[c]; }
        const float el = hist_entropy(l), er = hist_entropy(r);
        obj = el + er;
    }
    best_obj[b] = obj;
    best_bin[b] = nb >= 0 ? b : TR_BINS;
    __syncthreads();
    for (int s = TR_BINS / 2; s > 0; s >>= 1) {   // arg min; the lower value wins a tie (strict '<' in ascending order, :589)
        if (b < s) {
            const float o2 = best_obj[b + s];
            const int b2 = best_bin[b + s];
            if (o2 < best_obj[b] || (o2 == best_obj[b] && b2 < best_bin[b])) { best_obj[b] = o2; best_bin[b] = b2; }
        }
        __syncthreads();
    }
    if (best_bin[0] == TR_BINS) {
        if (b == 0) { CutResult r{}; r.objective = 1e35f; r.valid = 0; out[blockIdx.x] = r; }
        return;
    }
    if (b == best_bin[0]) {
        CutResult r;
        r.objective = obj;
        r.left_value = (float)b;
        r.right_value = (float)nb;
        r.left_mass = lm;
        r.right_mass = rm;
        r.valid = 1;
        out[blockIdx.x] = r;
    }
}

// float features: (segment << 32 | ordered value, example) for every (bootstrap example, sampled float feature of its node)
__global__ void __launch_bounds__(256)
train_nb_emit_kernel(int P, int K, size_t stride, const int* __restrict__ node_of, const int* __restrict__ slot_of, const unsigned* __restrict__ w,
                     const int* __restrict__ slot_feat, const int* __restrict__ nb_index, const float* __restrict__ Xf,
                     unsigned long long* __restrict__ keys, unsigned* __restrict__ vals, unsigned* __restrict__ counter, unsigned capacity) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    if (!w[i]) return;
    const int slot = slot_of[node_of[i]];
    if (slot < 0) return;
    for (int k = 0; k < K; k++) {
        const int nb = nb_index[slot_feat[slot * K + k]];
        if (nb < 0) continue;
        const unsigned pos = atomicAdd(counter, 1u);
        if (pos >= capacity) continue;   // cannot happen: capacity = P * min(K, n_nb)
        keys[pos] = ((unsigned long long)(unsigned)(slot * K + k) << 32) | f2ord(Xf[(size_t)nb * stride + i]);
        vals[pos] = (unsigned)i;
    }
}

// one wave per (slot, feature) segment of the sorted triples: ascending values, running class counts, every cut between
// two values at least 1e-6 apart (learning.cpp:560-604)
__global__ void __launch_bounds__(64)
train_nb_scan_kernel(int K, size_t stride, unsigned n_items, const unsigned long long* __restrict__ keys, const unsigned* __restrict__ vals,
                     const unsigned* __restrict__ w, const int* __restrict__ slot_layer, const int* __restrict__ slot_feat,
                     const int* __restrict__ nb_index, const int* __restrict__ lab, const unsigned* __restrict__ totals,
                     CutResult* __restrict__ out) {
    const unsigned seg = blockIdx.x;
    if (nb_index[slot_feat[seg]] < 0) return;   // a byte feature: the histogram path writes this record
    const int lane = threadIdx.x;
    const int slot = (int)(seg / (unsigned)K);
    // the segment's range in the sorted array (keys are unique per segment in their high word)
    auto lower = [&](unsigned long long key) {
        unsigned lo = 0, hi = n_items;
        while (lo < hi) { const unsigned mid = (lo + hi) >> 1; if (keys[mid] < key) lo = mid + 1; else hi = mid; }
        return lo;
    };
    const unsigned beg = lower((unsigned long long)seg << 32), end = lower((unsigned long long)(seg + 1u) << 32);
    unsigned tot[TR_CMAX], run[TR_CMAX];
    unsigned mass = 0;
#pragma unroll
    for (int c = 0; c < TR_CMAX; c++) { tot[c] = totals[slot * TR_CMAX + c]; run[c] = 0; mass += tot[c]; }
    const int* labl = lab + (size_t)slot_layer[slot] * stride;
    float best_obj = 1e35f, best_lv = 0.f, best_rv = 0.f;
    unsigned best_lm = 0;
    float prev_last = 0.f;   // value of the last element of the previous chunk
    for (unsigned base = beg; base < end; base += 64) {
        const unsigned m = base + (unsigned)lane;
        const bool in = m < end;
        float v = 0.f;
        int cls = -1;
        unsigned wt = 0;
        if (in) {
            v = ord2f_dev((unsigned)(keys[m] & 0xFFFFFFFFull));
            const unsigned e = vals[m];
            cls = labl[e];
            wt = w[e];
        }
        float vprev = __shfl_up(v, 1, 64);
        if (lane == 0) vprev = prev_last;
        // class counts of the elements before this lane's element
        unsigned left[TR_CMAX];
        unsigned lm = 0;
#pragma unroll
        for (int c = 0; c < TR_CMAX; c++) {
            const unsigned x = cls == c ? wt : 0u;
            unsigned incl = x;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned t = __shfl_up(incl, off, 64);
                if (lane >= off) incl += t;
            }
            left[c] = run[c] + incl - x;
            lm += left[c];
            run[c] += __shfl(incl, 63, 64);
        }
        float obj = 1e35f;
        if (in && m > beg && !((v - vprev) < 1e-6f)) {
            unsigned right[TR_CMAX];
#pragma unroll
            for (int c = 0; c < TR_CMAX; c++) right[c] = tot[c] - left[c];
            const float el = hist_entropy(left), er = hist_entropy(right);
            obj = el + er;
        }
        // the chunk's minimum, lowest position first; strict '<' against the running best
        float o = obj;
        int who = lane;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float o2 = __shfl_xor(o, off, 64);
            const int w2 = __shfl_xor(who, off, 64);
            if (o2 < o || (o2 == o && w2 < who)) { o = o2; who = w2; }
        }
        if (o < best_obj) {
            best_obj = o;
            best_lv = __shfl(vprev, who, 64);
            best_rv = __shfl(v, who, 64);
            best_lm = __shfl(lm, who, 64);
        }
        const unsigned last = (end - base < 64u ? end - base : 64u) - 1u;
        prev_last = __shfl(v, (int)last, 64);
    }
    if (lane == 0) {
        CutResult r;
        r.objective = best_obj;
        r.left_value = best_lv;
        r.right_value = best_rv;
        r.left_mass = best_lm;
        r.right_mass = mass - best_lm;
        r.valid = best_obj < 1e35f ? 1 : 0;
        out[seg] = r;
    }
}

// findLeafNode's rule on the freshly split nodes: every example (bootstrap or not) moves to a child
__global__ void __launch_bounds__(256)
train_route_kernel(int P, size_t stride, const uint8_t* __restrict__ Xb, const float* __restrict__ Xf, const int* __restrict__ nb_index,
                   int* __restrict__ node_of, const int* __restrict__ split_feat, const float* __restrict__ split_thr,
                   const int* __restrict__ split_left) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int node = node_of[i];
    const int f = split_feat[node];
    if (f < 0) return;
    const int nb = nb_index[f];
    const float v = nb < 0 ? (float)Xb[(size_t)f * stride + i] : Xf[(size_t)nb * stride + i];
    node_of[i] = v < split_thr[node] ? split_left[node] : split_left[node] + 1;   // classifier.cpp:105
}

// integer leaf counts over ALL examples: cnt[node][layer][class]
__global__ void __launch_bounds__(256)
train_leaf_count_kernel(int P, int L, size_t stride, const int* __restrict__ node_of, const int* __restrict__ lab, unsigned* __restrict__ cnt) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)P * L) return;
    const int l = (int)(gid / P), i = (int)(gid - (long long)l * P);
    atomicAdd(&cnt[((size_t)node_of[i] * L + l) * TR_CMAX + lab[(size_t)l * stride + i]], 1u);
}

// ---- the learner over a device-resident training set ---------------------------------------------------------------
rvseg_status train_core(rvseg_ctx* ctx, const TrainSet& T, const rvseg_train_params& tp, std::vector<uint8_t>& bytes_out, int* n_nodes_out) {
    const int P = T.P, D = T.D, L = T.L;
    const size_t stride = T.stride;
    const int K = std::min(D, tp.num_features > 0 ? tp.num_features : (int)std::ceil(std::sqrt((double)D)));
    hipStream_t s = ctx->stream;
    DevArena A;
    bool ok = true;
    unsigned* dW = A.alloc<unsigned>(ctx, P, &ok);
    int* dNode = A.alloc<int>(ctx, P, &ok);
    unsigned* dClassCnt = A.alloc<unsigned>(ctx, (size_t)L * TR_CMAX, &ok);
    const int SLOT_BATCH = 1024;
    unsigned* dHist = A.alloc<unsigned>(ctx, (size_t)SLOT_BATCH * K * TR_BINS * TR_CMAX, &ok);
    unsigned* dTotals = A.alloc<unsigned>(ctx, (size_t)SLOT_BATCH * TR_CMAX, &ok);
    CutResult* dCut = A.alloc<CutResult>(ctx, (size_t)SLOT_BATCH * K, &ok);
    int* dSlotLayer = A.alloc<int>(ctx, SLOT_BATCH, &ok);
    int* dSlotFeat = A.alloc<int>(ctx, (size_t)SLOT_BATCH * K, &ok);
    unsigned* dCounter = A.alloc<unsigned>(ctx, 4, &ok);
    // sorted path: at most min(K, n_nb) float features per example
    const size_t nb_cap = T.n_nb ? (size_t)P * (size_t)std::min(K, T.n_nb) : 0;
    unsigned long long *dKeysA = nullptr, *dKeysB = nullptr;
    unsigned *dValsA = nullptr, *dValsB = nullptr;
    void* dSortTemp = nullptr;
    size_t sort_temp_bytes = 0;
    if (nb_cap) {
        if (nb_cap >= (1ull << 32)) { ctx->err = "training set too large for the sorted feature path"; return RVSEG_ERR_CAPACITY; }
        dKeysA = A.alloc<unsigned long long>(ctx, nb_cap, &ok);
        dKeysB = A.alloc<unsigned long long>(ctx, nb_cap, &ok);
        dValsA = A.alloc<unsigned>(ctx, nb_cap, &ok);
        dValsB = A.alloc<unsigned>(ctx, nb_cap, &ok);
        (void)rocprim::radix_sort_pairs(nullptr, sort_temp_bytes, dKeysA, dKeysB, dValsA, dValsB, nb_cap, 0, 64, s);
        dSortTemp = A.alloc<uint8_t>(ctx, sort_temp_bytes, &ok);
    }
    if (!ok) return RVSEG_ERR_HIP;
    GrowBuf<int> gSlotOf, gSF, gSL;
    GrowBuf<float> gST;
    GrowBuf<unsigned> gCnt;

    // inverted class frequencies over the whole set (data.h:358-370): freq[c] = size / count_c, in float
    RV_HIP(ctx, hipMemsetAsync(dClassCnt, 0, (size_t)L * TR_CMAX * 4, s));
    train_class_count_kernel<<<dim3((unsigned)(((long long)P * L + 255) / 256)), dim3(256), 0, s>>>(P, L, stride, T.lab, dClassCnt);
    std::vector<unsigned> class_cnt((size_t)L * TR_CMAX);
    RV_HIP(ctx, hipMemcpyAsync(class_cnt.data(), dClassCnt, class_cnt.size() * 4, hipMemcpyDeviceToHost, s));
    RV_HIP(ctx, hipStreamSynchronize(s));
    RV_LAUNCH_OK(ctx);
    std::vector<std::vector<float>> freq(L);
    for (int l = 0; l < L; l++) {
        freq[l].assign(T.class_counts[l], 0.f);
        for (int c = 0; c < T.class_counts[l]; c++) {
            // "freq[label]++" per example on a float: exact up to 2^24, where the float stops growing
            const unsigned cn = class_cnt[(size_t)l * TR_CMAX + c];
            const float n = cn <= 16777216u ? (float)cn : 16777216.f;
            freq[l][c] = P / n;
        }
    }

    ForestModel model;
    model.raw.resize((size_t)tp.num_trees);
    std::vector<CutResult> cuts((size_t)SLOT_BATCH * K);
    std::vector<unsigned> totals((size_t)SLOT_BATCH * TR_CMAX);
    std::vector<int> perm(D);
    int total_nodes = 0;

    for (int t = 0; t < tp.num_trees; t++) {
        const uint64_t kt = mix64(tp.seed ^ mix64(0x74726565ull + (uint64_t)t));
        if (tp.use_bootstrap) {
            RV_HIP(ctx, hipMemsetAsync(dW, 0, (size_t)P * 4, s));
            train_bootstrap_kernel<<<dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s>>>(P, kt, dW);
        } else {
            std::vector<unsigned> ones((size_t)P, 1u);
            RV_HIP(ctx, hipMemcpyAsync(dW, ones.data(), (size_t)P * 4, hipMemcpyHostToDevice, s));
            RV_HIP(ctx, hipStreamSynchronize(s));
        }
        RV_HIP(ctx, hipMemsetAsync(dNode, 0, (size_t)P * 4, s));
        // the tree in level order while it grows; renumbered at the end
        std::vector<int> feat(1, 0), left(1, 0), depth(1, 0);
        std::vector<float> thr(1, 0.f);
        std::vector<uint64_t> nkey(1, mix64(kt ^ 0x726F6F74ull));
        std::vector<int> frontier(1, 0);
        while (!frontier.empty()) {
            const int n_nodes = (int)left.size();
            std::vector<int> slot_of(n_nodes, -1), split_feat(n_nodes, -1), split_left(n_nodes, 0);
            std::vector<float> split_thr(n_nodes, 0.f);
            std::vector<int> next_frontier;
            if (!gSlotOf.reserve(ctx, n_nodes)) return RVSEG_ERR_HIP;
            for (size_t base = 0; base < frontier.size(); base += SLOT_BATCH) {
                const int S = (int)std::min<size_t>(SLOT_BATCH, frontier.size() - base);
                std::vector<int> slot_layer(S), slot_feat((size_t)S * K);
                std::fill(slot_of.begin(), slot_of.end(), -1);
                bool any_nb = false;
                for (int q = 0; q < S; q++) {
                    const int node = frontier[base + q];
                    slot_of[node] = q;
                    const uint64_t key = nkey[node];
                    slot_layer[q] = (int)(draw64(key, 0) % (uint64_t)L);                       // "Pick a random class layer", :483-485
                    for (int f = 0; f < D; f++) perm[f] = f;                                    // numFeatures without replacement, :537
                    for (int k = 0; k < K; k++) {
                        const int j = k + (int)(draw64(key, 1 + (uint64_t)k) % (uint64_t)(D - k));
                        std::swap(perm[k], perm[j]);
                        slot_feat[(size_t)q * K + k] = perm[k];
                        any_nb = any_nb || T.nb_index[perm[k]] >= 0;
                    }
                }
                RV_HIP(ctx, hipMemcpyAsync(gSlotOf.p, slot_of.data(), (size_t)n_nodes * 4, hipMemcpyHostToDevice, s));
                RV_HIP(ctx, hipMemcpyAsync(dSlotLayer, slot_layer.data(), (size_t)S * 4, hipMemcpyHostToDevice, s));
                RV_HIP(ctx, hipMemcpyAsync(dSlotFeat, slot_feat.data(), (size_t)S * K * 4, hipMemcpyHostToDevice, s));
                RV_HIP(ctx, hipMemsetAsync(dHist, 0, (size_t)S * K * TR_BINS * TR_CMAX * 4, s));
                RV_HIP(ctx, hipMemsetAsync(dTotals, 0, (size_t)S * TR_CMAX * 4, s));
                const dim3 gridP((unsigned)((P + 255) / 256)), b256(256);
                train_slot_totals_kernel<<<gridP, b256, 0, s>>>(P, stride, dNode, gSlotOf.p, dW, dSlotLayer, T.lab, dTotals);
                train_hist_kernel<<<gridP, b256, 0, s>>>(P, K, stride, dNode, gSlotOf.p, dW, dSlotLayer, dSlotFeat, T.d_nb_index, T.lab, T.Xb, dHist);
                train_best_cut_kernel<<<dim3((unsigned)(S * K)), dim3(TR_BINS), 0, s>>>(K, dHist, dSlotFeat, T.d_nb_index, dCut);
                if (any_nb) {
                    RV_HIP(ctx, hipMemsetAsync(dCounter, 0, 16, s));
                    train_nb_emit_kernel<<<gridP, b256, 0, s>>>(P, K, stride, dNode, gSlotOf.p, dW, dSlotFeat, T.d_nb_index, T.Xf, dKeysA, dValsA,
                                                              dCounter, (unsigned)nb_cap);
                    unsigned n_items = 0;
                    RV_HIP(ctx, hipMemcpyAsync(&n_items, dCounter, 4, hipMemcpyDeviceToHost, s));
                    RV_HIP(ctx, hipStreamSynchronize(s));
                    RV_LAUNCH_OK(ctx);
                    n_items = (unsigned)std::min<size_t>(n_items, nb_cap);
                    const unsigned long long* keys_sorted = dKeysA;
                    const unsigned* vals_sorted = dValsA;
                    if (n_items > 1) {
                        int seg_bits = 1;
                        while ((1u << seg_bits) < (unsigned)(S * K)) seg_bits++;
                        size_t tb = sort_temp_bytes;
                        RV_HIP(ctx, rocprim::radix_sort_pairs(dSortTemp, tb, dKeysA, dKeysB, dValsA, dValsB, (size_t)n_items, 0,
                                                              (unsigned)(32 + seg_bits), s));
                        keys_sorted = dKeysB;
                        vals_sorted = dValsB;
                    }
                    train_nb_scan_kernel<<<dim3((unsigned)(S * K)), dim3(64), 0, s>>>(K, stride, n_items, keys_sorted, vals_sorted, dW, dSlotLayer,
                                                                                   dSlotFeat, T.d_nb_index, T.lab, dTotals, dCut);
                }
                RV_HIP(ctx, hipMemcpyAsync(cuts.data(), dCut, (size_t)S * K * sizeof(CutResult), hipMemcpyDeviceToHost, s));
                RV_HIP(ctx, hipMemcpyAsync(totals.data(), dTotals, (size_t)S * TR_CMAX * 4, hipMemcpyDeviceToHost, s));
                RV_HIP(ctx, hipStreamSynchronize(s));
                RV_LAUNCH_OK(ctx);
                for (int q = 0; q < S; q++) {
                    const int node = frontier[base + q];
                    unsigned mass = 0;
                    int present = 0;
                    for (int c = 0; c < TR_CMAX; c++) { const unsigned n = totals[(size_t)q * TR_CMAX + c]; mass += n; present += n ? 1 : 0; }
                    // stop rules of learning.cpp:521-527: too few examples, pure, too deep
                    if ((long long)mass < (long long)tp.min_split_examples || present <= 1 || depth[node] > tp.max_depth) continue;
                    int best_k = -1;
                    float best_obj = 1e35f;
                    for (int k = 0; k < K; k++) {   // features in sampled order, strict '<' keeps the first best (:589)
                        const CutResult& c = cuts[(size_t)q * K + k];
                        if (c.valid && c.objective < best_obj) { best_obj = c.objective; best_k = k; }
                    }
                    if (best_k < 0) continue;                                                   // bestFeature < 0, :611
                    const CutResult& c = cuts[(size_t)q * K + best_k];
                    if ((long long)c.left_mass < (long long)tp.min_child_split_examples || (long long)c.right_mass < (long long)tp.min_child_split_examples) continue;
                    const int f = slot_feat[(size_t)q * K + best_k];
                    float th = c.left_value + c.right_value;                                    // :592
                    th *= 0.5f;                                                                 // :607
                    if (!(c.left_value < th)) th = c.right_value;   // two adjacent floats: keep `x < th` separating them
                    const int lc = (int)left.size();
                    for (int side = 0; side < 2; side++) {                                      // DecisionTree::splitNode, classifier.cpp:77-95
                        feat.push_back(0); thr.push_back(0.f); left.push_back(0); depth.push_back(depth[node] + 1);
                        nkey.push_back(mix64(nkey[node] ^ (side == 0 ? 0x4Cull : 0x52ull)));
                    }
                    feat[node] = f; thr[node] = th; left[node] = lc;
                    split_feat[node] = f; split_thr[node] = th; split_left[node] = lc;
                    next_frontier.push_back(lc);
                    next_frontier.push_back(lc + 1);
                }
            }
            if (!next_frontier.empty()) {
                if (!gSF.reserve(ctx, n_nodes) || !gST.reserve(ctx, n_nodes) || !gSL.reserve(ctx, n_nodes)) return RVSEG_ERR_HIP;
                RV_HIP(ctx, hipMemcpyAsync(gSF.p, split_feat.data(), (size_t)n_nodes * 4, hipMemcpyHostToDevice, s));
                RV_HIP(ctx, hipMemcpyAsync(gST.p, split_thr.data(), (size_t)n_nodes * 4, hipMemcpyHostToDevice, s));
                RV_HIP(ctx, hipMemcpyAsync(gSL.p, split_left.data(), (size_t)n_nodes * 4, hipMemcpyHostToDevice, s));
                train_route_kernel<<<dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s>>>(P, stride, T.Xb, T.Xf, T.d_nb_index, dNode, gSF.p, gST.p, gSL.p);
                RV_HIP(ctx, hipStreamSynchronize(s));   // the host vectors above go out of scope
                RV_LAUNCH_OK(ctx);
            }
            frontier.swap(next_frontier);
        }
        // ---- leaf counts over ALL examples (updateMultiHistograms, learning.cpp:960-1012) ------------------
        const int n_nodes = (int)left.size();
        total_nodes += n_nodes;
        if (!gCnt.reserve(ctx, (size_t)n_nodes * L * TR_CMAX)) return RVSEG_ERR_HIP;
        RV_HIP(ctx, hipMemsetAsync(gCnt.p, 0, (size_t)n_nodes * L * TR_CMAX * 4, s));
        train_leaf_count_kernel<<<dim3((unsigned)(((long long)P * L + 255) / 256)), dim3(256), 0, s>>>(P, L, stride, dNode, T.lab, gCnt.p);
        std::vector<unsigned> cnt((size_t)n_nodes * L * TR_CMAX);
        RV_HIP(ctx, hipMemcpyAsync(cnt.data(), gCnt.p, cnt.size() * 4, hipMemcpyDeviceToHost, s));
        RV_HIP(ctx, hipStreamSynchronize(s));
        RV_LAUNCH_OK(ctx);
        // ---- the reference's node numbering: children are appended when their parent is popped, the right child is
        // popped first (learning.cpp:646-655)
        std::vector<int> new_id(n_nodes, -1), order;
        order.reserve(n_nodes);
        {
            std::vector<int> stack(1, 0);
            new_id[0] = 0;
            int next = 1;
            while (!stack.empty()) {
                const int v = stack.back();
                stack.pop_back();
                if (left[v] == 0) continue;
                new_id[left[v]] = next; new_id[left[v] + 1] = next + 1;
                next += 2;
                stack.push_back(left[v]);
                stack.push_back(left[v] + 1);
            }
        }
        RawTree& tree = model.raw[(size_t)t];
        tree.feat.assign(n_nodes, 0); tree.thr.assign(n_nodes, 0.f); tree.left.assign(n_nodes, 0);
        tree.hist.assign(n_nodes, {}); tree.mhist.assign(n_nodes, {});
        for (int v = 0; v < n_nodes; v++) {
            const int nv = new_id[v];
            tree.feat[nv] = feat[v]; tree.thr[nv] = thr[v];
            tree.left[nv] = left[v] ? new_id[left[v]] : 0;
            if (left[v] != 0) continue;
            tree.mhist[nv].resize((size_t)L);
            for (int l = 0; l < L; l++) {
                const int C = T.class_counts[l];
                std::vector<float>& h = tree.mhist[nv][l];
                h.assign(C, 0.f);
                for (int c = 0; c < C; c++) {
                    // "hist[l][classlabel] += freq[classlabel]" once per example (:989-991): n additions of the same addend
                    const unsigned n = cnt[((size_t)v * L + l) * TR_CMAX + c];
                    const float f = freq[l][c];
                    float acc = 0.f;
                    for (unsigned k = 0; k < n; k++) acc += f;
                    h[c] = acc;
                }
                float total = 0;
                for (int c = 0; c < C; c++) total += h[c];
                for (int c = 0; c < C; c++) h[c] = std::log((h[c] + tp.smoothing) / (total + C * tp.smoothing));   // :1004-1007
            }
            if (L == 1) tree.hist[nv] = tree.mhist[nv][0];   // a single-layer forest also serves classLogPosterior
        }
    }
    bytes_out = serialize_forest(model);
    if (n_nodes_out) *n_nodes_out = total_nodes;
    return RVSEG_OK;
}

rvseg_status check_train_args(rvseg_ctx* ctx, int32_t n_layers, const int32_t* class_counts, const rvseg_train_params& tp, int D) {
    if (!class_counts || n_layers < 1 || n_layers > RVSEG_MAX_LAYERS || tp.num_trees < 1 || tp.num_trees > kMaxTrees || tp.max_depth < 1 ||
        tp.min_split_examples < 0 || tp.min_child_split_examples < 0 || tp.num_features < 0 || tp.num_features > D || !(tp.smoothing >= 0.f)) {
        ctx->err = "bad arguments";
        return RVSEG_ERR_INVALID_ARG;
    }
    int sumC = 0;
    for (int l = 0; l < n_layers; l++) {
        if (class_counts[l] < 1 || class_counts[l] > TR_CMAX) { ctx->err = "the trainer handles 1..16 classes per layer"; return RVSEG_ERR_INVALID_ARG; }
        sumC += class_counts[l];
    }
    if (sumC > kMaxClasses) { ctx->err = "more than 64 classes over all layers"; return RVSEG_ERR_INVALID_ARG; }
    return RVSEG_OK;
}

// the model of the last training call on a context (so that a caller whose buffer was too small need not train again)
rvseg_status hand_out(rvseg_ctx* ctx, void* forest_out, size_t out_cap, size_t* size_out) {
    const std::vector<uint8_t>& b = ctx->trained_model;
    if (size_out) *size_out = b.size();
    if (!forest_out) return RVSEG_OK;
    if (out_cap < b.size()) { ctx->err = "output buffer too small (the model is kept: rvseg_forest_train_result)"; return RVSEG_ERR_INVALID_ARG; }
    std::memcpy(forest_out, b.data(), b.size());
    return RVSEG_OK;
}

// ---- frames -> training set on the device (src/train.cpp:115-147) -------------------------------------------------
// flags[p] = the stride-grid point has valid depth (extract's mask) and every label layer is >= 0 there
// (ExtractType::WITH_POSITIVE_LABEL, feature_extractor.h:93-121)
__global__ void __launch_bounds__(256)
train_frame_flags_kernel(FrameGeom g, const uint8_t* __restrict__ valid, const int8_t* __restrict__ labels, int L, int* __restrict__ flags) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= g.lw * g.lh) return;
    const int ly = p / g.lw, lx = p - ly * g.lw;
    const size_t px = (size_t)(ly * g.stride) * g.W + (size_t)lx * g.stride;
    int ok = valid[p] ? 1 : 0;
    for (int l = 0; l < L; l++) ok = ok && labels[(size_t)l * g.W * g.H + px] >= 0;
    flags[p] = ok;
}

__global__ void __launch_bounds__(256)
train_frame_scatter_kernel(FrameGeom g, const int* __restrict__ flags, const int* __restrict__ offs, const float* __restrict__ dump,
                           const int8_t* __restrict__ labels, int L, size_t base, size_t stride, const int* __restrict__ nb_index,
                           uint8_t* __restrict__ Xb, float* __restrict__ Xf, int* __restrict__ lab) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int Pg = g.lw * g.lh;
    if (gid >= (long long)Pg * (g.D + L)) return;
    const int col = (int)(gid / Pg), p = (int)(gid - (long long)col * Pg);
    if (!flags[p]) return;
    const size_t dst = base + (size_t)offs[p];
    if (col < g.D) {
        const float v = dump[(size_t)p * g.D + col];
        const int nb = nb_index[col];
        if (nb < 0) Xb[(size_t)col * stride + dst] = (uint8_t)(int)v;
        else Xf[(size_t)nb * stride + dst] = v;
    } else {
        const int l = col - g.D;
        const int ly = p / g.lw, lx = p - ly * g.lw;
        lab[(size_t)l * stride + dst] = labels[(size_t)l * g.W * g.H + (size_t)(ly * g.stride) * g.W + (size_t)lx * g.stride];
    }
}

}  // namespace
}  // namespace rvseg

using namespace rvseg;

extern "C" {

void rvseg_train_params_default(rvseg_train_params* tp) {
    if (!tp) return;
    std::memset(tp, 0, sizeof(*tp));
    tp->num_trees = 4;                  // resources/config.json:37
    tp->max_depth = 30;                 // :38
    tp->min_split_examples = 50;        // :39
    tp->min_child_split_examples = 1;   // learning.h:116
    tp->num_features = 0;               // ceil(sqrt(D)), DecisionTreeLearner::autoconf (learning.cpp:363-368)
    tp->use_bootstrap = 1;              // train.cpp:226
    tp->smoothing = 1.0f;               // learning.h:117
    tp->seed = 1;
}

rvseg_status rvseg_forest_train_result(rvseg_ctx* ctx, void* forest_out, size_t out_cap, size_t* size_out) {
    if (!ctx) return RVSEG_ERR_INVALID_ARG;
    if (ctx->trained_model.empty()) { ctx->err = "no model has been trained on this context"; return RVSEG_ERR_NO_FOREST; }
    return hand_out(ctx, forest_out, out_cap, size_out);
}

rvseg_status rvseg_forest_train(rvseg_ctx* ctx, const float* X, int32_t P, int32_t D, const int32_t* labels, int32_t n_layers,
                                const int32_t* class_counts, const rvseg_train_params* tp_in, void* forest_out, size_t out_cap,
                                size_t* size_out) {
    if (!ctx) return RVSEG_ERR_INVALID_ARG;
    rvseg_train_params tp;
    if (tp_in) tp = *tp_in; else rvseg_train_params_default(&tp);
    if (!X || !labels || !size_out || P < 1 || D < 1) { ctx->err = "bad arguments"; return RVSEG_ERR_INVALID_ARG; }
    rvseg_status st = check_train_args(ctx, n_layers, class_counts, tp, D);
    if (st != RVSEG_OK) return st;
    for (long long q = 0; q < (long long)P * n_layers; q++) {
        const int l = (int)(q % n_layers);
        if (labels[q] < 0 || labels[q] >= class_counts[l]) { ctx->err = "label outside its layer's class range"; return RVSEG_ERR_INVALID_ARG; }
    }
    RV_HIP(ctx, hipSetDevice(ctx->params.device));
    hipStream_t s = ctx->stream;
    DevArena A;
    bool ok = true;
    TrainSet T;
    T.P = P; T.D = D; T.L = n_layers; T.stride = (size_t)P;
    T.class_counts.assign(class_counts, class_counts + n_layers);
    float* dX = A.alloc<float>(ctx, (size_t)P * D, &ok);
    int* dNotByte = A.alloc<int>(ctx, D, &ok);
    int* dNotFinite = A.alloc<int>(ctx, 4, &ok);
    T.Xb = A.alloc<uint8_t>(ctx, (size_t)P * D, &ok);
    T.lab = A.alloc<int>(ctx, (size_t)P * n_layers, &ok);
    T.d_nb_index = A.alloc<int>(ctx, D, &ok);
    if (!ok) return RVSEG_ERR_HIP;
    std::vector<int> lab_lm((size_t)P * n_layers);
    for (int i = 0; i < P; i++)
        for (int l = 0; l < n_layers; l++) lab_lm[(size_t)l * P + i] = labels[(size_t)i * n_layers + l];
    RV_HIP(ctx, hipMemcpyAsync(dX, X, (size_t)P * D * 4, hipMemcpyHostToDevice, s));
    RV_HIP(ctx, hipMemcpyAsync(T.lab, lab_lm.data(), lab_lm.size() * 4, hipMemcpyHostToDevice, s));
    RV_HIP(ctx, hipMemsetAsync(dNotByte, 0, (size_t)D * 4, s));
    RV_HIP(ctx, hipMemsetAsync(dNotFinite, 0, 16, s));
    train_feature_stats_kernel<<<dim3((unsigned)D), dim3(256), 0, s>>>(dX, P, D, dNotByte, dNotFinite);
    std::vector<int> not_byte(D);
    int not_finite = 0;
    RV_HIP(ctx, hipMemcpyAsync(not_byte.data(), dNotByte, (size_t)D * 4, hipMemcpyDeviceToHost, s));
    RV_HIP(ctx, hipMemcpyAsync(&not_finite, dNotFinite, 4, hipMemcpyDeviceToHost, s));
    RV_HIP(ctx, hipStreamSynchronize(s));
    RV_LAUNCH_OK(ctx);
    if (not_finite) { ctx->err = "non-finite feature value in the training set"; return RVSEG_ERR_INVALID_ARG; }
    T.nb_index.assign(D, -1);
    for (int f = 0; f < D; f++) if (not_byte[f]) T.nb_index[f] = T.n_nb++;
    T.Xf = A.alloc<float>(ctx, (size_t)std::max(T.n_nb, 1) * P, &ok);
    if (!ok) return RVSEG_ERR_HIP;
    RV_HIP(ctx, hipMemcpyAsync(T.d_nb_index, T.nb_index.data(), (size_t)D * 4, hipMemcpyHostToDevice, s));
    train_pack_kernel<<<dim3((unsigned)(((long long)P * D + 255) / 256)), dim3(256), 0, s>>>(dX, P, D, T.stride, T.d_nb_index, T.Xb, T.Xf);
    RV_HIP(ctx, hipStreamSynchronize(s));
    RV_LAUNCH_OK(ctx);
    if ((st = train_core(ctx, T, tp, ctx->trained_model, nullptr)) != RVSEG_OK) return st;
    return hand_out(ctx, forest_out, out_cap, size_out);
}

rvseg_status rvseg_forest_train_frames(rvseg_ctx* ctx, int32_t n_frames, const uint8_t* rgb, const uint16_t* depth_mm, const float* calib,
                                       const int8_t* labels, int32_t n_layers, const int32_t* class_counts, int32_t augment,
                                       const rvseg_train_params* tp_in, void* forest_out, size_t out_cap, size_t* size_out,
                                       int32_t* n_examples_out) {
    if (!ctx) return RVSEG_ERR_INVALID_ARG;
    rvseg_train_params tp;
    if (tp_in) tp = *tp_in; else rvseg_train_params_default(&tp);
    if (!rgb || !depth_mm || !calib || !labels || !size_out || n_frames < 1) { ctx->err = "bad arguments"; return RVSEG_ERR_INVALID_ARG; }
    RV_HIP(ctx, hipSetDevice(ctx->params.device));
    rvseg_status st = pipeline_init(ctx);
    if (st != RVSEG_OK) return st;
    Pipeline* im = reinterpret_cast<Pipeline*>(ctx->impl);
    const FrameGeom& g = im->geom;
    const rvseg_params& p = ctx->params;
    if ((st = check_train_args(ctx, n_layers, class_counts, tp, g.D)) != RVSEG_OK) return st;
    const size_t npix = (size_t)g.W * g.H;
    const int Pg = g.lw * g.lh;
    const int n_var = augment ? 6 : 1;
    const size_t cap = (size_t)n_frames * n_var * Pg;
    if (cap >= (1ull << 31)) { ctx->err = "too many training points"; return RVSEG_ERR_CAPACITY; }
    hipStream_t s = ctx->stream;
    DevArena A;
    bool ok = true;
    TrainSet T;
    T.D = g.D; T.L = n_layers; T.stride = cap;
    T.class_counts.assign(class_counts, class_counts + n_layers);
    T.nb_index.assign(g.D, -1);
    for (int f = g.n_patch; f < g.D; f++) T.nb_index[f] = T.n_nb++;   // depth, height, normal: floats; the patch: Lab bytes
    T.Xb = A.alloc<uint8_t>(ctx, (size_t)g.D * cap, &ok);
    T.Xf = A.alloc<float>(ctx, (size_t)std::max(T.n_nb, 1) * cap, &ok);
    T.lab = A.alloc<int>(ctx, (size_t)n_layers * cap, &ok);
    T.d_nb_index = A.alloc<int>(ctx, g.D, &ok);
    int8_t* dLabels = A.alloc<int8_t>(ctx, (size_t)n_layers * npix, &ok);
    int* dFlags = A.alloc<int>(ctx, Pg, &ok);
    int* dOffs = A.alloc<int>(ctx, Pg + 1, &ok);
    size_t scan_bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, scan_bytes, dFlags, dOffs, 0, (size_t)Pg, rocprim::plus<int>(), s);
    void* dScan = A.alloc<uint8_t>(ctx, scan_bytes, &ok);
    if (!ok) return RVSEG_ERR_HIP;
    RV_HIP(ctx, hipMemcpyAsync(T.d_nb_index, T.nb_index.data(), (size_t)g.D * 4, hipMemcpyHostToDevice, s));
    if ((st = dev_reserve(ctx, im->in_rgb, npix * 3)) != RVSEG_OK || (st = dev_reserve(ctx, im->in_depth, npix * 2)) != RVSEG_OK ||
        (st = dev_reserve(ctx, im->lab, npix * 4)) != RVSEG_OK || (st = dev_reserve(ctx, im->cloud, npix * 16)) != RVSEG_OK ||
        (st = dev_reserve(ctx, im->rect, npix)) != RVSEG_OK || (st = dev_reserve(ctx, im->change, npix)) != RVSEG_OK ||
        (st = dev_reserve(ctx, im->nfeat, (size_t)Pg * 4)) != RVSEG_OK || (st = dev_reserve(ctx, im->dump, (size_t)Pg * g.D * 4)) != RVSEG_OK ||
        (st = dev_reserve(ctx, im->valid, (size_t)Pg)) != RVSEG_OK) return st;
    std::vector<uint8_t> h_rgb(npix * 3);
    std::vector<uint16_t> h_depth(npix);
    std::vector<int8_t> h_lab((size_t)n_layers * npix);
    size_t base = 0;
    static const int offsets[3] = {-20, 0, 20};   // train.cpp:115-117
    for (int fr = 0; fr < n_frames; fr++) {
        const uint8_t* src_rgb = rgb + (size_t)fr * npix * 3;
        const uint16_t* src_d = depth_mm + (size_t)fr * npix;
        const int8_t* src_l = labels + (size_t)fr * n_layers * npix;
        for (int var = 0; var < n_var; var++) {
            // the reference's order: for a in (-20, 0, +20): the frame, then its horizontal flip (train.cpp:119-147)
            const int a = augment ? offsets[var / 2] : 0;
            const bool flip = augment && (var & 1);
            for (int y = 0; y < g.H; y++)
                for (int x = 0; x < g.W; x++) {
                    const size_t d = (size_t)y * g.W + x, q = (size_t)y * g.W + (flip ? g.W - 1 - x : x);
                    // `color += a` on an 8UC3 cv::Mat: the scalar becomes cv::Scalar(a, 0, 0, 0) -- only channel 0 moves --
                    // with saturate_cast<uchar> (OpenCV's scalar rule; train.cpp:122)
                    int c0 = (int)src_rgb[q * 3] + a;
                    c0 = c0 < 0 ? 0 : (c0 > 255 ? 255 : c0);
                    h_rgb[d * 3] = (uint8_t)c0; h_rgb[d * 3 + 1] = src_rgb[q * 3 + 1]; h_rgb[d * 3 + 2] = src_rgb[q * 3 + 2];
                    h_depth[d] = src_d[q];
                    for (int l = 0; l < n_layers; l++) h_lab[(size_t)l * npix + d] = src_l[(size_t)l * npix + q];
                }
            RV_HIP(ctx, hipMemcpyAsync(im->in_rgb.p, h_rgb.data(), npix * 3, hipMemcpyHostToDevice, s));
            RV_HIP(ctx, hipMemcpyAsync(im->in_depth.p, h_depth.data(), npix * 2, hipMemcpyHostToDevice, s));
            RV_HIP(ctx, hipMemcpyAsync(dLabels, h_lab.data(), h_lab.size(), hipMemcpyHostToDevice, s));
            if ((st = upload_calib(ctx, im, calib + (size_t)fr * 21, 1, s)) != RVSEG_OK) return st;
            launch_prep(g, ctx->lab, im->in_rgb.as<uint8_t>(), im->in_depth.as<uint16_t>(), im->calibA.as<float>(), im->lab.as<uint32_t>(),
                        im->cloud.as<float4>(), p.feature_normal ? im->change.as<uint8_t>() : nullptr, 1, s);
            if (p.feature_normal) {
                launch_window_map(g, im->cloud.as<float4>(), im->change.as<uint8_t>(), im->rect.as<uint8_t>(), 1, s);
                launch_normal_feature(g, im->cloud.as<float4>(), im->rect.as<uint8_t>(), im->nfeat.as<float>(), 1, s);
            }
            launch_rf_frames(g, ctx->forest, im->resize_rows.as<ResizeRow>(), im->lab.as<uint32_t>(), im->in_depth.as<uint16_t>(),
                             im->cloud.as<float4>(), im->nfeat.as<float>(), nullptr, im->dump.as<float>(), im->valid.as<uint8_t>(), 1, s);
            train_frame_flags_kernel<<<dim3((unsigned)((Pg + 255) / 256)), dim3(256), 0, s>>>(g, im->valid.as<uint8_t>(), dLabels, n_layers, dFlags);
            size_t sb = scan_bytes;
            RV_HIP(ctx, rocprim::exclusive_scan(dScan, sb, dFlags, dOffs, 0, (size_t)Pg, rocprim::plus<int>(), s));
            const long long threads = (long long)Pg * (g.D + n_layers);
            train_frame_scatter_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s>>>(g, dFlags, dOffs, im->dump.as<float>(), dLabels,
                                                                                                  n_layers, base, T.stride, T.d_nb_index, T.Xb, T.Xf, T.lab);
            int last_off = 0, last_flag = 0;
            RV_HIP(ctx, hipMemcpyAsync(&last_off, dOffs + (Pg - 1), 4, hipMemcpyDeviceToHost, s));
            RV_HIP(ctx, hipMemcpyAsync(&last_flag, dFlags + (Pg - 1), 4, hipMemcpyDeviceToHost, s));
            RV_HIP(ctx, hipStreamSynchronize(s));   // (also: the host staging vectors are rewritten by the next variant)
            RV_LAUNCH_OK(ctx);
            base += (size_t)(last_off + last_flag);
        }
    }
    if (n_examples_out) *n_examples_out = (int32_t)base;
    if (base == 0) { ctx->err = "no labelled point with valid depth in the training frames"; return RVSEG_ERR_INVALID_ARG; }
    T.P = (int)base;
    // labels must lie inside their layer's class range (checked on the device copy: one small read-back)
    {
        std::vector<int> hl((size_t)T.P);
        for (int l = 0; l < n_layers; l++) {
            RV_HIP(ctx, hipMemcpy(hl.data(), T.lab + (size_t)l * T.stride, (size_t)T.P * 4, hipMemcpyDeviceToHost));
            for (int i = 0; i < T.P; i++)
                if (hl[i] < 0 || hl[i] >= class_counts[l]) { ctx->err = "label outside its layer's class range"; return RVSEG_ERR_INVALID_ARG; }
        }
    }
    if ((st = train_core(ctx, T, tp, ctx->trained_model, nullptr)) != RVSEG_OK) return st;
    return hand_out(ctx, forest_out, out_cap, size_out);
}

}  // extern "C"
