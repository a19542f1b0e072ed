// Forest training on the GPU: replaces, for the forests this path evaluates,
//   RandomForestLearner::learn                 third-party/libforest/src/learning.cpp:1031-1073
//   DecisionTreeLearner::learn (multi-layer)   learning.cpp:410-662   (single layer: :663-915, same search)
//   updateMultiHistograms / updateHistograms   learning.cpp:918-1012
//   forest->write                              src/train.cpp:244-249
//
// What the reference does per tree: bootstrap N examples; depth-first over an explicit stack, per node pick a
// random label layer, stop if mass < minSplitExamples / pure / depth > maxDepth, otherwise try numFeatures random
// features: sort the node's examples by the feature and take the threshold (midpoint of two adjacent distinct
// values) that minimises E(left) + E(right), E(h) = mass*log2(mass) - sum_c n_c*log2(n_c); split unless a child would
// have fewer than minChildSplitExamples.  Finally the leaf histograms are recomputed from ALL examples, each adding
// the inverted class frequency of its label, and stored as log((h + s) / (total + C*s)).
//
// MI355X design: level-wise instead of depth-first (the tree that results from a given sequence of random choices
// does not depend on the order nodes are visited in), and histograms instead of sorts: every feature is binned
// once into 256 bins -- the 363 colour-patch features are bytes, so their bins ARE their values and the candidate
// thresholds are exactly the reference's; float features (depth, height, normal) get 256 uniform bins between
// their extrema, thresholds at the midpoint between the largest value of one occupied bin and the smallest of the
// next.  Per level: one pass over the examples adds each bootstrap example into its node's (feature, bin, class)
// histograms; one block per (node, feature) scans the 256 bins for the best cut; the host picks per node, appends
// children, and one pass routes every example with the evaluator's own rule `x[f] < threshold`.  The leaf
// histograms come from integer counts (GPU) and the reference's own float accumulation order (host: n additions
// of the same addend, then the logarithm), so they are exactly what updateMultiHistograms would store for that tree.
//
// No parity oracle exists for training (the reference seeds from std::random_device, learning.cpp:18): tests pin the
// histogram definition, the split objective at the root (against brute force) and the stopping rules.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

#include "forest_model.h"
#include "rvseg_internal.h"

namespace rvseg {
namespace {

constexpr int TR_BINS = 256;
constexpr int TR_CMAX = 16;     // classes per layer the trainer handles (the reference's layers have 8 and 9)

// order-preserving map float -> uint so that atomicMin / atomicMax work on floats
__device__ __forceinline__ unsigned f2ord(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
inline float ord2f(unsigned o) {
    const unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    float v;
    std::memcpy(&v, &u, 4);
    return v;
}

// per feature: min, max (ordered uints) and whether every value is an integer in [0, 255]
__global__ void __launch_bounds__(256)
train_feature_stats_kernel(const float* __restrict__ X, int P, int D, unsigned* __restrict__ fmin, unsigned* __restrict__ fmax,
                           int* __restrict__ not_byte) {
    const int f = blockIdx.x;
    unsigned lo = 0xFFFFFFFFu, hi = 0u;
    int bad = 0;
    for (int i = threadIdx.x; i < P; i += 256) {
        const float v = X[(size_t)i * D + f];
        const unsigned o = f2ord(v);
        lo = o < lo ? o : lo;
        hi = o > hi ? o : hi;
        if (!(v >= 0.f && v <= 255.f && v == floorf(v))) bad = 1;
    }
    atomicMin(&fmin[f], lo);
    atomicMax(&fmax[f], hi);
    if (bad) not_byte[f] = 1;
}

// bins (feature-major, D x P bytes) and the smallest / largest value seen in every (feature, bin)
__global__ void __launch_bounds__(256)
train_bin_kernel(const float* __restrict__ X, int P, int D, const float* __restrict__ lo, const float* __restrict__ scale,
                 const int* __restrict__ not_byte, uint8_t* __restrict__ Xb, unsigned* __restrict__ bin_lo, unsigned* __restrict__ bin_hi) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)P * D) return;
    const int f = (int)(gid / P), i = (int)(gid - (long long)f * P);
    const float v = X[(size_t)i * D + f];
    int b;
    if (!not_byte[f]) b = (int)v;
    else {
        b = (int)((v - lo[f]) * scale[f]);
        b = b < 0 ? 0 : (b > TR_BINS - 1 ? TR_BINS - 1 : b);
    }
    Xb[gid] = (uint8_t)b;
    const unsigned o = f2ord(v);
    atomicMin(&bin_lo[f * TR_BINS + b], o);
    atomicMax(&bin_hi[f * TR_BINS + b], o);
}

// hist[slot][k][bin][class] += weight of every bootstrap example whose node is in the current frontier batch
__global__ void __launch_bounds__(256)
train_hist_kernel(int P, int K, const int* __restrict__ node_of, const int* __restrict__ slot_of, const uint16_t* __restrict__ w,
                  const int* __restrict__ slot_layer, const int* __restrict__ slot_feat, const int* __restrict__ labels /* L x P */,
                  const uint8_t* __restrict__ Xb, unsigned* __restrict__ hist) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const unsigned wi = w[i];
    if (!wi) return;
    const int slot = slot_of[node_of[i]];
    if (slot < 0) return;
    const int c = labels[(size_t)slot_layer[slot] * P + i];
    for (int k = 0; k < K; k++) {
        const int f = slot_feat[slot * K + k];
        const int b = Xb[(size_t)f * P + i];
        atomicAdd(&hist[(((size_t)slot * K + k) * TR_BINS + b) * TR_CMAX + c], wi);
    }
}

struct CutResult {
    float objective;     // E(left) + E(right) of the best cut, 1e35 when the feature has a single occupied bin
    int bin, next_bin;   // the cut lies between these two occupied bins
    unsigned left_mass, right_mass;
    unsigned mass;       // node mass
    int n_classes;       // classes present in the node (1 = pure)
};

__device__ __forceinline__ float nlog2n(unsigned n) { return n ? (float)n * log2f((float)n) : 0.f; }

// one block per (slot, feature), one thread per bin
__global__ void __launch_bounds__(TR_BINS)
train_best_cut_kernel(int K, const unsigned* __restrict__ hist, CutResult* __restrict__ out) {
    __shared__ unsigned h[TR_BINS][TR_CMAX + 1];
    __shared__ unsigned total[TR_CMAX];
    __shared__ unsigned occ[TR_BINS];
    __shared__ float best_obj[TR_BINS];
    __shared__ int best_bin[TR_BINS];
    const int b = threadIdx.x;
    const unsigned* src = hist + ((size_t)blockIdx.x * TR_BINS + b) * TR_CMAX;
    unsigned row = 0;
#pragma unroll
    for (int c = 0; c < TR_CMAX; c++) { const unsigned v = src[c]; h[b][c] = v; row += v; }
    occ[b] = row;
    __syncthreads();
    if (b < TR_CMAX) {
        unsigned t = 0;
        for (int q = 0; q < TR_BINS; q++) t += h[q][b];
        total[b] = t;
    }
    __syncthreads();
    // next occupied bin above b
    int nb = -1;
    if (row) for (int q = b + 1; q < TR_BINS; q++) if (occ[q]) { nb = q; break; }
    float obj = 1e35f;
    unsigned lm = 0, mass = 0;
    if (nb >= 0) {
        float e_left = 0.f, e_right = 0.f;
        unsigned rm = 0;
#pragma unroll
        for (int c = 0; c < TR_CMAX; c++) {
            unsigned l = 0;
            for (int q = 0; q <= b; q++) l += h[q][c];
            const unsigned r = total[c] - l;
            lm += l; rm += r;
            e_left -= nlog2n(l);
            e_right -= nlog2n(r);
        }
        obj = (e_left + nlog2n(lm)) + (e_right + nlog2n(rm));
        mass = lm + rm;
    }
    best_obj[b] = obj;
    best_bin[b] = b;
    __syncthreads();
    for (int s = TR_BINS / 2; s > 0; s >>= 1) {   // arg min, the lower bin wins a tie (the reference keeps the first)
        if (b < s) {
            const float o2 = best_obj[b + s];
            const int b2 = best_bin[b + s];
            if (o2 < best_obj[b] || (o2 == best_obj[b] && b2 < best_bin[b])) { best_obj[b] = o2; best_bin[b] = b2; }
        }
        __syncthreads();
    }
    if (b == best_bin[0]) {
        CutResult r;
        r.objective = obj;
        r.bin = b;
        r.next_bin = nb;
        r.left_mass = lm;
        unsigned m = 0;
        int ncls = 0;
        for (int c = 0; c < TR_CMAX; c++) { m += total[c]; ncls += total[c] ? 1 : 0; }
        r.mass = m;
        r.right_mass = m - lm;
        r.n_classes = ncls;
        if (nb < 0) { r.objective = 1e35f; r.left_mass = 0; r.right_mass = m; }
        (void)mass;
        out[blockIdx.x] = r;
    }
}

// findLeafNode's rule on the freshly split nodes: every example (bootstrap or not) moves to a child
__global__ void __launch_bounds__(256)
train_route_kernel(const float* __restrict__ X, int P, int D, int* __restrict__ node_of, const int* __restrict__ split_feat,
                   const float* __restrict__ split_thr, const int* __restrict__ split_left) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int node = node_of[i];
    const int f = split_feat[node];
    if (f < 0) return;
    const float v = X[(size_t)i * D + f];
    node_of[i] = v < split_thr[node] ? split_left[node] : split_left[node] + 1;   // classifier.cpp:105
}

// integer leaf counts over ALL examples: cnt[node][layer][class]
__global__ void __launch_bounds__(256)
train_leaf_count_kernel(int P, int L, const int* __restrict__ node_of, const int* __restrict__ labels, unsigned* __restrict__ cnt) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)P * L) return;
    const int l = (int)(gid / P), i = (int)(gid - (long long)l * P);
    atomicAdd(&cnt[((size_t)node_of[i] * L + l) * TR_CMAX + labels[gid]], 1u);
}

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

rvseg_status rvseg_forest_train(rvseg_ctx* ctx, const float* X, int32_t P, int32_t D, const int32_t* labels, int32_t n_layers,
                                const int32_t* class_counts, const rvseg_train_params* tp_in, void* forest_out, size_t out_cap,
                                size_t* size_out) {
    if (!ctx) return RVSEG_ERR_INVALID_ARG;
    rvseg_train_params tp;
    if (tp_in) tp = *tp_in; else rvseg_train_params_default(&tp);
    if (!X || !labels || !class_counts || !size_out || P < 1 || D < 1 || n_layers < 1 || n_layers > RVSEG_MAX_LAYERS ||
        tp.num_trees < 1 || tp.num_trees > kMaxTrees || tp.max_depth < 1 || tp.min_split_examples < 0 || tp.min_child_split_examples < 0 ||
        tp.num_features < 0 || tp.num_features > D || !(tp.smoothing >= 0.f)) {
        ctx->err = "bad arguments";
        return RVSEG_ERR_INVALID_ARG;
    }
    int sumC = 0;
    for (int l = 0; l < n_layers; l++) {
        if (class_counts[l] < 1 || class_counts[l] > TR_CMAX) { ctx->err = "the trainer handles 1..16 classes per layer"; return RVSEG_ERR_INVALID_ARG; }
        sumC += class_counts[l];
    }
    if (sumC > kMaxClasses) { ctx->err = "more than 64 classes over all layers"; return RVSEG_ERR_INVALID_ARG; }
    for (long long q = 0; q < (long long)P * n_layers; q++) {
        const int l = (int)(q % n_layers);
        if (labels[q] < 0 || labels[q] >= class_counts[l]) { ctx->err = "label outside its layer's class range"; return RVSEG_ERR_INVALID_ARG; }
    }
    const int K = tp.num_features > 0 ? tp.num_features : (int)std::ceil(std::sqrt((double)D));
    RV_HIP(ctx, hipSetDevice(ctx->params.device));
    hipStream_t s = ctx->stream;
    DevArena A;
    bool ok = true;

    // ---- data set on the device --------------------------------------------------------------------
    float* dX = A.alloc<float>(ctx, (size_t)P * D, &ok);
    uint8_t* dXb = A.alloc<uint8_t>(ctx, (size_t)P * D, &ok);
    int* dLab = A.alloc<int>(ctx, (size_t)P * n_layers, &ok);        // layer-major
    unsigned* dFmin = A.alloc<unsigned>(ctx, D, &ok);
    unsigned* dFmax = A.alloc<unsigned>(ctx, D, &ok);
    int* dNotByte = A.alloc<int>(ctx, D, &ok);
    float* dLo = A.alloc<float>(ctx, D, &ok);
    float* dScale = A.alloc<float>(ctx, D, &ok);
    unsigned* dBinLo = A.alloc<unsigned>(ctx, (size_t)D * TR_BINS, &ok);
    unsigned* dBinHi = A.alloc<unsigned>(ctx, (size_t)D * TR_BINS, &ok);
    uint16_t* dW = A.alloc<uint16_t>(ctx, P, &ok);
    int* dNode = A.alloc<int>(ctx, P, &ok);
    if (!ok) return RVSEG_ERR_HIP;
    std::vector<int> lab_lm((size_t)P * n_layers);
    for (int i = 0; i < P; i++)
        for (int l = 0; l < n_layers; l++) lab_lm[(size_t)l * P + i] = labels[(size_t)i * n_layers + l];
    RV_HIP(ctx, hipMemcpyAsync(dX, X, (size_t)P * D * 4, hipMemcpyHostToDevice, s));
    RV_HIP(ctx, hipMemcpyAsync(dLab, lab_lm.data(), lab_lm.size() * 4, hipMemcpyHostToDevice, s));
    RV_HIP(ctx, hipMemsetAsync(dFmin, 0xFF, (size_t)D * 4, s));
    RV_HIP(ctx, hipMemsetAsync(dFmax, 0x00, (size_t)D * 4, s));
    RV_HIP(ctx, hipMemsetAsync(dNotByte, 0, (size_t)D * 4, s));
    RV_HIP(ctx, hipMemsetAsync(dBinLo, 0xFF, (size_t)D * TR_BINS * 4, s));
    RV_HIP(ctx, hipMemsetAsync(dBinHi, 0x00, (size_t)D * TR_BINS * 4, s));
    train_feature_stats_kernel<<<dim3((unsigned)D), dim3(256), 0, s>>>(dX, P, D, dFmin, dFmax, dNotByte);
    std::vector<unsigned> fmin(D), fmax(D);
    std::vector<int> not_byte(D);
    RV_HIP(ctx, hipMemcpyAsync(fmin.data(), dFmin, (size_t)D * 4, hipMemcpyDeviceToHost, s));
    RV_HIP(ctx, hipMemcpyAsync(fmax.data(), dFmax, (size_t)D * 4, hipMemcpyDeviceToHost, s));
    RV_HIP(ctx, hipMemcpyAsync(not_byte.data(), dNotByte, (size_t)D * 4, hipMemcpyDeviceToHost, s));
    RV_HIP(ctx, hipStreamSynchronize(s));
    std::vector<float> lo(D), scale(D);
    for (int f = 0; f < D; f++) {
        const float a = ord2f(fmin[f]), b = ord2f(fmax[f]);
        if (!std::isfinite(a) || !std::isfinite(b)) { ctx->err = "non-finite feature value in the training set"; return RVSEG_ERR_INVALID_ARG; }
        lo[f] = a;
        scale[f] = b > a ? (float)TR_BINS / (b - a) : 0.f;
    }
    RV_HIP(ctx, hipMemcpyAsync(dLo, lo.data(), (size_t)D * 4, hipMemcpyHostToDevice, s));
    RV_HIP(ctx, hipMemcpyAsync(dScale, scale.data(), (size_t)D * 4, hipMemcpyHostToDevice, s));
    train_bin_kernel<<<dim3((unsigned)(((long long)P * D + 255) / 256)), dim3(256), 0, s>>>(dX, P, D, dLo, dScale, dNotByte, dXb, dBinLo, dBinHi);
    std::vector<unsigned> bin_lo((size_t)D * TR_BINS), bin_hi((size_t)D * TR_BINS);
    RV_HIP(ctx, hipMemcpyAsync(bin_lo.data(), dBinLo, bin_lo.size() * 4, hipMemcpyDeviceToHost, s));
    RV_HIP(ctx, hipMemcpyAsync(bin_hi.data(), dBinHi, bin_hi.size() * 4, hipMemcpyDeviceToHost, s));
    RV_HIP(ctx, hipStreamSynchronize(s));
    RV_LAUNCH_OK(ctx);

    // inverted class frequencies over the whole set (data.h:346-370): freq[c] = size / count_c, in float
    std::vector<std::vector<float>> freq(n_layers);
    for (int l = 0; l < n_layers; l++) {
        freq[l].assign(class_counts[l], 0.f);
        for (int i = 0; i < P; i++) freq[l][lab_lm[(size_t)l * P + i]]++;
        for (int c = 0; c < class_counts[l]; c++) freq[l][c] = P / freq[l][c];
    }

    std::mt19937_64 rng(tp.seed);
    ForestModel model;
    model.raw.resize((size_t)tp.num_trees);
    const int SLOT_BATCH = 1024;
    unsigned* dHist = A.alloc<unsigned>(ctx, (size_t)SLOT_BATCH * K * TR_BINS * TR_CMAX, &ok);
    CutResult* dCut = A.alloc<CutResult>(ctx, (size_t)SLOT_BATCH * K, &ok);
    int* dSlotLayer = A.alloc<int>(ctx, SLOT_BATCH, &ok);
    int* dSlotFeat = A.alloc<int>(ctx, (size_t)SLOT_BATCH * K, &ok);
    if (!ok) return RVSEG_ERR_HIP;
    std::vector<CutResult> cuts((size_t)SLOT_BATCH * K);
    std::vector<int> all_features(D);

    for (int t = 0; t < tp.num_trees; t++) {
        // bootstrap: N draws with replacement (DataStorage::bootstrapmulti) as per-example multiplicities
        std::vector<uint16_t> w(P, tp.use_bootstrap ? 0 : 1);
        if (tp.use_bootstrap) {
            std::uniform_int_distribution<int> pick(0, P - 1);
            for (int n = 0; n < P; n++) { uint16_t& x = w[pick(rng)]; if (x < 65535) x++; }
        }
        RV_HIP(ctx, hipMemcpyAsync(dW, w.data(), (size_t)P * 2, hipMemcpyHostToDevice, s));
        RV_HIP(ctx, hipMemsetAsync(dNode, 0, (size_t)P * 4, s));
        RawTree& tree = model.raw[(size_t)t];
        auto add_node = [&]() {   // DecisionTree::addNode, classifier.cpp:66-74
            tree.feat.push_back(0); tree.thr.push_back(0.f); tree.left.push_back(0);
            tree.hist.emplace_back(); tree.mhist.emplace_back();
        };
        add_node();
        std::vector<int> depth(1, 0);
        std::vector<int> frontier(1, 0);
        while (!frontier.empty()) {
            const int n_nodes = (int)tree.left.size();
            // per-level device tables over all nodes: slot of a frontier node, and the splits decided in this level
            std::vector<int> slot_of(n_nodes, -1), split_feat(n_nodes, -1), split_left(n_nodes, 0);
            std::vector<float> split_thr(n_nodes, 0.f);
            std::vector<int> next_frontier;
            int* dSlotOf = A.alloc<int>(ctx, n_nodes, &ok);
            if (!ok) return RVSEG_ERR_HIP;
            for (size_t base = 0; base < frontier.size(); base += SLOT_BATCH) {
                const int S = (int)std::min<size_t>(SLOT_BATCH, frontier.size() - base);
                std::vector<int> slot_layer(S), slot_feat((size_t)S * K);
                std::fill(slot_of.begin(), slot_of.end(), -1);
                for (int q = 0; q < S; q++) {
                    slot_of[frontier[base + q]] = q;
                    slot_layer[q] = (int)(rng() % (unsigned)n_layers);                       // "Pick a random class layer", :483-485
                    for (int f = 0; f < D; f++) all_features[f] = f;                        // sample numFeatures without replacement, :537
                    for (int k = 0; k < K; k++) {
                        const int j = k + (int)(rng() % (unsigned)(D - k));
                        std::swap(all_features[k], all_features[j]);
                        slot_feat[(size_t)q * K + k] = all_features[k];
                    }
                }
                RV_HIP(ctx, hipMemcpyAsync(dSlotOf, slot_of.data(), (size_t)n_nodes * 4, hipMemcpyHostToDevice, s));
                RV_HIP(ctx, hipMemcpyAsync(dSlotLayer, slot_layer.data(), (size_t)S * 4, hipMemcpyHostToDevice, s));
                RV_HIP(ctx, hipMemcpyAsync(dSlotFeat, slot_feat.data(), (size_t)S * K * 4, hipMemcpyHostToDevice, s));
                RV_HIP(ctx, hipMemsetAsync(dHist, 0, (size_t)S * K * TR_BINS * TR_CMAX * 4, s));
                train_hist_kernel<<<dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s>>>(P, K, dNode, dSlotOf, dW, dSlotLayer, dSlotFeat, dLab, dXb, dHist);
                train_best_cut_kernel<<<dim3((unsigned)(S * K)), dim3(TR_BINS), 0, s>>>(K, dHist, dCut);
                RV_HIP(ctx, hipMemcpyAsync(cuts.data(), dCut, (size_t)S * K * sizeof(CutResult), hipMemcpyDeviceToHost, s));
                RV_HIP(ctx, hipStreamSynchronize(s));
                RV_LAUNCH_OK(ctx);
                for (int q = 0; q < S; q++) {
                    const int node = frontier[base + q];
                    const CutResult& first = cuts[(size_t)q * K];
                    // stop rules of learning.cpp:521-527: too few examples, pure, too deep
                    if ((int)first.mass < tp.min_split_examples || first.n_classes <= 1 || depth[node] > tp.max_depth) continue;
                    int best_k = -1;
                    float best_obj = 1e35f;
                    for (int k = 0; k < K; k++) {   // features in sampled order, strict '<' keeps the first best (:589)
                        const CutResult& c = cuts[(size_t)q * K + k];
                        if (c.next_bin >= 0 && c.objective < best_obj) { best_obj = c.objective; best_k = k; }
                    }
                    if (best_k < 0) continue;                                                   // bestFeature < 0, :611
                    const CutResult& c = cuts[(size_t)q * K + best_k];
                    if ((int)c.left_mass < tp.min_child_split_examples || (int)c.right_mass < tp.min_child_split_examples) continue;
                    const int f = slot_feat[(size_t)q * K + best_k];
                    const float left_value = ord2f(bin_hi[(size_t)f * TR_BINS + c.bin]);        // largest value on the left
                    const float right_value = ord2f(bin_lo[(size_t)f * TR_BINS + c.next_bin]);  // smallest value on the right
                    float thr = left_value + right_value;                                       // :592
                    thr *= 0.5f;                                                                // :607
                    if (!(left_value < thr)) thr = right_value;   // two adjacent floats: keep `x < thr` separating them
                    const int left = (int)tree.left.size();
                    add_node(); add_node();                                                     // DecisionTree::splitNode, classifier.cpp:77-95
                    depth.push_back(depth[node] + 1); depth.push_back(depth[node] + 1);
                    tree.feat[node] = f; tree.thr[node] = thr; tree.left[node] = left;
                    split_feat[node] = f; split_thr[node] = thr; split_left[node] = left;
                    next_frontier.push_back(left);
                    next_frontier.push_back(left + 1);
                }
            }
            if (!next_frontier.empty()) {
                int* dSF = A.alloc<int>(ctx, n_nodes, &ok);
                float* dST = A.alloc<float>(ctx, n_nodes, &ok);
                int* dSL = A.alloc<int>(ctx, n_nodes, &ok);
                if (!ok) return RVSEG_ERR_HIP;
                RV_HIP(ctx, hipMemcpyAsync(dSF, split_feat.data(), (size_t)n_nodes * 4, hipMemcpyHostToDevice, s));
                RV_HIP(ctx, hipMemcpyAsync(dST, split_thr.data(), (size_t)n_nodes * 4, hipMemcpyHostToDevice, s));
                RV_HIP(ctx, hipMemcpyAsync(dSL, split_left.data(), (size_t)n_nodes * 4, hipMemcpyHostToDevice, s));
                train_route_kernel<<<dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s>>>(dX, P, D, dNode, dSF, dST, dSL);
                RV_HIP(ctx, hipStreamSynchronize(s));
                RV_LAUNCH_OK(ctx);
            }
            frontier.swap(next_frontier);
        }
        // ---- leaf histograms from ALL examples (updateMultiHistograms, learning.cpp:960-1012) --------------
        const int n_nodes = (int)tree.left.size();
        unsigned* dCnt = A.alloc<unsigned>(ctx, (size_t)n_nodes * n_layers * TR_CMAX, &ok);
        if (!ok) return RVSEG_ERR_HIP;
        RV_HIP(ctx, hipMemsetAsync(dCnt, 0, (size_t)n_nodes * n_layers * TR_CMAX * 4, s));
        train_leaf_count_kernel<<<dim3((unsigned)(((long long)P * n_layers + 255) / 256)), dim3(256), 0, s>>>(P, n_layers, dNode, dLab, dCnt);
        std::vector<unsigned> cnt((size_t)n_nodes * n_layers * TR_CMAX);
        RV_HIP(ctx, hipMemcpyAsync(cnt.data(), dCnt, cnt.size() * 4, hipMemcpyDeviceToHost, s));
        RV_HIP(ctx, hipStreamSynchronize(s));
        RV_LAUNCH_OK(ctx);
        for (int v = 0; v < n_nodes; v++) {
            if (tree.left[v] != 0) continue;
            tree.mhist[v].resize((size_t)n_layers);
            for (int l = 0; l < n_layers; l++) {
                const int C = class_counts[l];
                std::vector<float>& h = tree.mhist[v][l];
                h.assign(C, 0.f);
                for (int c = 0; c < C; c++) {
                    // "hist[l][classlabel] += freq[classlabel]" once per example (:989-991): n additions of the same addend
                    const unsigned n = cnt[((size_t)v * n_layers + l) * TR_CMAX + c];
                    const float f = freq[l][c];
                    float acc = 0.f;
                    for (unsigned k = 0; k < n; k++) acc += f;
                    h[c] = acc;
                }
                float total = 0;
                for (int c = 0; c < C; c++) total += h[c];
                for (int c = 0; c < C; c++) h[c] = std::log((h[c] + tp.smoothing) / (total + C * tp.smoothing));   // :1004-1007
            }
            if (n_layers == 1) tree.hist[v] = tree.mhist[v][0];   // a single-layer forest also serves classLogPosterior
        }
    }
    const std::vector<uint8_t> bytes = serialize_forest(model);
    *size_out = bytes.size();
    if (!forest_out) return RVSEG_OK;
    if (out_cap < bytes.size()) { ctx->err = "output buffer too small"; return RVSEG_ERR_INVALID_ARG; }
    std::memcpy(forest_out, bytes.data(), bytes.size());
    return RVSEG_OK;
}

}  // extern "C"
