/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle of the forest learner.
 *
 * A plain-C restatement of the reference's shared multi-layer learner, written from the cited lines
 * (all relative to /root/reference/third-party/libforest):
 *   DecisionTreeLearner::learn, multi-layer branch     src/learning.cpp:410-662
 *   EfficientEntropyHistogram (objective)              src/learning.cpp:27-295
 *   fastlog2                                           src/fastlog.h:47-58
 *   updateMultiHistograms                              src/learning.cpp:960-1012
 *   getInvertedClassFrequency                          include/libforest/data.h:358-370
 *   DecisionTree::splitNode / write                    src/classifier.cpp:77-95,144-152,210-220
 *
 * Depth-first over an explicit stack with one sort per (node, feature), like the reference.  It is never
 * linked or called by the product; tests compare the HIP trainer's forest.dat with this one byte for byte.
 *
 * PARITY UNPINNED at the reference level, by nature: the reference learner cannot be built here (data.cpp needs
 * Boost) and is not a function of its inputs -- it draws from std::random_device (learning.cpp:18,469,484,543) and
 * accumulates its objective incrementally in float over an std::sort order that leaves equal keys unspecified.
 * Build-owned definitions (shared with csrc/rvseg_train.hip, each side written independently from this text):
 *   1. Random choices come from a counter-based generator keyed by (seed, tree, node path), so the tree does not
 *      depend on the order nodes are visited in:
 *        mix(z):  z += 0x9E3779B97F4A7C15; z = (z ^ z>>30) * 0xBF58476D1CE4E5B9; z = (z ^ z>>27) * 0x94D049BB133111EB;
 *                 return z ^ z>>31                                   (splitmix64 finaliser)
 *        draw(key, i) = mix(key ^ mix(i + 0x632BE59BD9B4E019))
 *        tree key  kt = mix(seed ^ mix(0x74726565 + tree));  bootstrap draw n = draw(kt, 2^32 + n) mod P
 *        node key: root = mix(kt ^ 0x726F6F74), child = mix(parent ^ (0x4C for left, 0x52 for right))
 *        layer = draw(key, 0) mod L  (:483-485);  features: from the identity permutation, for k < K:
 *        j = k + draw(key, 1 + k) mod (D - k), swap(perm[k], perm[j])  (:537; the reference shuffles a persistent
 *        array, which makes the subset depend on the visiting order)
 *   2. The objective of a cut is E(left) + E(right) with E evaluated from the class counts by the expression of
 *      initEntropies (:279-293): E = -ENTROPY(mass) + sum over classes in ascending order with a non-zero count of
 *      ENTROPY(count), ENTROPY(p) = -(p) * fastlog2(p), all in float.  The reference's running value (addOne / subOne,
 *      :220-258) is the same quantity plus a rounding drift that depends on the unspecified order of equal keys.
 *   3. Bootstrap duplicates are kept as one example with a multiplicity (the reference stores copies: adjacent equal
 *      values, which never yield a cut, :578-585).
 *   4. threshold = (left + right) * 0.5f (:592,607); if that rounds down to `left` (adjacent floats) the threshold is
 *      `right`, so that `x < threshold` still separates the two (the reference would write past its child lists).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "rvseg_oracle.h"

static uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint64_t draw64(uint64_t key, uint64_t i) { return mix64(key ^ mix64(i + 0x632BE59BD9B4E019ull)); }

/* fastlog.h:47-58 */
static float fastlog2_f(float x) {
    union { float f; uint32_t i; } vx;
    union { uint32_t i; float f; } mx;
    vx.f = x;
    mx.i = (vx.i & 0x007FFFFFu) | 0x3f000000u;
    float y = (float)vx.i;
    y *= 1.1920928955078125e-7f;
    const float a = 1.498030302f * mx.f;
    const float b = 1.72587999f / (0.3520887068f + mx.f);
    float r = y - 124.22551499f;
    r = r - a;
    r = r - b;
    return r;
}
static float entropy_term(float p) { return (-p) * fastlog2_f(p); }   /* ENTROPY(p), learning.cpp:13 */

/* initEntropies (:279-293) from integer counts */
static float hist_entropy(const int *cnt, int C) {
    int mass = 0;
    for (int c = 0; c < C; c++) mass += cnt[c];
    float total = -entropy_term((float)mass);
    for (int c = 0; c < C; c++) {
        if (cnt[c] == 0) continue;
        total += entropy_term((float)cnt[c]);
    }
    return total;
}

typedef struct { float v; int idx; } sort_item;
static int cmp_item(const void *a, const void *b) {
    const sort_item *x = (const sort_item *)a, *y = (const sort_item *)b;
    if (x->v < y->v) return -1;
    if (x->v > y->v) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

typedef struct {
    int *feat; float *thr; int *left; int *depth; uint64_t *key;
    int **ex; int *n_ex;   /* example lists of nodes still to be split (freed when popped) */
    int n, cap;
} tree_build;

static int tb_add(tree_build *t, int depth, uint64_t key) {
    if (t->n == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 64;
        t->feat = realloc(t->feat, sizeof(int) * t->cap);
        t->thr = realloc(t->thr, sizeof(float) * t->cap);
        t->left = realloc(t->left, sizeof(int) * t->cap);
        t->depth = realloc(t->depth, sizeof(int) * t->cap);
        t->key = realloc(t->key, sizeof(uint64_t) * t->cap);
        t->ex = realloc(t->ex, sizeof(int *) * t->cap);
        t->n_ex = realloc(t->n_ex, sizeof(int) * t->cap);
    }
    const int v = t->n++;
    t->feat[v] = 0; t->thr[v] = 0.f; t->left[v] = 0; t->depth[v] = depth; t->key[v] = key;
    t->ex[v] = NULL; t->n_ex[v] = 0;
    return v;
}

/* growable byte buffer for the forest.dat stream (io.h:84-108: every vector is int32 n + n elements) */
typedef struct { uint8_t *p; size_t n, cap; } bytes;
static void put(bytes *b, const void *src, size_t n) {
    if (b->n + n > b->cap) { b->cap = (b->n + n) * 2 + 1024; b->p = realloc(b->p, b->cap); }
    memcpy(b->p + b->n, src, n);
    b->n += n;
}
static void put_i32(bytes *b, int32_t v) { put(b, &v, 4); }

int orc_forest_train(const float *X, int P, int D, const int32_t *labels, int n_layers, const int32_t *class_counts,
                     int num_trees, int max_depth, int min_split_examples, int min_child_split_examples, int num_features,
                     int use_bootstrap, float smoothing, uint64_t seed, void **forest_out, size_t *size_out) {
    if (!X || !labels || !class_counts || !forest_out || !size_out || P < 1 || D < 1 || n_layers < 1 || n_layers > 8) return -1;
    const int K = num_features > 0 ? num_features : (int)ceil(sqrt((double)D));   /* autoconf, learning.cpp:363-368 */
    int cmax = 0;
    for (int l = 0; l < n_layers; l++) cmax = class_counts[l] > cmax ? class_counts[l] : cmax;
    /* inverted class frequencies over the whole set, data.h:358-370 */
    float **freq = malloc(sizeof(float *) * n_layers);
    for (int l = 0; l < n_layers; l++) {
        freq[l] = calloc(class_counts[l], sizeof(float));
        for (int i = 0; i < P; i++) freq[l][labels[(size_t)i * n_layers + l]]++;
        for (int c = 0; c < class_counts[l]; c++) freq[l][c] = P / freq[l][c];
    }
    bytes out = {0};
    put_i32(&out, num_trees);
    int *w = malloc(sizeof(int) * P);
    int *perm = malloc(sizeof(int) * D);
    int *hist = malloc(sizeof(int) * cmax), *lh = malloc(sizeof(int) * cmax), *rh = malloc(sizeof(int) * cmax);
    sort_item *items = malloc(sizeof(sort_item) * P);
    int *stack = malloc(sizeof(int) * (size_t)(2 * (max_depth + 4) + 64));
    for (int t = 0; t < num_trees; t++) {
        const uint64_t kt = mix64(seed ^ mix64(0x74726565ull + (uint64_t)t));
        /* bootstrap: P draws with replacement (data.cpp:325-349, numBootstrapExamples = size: learning.cpp:366) */
        for (int i = 0; i < P; i++) w[i] = use_bootstrap ? 0 : 1;
        if (use_bootstrap) for (int n = 0; n < P; n++) w[draw64(kt, 0x100000000ull + (uint64_t)n) % (uint64_t)P]++;
        tree_build tb = {0};
        tb_add(&tb, 0, mix64(kt ^ 0x726F6F74ull));
        int n_root = 0;
        for (int i = 0; i < P; i++) n_root += w[i] ? 1 : 0;
        tb.ex[0] = malloc(sizeof(int) * (n_root ? n_root : 1));
        for (int i = 0, q = 0; i < P; i++) if (w[i]) tb.ex[0][q++] = i;
        tb.n_ex[0] = n_root;
        int sp = 0;
        stack[sp++] = 0;
        while (sp > 0) {
            const int node = stack[--sp];
            int *ex = tb.ex[node];
            const int n = tb.n_ex[node];
            tb.ex[node] = NULL;
            const uint64_t key = tb.key[node];
            const int layer = (int)(draw64(key, 0) % (uint64_t)n_layers);   /* :483-485 */
            const int C = class_counts[layer];
            int mass = 0, present = 0;
            for (int c = 0; c < C; c++) hist[c] = 0;
            for (int m = 0; m < n; m++) hist[labels[(size_t)ex[m] * n_layers + layer]] += w[ex[m]];
            for (int c = 0; c < C; c++) { mass += hist[c]; present += hist[c] ? 1 : 0; }
            /* :521-527: too few examples, pure, too deep */
            if (mass < min_split_examples || present <= 1 || tb.depth[node] > max_depth) { free(ex); continue; }
            float best_obj = 1e35f, best_left_v = 0.f, best_right_v = 0.f;
            int best_feature = -1, best_lm = 0, best_rm = mass;
            for (int d = 0; d < D; d++) perm[d] = d;
            for (int k = 0; k < K && k < D; k++) {
                const int j = k + (int)(draw64(key, 1 + (uint64_t)k) % (uint64_t)(D - k));
                const int tmp = perm[k]; perm[k] = perm[j]; perm[j] = tmp;
            }
            for (int k = 0; k < K && k < D; k++) {   /* :540-604 */
                const int f = perm[k];
                for (int m = 0; m < n; m++) { items[m].v = X[(size_t)ex[m] * D + f]; items[m].idx = ex[m]; }
                qsort(items, n, sizeof(sort_item), cmp_item);
                for (int c = 0; c < C; c++) { lh[c] = 0; rh[c] = hist[c]; }
                int lm = 0;
                for (int m = 1; m < n; m++) {
                    const int prev = items[m - 1].idx;
                    const int pc = labels[(size_t)prev * n_layers + layer];
                    lh[pc] += w[prev]; rh[pc] -= w[prev]; lm += w[prev];
                    const float diff = items[m].v - items[m - 1].v;
                    if (diff < 1e-6f) continue;                                 /* :578-585 */
                    const float obj = hist_entropy(lh, C) + hist_entropy(rh, C);   /* :588 */
                    if (obj < best_obj) {
                        best_obj = obj; best_feature = f;
                        best_left_v = items[m - 1].v; best_right_v = items[m].v;
                        best_lm = lm; best_rm = mass - lm;
                    }
                }
            }
            float thr = best_left_v + best_right_v;   /* :592 */
            thr *= 0.5f;                              /* :607 */
            if (best_feature >= 0 && !(best_left_v < thr)) thr = best_right_v;
            if (best_feature < 0 || best_lm < min_child_split_examples || best_rm < min_child_split_examples) { free(ex); continue; }   /* :610-617 */
            int nl = 0;
            for (int m = 0; m < n; m++) nl += X[(size_t)ex[m] * D + best_feature] < thr ? 1 : 0;
            const int left = tb_add(&tb, tb.depth[node] + 1, mix64(key ^ 0x4Cull));   /* splitNode: two nodes appended, classifier.cpp:77-95 */
            tb_add(&tb, tb.depth[node] + 1, mix64(key ^ 0x52ull));
            tb.ex[left] = malloc(sizeof(int) * (nl ? nl : 1));
            tb.ex[left + 1] = malloc(sizeof(int) * (n - nl ? n - nl : 1));
            int a = 0, b2 = 0;
            for (int m = 0; m < n; m++) {
                if (X[(size_t)ex[m] * D + best_feature] < thr) tb.ex[left][a++] = ex[m];   /* :632-643 */
                else tb.ex[left + 1][b2++] = ex[m];
            }
            tb.n_ex[left] = a; tb.n_ex[left + 1] = b2;
            tb.feat[node] = best_feature; tb.thr[node] = thr; tb.left[node] = left;
            stack[sp++] = left;          /* :654-655: the right child is popped first */
            stack[sp++] = left + 1;
            free(ex);
        }
        /* leaf histograms from ALL examples (updateMultiHistograms, :960-1012) */
        const int nn = tb.n;
        int *cnt = calloc((size_t)nn * n_layers * cmax, sizeof(int));
        for (int i = 0; i < P; i++) {
            int v = 0;
            while (tb.left[v] != 0) v = X[(size_t)i * D + tb.feat[v]] < tb.thr[v] ? tb.left[v] : tb.left[v] + 1;   /* classifier.cpp:97-117 */
            for (int l = 0; l < n_layers; l++) cnt[((size_t)v * n_layers + l) * cmax + labels[(size_t)i * n_layers + l]]++;
        }
        /* DecisionTree::write, classifier.cpp:144-152: splitFeatures, thresholds, leftChild, histograms, multi_histograms */
        put_i32(&out, nn); put(&out, tb.feat, sizeof(int) * nn);
        put_i32(&out, nn); put(&out, tb.thr, sizeof(float) * nn);
        put_i32(&out, nn); put(&out, tb.left, sizeof(int) * nn);
        float *h = malloc(sizeof(float) * cmax);
        for (int pass = 0; pass < 2; pass++) {
            put_i32(&out, nn);
            for (int v = 0; v < nn; v++) {
                const int leaf = tb.left[v] == 0;
                if (pass == 0) {   /* `histograms`: filled only by a single-layer forest (it also serves classLogPosterior) */
                    if (!leaf || n_layers != 1) { put_i32(&out, 0); continue; }
                } else {
                    if (!leaf) { put_i32(&out, 0); continue; }
                    put_i32(&out, n_layers);
                }
                for (int l = 0; l < (pass == 0 ? 1 : n_layers); l++) {
                    const int C = class_counts[l];
                    float total = 0;
                    for (int c = 0; c < C; c++) {
                        const int nrep = cnt[((size_t)v * n_layers + l) * cmax + c];
                        float acc = 0.f;
                        for (int r = 0; r < nrep; r++) acc += freq[l][c];   /* :989-991, one addition per example */
                        h[c] = acc;
                    }
                    for (int c = 0; c < C; c++) total += h[c];
                    for (int c = 0; c < C; c++) h[c] = logf((h[c] + smoothing) / (total + C * smoothing));   /* :1004-1007 */
                    put_i32(&out, C);
                    put(&out, h, sizeof(float) * C);
                }
            }
        }
        free(h); free(cnt);
        for (int v = 0; v < tb.n; v++) free(tb.ex[v]);
        free(tb.feat); free(tb.thr); free(tb.left); free(tb.depth); free(tb.key); free(tb.ex); free(tb.n_ex);
    }
    for (int l = 0; l < n_layers; l++) free(freq[l]);
    free(freq); free(w); free(perm); free(hist); free(lh); free(rh); free(items); free(stack);
    *forest_out = out.p;
    *size_out = out.n;
    return 0;
}

void orc_free(void *p) { free(p); }
