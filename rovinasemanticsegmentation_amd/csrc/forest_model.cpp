#include "forest_model.h"

#include <cstring>
#include <queue>

namespace rvseg {
namespace {

struct Reader {
    const uint8_t* p;
    size_t left;
    bool bad = false;
    int32_t i32() {
        int32_t v = 0;
        if (left < 4) { bad = true; return 0; }
        std::memcpy(&v, p, 4);
        p += 4; left -= 4;
        return v;
    }
    float f32() {
        float v = 0;
        if (left < 4) { bad = true; return 0; }
        std::memcpy(&v, p, 4);
        p += 4; left -= 4;
        return v;
    }
    // length prefix of a vector whose elements take at least `min_elem_bytes` each
    int32_t len(size_t min_elem_bytes) {
        int32_t n = i32();
        if (bad || n < 0 || (size_t)n * min_elem_bytes > left) { bad = true; return 0; }
        return n;
    }
};

bool read_tree(Reader& r, RawTree& t) {
    int32_t n = r.len(4);
    t.feat.resize(n);
    for (auto& v : t.feat) v = r.i32();
    n = r.len(4);
    t.thr.resize(n);
    for (auto& v : t.thr) v = r.f32();
    n = r.len(4);
    t.left.resize(n);
    for (auto& v : t.left) v = r.i32();
    n = r.len(4);
    t.hist.resize(n);
    for (auto& h : t.hist) {
        int32_t c = r.len(4);
        h.resize(c);
        for (auto& v : h) v = r.f32();
    }
    n = r.len(4);
    t.mhist.resize(n);
    for (auto& mh : t.mhist) {
        int32_t L = r.len(4);
        mh.resize(L);
        for (auto& h : mh) {
            int32_t c = r.len(4);
            h.resize(c);
            for (auto& v : h) v = r.f32();
        }
    }
    return !r.bad;
}

}  // namespace

bool parse_forest(const void* buf, size_t size, int feature_length, ForestModel& out,
                  std::string& err) {
    out = ForestModel();
    if (!buf) { err = "null forest buffer"; return false; }
    Reader r{static_cast<const uint8_t*>(buf), size};
    int32_t T = r.i32();
    if (r.bad || T < 0 || (size_t)T * 20 > size) { err = "truncated or corrupt forest header"; return false; }
    if (T == 0) { err = "forest has no trees"; return false; }
    if (T > kMaxTrees) {
        // libforest has no tree limit; the device evaluator keeps at most 16 leaf rows in each of a
        // point's 4 lanes (kernels_rf.hip), so larger ensembles are refused instead of mis-evaluated
        err = "forest has " + std::to_string(T) + " trees; at most " + std::to_string(kMaxTrees) + " are supported";
        return false;
    }
    out.n_trees = T;
    out.raw.resize((size_t)T);
    bool have_single = true, have_multi = true, first_leaf = true;

    for (int t = 0; t < T; t++) {
        RawTree& raw = out.raw[(size_t)t];
        if (!read_tree(r, raw)) { err = "truncated forest stream in tree " + std::to_string(t); return false; }
        const size_t n = raw.left.size();
        if (n == 0 || raw.feat.size() != n || raw.thr.size() != n || raw.hist.size() != n || raw.mhist.size() != n) {
            err = "inconsistent vector lengths in tree " + std::to_string(t);
            return false;
        }
        // breadth-first renumbering; also proves the child links form a tree
        std::vector<int32_t> new_id(n, -1), order;
        std::vector<int> depth(n, 0);
        order.reserve(n);
        std::queue<int32_t> q;
        q.push(0);
        new_id[0] = 0;
        int32_t next = 1;
        while (!q.empty()) {
            int32_t node = q.front();
            q.pop();
            order.push_back(node);
            int32_t l = raw.left[node];
            if (l == 0) continue;
            if (l < 0 || (size_t)l + 1 >= n || new_id[l] != -1 || new_id[l + 1] != -1) {
                err = "bad child link at node " + std::to_string(node) + " of tree " + std::to_string(t);
                return false;
            }
            if (raw.feat[node] < 0 || raw.feat[node] >= feature_length) {
                err = "split feature " + std::to_string(raw.feat[node]) + " outside the " +
                      std::to_string(feature_length) + "-dimensional feature vector (model/config mismatch)";
                return false;
            }
            new_id[l] = next++;
            new_id[l + 1] = next++;
            depth[l] = depth[l + 1] = depth[node] + 1;
            if (depth[l] > out.max_depth) out.max_depth = depth[l];
            q.push(l);
            q.push(l + 1);
        }
        const int32_t base = (int32_t)out.nodes.size();
        out.roots.push_back(base);
        out.nodes.resize(out.nodes.size() + order.size());
        for (size_t k = 0; k < order.size(); k++) {
            const int32_t node = order[k];
            DeviceNode dn{};
            dn.feature = raw.feat[node];
            dn.threshold = raw.thr[node];
            dn.left = 0;
            dn.leaf_row = -1;
            if (raw.left[node] != 0) {
                dn.left = base + new_id[raw.left[node]];
            } else {
                dn.feature = 0;
                dn.leaf_row = out.n_leaves++;
                // class counts must agree over all leaves of all trees
                const auto& h = raw.hist[node];
                const auto& mh = raw.mhist[node];
                if (first_leaf) {
                    out.single_classes = (int)h.size();
                    out.layer_classes.clear();
                    for (const auto& l : mh) out.layer_classes.push_back((int)l.size());
                    have_single = !h.empty();
                    have_multi = !mh.empty();
                    first_leaf = false;
                }
                if (have_single) {
                    if ((int)h.size() != out.single_classes) { err = "leaf histogram sizes differ"; return false; }
                    out.single_hist.insert(out.single_hist.end(), h.begin(), h.end());
                }
                if (have_multi) {
                    if (mh.size() != out.layer_classes.size()) { err = "leaf layer counts differ"; return false; }
                    for (size_t l = 0; l < mh.size(); l++) {
                        if ((int)mh[l].size() != out.layer_classes[l]) { err = "leaf multi-histogram sizes differ"; return false; }
                        out.multi_hist.insert(out.multi_hist.end(), mh[l].begin(), mh[l].end());
                    }
                }
            }
            out.nodes[(size_t)base + k] = dn;
        }
    }
    if (!have_single) out.single_classes = 0;
    if (!have_multi) out.layer_classes.clear();
    if (out.single_classes == 0 && out.layer_classes.empty()) { err = "forest carries no leaf histograms"; return false; }
    return true;
}

std::vector<uint8_t> serialize_forest(const ForestModel& m) {
    std::vector<uint8_t> o;
    auto put_i = [&](int32_t v) { uint8_t b[4]; std::memcpy(b, &v, 4); o.insert(o.end(), b, b + 4); };
    auto put_f = [&](float v) { uint8_t b[4]; std::memcpy(b, &v, 4); o.insert(o.end(), b, b + 4); };
    put_i((int32_t)m.raw.size());                      // writeBinary(stream, getSize()), classifier.cpp:213
    for (const RawTree& t : m.raw) {                   // DecisionTree::write, classifier.cpp:144-152
        put_i((int32_t)t.feat.size());
        for (int32_t v : t.feat) put_i(v);
        put_i((int32_t)t.thr.size());
        for (float v : t.thr) put_f(v);
        put_i((int32_t)t.left.size());
        for (int32_t v : t.left) put_i(v);
        put_i((int32_t)t.hist.size());
        for (const auto& h : t.hist) {
            put_i((int32_t)h.size());
            for (float v : h) put_f(v);
        }
        put_i((int32_t)t.mhist.size());
        for (const auto& mh : t.mhist) {
            put_i((int32_t)mh.size());
            for (const auto& h : mh) {
                put_i((int32_t)h.size());
                for (float v : h) put_f(v);
            }
        }
    }
    return o;
}

}  // namespace rvseg
