// TEST INFRASTRUCTURE ONLY -- never linked into the product library.
//
// Harness around the REFERENCE's own libforest evaluator, compiled in place from
// /root/reference/third-party/libforest/src/classifier.cpp (see ../Makefile, target _ref).
// Nothing of the reference is copied here: this file only calls the reference's public API
//   DecisionTree::{setSplitFeature,setThreshold,splitNode,getHistogram,getMultiHistogram}
//       (third-party/libforest/include/libforest/classifiers.h:70-150)
//   RandomForest::{addTree,write,read,classLogPosterior,multiClassLogPosterior}
//       (third-party/libforest/src/classifier.cpp:166-235)
// to (a) write seeded synthetic forests with the reference's own serialiser and (b) evaluate
// points with the reference's own tree walk, so that the C restatement in oracle/rvseg_oracle.c
// and the HIP path can be pinned bit-exactly against it.
//
// DataPoint's destructor lives in data.cpp, which needs Boost (absent in this image, and no
// stand-in header is written): the harness therefore never destroys a DataPoint and links with
// --unresolved-symbols=ignore-all; the only unresolved symbol is libf::DataPoint::freeData().
//
// usage:
//   libforest_ref gen  <out.dat> <seed> <trees> <leaves_per_tree> <max_depth> <D> <C_single> <L> <C_0> .. <C_{L-1}>
//   libforest_ref eval <forest.dat> <points.f32> <D> single|multi <out.f32>
#include "libforest/classifiers.h"
#include "libforest/data.h"
#include "libforest/io.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

namespace {

struct Rng {  // splitmix64: deterministic, implementation-independent
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uni() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
    int below(int n) { return (int)(next() % (uint64_t)n); }
};

// Feature value ranges of the Segmenter's 366-D vector: Lab bytes,
// depth [0.5,15] m, height [-1,3] m, normal angle [0,pi/2] (or -2).  For other D all features
// are treated as bytes.
void feature_range(int f, int D, float& lo, float& hi) {
    lo = 0.f; hi = 255.f;
    if (D == 366) {
        if (f == 363) { lo = 0.5f; hi = 15.f; }
        if (f == 364) { lo = -1.f; hi = 3.f; }
        if (f == 365) { lo = -2.f; hi = 1.5707964f; }
    }
}

void fill_hist(Rng& rng, std::vector<float>& h, int C) {
    // log of a normalised uniform(1e-3,1) draw: the shape of the trained leaf histograms,
    // which store log((h+1)/(tot+C)) (third-party/libforest/src/learning.cpp:1001-1008).
    std::vector<double> p(C);
    double tot = 0;
    for (int c = 0; c < C; c++) { p[c] = 1e-3 + (1.0 - 1e-3) * rng.uni(); tot += p[c]; }
    h.resize(C);
    for (int c = 0; c < C; c++) h[c] = (float)std::log(p[c] / tot);
}

int cmd_gen(int argc, char** argv) {
    if (argc < 10) { std::fprintf(stderr, "gen: too few arguments\n"); return 2; }
    const char* out = argv[2];
    uint64_t seed = std::strtoull(argv[3], 0, 10);
    int T = std::atoi(argv[4]), leaves = std::atoi(argv[5]), max_depth = std::atoi(argv[6]);
    int D = std::atoi(argv[7]), Csingle = std::atoi(argv[8]), L = std::atoi(argv[9]);
    if (argc < 10 + L) { std::fprintf(stderr, "gen: missing class counts\n"); return 2; }
    std::vector<int> CL(L);
    for (int l = 0; l < L; l++) CL[l] = std::atoi(argv[10 + l]);

    Rng rng(seed);
    libf::RandomForest* forest = new libf::RandomForest();
    for (int t = 0; t < T; t++) {
        libf::DecisionTree* tree = new libf::DecisionTree();
        // Grow like the learner does (learning.cpp:650-651): LIFO stack, children appended
        // pair-wise at split time, so node ids come out in DFS rather than breadth-first order.
        std::vector<int> stack, depth_of(1, 0);
        stack.push_back(0);
        int n_leaves = 1;
        std::vector<int> final_leaves;
        while (!stack.empty()) {
            // pick a random open node half of the time so that depths vary a lot
            size_t pick = stack.size() - 1;
            if (rng.uni() < 0.5) pick = (size_t)rng.below((int)stack.size());
            int node = stack[pick];
            stack.erase(stack.begin() + (long)pick);
            int dep = depth_of[node];
            bool can_split = n_leaves < leaves && dep < max_depth;
            if (!can_split) { final_leaves.push_back(node); continue; }
            int f = rng.below(D);
            float lo, hi; feature_range(f, D, lo, hi);
            float thr = lo + (hi - lo) * (float)rng.uni();
            if (f < 363 || D != 366) {
                // byte-valued features: put a third of the thresholds exactly on an integer so
                // that the strict '<' of findLeafNode (classifier.cpp:105) is exercised on ties
                if (rng.uni() < 0.34) thr = std::floor(thr);
            }
            tree->setSplitFeature(node, f);
            tree->setThreshold(node, thr);
            int left = tree->splitNode(node);
            depth_of.resize(left + 2);
            depth_of[left] = depth_of[left + 1] = dep + 1;
            stack.push_back(left);
            stack.push_back(left + 1);
            n_leaves++;
        }
        for (size_t i = 0; i < final_leaves.size(); i++) {
            int node = final_leaves[i];
            if (Csingle > 0) fill_hist(rng, tree->getHistogram(node), Csingle);
            if (L > 0) {
                std::vector<std::vector<float> >& mh = tree->getMultiHistogram(node);
                mh.resize(L);
                for (int l = 0; l < L; l++) fill_hist(rng, mh[l], CL[l]);
            }
        }
        forest->addTree(tree);
    }
    std::ofstream os(out, std::ios::binary);
    if (!os.is_open()) { std::fprintf(stderr, "cannot open %s\n", out); return 1; }
    forest->write(os);  // the reference's serialiser (classifier.cpp:210-220)
    os.close();
    return 0;
}

int cmd_eval(int argc, char** argv) {
    if (argc < 7) { std::fprintf(stderr, "eval: too few arguments\n"); return 2; }
    std::ifstream is(argv[2], std::ios::binary);
    if (!is.is_open()) { std::fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
    libf::RandomForest* forest = new libf::RandomForest();
    forest->read(is);  // the reference's reader (classifier.cpp:222-235)
    int D = std::atoi(argv[4]);
    bool multi = std::strcmp(argv[5], "multi") == 0;

    FILE* fp = std::fopen(argv[3], "rb");
    if (!fp) { std::fprintf(stderr, "cannot open %s\n", argv[3]); return 1; }
    std::fseek(fp, 0, SEEK_END);
    long bytes = std::ftell(fp);
    std::fseek(fp, 0, SEEK_SET);
    long P = bytes / (long)(sizeof(float) * D);
    std::vector<float> X((size_t)P * D);
    if (std::fread(X.data(), sizeof(float), X.size(), fp) != X.size()) return 1;
    std::fclose(fp);

    FILE* fo = std::fopen(argv[6], "wb");
    if (!fo) { std::fprintf(stderr, "cannot open %s\n", argv[6]); return 1; }
    for (long i = 0; i < P; i++) {
        libf::DataPoint* x = new libf::DataPoint(D);  // never deleted, see header
        for (int k = 0; k < D; k++) x->at(k) = X[(size_t)i * D + k];
        if (multi) {
            std::vector<std::vector<float> > post;
            forest->multiClassLogPosterior(x, post);  // classifier.cpp:187-208
            for (size_t l = 0; l < post.size(); l++)
                std::fwrite(post[l].data(), sizeof(float), post[l].size(), fo);
        } else {
            std::vector<float> post;
            forest->classLogPosterior(x, post);  // classifier.cpp:166-184
            std::fwrite(post.data(), sizeof(float), post.size(), fo);
        }
    }
    std::fclose(fo);
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc >= 2 && std::strcmp(argv[1], "gen") == 0) return cmd_gen(argc, argv);
    if (argc >= 2 && std::strcmp(argv[1], "eval") == 0) return cmd_eval(argc, argv);
    std::fprintf(stderr, "usage: libforest_ref gen|eval ...\n");
    return 2;
}
