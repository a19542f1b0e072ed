// Host-side forest model: parser for the libforest stream format and the breadth-first
// re-layout that the HIP traversal kernels consume.
//
// Format (reference: third-party/libforest/src/classifier.cpp:134-152,210-235 and
// include/libforest/io.h:34-108): int32 T; per tree five length-prefixed vectors
//   vec<int32> splitFeatures, vec<f32> thresholds, vec<int32> leftChild,
//   vec<vec<f32>> histograms, vec<vec<vec<f32>>> multi_histograms
// where vec = int32 n + n elements, native little-endian, no header.  Node 0 is the root,
// leftChild == 0 marks a leaf, the right child is left + 1 (classifiers.h:169-180).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace rvseg {

// One node of the breadth-first array, 16 bytes so that a lane fetches it with one dwordx4 load.
struct DeviceNode {
    int32_t feature;   // split feature index (inner nodes)
    float threshold;   // go left iff x[feature] < threshold  (classifier.cpp:105)
    int32_t left;      // absolute index of the left child in the node array; 0 <=> leaf
    int32_t leaf_row;  // leaves: row in the histogram tables; inner nodes on the device: packed patch
                       // coordinates of `feature` (channel << 16 | dy << 8 | dx), see upload_forest
};

// One tree exactly as the stream holds it (libforest classifiers.h:191-206): node order, leaf
// feature / threshold slots and inner-node histograms included, so that a model read from a file
// is written back byte for byte (RandomForest::write, classifier.cpp:144-152,210-220).
struct RawTree {
    std::vector<int32_t> feat, left;
    std::vector<float> thr;
    std::vector<std::vector<float>> hist;
    std::vector<std::vector<std::vector<float>>> mhist;
};

// Limits of the device evaluator (kernels_rf.hip: leaf_rows[16] x 4 lanes per point; one lane per class).
constexpr int kMaxTrees = 64;
constexpr int kMaxClasses = 64;

struct ForestModel {
    std::vector<RawTree> raw;             // the stream's own trees (what serialize_forest writes)
    int n_trees = 0;
    int max_depth = 0;                    // longest root->leaf path, in edges
    std::vector<int32_t> roots;           // per tree: index of its root in `nodes`
    std::vector<DeviceNode> nodes;        // all trees, each in breadth-first order
    int n_leaves = 0;
    // single-label histograms: n_leaves x single_classes (empty when the file has none)
    int single_classes = 0;
    std::vector<float> single_hist;
    // multi-layer histograms: n_leaves x sum(layer_classes), layers concatenated per leaf
    std::vector<int> layer_classes;
    std::vector<float> multi_hist;
};

// Parses and validates `buf`.  `feature_length` bounds the split feature indices (the reference
// segfaults on a model/config mismatch, README.md:30).  Returns false and sets `err`.
bool parse_forest(const void* buf, size_t size, int feature_length, ForestModel& out,
                  std::string& err);

// Serialises a forest in the reference's format (RandomForest::write, classifier.cpp:210-220;
// DecisionTree::write :144-152; writeBinary io.h:34-108): the trees of `m.raw`, node for node.
std::vector<uint8_t> serialize_forest(const ForestModel& m);

}  // namespace rvseg
