// Kernel-launch interface between the pipeline translation unit and the kernel files.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "rvseg_internal.h"

namespace rvseg {

// Geometry + feature layout of one camera configuration, passed to kernels by value.
struct FrameGeom {
    int W, H, stride;
    int lw, lh;                 // stride grid = low-resolution posterior image (W/stride x H/stride)
    float depth_min, depth_max; // metres (cloud NaN rule, feature_extractor.h:210)
    float dmin_mm, dmax_mm;     // float(d*1000.0) (mask rule, feature_extractor.h:43-44,60)
    int patch_size, r;          // patch_size, patch_size_reduce
    int n_patch;                // r*r*3 or 0 when the colour patch is disabled
    int pos_depth, pos_height, pos_normal; // index of each scalar feature in the vector or -1
    int D;                      // feature length
    float fill;                 // low-res image fill value
    int rt_rows;                // rows of the ResizeRow table (largest ROI half size + 1), 0 without colour patch
};

struct LabCoeffs { int c[9]; };

// cv::resize coefficient tables of the 8-bit patch path, one row per ROI half size
// (size = 2*half+1), host-computed with the formula the oracle states (OpenCV 2.4 imgwarp.cpp).
constexpr int RT_MAXR = 16;
struct ResizeRec { int16_t ofs, w0, w1, ofs1; };   // first tap, 11-bit weights of the two taps, second tap (both inside the ROI)
struct ResizeRow {
    ResizeRec x[RT_MAXR];   // x axis: weights clamped at the ROI border (OpenCV's xofs / alpha)
    ResizeRec y[RT_MAXR];   // y axis: weights kept, the two rows clipped to the ROI (OpenCV's yofs / beta + row clipping)
};

// float up-sampling tables (cv::resize INTER_LINEAR on CV_32FC(n), segmenter.cpp:380-382)
struct UpsampleTables {
    DevBuf xofs, ax0, ax1;  // W entries
    DevBuf yofs, ay0, ay1;  // H entries
};

// ---- kernels_features.hip --------------------------------------------------------------------
// d_lab2 (optional): the Lab image again with every pixel next to the one below it, {lab(y, x), lab(y + 1, x)} per
// pixel: a bilinear tap quad of the patch resize is then ONE 16-byte load (kernels_rf.hip, rf_frames_lazy_kernel)
void launch_prep(const FrameGeom& g, const LabTables& lab, const uint8_t* d_rgb, const uint16_t* d_depth,
                 const float* d_calibA, uint32_t* d_lab, float4* d_cloud, uint8_t* d_change, int n, hipStream_t s,
                 uint2* d_lab2 = nullptr);
void launch_window_map(const FrameGeom& g, const float4* d_cloud, uint8_t* d_change, uint8_t* d_rect, int n, hipStream_t s);
void launch_normal_feature(const FrameGeom& g, const float4* d_cloud, const uint8_t* d_rect, float* d_nfeat,
                           int n, hipStream_t s);

// ---- kernels_rf.hip --------------------------------------------------------------------------
// Fused per-point feature vector (LDS) + forest traversal over the stride grid of n frames.
//   d_low   : n x (lh*lw*S) low-resolution log-posteriors, layers concatenated per frame, each
//             [ly][lx][class]; invalid-depth cells receive g.fill
//   d_dump  : optional n x (lh*lw) x D materialised feature vectors (parity API only)
//   d_valid : optional n x (lh*lw) mask bytes
void launch_rf_frames(const FrameGeom& g, const DeviceForest& f, const ResizeRow* d_rt, const uint32_t* d_lab,
                      const uint16_t* d_depth, const float4* d_cloud, const float* d_nfeat, float* d_low,
                      float* d_dump, uint8_t* d_valid, int n, hipStream_t s, const uint2* d_lab2 = nullptr);
// true when the frame kernel can use the row-pair Lab image (8-byte nodes exist for this model)
bool rf_frames_wants_lab2(const DeviceForest& f);
// cv::resize to full resolution + pack [layer][y][x][class] (segmenter.cpp:380-431)
void launch_upsample_pack(const FrameGeom& g, const DeviceForest& f, const UpsampleTables& t,
                          const float* d_low, float* d_post, int n, hipStream_t s);
// label rules (rvseg_label_mode) over N points x C classes, class-contiguous
void launch_labels_frames(const float* d_values, int n_frames, int N, const DeviceForest& f, int mode, const int* unknown,
                          int8_t* d_labels, hipStream_t s);
void launch_labels(const float* d_values, size_t n_points, int C, int mode, int unknown, int8_t* d_labels,
                   hipStream_t s);

}  // namespace rvseg
