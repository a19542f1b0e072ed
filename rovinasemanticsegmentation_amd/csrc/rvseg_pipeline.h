// State of the frame / CRF pipelines (ctx->impl).
#pragma once
#include "rvseg_internal.h"
#include "rvseg_kernels.h"

namespace rvseg {

struct CrfState;     // rvseg_crf.hip
struct FusionState;  // rvseg_fusion.hip

// Host-buffer entry point (rvseg_segment_frames): two slots of pinned staging + device in/out
// buffers, so that the H2D copy of chunk k+1 and the D2H copy of chunk k-1 run under the compute of
// chunk k (three streams: in, compute, out).
struct HostStage {
    static constexpr int SLOTS = 2;
    void* h_rgb[SLOTS] = {}; void* h_depth[SLOTS] = {}; void* h_post[SLOTS] = {}; void* h_marg[SLOTS] = {}; void* h_lab[SLOTS] = {};
    size_t c_rgb[SLOTS] = {}, c_depth[SLOTS] = {}, c_post[SLOTS] = {}, c_marg[SLOTS] = {}, c_lab[SLOTS] = {};
    DevBuf d_rgb[SLOTS], d_depth[SLOTS], d_post[SLOTS], d_marg[SLOTS], d_lab[SLOTS];
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_in[SLOTS] = {}, ev_done[SLOTS] = {}, ev_out[SLOTS] = {};
    bool ready = false;
};

struct Pipeline {
    FrameGeom geom{};
    HostStage stage;
    DevBuf resize_rows;
    UpsampleTables up;
    // per-chunk device buffers (grow-only, sized for up to max_batch frames)
    DevBuf calibA, lab, lab2, cloud, change, rect, nfeat, low, post, marg, labels, in_rgb, in_depth, dump, valid;
    // pinned staging for the per-frame A = R*Kinv, t.  The device entry point returns without
    // synchronising, so a slot may only be rewritten once the copy that read it has run: a small ring,
    // each slot guarded by an event recorded behind its H2D copy
    static constexpr int CALIB_RING = 4;
    float* h_calibA[CALIB_RING] = {};
    size_t h_calibA_bytes[CALIB_RING] = {};
    hipEvent_t calib_ev[CALIB_RING] = {};
    bool calib_ev_live[CALIB_RING] = {};
    int calib_next = 0;
    // raised (by 3 = x8 slots) every time a lattice build overflows its hash table; applies to all later
    // builds of this context (include/rvseg.h, lattice_capacity_log2)
    int cap_boost = 0;
    CrfState* crf = nullptr;
    FusionState* fusion = nullptr;
    bool bare = false;  // created by a CRF entry point: frame tables not initialised yet
    // the lattice build depends only on the cloud and the colours, not on the forest: it runs on a
    // side stream beside feature extraction + forest evaluation (fork after prep, join before inference)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_entry = nullptr;   // ev_entry: everything before this chunk on the caller's stream
};

rvseg_status pipeline_init(rvseg_ctx* ctx);
// stages the per-frame A = R*Kinv, t of n calibrations (21 floats each) and enqueues their copy to im->calibA
rvseg_status upload_calib(rvseg_ctx* ctx, Pipeline* im, const float* calib, int n, hipStream_t s);
void timer_reset(rvseg_ctx* ctx);
void timer_mark(rvseg_ctx* ctx, const char* name, hipStream_t s);

// rvseg_fusion.hip
void fusion_state_free(Pipeline* im);
rvseg_status fusion_status(rvseg_ctx* ctx, Pipeline* im, bool wait);   // like crf_frames_status, for the index-range flag

// rvseg_crf.hip
void crf_state_free(Pipeline* im);
// DenseCRF on a cloud whose unaries / features live in HBM, for every label layer over ONE lattice:
// d_unaries = layers concatenated, each N x C_l accumulated log-posteriors (energy = -unary,
// src/segmenter.cpp:642); labels (optional) L x N; marginals stay in context memory
rvseg_status crf_cloud_layers(rvseg_ctx* ctx, int N, int n_layers, const int* class_counts, const float* d_unaries,
                              const float* d_features, float potts_w, int iterations, int label_mode, const int* unknown,
                              int8_t* d_labels, hipStream_t s);
// per-frame, per-layer DenseCRF on the frames of one chunk: unary = -(posteriors), features from
// the back-projected cloud and the colours
// part 1 (lattice + normaliser; needs the cloud only) and part 2 (mean field per layer + labels)
rvseg_status crf_frames_build_begin(rvseg_ctx* ctx, Pipeline* im, int n, hipStream_t s);
rvseg_status crf_frames_build(rvseg_ctx* ctx, Pipeline* im, int n, const uint8_t* d_rgb, hipStream_t s);
// Status of the last enqueued frame build (consumes it): RVSEG_OK, RVSEG_NOT_READY (only without `wait`)
// or RVSEG_ERR_CAPACITY after raising im->cap_boost.  RVSEG_OK when nothing is pending.
rvseg_status crf_frames_status(rvseg_ctx* ctx, Pipeline* im, bool wait);
rvseg_status crf_frames_infer(rvseg_ctx* ctx, Pipeline* im, int n, const float* d_post, float* d_marg, int8_t* d_labels,
                              hipStream_t s);

}  // namespace rvseg
