// State of the frame / CRF pipelines (ctx->impl).
#pragma once
#include "rvseg_internal.h"
#include "rvseg_kernels.h"

namespace rvseg {

struct CrfState;  // rvseg_crf.hip

struct Pipeline {
    FrameGeom geom{};
    DevBuf resize_rows;
    UpsampleTables up;
    // per-chunk device buffers (grow-only, sized for up to max_batch frames)
    DevBuf calibA, lab, cloud, change, rect, nfeat, low, post, marg, labels, in_rgb, in_depth, dump, valid;
    float* h_calibA = nullptr;  // pinned staging for the per-frame A = R*Kinv, t
    size_t h_calibA_bytes = 0;
    CrfState* crf = nullptr;
    bool bare = false;  // created by a CRF entry point: frame tables not initialised yet
    // the lattice build depends only on the cloud and the colours, not on the forest: it runs on a
    // side stream beside feature extraction + forest evaluation (fork after prep, join before inference)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};

rvseg_status pipeline_init(rvseg_ctx* ctx);
void timer_reset(rvseg_ctx* ctx);
void timer_mark(rvseg_ctx* ctx, const char* name, hipStream_t s);

// rvseg_crf.hip
void crf_state_free(Pipeline* im);
// per-frame, per-layer DenseCRF on the frames of one chunk: unary = -(posteriors), features from
// the back-projected cloud and the colours
// part 1 (lattice + normaliser; needs the cloud only) and part 2 (mean field per layer + labels)
rvseg_status crf_frames_build(rvseg_ctx* ctx, Pipeline* im, int n, const uint8_t* d_rgb, hipStream_t s);
rvseg_status crf_frames_infer(rvseg_ctx* ctx, Pipeline* im, int n, const float* d_post, float* d_marg, int8_t* d_labels,
                              hipStream_t s);

}  // namespace rvseg
