// Internal declarations shared by the C-ABI translation unit and the HIP kernel files.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rvseg.h"
#include "forest_model.h"

namespace rvseg {

// ---------------------------------------------------------------------------------------------
// device buffer with explicit ownership
// ---------------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// ---------------------------------------------------------------------------------------------
// device-side forest (breadth-first node array in HBM, leaf histogram table)
// ---------------------------------------------------------------------------------------------
struct DeviceForest {
    int n_trees = 0;
    int max_depth = 0;
    int n_nodes = 0;
    int n_leaves = 0;
    int n_layers = 0;                       // layers of the ACTIVE mode
    int class_counts[RVSEG_MAX_LAYERS] = {}; // per layer
    int sum_classes = 0;                    // S
    DevBuf nodes;                           // DeviceNode[n_nodes]
    DevBuf nodes8;                          // the same nodes in 8 bytes each (frame kernel), or empty: see upload_forest
    DevBuf roots;                           // int32[n_trees]
    DevBuf hist;                            // float[n_leaves * S] of the active mode
};

// Lab LUTs (gamma, cube root, 3x3 fixed-point matrix), uploaded once
struct LabTables {
    DevBuf gamma;   // uint16[256]
    DevBuf cbrt;    // uint16[3072]
    int coeffs[9];
};

struct StageTimer {
    std::vector<std::string> names;  // names[i] = stage that starts at events[i]
    std::vector<hipEvent_t> events;  // pool; events[i] .. events[i+1] bracket stage i
    size_t used = 0;
    std::vector<float> ms;
    // one stage may run on a side stream, overlapped with the stages above (the lattice build of the
    // frame path): its own pair of events, reported under side_name
    hipEvent_t side0 = nullptr, side1 = nullptr;
    std::string side_name;
    bool side_used = false;
};

}  // namespace rvseg

struct rvseg_ctx {
    rvseg_params params{};
    rvseg_schedule sched{};        // rvseg_set_schedule; defaults from rvseg_schedule_default
    int feature_length = 0;
    std::string err;
    rvseg::ForestModel host_forest;
    bool forest_loaded = false;
    rvseg::DeviceForest forest;
    rvseg::LabTables lab;
    hipStream_t stream = nullptr;  // ctx-owned stream for the host entry points
    // workspace: grows on demand, owned by the ctx
    std::vector<rvseg::DevBuf> pool;
    rvseg::StageTimer timer;
    struct Impl;
    Impl* impl = nullptr;  // frame / crf pipeline state (rvseg_pipeline.hip)
    std::vector<uint8_t> trained_model;   // forest.dat image of the last rvseg_forest_train* call (rvseg_forest_train_result)
    void* comm = nullptr;  // RCCL communicator of the local-map gather (rvseg_comm.cpp), or null
    int comm_rank = 0, comm_world = 0;
};

namespace rvseg {

// Error plumbing: every HIP call goes through this; failures land in ctx->err.
bool hip_ok(rvseg_ctx* ctx, hipError_t e, const char* what);
#define RV_HIP(ctx, call)                                           \
    do {                                                            \
        if (!::rvseg::hip_ok((ctx), (call), #call)) return RVSEG_ERR_HIP; \
    } while (0)

// Kernel launches happen inside void launch_* helpers.  Every launch group ends with RV_LAUNCHED(name): a launch the
// runtime refused (bad grid, too much LDS, ...) is parked per thread with the kernel's name -- the first one wins -- and
// the orchestration turns it into RVSEG_ERR_HIP with launch_error_take(), so no launch fails silently.
void launch_check(const char* what);
rvseg_status launch_error_take(rvseg_ctx* ctx);
#define RV_LAUNCHED(name) ::rvseg::launch_check(name)
#define RV_LAUNCH_OK(ctx)                                               \
    do {                                                                \
        const rvseg_status st__ = ::rvseg::launch_error_take(ctx);      \
        if (st__ != RVSEG_OK) return st__;                              \
    } while (0)

rvseg_status dev_alloc(rvseg_ctx* ctx, DevBuf& b, size_t bytes);
void dev_free(DevBuf& b);
// grow-only allocation: reallocates when the buffer is too small
rvseg_status dev_reserve(rvseg_ctx* ctx, DevBuf& b, size_t bytes);

// ---- kernels_rf.hip --------------------------------------------------------------------------
// P points with materialised D-dimensional features -> P x S log-posteriors
void launch_forest_eval(const DeviceForest& f, const float* d_X, int P, int D, float* d_out,
                        hipStream_t s);

}  // namespace rvseg
