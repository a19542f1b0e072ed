// Multi-GPU local-map label gather over RCCL: one process (or thread) per GPU, frames
// sharded across ranks with no data-path collective; the only exchange is the fixed-size label (or
// posterior) block every rank sends to the fusion rank.  xGMI is point to point, so a gather is one
// direct transfer per peer over that peer's own link -- no ring.
//
// librccl.so is opened on first use (dlopen): a single-GPU process never loads it.
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "rvseg_internal.h"

namespace {

// the few RCCL entry points used here, resolved at run time (signatures of rccl.h, ROCm 7.2)
struct Rccl {
    typedef struct { char internal[128]; } UniqueId;
    typedef void* Comm;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*Gather)(const void*, void*, size_t, int /*ncclDataType_t*/, int, Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    void* handle = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) return;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
        r.Gather = reinterpret_cast<decltype(r.Gather)>(dlsym(r.handle, "ncclGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.Gather;
    });
    return r;
}

rvseg_status rccl_fail(rvseg_ctx* ctx, const char* what, int rc) {
    const Rccl& r = rccl();
    if (ctx) ctx->err = std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error");
    return RVSEG_ERR_HIP;
}

}  // namespace

extern "C" {

rvseg_status rvseg_comm_unique_id(uint8_t id_out[RVSEG_COMM_ID_BYTES]) {
    if (!id_out) return RVSEG_ERR_INVALID_ARG;
    Rccl& r = rccl();
    if (!r.ok) return RVSEG_ERR_IO;   // librccl.so could not be opened
    Rccl::UniqueId id;
    if (r.GetUniqueId(&id) != 0) return RVSEG_ERR_HIP;
    static_assert(sizeof(id) == RVSEG_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    std::memcpy(id_out, &id, sizeof(id));
    return RVSEG_OK;
}

rvseg_status rvseg_comm_init(rvseg_ctx* ctx, int32_t rank, int32_t world, const uint8_t id[RVSEG_COMM_ID_BYTES]) {
    if (!ctx) return RVSEG_ERR_INVALID_ARG;
    if (!id || world < 1 || rank < 0 || rank >= world) { ctx->err = "bad arguments"; return RVSEG_ERR_INVALID_ARG; }
    if (ctx->comm) { ctx->err = "a communicator exists already on this context"; return RVSEG_ERR_INVALID_ARG; }
    Rccl& r = rccl();
    if (!r.ok) { ctx->err = "librccl.so could not be opened"; return RVSEG_ERR_IO; }
    RV_HIP(ctx, hipSetDevice(ctx->params.device));
    Rccl::UniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    Rccl::Comm comm = nullptr;
    const int rc = r.CommInitRank(&comm, world, uid, rank);
    if (rc != 0) return rccl_fail(ctx, "ncclCommInitRank", rc);
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return RVSEG_OK;
}

void rvseg_comm_destroy(rvseg_ctx* ctx) {
    if (!ctx || !ctx->comm) return;
    Rccl& r = rccl();
    if (r.ok) (void)r.CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_world = 0;
}

rvseg_status rvseg_gather_frames(rvseg_ctx* ctx, const void* d_local, size_t bytes_per_rank, void* d_recv, int32_t root, void* hip_stream) {
    if (!ctx) return RVSEG_ERR_INVALID_ARG;
    if (!ctx->comm) { ctx->err = "no communicator: call rvseg_comm_init first"; return RVSEG_ERR_INVALID_ARG; }
    if (root < 0 || root >= ctx->comm_world || (bytes_per_rank > 0 && !d_local) ||
        (ctx->comm_rank == root && bytes_per_rank > 0 && !d_recv)) { ctx->err = "bad arguments"; return RVSEG_ERR_INVALID_ARG; }
    if (bytes_per_rank == 0) return RVSEG_OK;
    Rccl& r = rccl();
    RV_HIP(ctx, hipSetDevice(ctx->params.device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
    const int rc = r.Gather(d_local, d_recv, bytes_per_rank, /*ncclInt8*/ 0, root, ctx->comm, s);
    if (rc != 0) return rccl_fail(ctx, "ncclGather", rc);
    return RVSEG_OK;
}

}  // extern "C"
