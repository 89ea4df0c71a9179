// The two exchanges of the multi-GPU hot path as plain C entry points over RCCL, for a caller that shards
// without torch: one process per GPU, one RCCL communicator per handle.
//   scd_allreduce_centroids   the per-Lloyd-iteration exchange: packed float64 [sums | counts | inertia] summed in place
//   scd_allgather_text        the vocabulary build: every rank contributes its fp16 shard of the name-major classifier
// The reference has no multi-GPU path (single process, main_unsup.py); scd_amd/kmeans.py and pipeline.py use the same
// pattern through torch.distributed (whose "nccl" backend is this library).  librccl is resolved with dlopen at
// scd_comm_init, so libscd_hip.so has no link-time dependency on it and single-GPU users never load it.
#include "common.h"
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>
#include <map>
#include <mutex>

namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_mu;
std::map<const scd_ctx*, ncclComm_t> g_comm;      // one communicator per handle
std::map<const scd_ctx*, int> g_world;

int load_rccl() {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_rccl.lib) return SCD_OK;
    void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
        scd_set_error("scd_comm: cannot load librccl.so.1: %s", dlerror());
        return SCD_ERCCL;
    }
#define SYM(field, name)                                                   \
    *(void**)(&g_rccl.field) = dlsym(lib, name);                           \
    if (!g_rccl.field) {                                                   \
        scd_set_error("scd_comm: librccl has no symbol %s", name);         \
        return SCD_ERCCL;                                                  \
    }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllReduce, "ncclAllReduce")
    SYM(AllGather, "ncclAllGather")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_rccl.lib = lib;
    return SCD_OK;
}

ncclComm_t comm_of(scd_handle h) {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_comm.find(h);
    return it == g_comm.end() ? nullptr : it->second;
}
}  // namespace

#define SCD_RCCL(expr)                                                                             \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess) {                                                                   \
            scd_set_error("%s failed: %s", #expr, g_rccl.GetErrorString(r_));                      \
            return SCD_ERCCL;                                                                      \
        }                                                                                          \
    } while (0)

extern "C" size_t scd_comm_unique_id_bytes(void) { return sizeof(ncclUniqueId); }

extern "C" int scd_comm_unique_id(void* id_out) {
    SCD_REQUIRE(id_out, "scd_comm_unique_id: null output");
    { const int rc = load_rccl(); if (rc) return rc; }
    SCD_RCCL(g_rccl.GetUniqueId((ncclUniqueId*)id_out));
    return SCD_OK;
}

extern "C" int scd_comm_init(scd_handle h, int rank, int world, const void* unique_id) {
    SCD_REQUIRE(h && unique_id && world >= 1 && rank >= 0 && rank < world, "scd_comm_init: bad arguments");
    { const int rc = scd_check_device(h, "scd_comm_init"); if (rc) return rc; }
    { const int rc = load_rccl(); if (rc) return rc; }
    SCD_REQUIRE(!comm_of(h), "scd_comm_init: the handle already has a communicator");
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t c = nullptr;
    SCD_RCCL(g_rccl.CommInitRank(&c, world, id, rank));
    std::lock_guard<std::mutex> lock(g_mu);
    g_comm[h] = c;
    g_world[h] = world;
    return SCD_OK;
}

extern "C" int scd_comm_destroy(scd_handle h) {
    ncclComm_t c = comm_of(h);
    if (!c) return SCD_OK;
    SCD_RCCL(g_rccl.CommDestroy(c));
    std::lock_guard<std::mutex> lock(g_mu);
    g_comm.erase(h);
    g_world.erase(h);
    return SCD_OK;
}

extern "C" int scd_allreduce_centroids(scd_handle h, double* packed, int64_t count, void* stream) {
    SCD_DEVICE_ENTRY(h, "scd_allreduce_centroids");
    SCD_REQUIRE(h && packed && count > 0, "scd_allreduce_centroids: bad arguments");
    ncclComm_t c = comm_of(h);
    SCD_REQUIRE(c, "scd_allreduce_centroids: scd_comm_init has not been called on this handle");
    SCD_RCCL(g_rccl.AllReduce(packed, packed, (size_t)count, ncclFloat64, ncclSum, c, (hipStream_t)stream));
    return SCD_OK;
}

extern "C" int scd_allgather_text(scd_handle h, const void* w_shard, int64_t shard_elems, void* w_full, void* stream) {
    SCD_DEVICE_ENTRY(h, "scd_allgather_text");
    SCD_REQUIRE(h && w_shard && w_full && shard_elems > 0, "scd_allgather_text: bad arguments");
    ncclComm_t c = comm_of(h);
    SCD_REQUIRE(c, "scd_allgather_text: scd_comm_init has not been called on this handle");
    SCD_RCCL(g_rccl.AllGather(w_shard, w_full, (size_t)shard_elems, ncclFloat16, c, (hipStream_t)stream));
    return SCD_OK;
}
