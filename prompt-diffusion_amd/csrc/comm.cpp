// The path's only exchange: ONE all-gather of the final latents, through an RCCL communicator the engine owns, so a
// host without torch can shard a batch over the GPUs of a node exactly as the reference shards its evaluation set
// (eval/distributed.py:25-27 sets up the process group, eval/evaluate_gen.py:55-57 takes every world_size-th batch).  One process per GPU.
// librccl is opened on first use (dlopen), so a single-GPU host never needs it.
#include "engine.h"
#include <cstring>
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*all_gather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*error_string)(ncclResult_t) = nullptr;
};

Rccl* rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r.lib ? &r : nullptr;
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) { pd_set_error("pd_comm: librccl not found (%s)", dlerror()); return nullptr; }
    r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(dlsym(r.lib, "ncclGetUniqueId"));
    r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(dlsym(r.lib, "ncclCommInitRank"));
    r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.all_gather = reinterpret_cast<decltype(r.all_gather)>(dlsym(r.lib, "ncclAllGather"));
    r.error_string = reinterpret_cast<decltype(r.error_string)>(dlsym(r.lib, "ncclGetErrorString"));
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_gather || !r.error_string) {
        pd_set_error("pd_comm: librccl lacks an expected entry point");
        dlclose(r.lib);
        r.lib = nullptr;
        return nullptr;
    }
    return &r;
}

#define RCCL_OK(R, expr)                                                                   \
    do {                                                                                   \
        const ncclResult_t res_ = (expr);                                                  \
        if (res_ != ncclSuccess) { pd_set_error("%s: %s", #expr, (R)->error_string(res_)); return 1; } \
    } while (0)

static_assert(PD_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "pd_comm id size");

}  // namespace

extern "C" {

int pd_comm_new_id(uint8_t id[PD_COMM_ID_BYTES]) {
    if (!id) { pd_set_error("null argument"); return 1; }
    Rccl* r = rccl();
    if (!r) return 1;
    ncclUniqueId u;
    RCCL_OK(r, r->get_unique_id(&u));
    memcpy(id, u.internal, PD_COMM_ID_BYTES);
    return 0;
}

int pd_comm_init(pd_engine* e, const uint8_t id[PD_COMM_ID_BYTES], int32_t world, int32_t rank) {
    if (!e || !id) { pd_set_error("null argument"); return 1; }
    if (world < 1 || rank < 0 || rank >= world) { pd_set_error("pd_comm_init: rank %d of %d", rank, world); return 1; }
    if (e->comm) { pd_set_error("pd_comm_init: the engine already owns a communicator"); return 1; }
    Rccl* r = rccl();
    if (!r) return 1;
    HIP_OK(hipSetDevice(e->device));
    ncclUniqueId u;
    memcpy(u.internal, id, PD_COMM_ID_BYTES);
    ncclComm_t c = nullptr;
    RCCL_OK(r, r->comm_init_rank(&c, world, u, rank));
    e->comm = c;
    e->comm_world = world;
    e->comm_rank = rank;
    return 0;
}

int pd_comm_world(pd_engine* e, int32_t* world, int32_t* rank) {
    if (!e) { pd_set_error("null engine"); return 1; }
    if (world) *world = e->comm ? e->comm_world : 1;
    if (rank) *rank = e->comm ? e->comm_rank : 0;
    return 0;
}

// recv[world][count] <- every rank's send[count], rank order; on the engine's stream, synchronised before returning like every
// call that fills a caller buffer.  Without a communicator (one GPU) this is a copy.
int pd_comm_all_gather(pd_engine* e, const float* send, float* recv, int64_t count, int32_t mem) {
    if (!e || !send || !recv) { pd_set_error("null argument"); return 1; }
    if (count < 0 || (mem != PD_MEM_HOST && mem != PD_MEM_DEVICE)) { pd_set_error("pd_comm_all_gather: bad count or mem"); return 1; }
    HIP_OK(hipSetDevice(e->device));
    const size_t bytes = (size_t)count * sizeof(float);
    if (!count) return 0;
    if (!e->comm) {
        if (mem == PD_MEM_HOST) { memcpy(recv, send, bytes); return 0; }
        HIP_OK(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, e->stream));
        HIP_OK(hipStreamSynchronize(e->stream));
        return 0;
    }
    Rccl* r = rccl();
    if (!r) return 1;
    const ncclComm_t c = static_cast<ncclComm_t>(e->comm);
    if (mem == PD_MEM_DEVICE) {
        RCCL_OK(r, r->all_gather(send, recv, (size_t)count, ncclFloat32, c, e->stream));
        HIP_OK(hipStreamSynchronize(e->stream));
        return 0;
    }
    float *ds = nullptr, *dr = nullptr;   // host buffers: staged (4 MB at the headline batch; once per sampling run)
    HIP_OK(hipMalloc(&ds, bytes));
    if (hipMalloc(&dr, bytes * e->comm_world) != hipSuccess) { (void)hipFree(ds); pd_set_error("pd_comm_all_gather: out of device memory"); return 1; }
    int rc = 1;
    do {
        if (hipMemcpyAsync(ds, send, bytes, hipMemcpyHostToDevice, e->stream) != hipSuccess) { pd_set_error("pd_comm_all_gather: copy in failed"); break; }
        const ncclResult_t res = r->all_gather(ds, dr, (size_t)count, ncclFloat32, c, e->stream);
        if (res != ncclSuccess) { pd_set_error("ncclAllGather: %s", r->error_string(res)); break; }
        if (hipMemcpyAsync(recv, dr, bytes * e->comm_world, hipMemcpyDeviceToHost, e->stream) != hipSuccess) { pd_set_error("pd_comm_all_gather: copy out failed"); break; }
        if (hipStreamSynchronize(e->stream) != hipSuccess) { pd_set_error("pd_comm_all_gather: %s", hipGetErrorString(hipGetLastError())); break; }
        rc = 0;
    } while (0);
    (void)hipFree(ds);
    (void)hipFree(dr);
    return rc;
}

int pd_comm_destroy(pd_engine* e) {
    if (!e) { pd_set_error("null engine"); return 1; }
    if (!e->comm) return 0;
    Rccl* r = rccl();
    if (!r) return 1;
    HIP_OK(hipSetDevice(e->device));
    (void)hipStreamSynchronize(e->stream);
    const ncclComm_t c = static_cast<ncclComm_t>(e->comm);
    e->comm = nullptr;
    RCCL_OK(r, r->comm_destroy(c));
    return 0;
}

}  // extern "C"
