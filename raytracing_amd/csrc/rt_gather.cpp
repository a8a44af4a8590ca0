// rt_gather.cpp -- the one exchange step of a frame rendered by several GPUs: every rank sends the
// compact slab of tiles it rendered to the root over RCCL (point-to-point sends, one per peer, so on
// an xGMI node each slab travels its own link into the root), and the root scatters the tiles into
// the frame. The reference renders on one host (src/main.rs:313-379) and has no counterpart.
//
// RCCL is bound at run time (dlopen), so that single-GPU users of the library do not load it. A copy
// already present in the process (for instance the one a host framework brought along) is used in
// preference to loading another.

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>

#include "rt_api_internal.h"
#include "rt_kernels.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

std::mutex g_rccl_mu;
Rccl g_rccl;
bool g_rccl_tried = false;
std::string g_rccl_why;   // why the first attempt to bind RCCL failed

template <class F>
bool bind(void* h, const char* name, F* out) {
    *out = reinterpret_cast<F>(dlsym(h, name));
    return *out != nullptr;
}

// The process-wide RCCL binding, or null with the reason in rt_last_error().
const Rccl* rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.handle) return &g_rccl;
    if (g_rccl_tried) {
        rt_fail(RT_ERR_UNSUPPORTED, "RCCL is not available in this process: %s", g_rccl_why.c_str());   // the first attempt's reason
        return nullptr;
    }
    g_rccl_tried = true;
    void* h = nullptr;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);   // a copy the process already holds
    for (const char* n : names)
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        const char* why = dlerror();
        g_rccl_why = std::string("cannot load librccl.so: ") + (why ? why : "unknown reason");
        rt_fail(RT_ERR_UNSUPPORTED, "%s", g_rccl_why.c_str());
        return nullptr;
    }
    Rccl r;
    r.handle = h;
    const bool ok = bind(h, "ncclGetUniqueId", &r.GetUniqueId) && bind(h, "ncclCommInitRank", &r.CommInitRank) &&
                    bind(h, "ncclCommDestroy", &r.CommDestroy) && bind(h, "ncclCommCount", &r.CommCount) &&
                    bind(h, "ncclCommUserRank", &r.CommUserRank) && bind(h, "ncclGroupStart", &r.GroupStart) &&
                    bind(h, "ncclGroupEnd", &r.GroupEnd) && bind(h, "ncclSend", &r.Send) && bind(h, "ncclRecv", &r.Recv) &&
                    bind(h, "ncclGetErrorString", &r.GetErrorString);
    if (!ok) {
        dlclose(h);
        g_rccl_why = "librccl.so lacks a symbol this library needs";
        rt_fail(RT_ERR_UNSUPPORTED, "%s", g_rccl_why.c_str());
        return nullptr;
    }
    g_rccl = r;
    return &g_rccl;
}

#define RCCL_TRY(R, expr)                                                                 \
    do {                                                                                  \
        ncclResult_t r_ = (expr);                                                         \
        if (r_ != ncclSuccess) return rt_fail(RT_ERR_DEVICE, "%s: %s", #expr, (R)->GetErrorString(r_)); \
    } while (0)

}  // namespace

extern "C" {

static_assert(sizeof(ncclUniqueId) == RT_COMM_ID_BYTES, "rt_comm_id must hold an ncclUniqueId");

int rt_comm_id_create(rt_comm_id* out) {
    if (!out) return rt_fail(RT_ERR_INVALID, "out is null");
    const Rccl* R = rccl();
    if (!R) return RT_ERR_UNSUPPORTED;
    ncclUniqueId id;
    RCCL_TRY(R, R->GetUniqueId(&id));
    std::memcpy(out->bytes, &id, sizeof(id));
    return RT_OK;
}

int rt_comm_create(rt_ctx* ctx, const rt_comm_id* id, int rank, int n_ranks, void** nccl_comm) {
    if (!ctx || !id || !nccl_comm) return rt_fail(RT_ERR_INVALID, "null argument");
    *nccl_comm = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return rt_fail(RT_ERR_INVALID, "rank %d out of range for %d ranks", rank, n_ranks);
    const Rccl* R = rccl();
    if (!R) return RT_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id->bytes, sizeof(uid));
    ncclComm_t comm = nullptr;
    RCCL_TRY(R, R->CommInitRank(&comm, n_ranks, uid, rank));
    *nccl_comm = comm;
    return RT_OK;
}

int rt_comm_destroy(void* nccl_comm) {
    if (!nccl_comm) return RT_OK;
    const Rccl* R = rccl();
    if (!R) return RT_ERR_UNSUPPORTED;
    RCCL_TRY(R, R->CommDestroy(static_cast<ncclComm_t>(nccl_comm)));
    return RT_OK;
}

int rt_comm_ranks(void* nccl_comm, int* rank, int* n_ranks) {
    if (!nccl_comm) return rt_fail(RT_ERR_INVALID, "communicator is null");
    const Rccl* R = rccl();
    if (!R) return RT_ERR_UNSUPPORTED;
    if (rank) RCCL_TRY(R, R->CommUserRank(static_cast<ncclComm_t>(nccl_comm), rank));
    if (n_ranks) RCCL_TRY(R, R->CommCount(static_cast<ncclComm_t>(nccl_comm), n_ranks));
    return RT_OK;
}

int rt_gather_tiles_device(rt_ctx* ctx, void* nccl_comm, const uint8_t* d_slab, uint32_t width, uint32_t height,
                           uint8_t* d_rgba8, void* stream) {
    if (!ctx || !nccl_comm || !d_slab) return rt_fail(RT_ERR_INVALID, "null argument");
    if (width == 0 || height == 0) return rt_fail(RT_ERR_INVALID, "frame size is zero");
    if (int g = rt_runtime_guard()) return g;
    const Rccl* R = rccl();
    if (!R) return RT_ERR_UNSUPPORTED;
    ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);
    int rank = 0, n = 0;
    RCCL_TRY(R, R->CommUserRank(comm, &rank));
    RCCL_TRY(R, R->CommCount(comm, &n));
    if (n < 1 || n > 4096) return rt_fail(RT_ERR_UNSUPPORTED, "communicator of %d ranks", n);
    if (rank == 0 && !d_rgba8) return rt_fail(RT_ERR_INVALID, "rank 0 needs the frame buffer");
    // Equal slabs: the largest shard's tile count (shards differ by a few tiles at most).
    uint32_t slab_tiles = 0;
    for (int r = 0; r < n; ++r) slab_tiles = std::max(slab_tiles, rt_shard_tiles(width, height, (uint32_t)r, (uint32_t)n));
    const size_t tile_bytes = (size_t)RT_TILE * RT_TILE * 4;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    if (rank != 0) {
        // This rank's own tiles only: the root knows how many each shard has.
        const size_t mine = (size_t)rt_shard_tiles(width, height, (uint32_t)rank, (uint32_t)n) * tile_bytes;
        if (mine) RCCL_TRY(R, R->Send(d_slab, mine, ncclUint8, 0, comm, s));
        return RT_OK;
    }
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        HIP_TRY(hipSetDevice(ctx->device));
        const size_t need = (size_t)n * slab_tiles * tile_bytes;
        if (ctx->gather_bytes < need) {
            // Work enqueued earlier may still read the old slab.
            HIP_TRY(hipStreamSynchronize(s));
            if (ctx->d_gather) (void)hipFree(ctx->d_gather);
            ctx->d_gather = nullptr;
            ctx->gather_bytes = 0;
            HIP_TRY(hipMalloc((void**)&ctx->d_gather, need));
            ctx->gather_bytes = need;
        }
    }
    uint8_t* base = ctx->d_gather;
    const size_t slab_bytes = (size_t)slab_tiles * tile_bytes;
    const size_t own = (size_t)rt_shard_tiles(width, height, 0u, (uint32_t)n) * tile_bytes;
    if (own) HIP_TRY(hipMemcpyAsync(base, d_slab, own, hipMemcpyDeviceToDevice, s));
    if (n > 1) {
        RCCL_TRY(R, R->GroupStart());
        for (int r = 1; r < n; ++r) {
            const size_t theirs = (size_t)rt_shard_tiles(width, height, (uint32_t)r, (uint32_t)n) * tile_bytes;
            if (!theirs) continue;
            const ncclResult_t rr = R->Recv(base + (size_t)r * slab_bytes, theirs, ncclUint8, r, comm, s);
            if (rr != ncclSuccess) {
                (void)R->GroupEnd();
                return rt_fail(RT_ERR_DEVICE, "ncclRecv from rank %d: %s", r, R->GetErrorString(rr));
            }
        }
        RCCL_TRY(R, R->GroupEnd());
    }
    return rt_deinterleave_tiles_device(ctx, base, slab_tiles, width, height, (uint32_t)n, d_rgba8, stream);
}

}  // extern "C"
