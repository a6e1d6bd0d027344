// rsk_comm.hip -- the path's one collective at the C-ABI: the all-gather of the ranks' hit records over RCCL / xGMI
// (north_star: "candidate pairs shard across the GPUs of one node with an RCCL gather of hit buffers").
// The reference has no collective -- it deals pairs to threads through one locked counter (runself.cpp:72-99) and every
// thread writes its hits to one file under a lock (dbsearcher.cpp:98-106); one process per GPU needs the exchange instead.
// A multi-process C++ caller (one DBSearcher per rank) uses this without Python: rank 0 makes the id, hands its 128 bytes to
// the other ranks by whatever it has (a file, MPI, an environment variable), every rank creates the communicator, and
// rsk_gather_hits concatenates the ranks' device buffers in rank order on every rank, device to device.
// librccl.so is opened at run time (dlopen): librsk.so itself keeps depending on libamdhip64 only, and a box without RCCL
// runs everything but these three calls.
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <vector>

#include "rsk_internal.h"

struct rccl_id { char internal[RSK_COMM_ID_BYTES]; };      // ncclUniqueId, passed by value

namespace {
// the few RCCL entry points used, with RCCL's own signatures (rccl.h:187-933; ncclResult_t / ncclDataType_t are ints)
struct rccl_api {
    void *lib = nullptr;
    int (*GetUniqueId)(void *id) = nullptr;
    int (*CommInitRank)(void **comm, int nranks, rccl_id id, int rank) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*AllGather)(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t s) = nullptr;
    int (*Broadcast)(const void *send, void *recv, size_t count, int dtype, int root, void *comm, hipStream_t s) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
rccl_api g_rccl;
std::mutex g_rccl_mutex;

int rccl_load()
{
    std::lock_guard<std::mutex> g(g_rccl_mutex);
    if (g_rccl.lib) return RSK_OK;
    void *h = nullptr;
    for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) { rsk_set_error("rsk_comm: librccl.so not found (%s)", dlerror()); return RSK_E_INVALID; }
    rccl_api a;
    a.lib = h;
    a.GetUniqueId = (int (*)(void *)) dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (int (*)(void **, int, rccl_id, int)) dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (int (*)(void *)) dlsym(h, "ncclCommDestroy");
    a.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t)) dlsym(h, "ncclAllGather");
    a.Broadcast = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t)) dlsym(h, "ncclBroadcast");
    a.GroupStart = (int (*)()) dlsym(h, "ncclGroupStart");
    a.GroupEnd = (int (*)()) dlsym(h, "ncclGroupEnd");
    a.GetErrorString = (const char *(*)(int)) dlsym(h, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.Broadcast || !a.GroupStart || !a.GroupEnd) {
        rsk_set_error("rsk_comm: librccl.so lacks an entry point");
        dlclose(h);
        return RSK_E_INVALID;
    }
    g_rccl = a;
    return RSK_OK;
}

int rccl_fail(int r, const char *what)
{
    rsk_set_error("rsk_comm: %s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
    return RSK_E_DEVICE;
}
#define RSK_RCCL(call, what) do { const int r_ = (call); if (r_ != 0) return rccl_fail(r_, what); } while (0)
enum { RCCL_UINT8 = 1, RCCL_UINT64 = 5 };      // ncclUint8, ncclUint64 (rccl.h ncclDataType_t)
}   // namespace

struct rsk_comm {
    rsk_ctx *ctx = nullptr;
    void *comm = nullptr;
    int rank = 0, world = 1;
    uint64_t *d_counts = nullptr;          // [world + 1]: this rank's count at [world], everyone's at [0, world)
    void *d_all = nullptr;                 // the gathered records (grow-only)
    size_t all_bytes = 0;
};

extern "C" int rsk_comm_unique_id(unsigned char *id)
{
    if (!id) { rsk_set_error("rsk_comm_unique_id: NULL argument"); return RSK_E_INVALID; }
    int rc = rccl_load();
    if (rc != RSK_OK) return rc;
    rccl_id u;
    RSK_RCCL(g_rccl.GetUniqueId(&u), "ncclGetUniqueId");
    memcpy(id, u.internal, RSK_COMM_ID_BYTES);
    return RSK_OK;
}

extern "C" int rsk_comm_create(rsk_ctx *ctx, const unsigned char *id, int rank, int world, rsk_comm **out)
{
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) { rsk_set_error("rsk_comm_create: invalid argument"); return RSK_E_INVALID; }
    int rc = rccl_load();
    if (rc != RSK_OK) return rc;
    RSK_HIP(hipSetDevice(ctx->device));
    rsk_comm *c = new rsk_comm;
    c->ctx = ctx; c->rank = rank; c->world = world;
    rccl_id u;
    memcpy(u.internal, id, RSK_COMM_ID_BYTES);
    const int r = g_rccl.CommInitRank(&c->comm, world, u, rank);
    if (r != 0) { delete c; return rccl_fail(r, "ncclCommInitRank"); }
    if (hipMalloc((void **) &c->d_counts, (size_t) (world + 1) * 8) != hipSuccess) {
        g_rccl.CommDestroy(c->comm);
        delete c;
        rsk_set_error("rsk_comm_create: out of device memory");
        return RSK_E_NOMEM;
    }
    *out = c;
    return RSK_OK;
}

extern "C" void rsk_comm_destroy(rsk_comm *c)
{
    if (!c) return;
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    if (c->d_counts) (void) hipFree(c->d_counts);
    if (c->d_all) (void) hipFree(c->d_all);
    delete c;
}

extern "C" int rsk_comm_rank(const rsk_comm *c) { return c ? c->rank : -1; }
extern "C" int rsk_comm_world(const rsk_comm *c) { return c ? c->world : 0; }

// counts by one ncclAllGather, then one ncclBroadcast per rank inside a group (each rank's records land at its offset of
// the result: no padding to the largest count, nothing through the host but the world counts)
extern "C" int rsk_gather_hits(rsk_comm *c, const void *d_local, uint64_t n_local, uint32_t rec_bytes, void **d_all, uint64_t *n_all,
                               uint64_t *counts)
{
    if (!c || !d_all || !n_all || rec_bytes == 0 || (n_local && !d_local)) { rsk_set_error("rsk_gather_hits: invalid argument"); return RSK_E_INVALID; }
    rsk_ctx *ctx = c->ctx;
    RSK_HIP(hipSetDevice(ctx->device));
    RSK_HIP(hipMemcpyAsync(c->d_counts + c->world, &n_local, 8, hipMemcpyHostToDevice, ctx->stream));
    RSK_RCCL(g_rccl.AllGather(c->d_counts + c->world, c->d_counts, 1, RCCL_UINT64, c->comm, ctx->stream), "ncclAllGather (counts)");
    std::vector<uint64_t> h((size_t) c->world);
    RSK_HIP(hipMemcpyAsync(h.data(), c->d_counts, (size_t) c->world * 8, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipStreamSynchronize(ctx->stream));
    uint64_t total = 0;
    for (uint64_t x : h) total += x;
    const size_t need = (size_t) std::max<uint64_t>(total, 1) * rec_bytes;
    if (need > c->all_bytes) {
        if (c->d_all) (void) hipFree(c->d_all);
        c->d_all = nullptr; c->all_bytes = 0;
        const int rc = rsk_dev_malloc(ctx, &c->d_all, need);
        if (rc != RSK_OK) return rc;
        c->all_bytes = need;
    }
    RSK_RCCL(g_rccl.GroupStart(), "ncclGroupStart");
    uint64_t off = 0;
    for (int r = 0; r < c->world; ++r) {
        if (h[(size_t) r]) {
            char *dst = (char *) c->d_all + off * rec_bytes;            // (a rank that is not the root passes its receive buffer as both)
            const int rr = g_rccl.Broadcast(r == c->rank ? d_local : (const void *) dst, dst, (size_t) h[(size_t) r] * rec_bytes, RCCL_UINT8, r, c->comm,
                                            ctx->stream);
            if (rr != 0) { g_rccl.GroupEnd(); return rccl_fail(rr, "ncclBroadcast"); }
        }
        off += h[(size_t) r];
    }
    RSK_RCCL(g_rccl.GroupEnd(), "ncclGroupEnd");
    // "on return *d_all holds all ranks' records" (include/reseek_amd.h): the broadcasts are queued on the context's stream, which may
    // be a non-blocking user stream a plain hipMemcpy does not order with -- wait here
    RSK_HIP(hipStreamSynchronize(ctx->stream));
    *d_all = c->d_all;
    *n_all = total;
    if (counts) memcpy(counts, h.data(), (size_t) c->world * 8);
    return RSK_OK;
}
