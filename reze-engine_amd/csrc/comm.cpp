// comm.cpp — multi-GPU: the lazily bound RCCL, communicators, the all-gather, the peer-direct gather (ctx.h).
#include "ctx.h"

#include <dlfcn.h>

using namespace rzi;

namespace rzi {

Rccl g_rccl;

int rccl_bind()
{
    if (g_rccl.h) return RZ_OK;
    // ONE RCCL per process. A host that already carries a copy (PyTorch bundles its own librccl.so, soname librccl.so.1,
    // and loads it with libtorch_hip) must not get a second one next to it — two RCCL runtimes in one process each
    // start their own proxy threads and IPC state. RTLD_NOLOAD returns the already-mapped object with that soname, if
    // there is one; only a process without RCCL loads ROCm's.
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    bool reused = h != nullptr;
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD), reused = h != nullptr;
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(RZ_ERR_UNSUPPORTED, "RCCL not available: %s", dlerror());
    Rccl r;
    r.h = h;
    r.reused = reused;
    r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(h, "ncclGetVersion"));
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(h, "ncclCommCount"));
    r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString || !r.CommInitAll ||
        !r.GroupStart || !r.GroupEnd)
        return fail(RZ_ERR_UNSUPPORTED, "RCCL symbols missing");
    g_rccl = r;
    return RZ_OK;
}

// Undo rz_gather_direct for everything `c` takes part in: as a root, every contributor goes back to its own output
// buffers (after draining, so no kernel is still storing into memory about to be freed); as a contributor, it leaves
// the root's list.
void drop_direct_gather(rz_ctx *c)
{
    drop_graph(c);
    for (rz_ctx *k : c->contributors) drop_graph(k);
    for (rz_ctx *k : c->contributors) {
        if (k != c) { (void)hipSetDevice(k->device); if (k->stream) (void)hipStreamSynchronize(k->stream); }
        k->ext_pos = k->ext_nrm = nullptr;
        k->gather_root = nullptr;
    }
    c->contributors.clear();
    if (c->gather_root) {
        auto &v = c->gather_root->contributors;
        v.erase(std::remove(v.begin(), v.end(), c), v.end());
        c->gather_root = nullptr;
        c->ext_pos = c->ext_nrm = nullptr;
    }
    (void)hipSetDevice(c->device);
}

}  // namespace rzi

extern "C" {

static int comm_buffers(rz_ctx *c, int nranks, int rank, uint32_t v_total);

int rz_comm_unique_id(char id[128])
{
    if (!id) return fail(RZ_ERR_INVALID, "null id");
    if (int r = rccl_bind()) return r;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    NCCL_TRY(g_rccl.GetUniqueId(&u));
    memcpy(id, &u, 128);
    return RZ_OK;
}

int rz_rccl_info(char *path, size_t path_bytes, int *version, int *reused)
{
    if (int r = rccl_bind()) return r;
    if (path && path_bytes) {
        Dl_info di;
        memset(&di, 0, sizeof di);
        path[0] = 0;
        if (dladdr(reinterpret_cast<void *>(g_rccl.AllGather), &di) && di.dli_fname) snprintf(path, path_bytes, "%s", di.dli_fname);
    }
    if (version) { *version = 0; if (g_rccl.GetVersion) (void)g_rccl.GetVersion(version); }
    if (reused) *reused = g_rccl.reused ? 1 : 0;
    return RZ_OK;
}

int rz_comm_info(rz_ctx *c, int *comm_count, int *comm_user_rank)
{
    if (int r = use(c)) return r;
    if (!c->comm) return fail(RZ_ERR_INVALID, "rz_comm_init has not been called");
    if (!g_rccl.CommCount || !g_rccl.CommUserRank) return fail(RZ_ERR_UNSUPPORTED, "this RCCL lacks ncclCommCount / ncclCommUserRank");
    int n = 0, u = -1;
    NCCL_TRY(g_rccl.CommCount(c->comm, &n));
    NCCL_TRY(g_rccl.CommUserRank(c->comm, &u));
    if (comm_count) *comm_count = n;
    if (comm_user_rank) *comm_user_rank = u;
    return RZ_OK;
}

int rz_comm_init(rz_ctx *c, int nranks, int rank, const char id[128], uint32_t v_total)
{
    if (int r = use(c)) return r;
    if (nranks < 1 || rank < 0 || rank >= nranks || !id) return fail(RZ_ERR_INVALID, "bad communicator arguments");
    if (c->I != 1) return fail(RZ_ERR_UNSUPPORTED, "instancing and vertex sharding are exclusive: a crowd shards along the instance axis (rz_instance_range), with nothing to gather");
    if (c->V == 0) return fail(RZ_ERR_INVALID, "upload this rank's mesh shard before rz_comm_init");
    uint32_t b = 0, n = 0;
    if (int r = rz_shard_range(v_total, nranks, rank, &b, &n)) return r;
    if (n != c->V) return fail(RZ_ERR_INVALID, "rank %d holds %u vertices but rz_shard_range assigns %u", rank, c->V, n);
    if (int r = rccl_bind()) return r;
    if (c->comm) { g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
    ncclUniqueId u;
    memcpy(&u, id, 128);
    NCCL_TRY(g_rccl.CommInitRank(&c->comm, nranks, u, rank));
    return comm_buffers(c, nranks, rank, v_total);
}

int rz_allgather(rz_ctx *c, int with_normals)
{
    if (int r = use(c)) return r;
    if (!c->comm) return fail(RZ_ERR_INVALID, "rz_comm_init has not been called");
    const size_t count = (size_t)c->chunk * 3;
    NCCL_TRY(g_rccl.AllGather(c->out_pos, c->g_pos, count, ncclFloat, c->comm, c->stream));
    if (with_normals) NCCL_TRY(g_rccl.AllGather(c->out_nrm, c->g_nrm, count, ncclFloat, c->comm, c->stream));
    return RZ_OK;
}

static int comm_buffers(rz_ctx *c, int nranks, int rank, uint32_t v_total)
{
    c->nranks = nranks; c->rank = rank; c->v_total = v_total;
    if (int r = rz_gather_chunk(v_total, nranks, &c->chunk)) return r;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    dfree(c->g_pos); dfree(c->g_nrm);
    const size_t g = (size_t)nranks * c->chunk * 3 * sizeof(float);
    HIP_TRY(hipMalloc(&c->g_pos, g));
    HIP_TRY(hipMalloc(&c->g_nrm, g));
    return ensure_outputs(c);
}

int rz_comm_init_all(rz_ctx **ctxs, int n, uint32_t v_total)
{
    if (!ctxs || n < 1 || n > 64) return fail(RZ_ERR_INVALID, "bad context list");
    if (int r = rccl_bind()) return r;
    int devs[64];
    for (int r = 0; r < n; ++r) {
        rz_ctx *c = ctxs[r];
        if (!c) return fail(RZ_ERR_INVALID, "null context in list");
        if (c->I != 1) return fail(RZ_ERR_UNSUPPORTED, "instancing and vertex sharding are exclusive: a crowd shards along the instance axis (rz_instance_range), with nothing to gather");
        uint32_t b = 0, cnt = 0;
        if (int e = rz_shard_range(v_total, n, r, &b, &cnt)) return e;
        if (cnt != c->V) return fail(RZ_ERR_INVALID, "context %d holds %u vertices but rz_shard_range assigns %u", r, c->V, cnt);
        for (int k = 0; k < r; ++k)
            if (devs[k] == c->device) return fail(RZ_ERR_INVALID, "contexts %d and %d share device %d: RCCL needs one GPU per rank", k, r, c->device);
        devs[r] = c->device;
        if (c->comm) { g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
    }
    ncclComm_t comms[64];
    NCCL_TRY(g_rccl.CommInitAll(comms, n, devs));
    for (int r = 0; r < n; ++r) {
        ctxs[r]->comm = comms[r];
        if (int e = comm_buffers(ctxs[r], n, r, v_total)) return e;
    }
    return RZ_OK;
}

int rz_allgather_all(rz_ctx **ctxs, int n, int with_normals)
{
    if (!ctxs || n < 1) return fail(RZ_ERR_INVALID, "bad context list");
    for (int r = 0; r < n; ++r)
        if (!ctxs[r] || !ctxs[r]->comm || ctxs[r]->nranks != n) return fail(RZ_ERR_INVALID, "rz_comm_init_all has not been called on this list");
    NCCL_TRY(g_rccl.GroupStart());
    for (int r = 0; r < n; ++r) {
        rz_ctx *c = ctxs[r];
        const size_t count = (size_t)c->chunk * 3;
        ncclResult_t a = g_rccl.AllGather(c->out_pos, c->g_pos, count, ncclFloat, c->comm, c->stream);
        if (a == ncclSuccess && with_normals) a = g_rccl.AllGather(c->out_nrm, c->g_nrm, count, ncclFloat, c->comm, c->stream);
        if (a != ncclSuccess) { g_rccl.GroupEnd(); return fail(RZ_ERR_RCCL, "ncclAllGather failed: %s", g_rccl.GetErrorString(a)); }
    }
    NCCL_TRY(g_rccl.GroupEnd());
    return RZ_OK;
}

static int gather_direct_attach(rz_ctx **ctxs, int n, uint32_t v_total, int root, bool *started)
{
    if (!ctxs || n < 1 || n > 64 || root < 0 || root >= n) return fail(RZ_ERR_INVALID, "bad context list / root");
    for (int r = 0; r < n; ++r) {
        rz_ctx *c = ctxs[r];
        if (!c) return fail(RZ_ERR_INVALID, "null context in list");
        if (c->I != 1) return fail(RZ_ERR_UNSUPPORTED, "instancing and vertex sharding are exclusive: a crowd shards along the instance axis (rz_instance_range), with nothing to gather");
        uint32_t b = 0, cnt = 0;
        if (int e = rz_shard_range(v_total, n, r, &b, &cnt)) return e;
        if (cnt != c->V) return fail(RZ_ERR_INVALID, "context %d holds %u vertices but rz_shard_range assigns %u", r, c->V, cnt);
        for (int k = 0; k < r; ++k)
            if (ctxs[k] == c) return fail(RZ_ERR_INVALID, "context listed twice");
    }
    rz_ctx *rt = ctxs[root];
    *started = true;                    // validation passed: from here on state changes
    for (int r = 0; r < n; ++r) drop_direct_gather(ctxs[r]);
    // the gathered buffer lives on the root's GPU (rz_read_gathered, or a renderer there, consumes it)
    rt->nranks = n; rt->rank = root; rt->v_total = v_total;
    uint32_t chunk = 0;
    if (int r = rz_gather_chunk(v_total, n, &chunk)) return r;
    HIP_TRY(hipSetDevice(rt->device));
    HIP_TRY(hipStreamSynchronize(rt->stream));
    dfree(rt->g_pos); dfree(rt->g_nrm);
    const size_t g = (size_t)n * chunk * 3 * sizeof(float);
    HIP_TRY(hipMalloc(&rt->g_pos, g));
    HIP_TRY(hipMalloc(&rt->g_nrm, g));
    // on the root's stream and drained: a plain hipMemset is asynchronous to the host and rides the NULL stream, which the
    // contexts' non-blocking streams do not wait for — it could land on top of the first frame's output
    HIP_TRY(hipMemsetAsync(rt->g_pos, 0, g, rt->stream));
    HIP_TRY(hipMemsetAsync(rt->g_nrm, 0, g, rt->stream));
    HIP_TRY(hipStreamSynchronize(rt->stream));
    for (int r = 0; r < n; ++r) {
        rz_ctx *c = ctxs[r];
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->device != rt->device) {
            int can = 0;
            HIP_TRY(hipDeviceCanAccessPeer(&can, c->device, rt->device));
            if (!can) return fail(RZ_ERR_UNSUPPORTED, "GPU %d cannot store into GPU %d's memory (no peer access)", c->device, rt->device);
            hipError_t pe = hipDeviceEnablePeerAccess(rt->device, 0);
            if (pe == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            else if (pe != hipSuccess) return fail(RZ_ERR_HIP, "hipDeviceEnablePeerAccess(%d -> %d): %s", c->device, rt->device, hipGetErrorString(pe));
        }
        if (!c->ev_done) HIP_TRY(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
        c->nranks = n; c->rank = r; c->v_total = v_total; c->chunk = chunk;
        c->ext_pos = rt->g_pos + (size_t)r * chunk * 3;
        c->ext_nrm = rt->g_nrm + (size_t)r * chunk * 3;
        c->gather_root = rt;
        rt->contributors.push_back(c);
    }
    HIP_TRY(hipSetDevice(rt->device));
    return RZ_OK;
}

int rz_gather_direct(rz_ctx **ctxs, int n, uint32_t v_total, int root)
{
    bool started = false;
    const int rc = gather_direct_attach(ctxs, n, v_total, root, &started);
    if (rc != RZ_OK && started) {
        // all or nothing: a failure half-way (no peer access from one of the GPUs, out of memory ...) must not leave some
        // shards storing into the root's buffer and others not
        const std::string msg = rz_last_error();
        for (int r = 0; r < n; ++r)
            if (ctxs[r]) drop_direct_gather(ctxs[r]);
        return fail(rc, "%s", msg.c_str());
    }
    return rc;
}

int rz_gather_fence(rz_ctx *root)
{
    if (int r = use(root)) return r;
    if (root->contributors.empty()) return fail(RZ_ERR_INVALID, "rz_gather_direct has not been called with this root");
    for (rz_ctx *k : root->contributors) {
        if (k == root) continue;
        HIP_TRY(hipSetDevice(k->device));
        HIP_TRY(hipEventRecord(k->ev_done, k->stream));
        HIP_TRY(hipSetDevice(root->device));
        HIP_TRY(hipStreamWaitEvent(root->stream, k->ev_done, 0));
    }
    HIP_TRY(hipSetDevice(root->device));
    return RZ_OK;
}

int rz_read_gathered(rz_ctx *c, uint32_t v0, uint32_t n, float *pos3, float *nrm3)
{
    if (int r = use(c)) return r;
    if (!c->g_pos) return fail(RZ_ERR_INVALID, "no gathered buffer");
    if ((uint64_t)v0 + n > c->v_total) return fail(RZ_ERR_INVALID, "range exceeds the full mesh");
    if (!c->contributors.empty())
        if (int r = rz_gather_fence(c)) return r;    // peer-direct: the other GPUs' frames must have landed
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (pos3 && n) HIP_TRY(hipMemcpy(pos3, c->g_pos + (size_t)v0 * 3, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (nrm3 && n) HIP_TRY(hipMemcpy(nrm3, c->g_nrm + (size_t)v0 * 3, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return RZ_OK;
}

}  // extern "C"
