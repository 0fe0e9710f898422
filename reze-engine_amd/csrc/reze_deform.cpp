// reze_deform.cpp — implementation of the C ABI in include/reze_deform.h.
//
// One rz_ctx = one MI355X: a HIP stream, the static mesh shard re-laid-out as planar SoA, the
// skeleton, optional morph targets, per-frame pose staging, output buffers, and (optionally) an
// RCCL communicator for the all-gather of deformed positions. Every entry point cites, in the
// header, the reference call site it replaces; this file is only plumbing around
// deform_kernels.hip. There is NO CPU fallback: without a working HIP device every call fails.
#include "../../include/reze_deform.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "deform_kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(e_ == hipErrorOutOfMemory ? RZ_ERR_OOM : RZ_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                 \
    } while (0)

uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

constexpr uint32_t kVertPad = 1024;   // planes are padded to a whole S=1 tile (256 quads)
// Shards of one mesh are equal-sized except the last; their size is a multiple of 256 vertices (whole quads, whole S = 1 wave
// steps, 16-byte aligned float3 boundaries in the gathered buffer). 1024 (rounds 1-3) made the ranks of a 1 M-vertex mesh over
// 8 GPUs carry 125 952 vertices and the last 118 336; now 125 184 / 123 712: the slowest rank has 0.6 % less to do.
constexpr uint32_t kShardGrain = 256;
constexpr int kStageSlots = 8;

// ---- lazily bound RCCL (librccl.so.1 is only needed by the multi-GPU entry points) ----
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    bool reused = false;                // bound to a copy the process had already loaded (e.g. PyTorch's)
};
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

#define NCCL_TRY(expr)                                                                              \
    do {                                                                                            \
        ncclResult_t r_ = (expr);                                                                   \
        if (r_ != ncclSuccess)                                                                      \
            return fail(RZ_ERR_RCCL, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));            \
    } while (0)

}  // namespace

// Slots of the zero-copy pose ring. One hipEventRecord per RZ_ZC_SLOTS / 2 uploads guards slot reuse, and a record costs ~1.4 us of
// stream time: measured on a 1/8 shard of C5 (tools/live_shard.py, per-frame-pose loop over the resident replay): 8 slots +0.86 us
// per frame, 16 slots +0.50 us, 32 slots +0.37 us. 32 x <= 256 KB of pinned memory per context.
#ifndef RZ_ZC_SLOTS
#define RZ_ZC_SLOTS 32
#endif

struct rz_ctx {
    int device = 0;
    int n_cu = 256;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    // static mesh shard
    uint32_t V = 0, Vp = 0;
    float *geom = nullptr;              // 6 x Vp
    uint32_t *j01 = nullptr, *j23 = nullptr, *wq = nullptr;

    // skeleton
    uint32_t B = 0;
    float *inv_bind = nullptr;          // B x 16
    // optional topology for on-device FK
    bool has_topology = false;
    uint4 *fk_rec = nullptr;            // [B][2] one 32-byte record per bone (deform_kernels.h: RzFkParams::bone_rec)
    bool pose_local_t = false;
    // device-side motion sampling (rz_upload_animation / rz_set_pose_sampled)
    bool has_animation = false, pose_sampled = false;
    uint4 *an_bone_range = nullptr, *an_feed_range = nullptr;     // (first key, end, first frame, last frame) per bone / per morph feed
    uint32_t *an_feed_off = nullptr;
    float *an_key_frame = nullptr, *an_key_pos = nullptr, *an_mkey_frame = nullptr, *an_mkey_weight = nullptr, *an_feed_ratio = nullptr;
    float4 *an_key_rot = nullptr;
    uint4 *an_key_interp = nullptr;
    uint32_t an_M = 0;                  // vertex-morph count the feeds were built for
    float *an_frames = nullptr;         // [I]
    bool frames_inline = false;         // one character: the frame rides in the kernel arguments (frame0), nothing is uploaded
    float frame0 = 0.0f;
    size_t an_frames_alloc = 0;          // the current local pose carries translations (behind the rotations in its slot)
    int fk_levels = 0;                  // depth of the hierarchy
    float4 *local_q = nullptr;          // I x B   (current pose slot)

    bool pose_local = false;            // the current pose came from rz_set_pose_local
    // physics hand-off for device-solved frames (rz_override_world)
    int *ovr_off = nullptr, *ovr_bone = nullptr;
    float *ovr_world = nullptr;
    uint32_t ovr_count = 0;
    size_t ovr_alloc = 0, ovr_off_alloc = 0;
    // PMX bone morphs (rz_upload_bone_morphs): entries grouped by bone, ascending morph index inside a bone
    uint32_t *bm_off = nullptr, *bm_morph = nullptr;
    float4 *bm_rot = nullptr, *bm_tr = nullptr;
    uint32_t bm_count = 0;

    // morphs
    int morph_mode = 0;                 // 0 none, 1 dense, 2 sparse
    uint32_t M = 0, Mpad = 12;
    float *dense = nullptr;             // M x 3 x Vp
    uint32_t *sp_ptr = nullptr;         // Vp + 1
    float4 *sp_entries = nullptr;
    uint64_t sp_count = 0;

    // per-frame state
    uint32_t I = 1;
    float *world = nullptr;             // I x B x 16   (current pose slot)
    // Per-frame INPUTS are double-buffered and uploaded on their own stream, so the upload of pose f+1 overlaps the
    // kernels of pose f: ev_up[k] = slot k has landed (the compute stream waits for it), ev_free[k] = everything
    // that reads slot k has been enqueued up to here (the upload stream waits for it before overwriting the slot).
    // One device block per pose slot: [world I*B*16 | morph weights pad4(I*max(M,1)) | local rotations I*B*4 | local
    // translations I*B*3] floats. A world-matrix pose fills [world | weights], a local pose [weights | rotations (|
    // translations)] — each a CONTIGUOUS range, so every upload is one copy (measured: a small H2D copy is a 4.5 us blit
    // kernel on this runtime, and a second one for 256 bytes of morph weights cost as much as the first).
    // The offsets are those of the instance / bone / morph counts AT UPLOAD TIME (point_pose_slot): shrinking the crowd
    // afterwards leaves the resident pose where it is.
    float *pose_blk[2] = {nullptr, nullptr};
    size_t mw_pad = 0;                              // floats reserved for the morph weights in the current layout (multiple of 4)
    int pose_slot = 0;
    hipStream_t up_stream = nullptr;
    hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr};
    bool free_recorded[2] = {false, false};
    float4 *palette = nullptr;          // I x B x 3   (current ring slot)
    float *morph_w = nullptr;           // I x M   (current pose slot)
    uint32_t *act_idx = nullptr;        // I x Mpad    (current ring slot)
    float *act_w = nullptr;             // I x Mpad    (current ring slot)
    int *act_count = nullptr;           // I           (current ring slot)
    // Everything the FRONT kernels (rz_prep_kernel / rz_fk_kernel) hand to the skin kernel lives in a 2-slot ring, so the
    // front kernels of frame f+1 can run on the upload stream while the skin kernel of frame f is still reading slot f:
    // ev_front[s] = slot s is ready (compute stream waits), ev_skin[s] = the skin kernel that read slot s has been enqueued
    // up to here (the front stream waits before overwriting the slot, two frames later). Used by crowds (overlap_on).
    float4 *palette_ring[2] = {nullptr, nullptr};
    uint32_t *act_idx_ring[2] = {nullptr, nullptr};
    float *act_w_ring[2] = {nullptr, nullptr};
    int *act_count_ring[2] = {nullptr, nullptr};
    int ring_slot = 0;
    hipEvent_t ev_front[2] = {nullptr, nullptr}, ev_skin[2] = {nullptr, nullptr};
    bool skin_recorded[2] = {false, false};
    bool overlap_on = false;            // the two streams currently follow the overlapped-front protocol
    bool pose_set = false;
    uint32_t pose_I = 0;                // instance count the current pose was uploaded for
    size_t pose_alloc_I = 0, pose_alloc_B = 0, pose_alloc_M = 0;

    // outputs
    float *out_pos = nullptr, *out_nrm = nullptr;
    size_t out_alloc_floats = 0;
    // fused consumers
    float *edge = nullptr;              // Vp
    float *out_hull = nullptr;          // I x Vp x 3
    size_t hull_alloc_floats = 0;
    uint32_t *aabb = nullptr;           // I x 2 x 6 keys
    size_t aabb_alloc_inst = 0;
    bool aabb_on = false;
    int aabb_slot = 0;                  // slot the NEXT frame accumulates into
    bool aabb_rearm = false;            // both slots must be armed again before the next frame

    // pinned staging ring for rz_set_pose
    void *stage[kStageSlots] = {};
    size_t stage_bytes = 0;
    hipEvent_t stage_ev[kStageSlots] = {};
    bool stage_used[kStageSlots] = {};
    int stage_next = 0;

    // Zero-copy poses (one character, <= 256 KB): rz_set_pose* only writes the pose into a slot of this pinned,
    // device-mapped ring — no copy is enqueued at all. The first frame's kernels read it over the host link (rz_fk_kernel
    // the local pose; the one-launch deform kernel the world matrices, whose workgroup 0 also leaves them in the device
    // pose block for the frames that replay the pose); anything that needs a device-resident pose first (rz_prep_kernel)
    // gets it through make_resident(). Measured on MI355X (tools/uploadbench): a 16.6 KB hipMemcpyAsync in front of a
    // frame costs 18 us, two of <= 16 KB 9.5 us, reading the pinned slot from the kernel 7 us with the loads fully exposed.
    // A slot is reused kZcSlots (32) uploads later; one event per kZcSlots / 2 uploads (recorded on the compute stream at upload time) proves
    // its readers are done, so there is no per-frame marker either.
    static constexpr int kZcSlots = RZ_ZC_SLOTS;    // a slot is reused kZcSlots uploads later; one event per kZcSlots / 2 uploads guards the reuse
    void *zc_host[kZcSlots] = {};
    void *zc_dev[kZcSlots] = {};
    size_t zc_bytes = 0;
    uint64_t zc_uploads = 0;
    hipEvent_t zc_ev[2] = {nullptr, nullptr};
    uint64_t zc_ev_seq[2] = {~0ull, ~0ull};
    // Pose prefetch (deform_kernels.h: pf_*): every slot carries a header behind its payload — the sequence number of the pose
    // it holds, written LAST — and the device keeps one tag per pose block: the sequence number a frame's helper workgroup
    // staged there. Sequence numbers are (ring epoch << 32 | upload index + 1): a re-allocated ring never matches old tags.
    size_t zc_hdr_off = 0;              // byte offset of the header inside a slot
    uint32_t zc_epoch = 0;
    uint64_t zc_seq_cur = 0;            // sequence number of the current pose (0 = not prefetchable)
    uint64_t *zc_tag = nullptr;         // device: [2], one per pose block
    int t_prefetch = -1;                // "pose_prefetch": -1 / 1 on, 0 off
    int zc_cur = -1;                    // slot of the current pose, -1 = the current pose came down as a copy
    bool zc_local = false;              // layout of that slot: [weights | rotations | translations] or [world | weights]
    int zc_kind = 0;                    // 0 world, 1 local rotations, 2 local rotations + translations (part of the sequence number)
    size_t zc_total = 0, zc_mw_off = 0, zc_lq_off = 0;     // bytes in the slot, and where the weights / rotations sit in it
    bool world_resident = true, mw_resident = true, local_resident = true;   // which parts the device pose block holds

    // host-compacted active-morph list of the current pose (single-instance FAST path);
    // count < 0 means "more than kKargMorphs active: use the prep kernel"
    RzMorphList ml;

    // tuning (0 / -1 = automatic)
    int t_split = 0, t_unroll = 0, t_grid_cap = 0, t_nt = 1, t_nts = -1, t_geo = 0, t_fast = -1, t_instloop = -1, t_dbg = 0, t_outcap = -1, t_instblock = 0, t_instorder = 1, t_overlap = -1, t_zerocopy = -1, t_fusefk = -1;
    // Bone-subset crowd frames: per vertex run of the CURRENT launch shape, the ascending list of bones the run's
    // vertices name, and the joints rewritten as slots of that list. Derived from the static mesh, rebuilt (one small kernel +
    // one readback of the counts) whenever the shape (vertices per run, runs), the mesh or the skeleton changes.
    uint32_t *rj01 = nullptr, *rj23 = nullptr;      // [Vp]
    uint16_t *sub_list = nullptr;                   // [sub_runs][sub_B]
    uint32_t *sub_count = nullptr;                  // [sub_runs]
    size_t sub_list_alloc = 0, sub_count_alloc = 0;
    uint32_t sub_per = 0, sub_runs = 0, sub_B = 0, sub_max = 0;
    bool sub_valid = false;
    int t_subsets = -1;                 // "inst_subsets": -1 / 1 = stage only the bones a vertex run names when that is a gain, 0 = always the whole palette
    bool palette_stale = false;         // the last crowd frame formed its palettes in LDS only (subset form): rz_read_palette forms them on demand
    int t_graph = 0;                    // "graph" tuning key: rz_deform_n replays captured hipGraphs of kGraphFrames frames
    hipGraphExec_t graph_exec = nullptr;
    uint64_t graph_sig = 0;             // signature of everything the captured launches depend on
    bool tuned_by_search = false;       // rz_autotune set morph_split / grid_cap / inst_loop for the CURRENT mesh, morphs and instance count

    // multi-GPU
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    uint32_t v_total = 0, chunk = 0;
    float *g_pos = nullptr, *g_nrm = nullptr;   // nranks x chunk x 3
    // peer-direct gather (rz_gather_direct): this context's kernels store straight into the root's gathered buffer
    float *ext_pos = nullptr, *ext_nrm = nullptr;
    rz_ctx *gather_root = nullptr;              // set on every contributor (the root contributes too)
    // rz_fork: a fork borrows every STATIC device buffer of its lender (mesh, skeleton, topology, morph targets, bone morphs,
    // motion, edge scale) and owns everything per-frame (streams, pose slots, palettes, outputs). While forks exist neither
    // side may replace static data.
    rz_ctx *lender = nullptr;
    int n_forks = 0;
#ifdef RZ_ABLATE
    unsigned long long *tl = nullptr;           // tools-only build: per-wave timeline of the last frame (dbg = 100)
    size_t tl_waves = 0;
#endif
#ifdef RZ_ALL_VARIANTS
    uint32_t *gate_host = nullptr, *gate_dev = nullptr;     // tools-only build: rz_debug_gate
#endif
    std::vector<rz_ctx *> contributors;         // set on the root
    hipEvent_t ev_done = nullptr;               // "my last frame has been enqueued up to here" for rz_gather_fence
};

namespace {

int use(rz_ctx *c)
{
    if (!c) return fail(RZ_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    return RZ_OK;
}

// static data shared between a context and its forks cannot be replaced
int static_unlocked(const rz_ctx *c, const char *what)
{
    if (c->lender) return fail(RZ_ERR_INVALID, "%s on a fork: static data belongs to the context it was forked from", what);
    if (c->n_forks) return fail(RZ_ERR_INVALID, "%s while %d fork(s) of this context share its static data: destroy them first", what, c->n_forks);
    return RZ_OK;
}

template <class T> void dfree(T *&p)
{
    if (p) { (void)hipFree(p); p = nullptr; }
}

// Poll an event (no sleep: a blocking wait wakes tens of microseconds late, which starves a GPU whose frames are 16 us long
// — measured: 36 us per frame). Bounded: a GPU that stops making progress turns into an error after 10 s, not a hang.
int poll_event(hipEvent_t ev, const char *what)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; ++spins) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) return RZ_OK;
        if (q != hipErrorNotReady) return fail(RZ_ERR_HIP, "%s: %s", what, hipGetErrorString(q));
        if ((spins & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10))
            return fail(RZ_ERR_HIP, "%s: the GPU made no progress for 10 s", what);
    }
}

// A captured hipGraph bakes device pointers and launch shapes in. The replay key (frame_signature) covers all of them; on
// top of that every entry point that frees or re-shapes something a frame reads drops the graph outright.
void drop_graph(rz_ctx *c)
{
    if (c->graph_exec) { (void)hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
    c->graph_sig = 0;
}

// upload-time scratch that is released on every exit path (HIP_TRY returns early on failure)
template <class T> struct Scratch {
    T *p = nullptr;
    ~Scratch() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count) { return hipMalloc(&p, count * sizeof(T)); }
};

int ensure_outputs(rz_ctx *c)
{
    // one instance: room for a whole all-gather chunk; instances are strided by Vp
    const size_t need = (c->I == 1) ? std::max<size_t>(c->Vp, c->chunk) * 3 : (size_t)c->I * c->Vp * 3;
    if (need == 0) return RZ_OK;
    if (need > c->out_alloc_floats) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        dfree(c->out_pos);
        dfree(c->out_nrm);
        HIP_TRY(hipMalloc(&c->out_pos, need * sizeof(float)));
        HIP_TRY(hipMalloc(&c->out_nrm, need * sizeof(float)));
        HIP_TRY(hipMemsetAsync(c->out_pos, 0, need * sizeof(float), c->stream));
        HIP_TRY(hipMemsetAsync(c->out_nrm, 0, need * sizeof(float), c->stream));
        c->out_alloc_floats = need;
    }
    if (c->edge && need > c->hull_alloc_floats) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        dfree(c->out_hull);
        HIP_TRY(hipMalloc(&c->out_hull, need * sizeof(float)));
        HIP_TRY(hipMemsetAsync(c->out_hull, 0, need * sizeof(float), c->stream));
        c->hull_alloc_floats = need;
    }
    // Bounding-box keys: a frame accumulates into one slot and re-arms the other FOR THE INSTANCES IT LAUNCHES, so the
    // buffer is armed from scratch whenever the reduction is switched on or the instance count changes (aabb_rearm) —
    // otherwise an instance that sat out some frames would come back onto a slot still holding its old extents.
    if (c->aabb_on && (c->I > c->aabb_alloc_inst || c->aabb_rearm)) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->I > c->aabb_alloc_inst) {
            dfree(c->aabb);
            HIP_TRY(hipMalloc(&c->aabb, (size_t)c->I * 12 * sizeof(uint32_t)));
            c->aabb_alloc_inst = c->I;
        }
        std::vector<uint32_t> init((size_t)c->aabb_alloc_inst * 12);
        for (size_t i = 0; i < init.size(); ++i) init[i] = (i % 6) < 3 ? 0xffffffffu : 0u;
        HIP_TRY(hipMemcpy(c->aabb, init.data(), init.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        c->aabb_slot = 0;
        c->aabb_rearm = false;
    }
    return RZ_OK;
}

void set_ring(rz_ctx *c, int slot)
{
    c->ring_slot = slot;
    c->palette = c->palette_ring[slot]; c->act_idx = c->act_idx_ring[slot]; c->act_w = c->act_w_ring[slot]; c->act_count = c->act_count_ring[slot];
}

// Lay the CURRENT instance / bone / morph counts out over pose slot k and make it the current slot.
void point_pose_slot(rz_ctx *c, int k)
{
    const size_t I = c->I, B = c->B, Mq = std::max<uint32_t>(c->M, 1);
    c->mw_pad = (I * Mq + 3) / 4 * 4;
    c->pose_slot = k;
    c->world = c->pose_blk[k];
    c->morph_w = c->pose_blk[k] + I * B * 16;
    c->local_q = reinterpret_cast<float4 *>(c->pose_blk[k] + I * B * 16 + c->mw_pad);
}

int ensure_pose_buffers(rz_ctx *c)
{
    if (c->B == 0) return RZ_OK;
    const uint32_t Mq = std::max<uint32_t>(c->M, 1);
    if (c->I <= c->pose_alloc_I && c->B <= c->pose_alloc_B && Mq <= c->pose_alloc_M && c->world) return RZ_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->up_stream));
    for (int k = 0; k < 2; ++k) { dfree(c->pose_blk[k]); c->free_recorded[k] = false; }
    c->world = nullptr; c->morph_w = nullptr; c->local_q = nullptr;
    for (int k = 0; k < 2; ++k) { dfree(c->palette_ring[k]); dfree(c->act_idx_ring[k]); dfree(c->act_w_ring[k]); dfree(c->act_count_ring[k]); c->skin_recorded[k] = false; }
    c->palette = nullptr; c->act_idx = nullptr; c->act_w = nullptr; c->act_count = nullptr;
    const size_t I = c->I, B = c->B;
    const size_t Mpad = round_up(Mq + 8, 4);
    const size_t blk_floats = I * B * 16 + ((I * Mq + 3) / 4 * 4) + I * B * 7 + 4;      // + 4: the prefetch helper copies whole 16-byte cells
    for (int k = 0; k < 2; ++k) {
        HIP_TRY(hipMalloc(&c->pose_blk[k], blk_floats * sizeof(float)));
        HIP_TRY(hipMemsetAsync(c->pose_blk[k], 0, blk_floats * sizeof(float), c->stream));
    }
    point_pose_slot(c, 0);
    for (int k = 0; k < 2; ++k) {
        HIP_TRY(hipMalloc(&c->palette_ring[k], I * B * 3 * sizeof(float4)));
        HIP_TRY(hipMalloc(&c->act_idx_ring[k], I * Mpad * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&c->act_w_ring[k], I * Mpad * sizeof(float)));
        HIP_TRY(hipMalloc(&c->act_count_ring[k], I * sizeof(int)));
        HIP_TRY(hipMemsetAsync(c->palette_ring[k], 0, I * B * 3 * sizeof(float4), c->stream));
        HIP_TRY(hipMemsetAsync(c->act_idx_ring[k], 0, I * Mpad * sizeof(uint32_t), c->stream));
        HIP_TRY(hipMemsetAsync(c->act_w_ring[k], 0, I * Mpad * sizeof(float), c->stream));
        HIP_TRY(hipMemsetAsync(c->act_count_ring[k], 0, I * sizeof(int), c->stream));
    }
    set_ring(c, 0);
    if (!c->zc_tag) HIP_TRY(hipMalloc(&c->zc_tag, 2 * sizeof(uint64_t)));
    HIP_TRY(hipMemsetAsync(c->zc_tag, 0, 2 * sizeof(uint64_t), c->stream));
    c->zc_epoch++;                      // poses staged under the old layout must never match again
    c->zc_seq_cur = 0;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->pose_alloc_I = I; c->pose_alloc_B = B; c->pose_alloc_M = Mq;
    c->pose_set = false;
    return RZ_OK;
}

void free_animation(rz_ctx *c)
{
    drop_graph(c);
    dfree(c->an_bone_range); dfree(c->an_feed_range); dfree(c->an_feed_off);
    dfree(c->an_key_frame); dfree(c->an_key_pos); dfree(c->an_mkey_frame); dfree(c->an_mkey_weight); dfree(c->an_feed_ratio);
    dfree(c->an_key_rot); dfree(c->an_key_interp);
    c->has_animation = false;
    if (c->pose_sampled) { c->pose_sampled = false; c->pose_set = false; }
}

void free_bone_morphs(rz_ctx *c)
{
    if (!c->bm_off) return;
    drop_graph(c);
    dfree(c->bm_off); dfree(c->bm_morph); dfree(c->bm_rot); dfree(c->bm_tr);
    c->bm_count = 0;
}

template <typename T> int to_device(T **dst, const void *src, size_t count)
{
    *dst = nullptr;
    HIP_TRY(hipMalloc(dst, std::max<size_t>(count, 1) * sizeof(T)));
    if (count) HIP_TRY(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    return RZ_OK;
}

// A launch shape found by rz_autotune belongs to the workload it was timed on.
void forget_search(rz_ctx *c)
{
    if (!c->tuned_by_search) return;
    c->t_split = 0; c->t_grid_cap = 0; c->t_instloop = -1;
    c->tuned_by_search = false;
}

void free_morphs(rz_ctx *c)
{
    forget_search(c);
    drop_graph(c);
    dfree(c->dense); dfree(c->sp_ptr); dfree(c->sp_entries);
    free_bone_morphs(c);                  // their entries name morphs of the old set
    c->morph_mode = 0; c->M = 0; c->Mpad = 12; c->sp_count = 0;
    c->zc_epoch++; c->zc_seq_cur = 0;     // ... and so does a pose staged ahead of its frame
    c->pose_set = false;                  // morph weights belong to the old target set
}

int auto_split(const rz_ctx *c)
{
    if (c->morph_mode != 1) return 1;
    // S lanes share a quad, so waves = quads * S / 64. Measured on MI355X (profiles/r1_a_sweep*, and the search tables of
    // profiles/r3_bench_*.json): 126 k verts -> S = 4, 30 k -> S = 8, i.e. aim for ~1500 waves, never beyond 8; and even
    // the 1 M-vertex mesh (3 906 waves at S = 1) streams 3 % faster with two lanes per quad (122.8 vs 127.0 us; S = 4 is
    // within 0.4 % of S = 2), so a dense frame never runs below S = 2 — which also keeps rz_autotune's pick on the
    // heuristic plan instead of flipping between two near-equal candidates from run to run.
    const uint64_t quads = (uint64_t)c->Vp / 4 * c->I;
    const uint64_t want = 1500;
    int S = 2;
    while (S < 8 && quads * S / 64 < want) S <<= 1;
    while (S > 1 && (uint32_t)S > c->M) S >>= 1;
    return S;
}

struct Plan { RzVariant v; uint32_t grid_x, n_quads, quads_per_wave; bool prep, dma; int inst_group; uint32_t verts_per_wg; int poses_per_wg; uint32_t out_cap; int inst_block; bool fuse_fk; bool subsets; uint32_t sub_bones; uint64_t inst_lds; bool pf; uint32_t sp_cap; };

// Where the kernels read the current pose from: the device pose block, or (zero-copy, not yet resident) the pinned slot.
const float *src_world(const rz_ctx *c)
{
    return (c->zc_cur >= 0 && !c->world_resident && !c->zc_local) ? static_cast<const float *>(c->zc_dev[c->zc_cur]) : c->world;
}
const float *src_morph_w(const rz_ctx *c)
{
    if (c->zc_cur < 0 || c->mw_resident) return c->morph_w;
    const char *base = static_cast<const char *>(c->zc_dev[c->zc_cur]);
    return reinterpret_cast<const float *>(base + c->zc_mw_off);
}
const float4 *src_local_q(const rz_ctx *c)
{
    if (c->zc_cur < 0 || c->local_resident || !c->zc_local) return c->local_q;
    return reinterpret_cast<const float4 *>(static_cast<const char *>(c->zc_dev[c->zc_cur]) + c->zc_lq_off);
}

// Bring every part of a zero-copy pose into the device pose block (one copy out of the pinned slot, stream-ordered).
int make_resident(rz_ctx *c)
{
    if (c->zc_cur < 0 || (c->world_resident && c->mw_resident && c->local_resident)) return RZ_OK;
    void *dst = c->zc_local ? static_cast<void *>(c->morph_w) : static_cast<void *>(c->world);
    HIP_TRY(hipMemcpyAsync(dst, c->zc_host[c->zc_cur], c->zc_total, hipMemcpyHostToDevice, c->stream));
    c->world_resident = c->mw_resident = c->local_resident = true;
    return RZ_OK;
}

RzFkParams fk_params(const rz_ctx *c);

// Sequence number of a zero-copy pose, the value its slot header and — once staged — its device tag hold:
// ring epoch << 32 | pose kind << 30 | upload index (1-based, 30 bits). The KIND (0 world matrices, 1 local rotations, 2 local
// rotations + translations) is part of the number because a helper workgroup stages the next slot assuming the next pose has the
// layout and size of its own frame's: a pose of another kind can then never match what the helper expected. Sizes inside a kind
// only change with the skeleton / morph set / instance count, which start a new epoch.
uint64_t zc_seq(const rz_ctx *c, uint64_t upload_index_1, int kind)
{
    return ((uint64_t)c->zc_epoch << 32) | ((uint64_t)(kind & 3) << 30) | (upload_index_1 & 0x3fffffffull);
}

RzDeformParams deform_params(const rz_ctx *c, const Plan &pl)
{
    RzDeformParams p;
    memset(&p, 0, sizeof p);
    p.geom = c->geom; p.joints01 = c->j01; p.joints23 = c->j23; p.weights = c->wq;
    p.palette = c->palette; p.world = src_world(c); p.inv_bind = c->inv_bind; p.dense = c->dense;
    p.act_idx = c->act_idx; p.act_w = c->act_w; p.act_count = c->act_count; p.morph_w = src_morph_w(c);
    if (pl.v.fast && c->zc_cur >= 0) {            // the one-launch kernel's workgroup 0 makes the pose resident
        if (!c->world_resident && !c->zc_local) p.world_copy = c->world;
        if (!c->mw_resident && pl.v.mode == 2) p.morph_w_copy = c->morph_w;
        // Pose prefetch, first frame of a zero-copy world pose only. This frame: was the pose staged by the previous frame's
        // helper? Next frame: a helper workgroup looks at the slot the next upload will use (deform_kernels.h: pf_*).
        if (pl.pf && p.world_copy && c->zc_tag && c->zc_seq_cur) {
            p.st_tag = c->zc_tag + c->pose_slot; p.st_expect = c->zc_seq_cur;
            p.st_world = c->world; p.st_morph_w = (pl.v.mode == 2 && c->M > 0) ? c->morph_w : nullptr;
            const int nxt = (c->zc_cur + 1) % rz_ctx::kZcSlots;
            p.pf_src = static_cast<const float *>(c->zc_dev[nxt]);
            p.pf_src_seq = reinterpret_cast<const uint64_t *>(static_cast<const char *>(c->zc_dev[nxt]) + c->zc_hdr_off);
            p.pf_dst = c->pose_blk[c->pose_slot ^ 1];
            p.pf_tag = c->zc_tag + (c->pose_slot ^ 1);
            p.pf_expect = zc_seq(c, c->zc_uploads + 1, c->zc_kind);
            p.pf_bytes = (uint32_t)((c->zc_total + 15) / 16 * 16);
        }
    }
    p.sp_ptr = c->sp_ptr; p.sp_entries = c->sp_entries;
    p.out_pos = c->ext_pos ? c->ext_pos : c->out_pos; p.out_nrm = c->ext_nrm ? c->ext_nrm : c->out_nrm;
    p.edge = c->edge; p.out_hull = c->out_hull; p.aabb = c->aabb_on ? c->aabb : nullptr; p.aabb_slot = c->aabb_slot;
    p.n_verts = c->V;
    p.Vp = c->Vp; p.n_quads = pl.n_quads; p.quads_per_wave = pl.quads_per_wave; p.dma = pl.dma ? 1 : 0;
    p.B = (int)c->B; p.M = (int)c->M; p.Mpad = (int)c->Mpad;
    p.inst_order = c->t_instorder;
#ifdef RZ_ABLATE
    p.dbg = c->t_dbg == 100 ? 0 : c->t_dbg;
    p.tl = c->t_dbg == 100 ? c->tl : nullptr;      // (allocated by rz_debug_timeline_arm)
#endif
    p.out_cap = pl.out_cap;
    p.sp_cap = pl.sp_cap;
    if (pl.subsets) { p.rj01 = c->rj01; p.rj23 = c->rj23; p.sub_list = c->sub_list; p.sub_count = c->sub_count; p.sub_stride = (int)c->sub_B; }
    if (pl.fuse_fk) {
        p.fk = fk_params(c); p.fk_on = 1;
        if (c->zc_cur >= 0 && c->zc_local && !c->pose_sampled && (!c->local_resident || !c->mw_resident)) {
            // zero-copy local pose, first frame: workgroup 0 makes it resident (every later frame of this pose reads the device
            // block instead of pulling the pose over the host link in every workgroup), and — like the one-launch frame of a
            // world pose — the frame looks for a copy the previous frame's helper staged and carries a helper for the next upload
            float *blk_t = c->pose_local_t ? reinterpret_cast<float *>(c->local_q + c->B) : nullptr;
            p.fk.copy_q = c->local_q; p.fk.copy_t = blk_t;
            if (c->M > 0) p.morph_w_copy = c->morph_w;
            if (pl.pf && c->zc_tag && c->zc_seq_cur) {
                p.st_tag = c->zc_tag + c->pose_slot; p.st_expect = c->zc_seq_cur; p.fk.st_expect = c->zc_seq_cur;
                p.fk.st_local_q = c->local_q; p.fk.st_local_t = blk_t; p.st_morph_w = c->morph_w;
                const int nxt = (c->zc_cur + 1) % rz_ctx::kZcSlots;
                p.pf_src = static_cast<const float *>(c->zc_dev[nxt]);
                p.pf_src_seq = reinterpret_cast<const uint64_t *>(static_cast<const char *>(c->zc_dev[nxt]) + c->zc_hdr_off);
                p.pf_dst = c->pose_blk[c->pose_slot ^ 1] + (c->morph_w - c->pose_blk[c->pose_slot]);     // the local range of the other block
                p.pf_tag = c->zc_tag + (c->pose_slot ^ 1);
                p.pf_expect = zc_seq(c, c->zc_uploads + 1, c->zc_kind);
                p.pf_bytes = (uint32_t)((c->zc_total + 15) / 16 * 16);
            }
        }
    }
    return p;
}

// Launch shape of an instanced, morph-free crowd frame (rz_skin_instances_kernel): G poses per workgroup share one decode of each
// vertex; the grid is (vertex runs, pose groups), `total` workgroups in all.
struct InstShape { int G, blk, blk_full; bool want_in_kernel; uint32_t per, runs; };

void inst_runs(const rz_ctx *c, int G, int blk, bool for_subsets, uint32_t *per, uint32_t *runs)
{
    // 256 threads = two workgroups per CU, 512 / 1024 = one whose 8 / 16 waves share one staged palette group — ONE round of
    // workgroups. With bone subsets (15-30 KB of LDS) two 512-thread workgroups fit a CU and more, shorter runs name fewer
    // bones (36 -> 17 per run at 1024 workgroups), but whether that pays depends on the box: tools/c4_subsets.py measured
    // 256 / 512 / 768 / 1024 workgroups at 33.4 / 34.3 / 32.9 / 32.6 us on one MI355X and 33.1-33.4 / 42 / 42 / 42 us on two
    // others (profiles/r3_c4_subsets.txt). One workgroup per CU is the shape that is good everywhere, so it is the default;
    // rz_autotune tries the others on the box it runs on.
    (void)for_subsets;
    const uint32_t wg_per_cu = blk == 256 ? 2u : 1u;
    const uint32_t groups = (c->I + G - 1) / G;
    const uint32_t total = c->t_grid_cap > 0 ? (uint32_t)c->t_grid_cap : wg_per_cu * (uint32_t)c->n_cu;
    const uint32_t gxi = std::max<uint32_t>(1, total / groups);
    *per = round_up((c->V + gxi - 1) / gxi, 64);
    *runs = (c->V + *per - 1) / *per;
    if (*runs > 0xffffu) {                  // the crowd kernel takes its launch shape as two 16-bit fields of one preloaded argument
        *per = round_up((c->V + 0xfffeu) / 0xffffu, 64);
        *runs = (c->V + *per - 1) / *per;
    }
}

// The shape the caller ASKS for (inst_loop / inst_block / grid_cap or the defaults), before LDS limits the group size; when
// bone subsets are allowed it is the shape of the SUBSET form (the whole-palette fallback sizes its own grid).
bool inst_shape(const rz_ctx *c, InstShape *s)
{
    const bool epilogues = c->edge != nullptr || c->aabb_on;   // only the generic kernel carries the fused consumers
    if (!(c->morph_mode == 0 && c->I > 1 && c->t_instloop != 0 && c->t_instloop != 9 && !epilogues) || c->B == 0 || c->V == 0) return false;
    s->want_in_kernel = c->t_fast != 0 && !c->pose_local;
    // workgroup size. Whole palettes: 512 threads for the one-launch frame (one workgroup of 8 waves per CU shares the 102 KB
    // group), 256 behind rz_prep_kernel / rz_fk_kernel (two workgroups of 80 KB each: measured best in round 2). Bone subsets
    // (30 KB): 512 threads in both forms — with finished rows staged the 512-thread kernel runs C4 in 32.1 us against 34.5 us
    // for 256 threads (tools/c4_subsets.py, fast = 0 rows of profiles/r3_c4_subsets.txt).
    const bool forced = c->t_instblock == 256 || c->t_instblock == 512 || c->t_instblock == 1024;
    s->blk_full = forced ? c->t_instblock : (s->want_in_kernel ? 512 : 256);
    s->blk = forced ? c->t_instblock : (c->t_subsets != 0 ? 512 : s->blk_full);
    s->G = (int)std::min<uint32_t>(c->t_instloop > 0 ? (uint32_t)c->t_instloop : 8u, c->I);
    if (s->G < 2) return false;
    inst_runs(c, s->G, s->blk, c->t_subsets != 0, &s->per, &s->runs);
    return true;
}

Plan make_plan(const rz_ctx *c)
{
    Plan pl;
    memset(&pl, 0, sizeof pl);          // padding too: frame_signature() hashes the struct
    RzVariant &v = pl.v;
    v.mode = c->morph_mode;
    // Without dense targets S only sets the size of a wave step: S = 4 makes it 64 vertices instead of 256, so a small
    // mesh (one 30 k-vertex character is 118 wave steps at S = 1) reaches four times as many CUs, and a region where
    // every vertex carries dozens of sparse entries (a face) spreads over four times as many waves. 1 or 4 there.
    v.S = (v.mode == 1) ? (c->t_split > 0 ? c->t_split : auto_split(c))
                        : (c->t_split > 0 ? (c->t_split >= 4 ? 4 : 1)
                                          : ((uint64_t)c->V * c->I <= (v.mode == 2 ? (256u << 10) : (64u << 10)) ? 4 : 1));   // measured: 30 k verts 5.5 -> 4.5 us, 126 k 6.0 -> 6.6
    v.U = c->t_unroll > 0 ? c->t_unroll : 8;    // 24 loads in flight per lane: best or tied at every size measured
    v.nt = c->t_nt != 0;
    // streaming stores pay once the frame's output no longer fits the L2s (measured: 1 M verts yes, 126 k no)
    // and the morph stream is flowing too; a morph-free instanced frame (184 MB of output, MALL-absorbed) is faster
    // with plain stores: 6.9 vs 5.7 TB/s in tools/membench
    v.nts = c->t_nts < 0 ? (v.mode == 1 && (uint64_t)c->V * c->I * 24 >= (16u << 20)) : c->t_nts != 0;
    v.geo = c->t_geo != 0;
    // one-launch frame: single instance, and (dense) the active list fits the kernel arguments
    // Device-animated single character: the hierarchy solve (and the motion sampling) runs as the prologue of every
    // workgroup of the deform kernel — one launch per frame, no rz_fk_kernel / rz_prep_kernel in front of it. Measured
    // (tools/fk_fuse_bench.py, 30 k vertices / 200 bones): sampled poses 11.5-17.3 -> 9.2-13.5 us per frame in every morph
    // mode (and 27.7 -> 24.5 us on a 1/8 shard of C5); local poses 12.7-14.4 -> 9.8-12.8 us without dense morphs, no gain
    // with them (there the three-kernel frame keeps its kernel-argument morph list and streams from its first instruction).
    // Automatic mode follows that; "fuse_fk" = 0 / 1 forces it.
    pl.fuse_fk = c->I == 1 && c->pose_local && c->has_topology && (size_t)c->B * 48 + rz_fk_scratch_bytes((int)c->B) + (size_t)c->M * 12 + 4096 <= 160 * 1024 &&
                 (c->t_fusefk == 1 || (c->t_fusefk < 0 && (c->pose_sampled || c->morph_mode != 1)));
    const bool can_fast = c->I == 1 && !pl.fuse_fk && (v.mode != 1 || c->ml.count >= 0);
    v.fast = can_fast && c->t_fast != 0;
    pl.dma = false;
    pl.inst_group = 0;
    pl.verts_per_wg = 0;
    pl.poses_per_wg = 0;
    pl.out_cap = 0;
    pl.n_quads = (c->V + 3) / 4;
    pl.quads_per_wave = 8;
    pl.grid_x = 1;
    pl.prep = !v.fast && !pl.fuse_fk;
    // persistent, balanced grid: `cap` workgroups in total, every wave owns an equal contiguous run of quads
    const uint32_t waves_per_wg = 4, qpw_step = 64 / (uint32_t)v.S;
    // measured (profiles/r1_*sweep*): 2 workgroups per CU for one big mesh, 8 per instance when instanced
    uint32_t cap = c->t_grid_cap > 0 ? (uint32_t)c->t_grid_cap : std::max(2u * (uint32_t)c->n_cu, 8u * c->I);
    // Pose prefetch: the first frame of a zero-copy world pose carries one helper workgroup that stages the NEXT pose (if the
    // host has written it already) — it takes one of the grid's slots, the workers share the mesh among cap - 1.
    pl.pf = c->I == 1 && c->t_prefetch != 0 && c->zc_cur >= 0 && c->zc_seq_cur != 0 && c->zc_tag &&
            ((v.fast && !c->zc_local && !c->world_resident) ||
             (pl.fuse_fk && c->zc_local && !c->pose_sampled && (!c->local_resident || !c->mw_resident)));
    if (pl.pf && cap > 1) cap -= 1;
    uint32_t gx = std::max<uint32_t>(1, cap / std::max<uint32_t>(1, c->I));
    const uint32_t max_useful = (pl.n_quads + waves_per_wg * qpw_step - 1) / (waves_per_wg * qpw_step);
    gx = std::max<uint32_t>(1, std::min(gx, max_useful));
    uint32_t per_wave = (pl.n_quads + gx * waves_per_wg - 1) / (gx * waves_per_wg);
    per_wave = std::max<uint32_t>(8, round_up(per_wave, 8));
    pl.quads_per_wave = per_wave;
    pl.grid_x = std::max<uint32_t>(1, (pl.n_quads + per_wave * waves_per_wg - 1) / (per_wave * waves_per_wg));
    // LDS write batching. Measured (tools/ablate_c5.py): parking a wave's WHOLE run and writing it once at the end
    // takes C5 from 127.6 to 124.5 us (the stores leave the read stream alone until the kernel's tail); flushing every
    // step gains nothing. So automatic mode turns it on exactly when the run fits the buffer (<= 640 vertices per wave).
    {
        const uint32_t step = 256u / (uint32_t)v.S;
        const uint32_t run = round_up(pl.quads_per_wave * 4, 64);
        if (c->t_outcap > 0) pl.out_cap = std::max(std::min<uint32_t>(round_up((uint32_t)c->t_outcap, 64), 640), step);
        else if (c->t_outcap < 0 && run <= 640) pl.out_cap = std::max(run, step);
        // a skeleton near the LDS limit (3 242 bones = 152 KB of palette) leaves no room for the write-batching buffer: do without
        if (pl.out_cap) {
            RzDeformParams q;
            memset(&q, 0, sizeof q);
            q.B = (int)c->B; q.M = (int)c->M; q.Mpad = (int)c->Mpad; q.out_cap = pl.out_cap; q.fk_on = pl.fuse_fk ? 1 : 0;
            if (rz_deform_lds_bytes(q, v) > 160 * 1024) pl.out_cap = 0;
        }
    }
    // Sparse targets: every wave stages its step's piece of the CSR in LDS (deform_kernels.hip, MODE 2). The buffer takes what the
    // launch leaves of the LDS — a frame of at most one workgroup per CU (a single character: the demo model is 113 workgroups)
    // has the CU's 160 KB to itself, larger frames plan for two workgroups per CU — up to a whole step of M-entry rows or 2 048
    // entries (32 KB per wave); longer ranges go through it in pieces.
    if (v.mode == 2) {
        RzDeformParams q;
        memset(&q, 0, sizeof q);
        q.B = (int)c->B; q.M = (int)c->M; q.Mpad = (int)c->Mpad; q.out_cap = pl.out_cap; q.fk_on = pl.fuse_fk ? 1 : 0;
        const size_t base = rz_deform_lds_bytes(q, v);
        const size_t budget = ((uint64_t)pl.grid_x * c->I <= (uint64_t)c->n_cu ? 160u : 80u) * 1024u;
        const size_t avail = budget > base + 4096 ? (budget - base - 1024) / (waves_per_wg * 16) : 64;
        const uint32_t want = std::min<uint32_t>(2048u, (256u / (uint32_t)v.S) * std::max<uint32_t>(c->M, 1u));
        pl.sp_cap = std::max<uint32_t>(64u, std::min<uint32_t>(round_up(want, 256), (uint32_t)(avail / 64 * 64)));         // whole 1 KiB bursts (the kernel issues them in groups of four)
    }
    // instanced, morph-free frames: G poses per workgroup share one decode of each vertex, their palettes live in LDS.
    // Where the palettes come from (measured on C4, tools/ablate_c4.py, frame = everything a frame launches):
    //   prep kernel + 16-byte LDS-DMA of finished palettes (default, and always behind the on-device FK, which writes
    //       the palettes itself): 38.7-39.5 us (kernel 34 + 2.7 us prep + launch boundary);
    //   in-kernel (fast = 1): the skin kernel stages the group's world matrices (64-byte slots, same LDS-DMA) and
    //       multiplies by the inverse bind matrices in place — one launch per frame, 39.6-40 us: the staging costs
    //       what the extra launch did, so it is opt-in.
    const bool epilogues = c->edge != nullptr || c->aabb_on;   // only the generic kernel carries the fused consumers
    InstShape is;
    if (inst_shape(c, &is)) {
        // Where the palettes come from. World-matrix poses (rz_set_pose): the skin kernel forms them itself — ONE launch per
        // frame (fast = 0 forces rz_prep_kernel in front). Device-solved poses: rz_fk_kernel has written the palettes already,
        // the skin kernel copies them in (48-byte rows, LDS-DMA).
        // What is staged: by default only the bones the workgroup's vertex run names (bone-subset form) — on C4
        // ~34 of 200 bones, 30 KB of LDS instead of 102 KB and a front of 1.3 us instead of 2.9 us per workgroup. It needs the
        // run lists of exactly this launch shape (ensure_run_subsets, called by every entry point that launches frames) and
        // is a gain only when the largest list is shorter than the skeleton; otherwise the whole palette is staged: 64-byte
        // slots re-packed in place for the one-launch frame, which wants B <= block threads and, at 8 poses x 200 bones =
        // 102 KB, one 512-thread workgroup per CU (measured 35.5 us on C4 against 39.0 us for prep kernel + launch boundary +
        // 256-thread skin kernel, and 39.5 us for two 256-thread workgroups of 6 poses).
        const int blk = is.blk;
        const uint32_t lds_budget = (blk == 256 ? 80u : 156u) * 1024u;
        bool sub = c->t_subsets != 0 && c->sub_valid && c->sub_per == is.per && c->sub_runs == is.runs && c->sub_B == c->B &&
                   c->sub_max < c->B && c->sub_max <= (uint32_t)blk;
        if (sub) {
            const size_t lds = rz_skin_instances_lds_bytes(is.G, c->sub_max, !is.want_in_kernel, true);
            if (lds > lds_budget) sub = false;
            else {
                pl.subsets = true; pl.sub_bones = c->sub_max; pl.inst_lds = lds;
                pl.inst_group = is.G; pl.inst_block = blk; pl.verts_per_wg = is.per; pl.grid_x = is.runs;
                pl.prep = !is.want_in_kernel; pl.dma = !is.want_in_kernel;
            }
        }
        if (!sub) {
            const int blk_f = is.blk_full;
            const uint32_t budget_f = (blk_f == 256 ? 80u : 156u) * 1024u;
            const bool in_kernel = is.want_in_kernel && c->B <= (uint32_t)blk_f;   // the in-place product gives every bone its own thread
            const uint32_t g_lds = budget_f / (c->B * (in_kernel ? 64u : 48u));
            int G = (int)std::min<uint32_t>((uint32_t)is.G, g_lds);
            if (G >= 2) {
                uint32_t per = 0, runs = 0;
                inst_runs(c, G, blk_f, false, &per, &runs);
                pl.inst_group = G; pl.inst_block = blk_f; pl.verts_per_wg = per; pl.grid_x = runs;
                pl.prep = !in_kernel; pl.dma = !in_kernel;
                pl.inst_lds = rz_skin_instances_lds_bytes(G, c->B, !in_kernel, false);
            }
        }
    }
    // register-resident instanced form: 2048-vertex runs, pose ranges sized for ~2 WGs per CU
    // (measured slower than the LDS pose-group form on C4 — 37 vs 34 us — so it is opt-in: inst_loop = 9)
    if (v.mode == 0 && c->I > 1 && c->t_instloop == 9 && c->B * 3 <= 65535u && !epilogues) {
        const uint32_t runs = (c->V + 2047) / 2048;
        uint32_t total = c->t_grid_cap > 0 ? (uint32_t)c->t_grid_cap : 2u * (uint32_t)c->n_cu;
        uint32_t ranges = std::max<uint32_t>(1, std::min<uint32_t>(c->I, total / std::max<uint32_t>(1, runs)));
        pl.poses_per_wg = (int)((c->I + ranges - 1) / ranges);
        pl.inst_group = 0;
        pl.grid_x = runs;
        pl.prep = true;
        pl.dma = true;
    }
    return pl;
}

// Bring the run lists of the bone-subset crowd frame in line with the launch shape the next frame asks for. One small kernel
// and one readback of `runs` counters, only when the shape, the mesh or the skeleton changed — never per frame.
int ensure_run_subsets(rz_ctx *c)
{
    InstShape is;
    if (c->t_subsets == 0 || !inst_shape(c, &is)) return RZ_OK;
    if (c->sub_valid && c->sub_per == is.per && c->sub_runs == is.runs && c->sub_B == c->B) return RZ_OK;
    c->sub_valid = false;
    HIP_TRY(hipStreamSynchronize(c->stream));       // a frame in flight may still read the old lists
    drop_graph(c);
    if (!c->rj01) HIP_TRY(hipMalloc(&c->rj01, (size_t)c->Vp * 4));
    if (!c->rj23) HIP_TRY(hipMalloc(&c->rj23, (size_t)c->Vp * 4));
    const size_t need_list = (size_t)is.runs * c->B;
    if (need_list > c->sub_list_alloc) {
        dfree(c->sub_list);
        HIP_TRY(hipMalloc(&c->sub_list, need_list * sizeof(uint16_t)));
        c->sub_list_alloc = need_list;
    }
    if (is.runs > c->sub_count_alloc) {
        dfree(c->sub_count);
        HIP_TRY(hipMalloc(&c->sub_count, (size_t)is.runs * sizeof(uint32_t)));
        c->sub_count_alloc = is.runs;
    }
    const uint32_t v_lim = (c->V + 3) / 4 * 4;      // what the skin kernel walks: whole quads
    HIP_TRY(hipMemsetAsync(c->rj01, 0, (size_t)c->Vp * 4, c->stream));
    HIP_TRY(hipMemsetAsync(c->rj23, 0, (size_t)c->Vp * 4, c->stream));
    HIP_TRY(rz_launch_run_subsets(c->j01, c->j23, v_lim, is.per, is.runs, c->B, c->sub_list, c->sub_count, c->rj01, c->rj23, c->stream));
    std::vector<uint32_t> counts(is.runs);
    HIP_TRY(hipMemcpyAsync(counts.data(), c->sub_count, (size_t)is.runs * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    uint32_t mx = 0;
    for (uint32_t n : counts) mx = std::max(mx, n);
    c->sub_per = is.per; c->sub_runs = is.runs; c->sub_B = c->B; c->sub_max = mx;
    c->sub_valid = true;
    return RZ_OK;
}

// Plan of the next frame: the run lists first (the plan only takes the subset form when they match its shape).
int frame_plan(rz_ctx *c, Plan *pl)
{
    if (int r = ensure_run_subsets(c)) return r;
    *pl = make_plan(c);
    return RZ_OK;
}

RzPrepParams prep_params(const rz_ctx *c)
{
    RzPrepParams p;
    memset(&p, 0, sizeof p);
    p.world = c->world; p.inv_bind = c->inv_bind; p.palette = c->palette; p.morph_w = c->morph_w;
    p.act_idx = c->act_idx; p.act_w = c->act_w; p.act_count = c->act_count;
    p.B = (int)c->B; p.M = (int)c->M; p.Mpad = (int)c->Mpad;
    return p;
}

int check_ready(rz_ctx *c)
{
    if (c->V == 0 || !c->geom) return fail(RZ_ERR_INVALID, "no mesh uploaded (rz_upload_mesh)");
    if (c->B == 0 || !c->inv_bind) return fail(RZ_ERR_INVALID, "no skeleton uploaded (rz_upload_skeleton)");
    if (!c->pose_set) return fail(RZ_ERR_INVALID, "no pose set (rz_set_pose)");
    return RZ_OK;
}

RzFkParams fk_params(const rz_ctx *c)
{
    RzFkParams p;
    memset(&p, 0, sizeof p);            // padding too: frame_signature() hashes the struct
    p.local_q = src_local_q(c);
    p.local_t = c->pose_local_t ? reinterpret_cast<const float *>(p.local_q + (size_t)c->I * c->B) : nullptr;
    p.bone_rec = c->fk_rec; p.inv_bind = c->inv_bind;
    p.world = c->world; p.palette = c->palette; p.B = (int)c->B; p.n_levels = c->fk_levels;
    if (c->ovr_count) { p.ovr_off = c->ovr_off; p.ovr_bone = c->ovr_bone; p.ovr_world = c->ovr_world; }
    if (c->bm_count && c->M) {
        p.bm_off = c->bm_off; p.bm_morph = c->bm_morph; p.bm_rot = c->bm_rot; p.bm_tr = c->bm_tr;
        p.bm_w = src_morph_w(c); p.bm_M = (int)c->M;
    }
    if (c->pose_sampled) {
        RzSampleParams &q = p.sample;
        q.frames = c->frames_inline ? nullptr : c->an_frames; q.frames_inline = c->frames_inline ? 1 : 0; q.frame0 = c->frame0;
        q.bone_range = c->an_bone_range; q.key_frame = c->an_key_frame;
        q.key_rot = c->an_key_rot; q.key_pos = c->an_key_pos; q.key_interp = c->an_key_interp;
        q.mkey_frame = c->an_mkey_frame; q.mkey_weight = c->an_mkey_weight;
        q.feed_off = c->an_feed_off; q.feed_range = c->an_feed_range; q.feed_ratio = c->an_feed_ratio;
        q.morph_w = c->morph_w; q.M = (int)c->M;
    }
    return p;
}

int launch_fk(rz_ctx *c, hipStream_t st)
{
    HIP_TRY(rz_launch_fk(fk_params(c), c->I, st));
    c->palette_stale = false;
    return RZ_OK;
}

int launch_prep(rz_ctx *c, hipStream_t st)
{
    HIP_TRY(rz_launch_prep(prep_params(c), c->I, st));
    c->palette_stale = false;
    return RZ_OK;
}

// Everything a frame launches in front of the deform kernel: on-device FK (local-rotation poses) and/or the prep
// kernel. The FK kernel already writes the palette, so prep is only still needed for its morph compaction.
int launch_front(rz_ctx *c, const Plan &pl, hipStream_t st)
{
    if (pl.fuse_fk) return RZ_OK;       // the deform kernel solves the hierarchy itself
    if (c->pose_local) {
        if (int r = launch_fk(c, st)) return r;
        if (pl.prep && c->morph_mode == 1)
            if (int r = launch_prep(c, st)) return r;
        return RZ_OK;
    }
    if (pl.prep) return launch_prep(c, st);
    return RZ_OK;
}

int launch_deform(rz_ctx *c, const Plan &pl)
{
    RzDeformParams p = deform_params(c, pl);
    if (pl.poses_per_wg > 0) {
        HIP_TRY(rz_launch_skin_instances_reg(p, (int)c->I, pl.poses_per_wg, pl.grid_x, pl.v.nts, c->stream));
        return RZ_OK;
    }
    if (pl.inst_group > 0) {
        HIP_TRY(rz_launch_skin_instances(p, pl.inst_group, (int)c->I, pl.verts_per_wg, pl.grid_x, pl.inst_block, pl.v.nts, (size_t)pl.inst_lds, c->stream));
        if (!pl.dma) c->palette_stale = pl.subsets;    // the whole-palette one-launch frame copies its palettes out, the subset form cannot
        return RZ_OK;
    }
    size_t lds = rz_deform_lds_bytes(p, pl.v);
    if (lds > 160 * 1024) return fail(RZ_ERR_UNSUPPORTED, "skeleton too large for the LDS palette (%zu B)", lds);
    HIP_TRY(rz_launch_deform(p, c->ml, pl.v, pl.grid_x + (p.pf_src ? 1u : 0u), c->I, c->stream));
    c->palette_stale = false;                          // this frame's palette is in memory: written by its front kernels or by workgroup 0
    if (p.world_copy) c->world_resident = true;        // workgroup 0 of that launch left the pose in the device block
    if (p.morph_w_copy) c->mw_resident = true;
    if (p.fk.copy_q) c->local_resident = c->mw_resident = true;
    if (c->aabb_on) c->aabb_slot ^= 1;     // this launch re-armed the other slot for the next frame
    return RZ_OK;
}

// Crowds overlap the front kernels of a frame with the skin kernel of the frame before it. The protocol
// needs a frame that HAS front kernels and a skin kernel that reads nothing of the pose slots themselves (sparse morph
// frames read the uploaded weights directly), and plain stream capture (the graph key) stays single-stream.
bool want_overlap(const rz_ctx *c, const Plan &pl)
{
    // OPT-IN (overlap = 1): measured on MI355X / ROCm 7.2 the two cross-stream hand-offs per frame cost more than the front
    // kernels they hide — C4 39.0 -> 44.8 us with rz_prep_kernel in front, 43.4 -> 62.1 us with rz_fk_kernel
    return c->t_overlap == 1 && c->I > 1 && c->morph_mode != 2 && !c->t_graph && (pl.prep || c->pose_local);
}

// Switching protocols is rare (instance count, tuning keys): drain both streams so that nothing enqueued under the old
// rules is still running when the first frame under the new ones starts.
int set_overlap(rz_ctx *c, bool on)
{
    if (c->overlap_on == on) return RZ_OK;
    HIP_TRY(hipStreamSynchronize(c->up_stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->overlap_on = on;
    c->skin_recorded[0] = c->skin_recorded[1] = false;
    c->free_recorded[0] = c->free_recorded[1] = false;
    return RZ_OK;
}

// One whole frame: front kernels (if the plan has any) + the deform / skin kernel.
int run_frame(rz_ctx *c, const Plan &pl)
{
    // rz_prep_kernel (and a skin kernel that is not the one-launch form) reads a device-resident pose
    if (c->zc_cur >= 0 && (pl.prep || (!pl.v.fast && !c->pose_local)))
        if (int r = make_resident(c)) return r;
    if (c->overlap_on) {
        const int s = c->ring_slot ^ 1;                       // the slot the skin kernel of two frames ago read
        if (c->skin_recorded[s]) HIP_TRY(hipStreamWaitEvent(c->up_stream, c->ev_skin[s], 0));
        set_ring(c, s);
        if (int r = launch_front(c, pl, c->up_stream)) return r;
        HIP_TRY(hipEventRecord(c->ev_front[s], c->up_stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_front[s], 0));
        if (int r = launch_deform(c, pl)) return r;
        HIP_TRY(hipEventRecord(c->ev_skin[s], c->stream));
        c->skin_recorded[s] = true;
        return RZ_OK;
    }
    if (int r = launch_front(c, pl, c->stream)) return r;
    return launch_deform(c, pl);
}

// the stream per-frame inputs travel on and front kernels run on
hipStream_t front_stream(const rz_ctx *c) { return c->overlap_on ? c->up_stream : c->stream; }

uint64_t algorithmic_bytes(const rz_ctx *c)
{
    // 36 B read + 24 B written per vertex, 12*M B of dense morph targets per vertex,
    // world + inverse-bind matrices, morph weights. The static mesh is counted once for instances.
    const uint64_t V = c->V, I = c->I, B = c->B, M = c->M;
    uint64_t bytes = V * 36 + I * (V * 24 + B * 64) + B * 64;
    if (c->morph_mode == 1) bytes += I * (V * 12 * M + M * 4);
    if (c->morph_mode == 2) bytes += V * 4 + I * (c->sp_count * 16 + M * 4);
    return bytes;
}

int upload_skinning(rz_ctx *c, uint32_t V, const uint16_t *joints4, const uint8_t *weights4)
{
    Scratch<uint16_t> dj;
    Scratch<uint8_t> dw;
    HIP_TRY(dj.alloc((size_t)V * 4));
    HIP_TRY(dw.alloc((size_t)V * 4));
    HIP_TRY(hipMemcpy(dj.p, joints4, (size_t)V * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dw.p, weights4, (size_t)V * 4, hipMemcpyHostToDevice));
    HIP_TRY(rz_launch_pack_skinning(dj.p, dw.p, V, c->j01, c->j23, c->wq, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
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

int alloc_mesh(rz_ctx *c, uint32_t V)
{
    HIP_TRY(hipStreamSynchronize(c->stream));
    drop_direct_gather(c);                // shard sizes are about to change: back to private output buffers
    dfree(c->geom); dfree(c->j01); dfree(c->j23); dfree(c->wq); dfree(c->edge);
    dfree(c->rj01); dfree(c->rj23); c->sub_valid = false;      // the run lists name this mesh's joints
    free_morphs(c);                       // morph targets are per-vertex: a new mesh invalidates them
    c->V = V;
    c->Vp = round_up(V, kVertPad);
    const size_t Vp = c->Vp;
    HIP_TRY(hipMalloc(&c->geom, 6 * Vp * sizeof(float)));
    HIP_TRY(hipMalloc(&c->j01, Vp * 4));
    HIP_TRY(hipMalloc(&c->j23, Vp * 4));
    HIP_TRY(hipMalloc(&c->wq, Vp * 4));
    // padding vertices: zero position/normal, joint 0, weights 0 (takes the (1,0,0,0) branch)
    HIP_TRY(hipMemsetAsync(c->geom, 0, 6 * Vp * sizeof(float), c->stream));
    HIP_TRY(hipMemsetAsync(c->j01, 0, Vp * 4, c->stream));
    HIP_TRY(hipMemsetAsync(c->j23, 0, Vp * 4, c->stream));
    HIP_TRY(hipMemsetAsync(c->wq, 0, Vp * 4, c->stream));
    return RZ_OK;
}

}  // namespace

extern "C" {

const char *rz_last_error(void) { return g_err.c_str(); }
int rz_abi_version(void) { return RZ_ABI_VERSION; }

int rz_device_count(int *count)
{
    if (!count) return fail(RZ_ERR_INVALID, "null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(RZ_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return RZ_OK;
}

int rz_create(int device, rz_ctx **out)
{
    if (!out) return fail(RZ_ERR_INVALID, "null out");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(RZ_ERR_NO_DEVICE, "no HIP device: %s", e == hipSuccess ? "count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(RZ_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RZ_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 (MI355X) code only", device,
                    prop.gcnArchName);
    rz_ctx *c = new rz_ctx();
    memset(&c->ml, 0, sizeof c->ml);
    c->device = device;
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (se == hipSuccess) se = hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking);
    for (int k = 0; k < 2 && se == hipSuccess; ++k) {
        se = hipEventCreateWithFlags(&c->ev_up[k], hipEventDisableTiming);
        if (se == hipSuccess) se = hipEventCreateWithFlags(&c->ev_free[k], hipEventDisableTiming);
        if (se == hipSuccess) se = hipEventCreateWithFlags(&c->ev_front[k], hipEventDisableTiming);
        if (se == hipSuccess) se = hipEventCreateWithFlags(&c->ev_skin[k], hipEventDisableTiming);
    }
    if (se == hipSuccess) se = hipEventCreate(&c->ev0);
    if (se == hipSuccess) se = hipEventCreate(&c->ev1);
    for (int i = 0; i < kStageSlots && se == hipSuccess; ++i)
        se = hipEventCreateWithFlags(&c->stage_ev[i], hipEventDisableTiming);
    if (se != hipSuccess) {
        rz_destroy(c);
        return fail(RZ_ERR_HIP, "context setup failed: %s", hipGetErrorString(se));
    }
    *out = c;
    return RZ_OK;
}

int rz_destroy(rz_ctx *c)
{
    if (!c) return RZ_OK;
    if (c->n_forks) return fail(RZ_ERR_INVALID, "%d fork(s) still borrow this context's static data: destroy them first", c->n_forks);
    (void)hipSetDevice(c->device);
#ifdef RZ_ALL_VARIANTS
    if (c->gate_host) *reinterpret_cast<volatile uint32_t *>(c->gate_host) = 1u;      // a test died with the gate closed: open it before draining the stream
#endif
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->up_stream) (void)hipStreamSynchronize(c->up_stream);
    if (c->lender) {                      // a fork frees nothing it borrowed
        c->geom = nullptr; c->j01 = c->j23 = c->wq = nullptr; c->inv_bind = nullptr;
        c->fk_rec = nullptr;
        c->an_bone_range = c->an_feed_range = nullptr; c->an_feed_off = nullptr;
        c->an_key_frame = c->an_key_pos = c->an_mkey_frame = c->an_mkey_weight = c->an_feed_ratio = nullptr; c->an_key_rot = nullptr; c->an_key_interp = nullptr;
        c->bm_off = c->bm_morph = nullptr; c->bm_rot = c->bm_tr = nullptr;
        c->dense = nullptr; c->sp_ptr = nullptr; c->sp_entries = nullptr; c->edge = nullptr;
        c->lender->n_forks--;
        c->lender = nullptr;
    }
    drop_direct_gather(c);
    drop_graph(c);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    dfree(c->geom); dfree(c->j01); dfree(c->j23); dfree(c->wq); dfree(c->inv_bind);
    dfree(c->rj01); dfree(c->rj23); dfree(c->sub_list); dfree(c->sub_count); dfree(c->zc_tag);
    dfree(c->fk_rec);
    free_animation(c); dfree(c->an_frames);
    dfree(c->pose_blk[0]); dfree(c->pose_blk[1]);
    dfree(c->ovr_off); dfree(c->ovr_bone); dfree(c->ovr_world);
    free_bone_morphs(c);
    free_morphs(c);
    for (int k = 0; k < 2; ++k) {
        if (c->ev_up[k]) (void)hipEventDestroy(c->ev_up[k]);
        if (c->ev_free[k]) (void)hipEventDestroy(c->ev_free[k]);
    }
    for (int k = 0; k < 2; ++k) {
        dfree(c->palette_ring[k]); dfree(c->act_idx_ring[k]); dfree(c->act_w_ring[k]); dfree(c->act_count_ring[k]);
        if (c->ev_front[k]) (void)hipEventDestroy(c->ev_front[k]);
        if (c->ev_skin[k]) (void)hipEventDestroy(c->ev_skin[k]);
    }
    dfree(c->out_pos); dfree(c->out_nrm); dfree(c->g_pos); dfree(c->g_nrm);
#ifdef RZ_ABLATE
    dfree(c->tl);
#endif
#ifdef RZ_ALL_VARIANTS
    if (c->gate_host) { (void)hipHostFree(c->gate_host); c->gate_host = nullptr; }
#endif
    dfree(c->edge); dfree(c->out_hull); dfree(c->aabb);
    for (int i = 0; i < kStageSlots; ++i) {
        if (c->stage[i]) (void)hipHostFree(c->stage[i]);
        if (c->stage_ev[i]) (void)hipEventDestroy(c->stage_ev[i]);
    }
    for (int i = 0; i < rz_ctx::kZcSlots; ++i)
        if (c->zc_host[i]) (void)hipHostFree(c->zc_host[i]);
    for (int e = 0; e < 2; ++e)
        if (c->zc_ev[e]) (void)hipEventDestroy(c->zc_ev[e]);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return RZ_OK;
}

int rz_fork(rz_ctx *parent, rz_ctx **out)
{
    if (!out) return fail(RZ_ERR_INVALID, "null out");
    *out = nullptr;
    if (int r = use(parent)) return r;
    if (parent->lender) return fail(RZ_ERR_INVALID, "rz_fork of a fork: fork the context that owns the static data");
    if (parent->V == 0 || !parent->geom || parent->B == 0 || !parent->inv_bind) return fail(RZ_ERR_INVALID, "rz_fork needs a mesh and a skeleton (rz_upload_mesh, rz_upload_skeleton)");
    if (parent->comm || parent->gather_root) return fail(RZ_ERR_UNSUPPORTED, "rz_fork of a context that takes part in a gather");
    HIP_TRY(hipStreamSynchronize(parent->stream));        // every static upload of the lender has landed (before anything is created: nothing to undo on failure)
    rz_ctx *c = nullptr;
    if (int r = rz_create(parent->device, &c)) return r;
    c->V = parent->V; c->Vp = parent->Vp; c->geom = parent->geom; c->j01 = parent->j01; c->j23 = parent->j23; c->wq = parent->wq;
    c->B = parent->B; c->inv_bind = parent->inv_bind;
    c->has_topology = parent->has_topology; c->fk_rec = parent->fk_rec; c->fk_levels = parent->fk_levels;
    c->has_animation = parent->has_animation; c->an_bone_range = parent->an_bone_range; c->an_feed_range = parent->an_feed_range;
    c->an_feed_off = parent->an_feed_off; c->an_key_frame = parent->an_key_frame; c->an_key_pos = parent->an_key_pos;
    c->an_mkey_frame = parent->an_mkey_frame; c->an_mkey_weight = parent->an_mkey_weight; c->an_feed_ratio = parent->an_feed_ratio;
    c->an_key_rot = parent->an_key_rot; c->an_key_interp = parent->an_key_interp; c->an_M = parent->an_M;
    c->bm_off = parent->bm_off; c->bm_morph = parent->bm_morph; c->bm_rot = parent->bm_rot; c->bm_tr = parent->bm_tr; c->bm_count = parent->bm_count;
    c->morph_mode = parent->morph_mode; c->M = parent->M; c->Mpad = parent->Mpad; c->dense = parent->dense;
    c->sp_ptr = parent->sp_ptr; c->sp_entries = parent->sp_entries; c->sp_count = parent->sp_count;
    c->edge = parent->edge; c->aabb_on = parent->aabb_on; c->aabb_rearm = parent->aabb_on;
    c->I = parent->I;
    c->t_split = parent->t_split; c->t_unroll = parent->t_unroll; c->t_grid_cap = parent->t_grid_cap; c->t_nt = parent->t_nt; c->t_nts = parent->t_nts;
    c->t_geo = parent->t_geo; c->t_fast = parent->t_fast; c->t_instloop = parent->t_instloop; c->t_outcap = parent->t_outcap; c->t_instblock = parent->t_instblock;
    c->t_instorder = parent->t_instorder; c->t_overlap = parent->t_overlap; c->t_zerocopy = parent->t_zerocopy; c->t_fusefk = parent->t_fusefk;
    c->t_graph = parent->t_graph; c->tuned_by_search = parent->tuned_by_search; c->t_subsets = parent->t_subsets; c->t_prefetch = parent->t_prefetch;
    c->lender = parent;
    parent->n_forks++;
    int rc = ensure_pose_buffers(c);
    if (rc == RZ_OK) rc = ensure_outputs(c);
    if (rc != RZ_OK) { rz_destroy(c); return rc; }
    *out = c;
    return RZ_OK;
}

int rz_shard_range(uint32_t v_total, int nranks, int rank, uint32_t *begin, uint32_t *count)
{
    if (nranks < 1 || rank < 0 || rank >= nranks || !begin || !count)
        return fail(RZ_ERR_INVALID, "bad shard query (nranks=%d rank=%d)", nranks, rank);
    const uint64_t per = ((uint64_t)v_total + nranks - 1) / nranks;
    const uint64_t chunk = (per + kShardGrain - 1) / kShardGrain * kShardGrain;
    uint64_t b = std::min<uint64_t>(v_total, chunk * (uint64_t)rank);
    uint64_t n = std::min<uint64_t>(chunk, v_total - b);
    *begin = (uint32_t)b;
    *count = (uint32_t)n;
    return RZ_OK;
}

int rz_upload_mesh(rz_ctx *c, uint32_t V, const float *interleaved8, const uint16_t *joints4, const uint8_t *weights4)
{
    if (int r = use(c)) return r;
    if (int r = static_unlocked(c, "rz_upload_mesh")) return r;
    if (V == 0 || !interleaved8 || !joints4 || !weights4) return fail(RZ_ERR_INVALID, "rz_upload_mesh: empty mesh or null array");
    if (int r = alloc_mesh(c, V)) return r;
    Scratch<float> scratch;
    HIP_TRY(scratch.alloc((size_t)V * 8));
    float *tmp = scratch.p;
    HIP_TRY(hipMemcpy(tmp, interleaved8, (size_t)V * 8 * sizeof(float), hipMemcpyHostToDevice));
    const size_t Vp = c->Vp;
    HIP_TRY(rz_launch_deinterleave(tmp, 8, 0, V, c->geom, c->geom + Vp, c->geom + 2 * Vp, c->stream));
    HIP_TRY(rz_launch_deinterleave(tmp, 8, 3, V, c->geom + 3 * Vp, c->geom + 4 * Vp, c->geom + 5 * Vp, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (int r = upload_skinning(c, V, joints4, weights4)) return r;
    return ensure_outputs(c);
}

int rz_upload_mesh_soa(rz_ctx *c, uint32_t V, const float *pos3, const float *nrm3, const uint16_t *joints4,
                       const uint8_t *weights4)
{
    if (int r = use(c)) return r;
    if (int r = static_unlocked(c, "rz_upload_mesh_soa")) return r;
    if (V == 0 || !pos3 || !nrm3 || !joints4 || !weights4) return fail(RZ_ERR_INVALID, "rz_upload_mesh_soa: empty mesh or null array");
    if (int r = alloc_mesh(c, V)) return r;
    Scratch<float> scratch;
    HIP_TRY(scratch.alloc((size_t)V * 3));
    float *tmp = scratch.p;
    const size_t Vp = c->Vp;
    HIP_TRY(hipMemcpy(tmp, pos3, (size_t)V * 3 * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(rz_launch_deinterleave(tmp, 3, 0, V, c->geom, c->geom + Vp, c->geom + 2 * Vp, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(tmp, nrm3, (size_t)V * 3 * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(rz_launch_deinterleave(tmp, 3, 0, V, c->geom + 3 * Vp, c->geom + 4 * Vp, c->geom + 5 * Vp, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (int r = upload_skinning(c, V, joints4, weights4)) return r;
    return ensure_outputs(c);
}

int rz_upload_skeleton(rz_ctx *c, uint32_t B, const float *inverse_bind16)
{
    if (int r = use(c)) return r;
    if (int r = static_unlocked(c, "rz_upload_skeleton")) return r;
    if (B == 0 || !inverse_bind16) return fail(RZ_ERR_INVALID, "rz_upload_skeleton: model has no bones");
    if ((size_t)B * 48 + 8192 > 160 * 1024) return fail(RZ_ERR_UNSUPPORTED, "more than %d bones do not fit the LDS palette", (160 * 1024 - 8192) / 48);
    HIP_TRY(hipStreamSynchronize(c->stream));
    drop_graph(c);
    c->ovr_count = 0;
    dfree(c->inv_bind);
    HIP_TRY(hipMalloc(&c->inv_bind, (size_t)B * 16 * sizeof(float)));
    HIP_TRY(hipMemcpy(c->inv_bind, inverse_bind16, (size_t)B * 16 * sizeof(float), hipMemcpyHostToDevice));
    c->B = B;
    c->zc_epoch++; c->zc_seq_cur = 0;   // a pose staged for the old skeleton must never match
    c->sub_valid = false;               // joints are clamped to the bone count when the run lists are built
    c->palette_stale = false;
    c->pose_set = false;
    c->has_topology = false;            // belongs to the previous skeleton
    free_bone_morphs(c);                // ... as do bone morphs (their entries name its bones)
    free_animation(c);                  // ... and so does an uploaded motion (its tracks name bones of that skeleton)
    return ensure_pose_buffers(c);
}

int rz_upload_morphs_dense(rz_ctx *c, uint32_t M, const float *deltas)
{
    if (int r = use(c)) return r;
    if (int r = static_unlocked(c, "rz_upload_morphs_dense")) return r;
    if (c->V == 0) return fail(RZ_ERR_INVALID, "upload the mesh before its morph targets");
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_morphs(c);
    if (M == 0) return ensure_pose_buffers(c);
    if (!deltas) return fail(RZ_ERR_INVALID, "null morph deltas");
    const size_t Vp = c->Vp, V = c->V;
    HIP_TRY(hipMalloc(&c->dense, (size_t)M * 3 * Vp * sizeof(float)));
    if (Vp != V) HIP_TRY(hipMemsetAsync(c->dense, 0, (size_t)M * 3 * Vp * sizeof(float), c->stream));
    // stream the host array through a bounded device staging buffer, re-laying each morph into planes
    const uint32_t batch = (uint32_t)std::max<size_t>(1, std::min<size_t>(M, (64u << 20) / (V * 12)));
    Scratch<float> scratch;
    HIP_TRY(scratch.alloc((size_t)batch * V * 3));
    float *tmp = scratch.p;
    for (uint32_t m0 = 0; m0 < M; m0 += batch) {
        const uint32_t nb = std::min(batch, M - m0);
        HIP_TRY(hipMemcpy(tmp, deltas + (size_t)m0 * V * 3, (size_t)nb * V * 3 * sizeof(float), hipMemcpyHostToDevice));
        for (uint32_t k = 0; k < nb; ++k) {
            float *pl = c->dense + (size_t)(m0 + k) * 3 * Vp;
            HIP_TRY(rz_launch_deinterleave(tmp + (size_t)k * V * 3, 3, 0, (uint32_t)V, pl, pl + Vp, pl + 2 * Vp, c->stream));
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    c->morph_mode = 1;
    c->M = M;
    c->Mpad = round_up(M + 8, 4);
    return ensure_pose_buffers(c);
}

int rz_upload_morphs_sparse(rz_ctx *c, uint32_t M, const uint32_t *morph_off, const uint32_t *vert_idx, const float *delta3)
{
    if (int r = use(c)) return r;
    if (int r = static_unlocked(c, "rz_upload_morphs_sparse")) return r;
    if (c->V == 0) return fail(RZ_ERR_INVALID, "upload the mesh before its morph targets");
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_morphs(c);
    if (M == 0) return ensure_pose_buffers(c);
    if (!morph_off) return fail(RZ_ERR_INVALID, "null morph offsets");
    const uint32_t E = morph_off[M];
    if (E > 0 && (!vert_idx || !delta3)) return fail(RZ_ERR_INVALID, "null morph entries");
    for (uint32_t m = 0; m < M; ++m)
        if (morph_off[m] > morph_off[m + 1]) return fail(RZ_ERR_INVALID, "morph offsets must be non-decreasing");
    // transpose morph-major (PMX file order) into a per-vertex CSR; a vertex's entries keep
    // ascending morph order (then file order), which is the oracle's accumulation order
    const size_t Vp = c->Vp;
    std::vector<uint32_t> ptr(Vp + 1, 0);
    for (uint32_t e = 0; e < E; ++e)
        if (vert_idx[e] < c->V) ptr[vert_idx[e] + 1]++;
    for (size_t v = 0; v < Vp; ++v) ptr[v + 1] += ptr[v];
    const uint32_t kept = ptr[Vp];
    std::vector<float4> ent(std::max<uint32_t>(kept, 1));
    std::vector<uint32_t> cur(ptr.begin(), ptr.end() - 1);
    for (uint32_t m = 0; m < M; ++m)
        for (uint32_t e = morph_off[m]; e < morph_off[m + 1]; ++e) {
            const uint32_t v = vert_idx[e];
            if (v >= c->V) continue;
            float4 x;
            x.x = delta3[(size_t)e * 3]; x.y = delta3[(size_t)e * 3 + 1]; x.z = delta3[(size_t)e * 3 + 2];
            memcpy(&x.w, &m, 4);
            ent[cur[v]++] = x;
        }
    HIP_TRY(hipMalloc(&c->sp_ptr, (Vp + 1) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&c->sp_entries, ent.size() * sizeof(float4)));
    HIP_TRY(hipMemcpy(c->sp_ptr, ptr.data(), (Vp + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->sp_entries, ent.data(), ent.size() * sizeof(float4), hipMemcpyHostToDevice));
    c->sp_count = kept;
    c->morph_mode = 2;
    c->M = M;
    c->Mpad = round_up(M + 8, 4);
    return ensure_pose_buffers(c);
}

int rz_set_instances(rz_ctx *c, uint32_t I)
{
    if (int r = use(c)) return r;
    if (I == 0 || I > 65535) return fail(RZ_ERR_INVALID, "instance count must be 1..65535");
    if (I > 1 && (c->comm || c->gather_root)) return fail(RZ_ERR_UNSUPPORTED, "instancing and vertex sharding are exclusive");
    if (I != c->I) {
        forget_search(c);
        drop_graph(c);
        c->aabb_rearm = true;
        c->ovr_count = 0;                 // overrides name (instance, bone) pairs of the old crowd
        // The host-compacted active-morph list is only maintained while I == 1 (upload_pose). Coming back to one instance
        // from a crowd it is stale (zeroed): let the prep kernel compact instance 0's weights, which are still on the device.
        if (c->M > 0 && c->morph_mode == 1) c->ml.count = -1;
        // A crowd larger than the one the resident pose was uploaded for has no pose for its new members (and a
        // single-character pose may still sit in its pinned slot, which holds exactly one instance): ask for a new one.
        if (I > c->pose_I) c->pose_set = false;
        c->zc_epoch++; c->zc_seq_cur = 0;
    }
    c->I = I;
    if (int r = ensure_pose_buffers(c)) return r;
    return ensure_outputs(c);
}

// Pinned staging ring for per-frame inputs: a slot is reused only after the copy that read it has completed.
static int stage_acquire(rz_ctx *c, size_t need, int *slot_out)
{
    if (need > c->stage_bytes) {
        HIP_TRY(hipStreamSynchronize(c->up_stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int i = 0; i < kStageSlots; ++i) {
            if (c->stage[i]) { (void)hipHostFree(c->stage[i]); c->stage[i] = nullptr; }
            HIP_TRY(hipHostMalloc(&c->stage[i], need, hipHostMallocDefault));
            c->stage_used[i] = false;
        }
        c->stage_bytes = need;
    }
    const int slot = c->stage_next;
    c->stage_next = (slot + 1) % kStageSlots;
    if (c->stage_used[slot]) {
        // the copy that read this slot kStageSlots uploads ago: normally long done
        if (int r = poll_event(c->stage_ev[slot], "pinned staging slot")) return r;
    }
    *slot_out = slot;
    return RZ_OK;
}

// A slot of the zero-copy ring for the next upload: (re)allocate the ring when the pose outgrew it, record the 1-in-4 event,
// and make sure the readers of the slot's previous tenant (8 uploads ago) are done.
static int zc_acquire(rz_ctx *c, size_t need, int *slot_out)
{
    if (need > c->zc_bytes) {
        HIP_TRY(hipStreamSynchronize(c->up_stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        drop_graph(c);
        const size_t hdr_off = (need + 63) / 64 * 64;
        for (int i = 0; i < rz_ctx::kZcSlots; ++i) {
            if (c->zc_host[i]) { (void)hipHostFree(c->zc_host[i]); c->zc_host[i] = nullptr; c->zc_dev[i] = nullptr; }
            // no pinned, device-mapped memory to be had (locked-memory limits ...): not an error, the caller copies instead
            if (hipHostMalloc(&c->zc_host[i], hdr_off + 64, hipHostMallocMapped) != hipSuccess ||
                hipHostGetDevicePointer(&c->zc_dev[i], c->zc_host[i], 0) != hipSuccess) {
                (void)hipGetLastError();
                for (int j = 0; j <= i; ++j)
                    if (c->zc_host[j]) { (void)hipHostFree(c->zc_host[j]); c->zc_host[j] = nullptr; c->zc_dev[j] = nullptr; }
                c->zc_bytes = 0; c->zc_cur = -1;
                return RZ_ERR_UNSUPPORTED;
            }
        }
        for (int i = 0; i < rz_ctx::kZcSlots; ++i) memset(static_cast<char *>(c->zc_host[i]) + hdr_off, 0, 64);
        c->zc_bytes = need;
        c->zc_hdr_off = hdr_off;
        c->zc_epoch++;
        c->zc_seq_cur = 0;
        c->zc_uploads = 0;
        c->zc_ev_seq[0] = c->zc_ev_seq[1] = ~0ull;
        c->zc_cur = -1;
    }
    const uint64_t u = c->zc_uploads;
    constexpr uint64_t P = rz_ctx::kZcSlots / 2;        // event period
    if (u % P == 0) {
        const int e = (int)((u / P) & 1);
        if (!c->zc_ev[e]) HIP_TRY(hipEventCreateWithFlags(&c->zc_ev[e], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c->zc_ev[e], c->stream));      // everything launched before upload u, i.e. every reader of uploads < u
        c->zc_ev_seq[e] = u;
    }
    if (u >= (uint64_t)rz_ctx::kZcSlots) {
        // previous tenant = upload u - 2P, read by frames launched before upload u - 2P + 1 (and, speculatively, by the helper
        // workgroup of the frame before it): covered by the event of the first multiple of P that is >= u - 2P + 1 — it is
        // <= u - P, so it was recorded at least P uploads ago, and it is the older of the two events kept
        const uint64_t cand = (u - (2 * P - 1) + (P - 1)) / P * P;
        const int e = (int)((cand / P) & 1);
        if (c->zc_ev_seq[e] != cand) return fail(RZ_ERR_HIP, "zero-copy ring bookkeeping is inconsistent (upload %llu)", (unsigned long long)u);
        if (int r = poll_event(c->zc_ev[e], "zero-copy pose ring")) return r;
    }
    *slot_out = (int)(u % rz_ctx::kZcSlots);
    c->zc_uploads = u + 1;
    return RZ_OK;
}

// One per-frame pose as the host handed it over, and where its parts go inside a slot (pinned slot and device pose block share
// the layout):   world pose [world | weights]     local pose [weights | rotations | translations]
struct PoseParts {
    const void *primary; size_t pbytes;       // world matrices or local rotations
    const void *secondary; size_t sbytes;     // local translations (local poses only, may be absent)
    const float *morph_weights;               // may be null (= all zero)
    bool local;
    size_t mb, mwb, total;                    // weight bytes handed over / their padded place / bytes of the whole range
};

static void lay_out_pose(const rz_ctx *c, const PoseParts &pp, char *st)
{
    char *st_mw = pp.local ? st : st + pp.pbytes;
    char *st_pr = pp.local ? st + pp.mwb : st;
    memcpy(st_pr, pp.primary, pp.pbytes);
    if (pp.sbytes) memcpy(st_pr + pp.pbytes, pp.secondary, pp.sbytes);
    if (pp.local || c->M > 0) {
        if (pp.morph_weights && pp.mb) memcpy(st_mw, pp.morph_weights, pp.mb); else memset(st_mw, 0, pp.mb);
        if (pp.mwb > pp.mb) memset(st_mw + pp.mb, 0, pp.mwb - pp.mb);
    }
}

// One character: no copy at all. The pose is laid out in a pinned, device-mapped slot; the frame's own kernels read it.
// Returns RZ_ERR_UNSUPPORTED when no such memory can be had (the caller copies instead).
static int upload_pose_zero_copy(rz_ctx *c, const PoseParts &pp)
{
    int zs = 0;
    if (int r = zc_acquire(c, std::max<size_t>(std::max<size_t>((size_t)c->B * 64 + pp.mwb, pp.mwb + (size_t)c->B * 28), 4096), &zs)) return r;
    // header protocol of the pose prefetch: invalid while the pose is being written, its sequence number once it is complete
    // (x86 stores retire in program order; the fences keep the compiler from moving them). World-matrix poses are prefetched
    // by the one-launch frame's helper, local poses by the fused-hierarchy frame's (zc_seq: the kind is part of the number).
    char *slot = static_cast<char *>(c->zc_host[zs]);
    volatile uint64_t *hdr = reinterpret_cast<volatile uint64_t *>(slot + c->zc_hdr_off);
    *hdr = 0;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    lay_out_pose(c, pp, slot);
    std::atomic_thread_fence(std::memory_order_seq_cst);
    const int kind = pp.local ? (pp.sbytes ? 2 : 1) : 0;
    const uint64_t seq = zc_seq(c, c->zc_uploads, kind);       // zc_uploads is already this upload's index + 1
    *hdr = seq;
    c->zc_seq_cur = seq;
    point_pose_slot(c, c->pose_slot ^ 1);   // where the pose will live once something makes it resident
    c->free_recorded[c->pose_slot] = false;
    c->zc_cur = zs; c->zc_local = pp.local; c->zc_kind = kind; c->zc_total = pp.total;
    c->zc_mw_off = pp.local ? 0 : pp.pbytes; c->zc_lq_off = pp.mwb;
    c->world_resident = pp.local;           // a local pose has no world matrices to bring over: rz_fk_kernel writes them
    c->mw_resident = false;
    c->local_resident = !pp.local;
    return RZ_OK;
}

// The pose goes through a pinned ring slot into the OTHER device slot as ONE copy — on the upload stream for big poses, so
// the upload overlaps whatever the compute stream is still running on the current slot; the compute stream then waits for it.
static int upload_pose_copy(rz_ctx *c, const PoseParts &pp)
{
    c->zc_cur = -1;
    c->zc_seq_cur = 0;
    c->zc_epoch++;        // this copy overwrites a pose block a helper may have staged and tagged: no later zero-copy pose may match that tag
    c->world_resident = c->mw_resident = c->local_resident = true;
    int slot = 0;
    if (int r = stage_acquire(c, std::max<size_t>((size_t)c->I * c->B * 64 + pp.mwb, pp.mwb + (size_t)c->I * c->B * 28), &slot)) return r;
    // Large poses (instanced crowds: MBs) take the upload stream: everything enqueued so far reads the current device
    // slot, so mark it, fill the other slot once ITS last readers are done, and make the compute stream wait for it.
    // Small ones that are copied at all (zero_copy = 0, small crowds) go down the compute stream itself — measured on C5,
    // the two extra packets of the cross-stream hand-off (marker + barrier) cost 3 us more per frame than the copy they hide.
    const int cur = c->pose_slot, k = cur ^ 1;
    // Overlapped-front protocol (crowds, opt-in): EVERY per-frame input travels on the upload stream and is consumed there,
    // by the front kernels — stream order is the only ordering needed, no event at all.
    const bool piped = !c->overlap_on && pp.total > (256u << 10);
    hipStream_t us = (piped || c->overlap_on) ? c->up_stream : c->stream;
    if (c->overlap_on) {
        c->free_recorded[0] = c->free_recorded[1] = false;
    } else if (piped) {
        HIP_TRY(hipEventRecord(c->ev_free[cur], c->stream));
        c->free_recorded[cur] = true;
        // Slot k was last current two uploads ago; its readers (and the FK kernel that WRITES its world matrices) were
        // all enqueued before the upload after it. If that upload was a piped one it left ev_free[k] behind them; if it
        // was a small in-stream one it recorded nothing, so fall back to "everything enqueued so far" (no overlap for
        // this one frame, but never a torn or clobbered pose).
        if (!c->free_recorded[k]) HIP_TRY(hipEventRecord(c->ev_free[k], c->stream));
        HIP_TRY(hipStreamWaitEvent(c->up_stream, c->ev_free[k], 0));
    } else {
        c->free_recorded[cur] = false;      // the slot's readers are about to be enqueued and nothing will mark their end
    }
    char *st = static_cast<char *>(c->stage[slot]);
    lay_out_pose(c, pp, st);
    point_pose_slot(c, k);                  // c->world / c->morph_w / c->local_q now name slot k under the current counts
    void *dst = pp.local ? static_cast<void *>(c->morph_w) : static_cast<void *>(c->world);
    HIP_TRY(hipMemcpyAsync(dst, st, pp.total, hipMemcpyHostToDevice, us));
    HIP_TRY(hipEventRecord(c->stage_ev[slot], us));
    c->stage_used[slot] = true;
    if (piped) {
        HIP_TRY(hipEventRecord(c->ev_up[k], c->up_stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_up[k], 0));
    }
    c->free_recorded[k] = false;            // slot k gets new readers from here on: its old end-of-readers mark is void
    return RZ_OK;
}

// Shared tail of rz_set_pose / rz_set_pose_local.
static int upload_pose(rz_ctx *c, const void *primary, size_t pbytes, const void *secondary, size_t sbytes, bool local,
                       const float *morph_weights)
{
    // the pose kind decides the plan, the plan decides which stream protocol the frame (and therefore this upload) follows
    c->pose_set = false;
    c->pose_local = local;
    c->pose_sampled = false;
    Plan upl;
    if (int r = frame_plan(c, &upl)) return r;          // the plan frames will use (run lists first), not the whole-palette fallback
    if (int r = set_overlap(c, want_overlap(c, upl))) return r;
    PoseParts pp;
    pp.primary = primary; pp.pbytes = pbytes; pp.secondary = secondary; pp.sbytes = sbytes; pp.morph_weights = morph_weights; pp.local = local;
    pp.mb = (size_t)c->I * c->M * sizeof(float);
    pp.mwb = ((size_t)c->I * std::max<uint32_t>(c->M, 1) + 3) / 4 * 4 * sizeof(float);
    pp.total = local ? pp.mwb + pbytes + sbytes : pbytes + (c->M > 0 ? pp.mwb : 0);
    int rc = RZ_ERR_UNSUPPORTED;
    if (!c->overlap_on && c->I == 1 && pp.total <= (256u << 10) && c->t_zerocopy != 0) {
        rc = upload_pose_zero_copy(c, pp);
        if (rc == RZ_ERR_UNSUPPORTED) c->t_zerocopy = 0;      // no pinned device-mapped memory: from now on every pose is copied
    }
    if (rc == RZ_ERR_UNSUPPORTED) rc = upload_pose_copy(c, pp);
    if (rc) return rc;
    c->pose_I = c->I;
    // ordered compaction of the non-zero weights for the one-launch path (instance 0)
    memset(&c->ml, 0, sizeof c->ml);
    if (c->M > 0 && morph_weights && c->I == 1) {
        int n = 0;
        for (uint32_t m = 0; m < c->M; ++m) {
            const float w = morph_weights[m];
            if (w == 0.0f) continue;
            if (n < kKargMorphs) { c->ml.idx[n] = m; c->ml.w[n] = w; }
            ++n;
        }
        c->ml.count = n <= kKargMorphs ? n : -1;
    }
    c->pose_set = true;
    return RZ_OK;
}

int rz_set_pose(rz_ctx *c, const float *world, const float *morph_weights)
{
    if (int r = use(c)) return r;
    if (c->B == 0) return fail(RZ_ERR_INVALID, "no skeleton uploaded");
    if (!world) return fail(RZ_ERR_INVALID, "null world matrices");
    if (int r = ensure_pose_buffers(c)) return r;
    return upload_pose(c, world, (size_t)c->I * c->B * 16 * sizeof(float), nullptr, 0, false, morph_weights);
}

int rz_upload_skeleton_topology(rz_ctx *c, uint32_t B, const int32_t *parents, const float *bind_translation3,
                                const int32_t *append_parent, const float *append_ratio, const uint8_t *append_move)
{
    if (int r = use(c)) return r;
    if (int r = static_unlocked(c, "rz_upload_skeleton_topology")) return r;
    if (B == 0 || B != c->B) return fail(RZ_ERR_INVALID, "topology has %u bones but the uploaded skeleton has %u", B, c->B);
    if (!parents || !bind_translation3) return fail(RZ_ERR_INVALID, "null topology arrays");
    // hierarchy levels (parents may come in any order, like the reference's recursive solve; cycles are an error)
    std::vector<int> level(B, -1);
    for (uint32_t b = 0; b < B; ++b) {
        if (parents[b] >= (int32_t)B) return fail(RZ_ERR_INVALID, "bone %u parent %d out of range", b, parents[b]);
        std::vector<uint32_t> chain;
        uint32_t cur = b;
        while (level[cur] < 0) {
            chain.push_back(cur);
            if (chain.size() > B) return fail(RZ_ERR_INVALID, "bone hierarchy has a cycle through bone %u", b);
            if (parents[cur] < 0) { level[cur] = 0; chain.pop_back(); break; }
            cur = (uint32_t)parents[cur];
        }
        for (size_t k = chain.size(); k-- > 0;) level[chain[k]] = level[(uint32_t)parents[chain[k]]] + 1;
    }
    int n_levels = 0;
    for (uint32_t b = 0; b < B; ++b) n_levels = std::max(n_levels, level[b] + 1);
    // one 32-byte record per bone: (parent, append parent, bits(append ratio), flags) (bits(bind x y z), 0)
    std::vector<uint4> rec((size_t)B * 2);
    for (uint32_t b = 0; b < B; ++b) {
        const int32_t ap = (append_parent && append_parent[b] >= 0 && append_parent[b] < (int32_t)B) ? append_parent[b] : -1;
        const float ratio = append_ratio ? append_ratio[b] : 1.0f;
        uint32_t rb, bx, by, bz;
        memcpy(&rb, &ratio, 4);
        memcpy(&bx, bind_translation3 + (size_t)b * 3, 4); memcpy(&by, bind_translation3 + (size_t)b * 3 + 1, 4); memcpy(&bz, bind_translation3 + (size_t)b * 3 + 2, 4);
        rec[2 * b] = make_uint4((uint32_t)(parents[b] < 0 ? -1 : parents[b]), (uint32_t)ap, rb, (append_move && append_move[b]) ? 1u : 0u);
        rec[2 * b + 1] = make_uint4(bx, by, bz, 0u);
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    drop_graph(c);
    c->ovr_count = 0;
    dfree(c->fk_rec);
    if (int r = to_device(&c->fk_rec, rec.data(), rec.size())) return r;
    c->fk_levels = n_levels;
    c->has_topology = true;
    return RZ_OK;
}

int rz_upload_bone_morphs(rz_ctx *c, uint32_t n, const uint32_t *morph, const uint32_t *bone, const float *translation3, const float *rotation4)
{
    if (int r = use(c)) return r;
    if (int r = static_unlocked(c, "rz_upload_bone_morphs")) return r;
    if (n == 0) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        free_bone_morphs(c);
        return RZ_OK;
    }
    if (!c->has_topology) return fail(RZ_ERR_INVALID, "bone morphs act on device-solved poses: call rz_upload_skeleton_topology first");
    if (c->M == 0) return fail(RZ_ERR_INVALID, "upload the morph set first (rz_upload_morphs_*): bone-morph entries name its morphs");
    if (!morph || !bone || !translation3 || !rotation4) return fail(RZ_ERR_INVALID, "null bone-morph arrays");
    for (uint32_t k = 0; k < n; ++k) {
        if (morph[k] >= c->M) return fail(RZ_ERR_INVALID, "bone-morph entry %u names morph %u of %u", k, morph[k], c->M);
        if (bone[k] >= c->B) return fail(RZ_ERR_INVALID, "bone-morph entry %u names bone %u of %u", k, bone[k], c->B);
        for (int j = 0; j < 7; ++j) {
            const float x = j < 3 ? translation3[(size_t)k * 3 + j] : rotation4[(size_t)k * 4 + j - 3];
            if (!(x == x) || x - x != 0.0f) return fail(RZ_ERR_INVALID, "bone-morph entry %u is not finite", k);
        }
    }
    // group by bone; inside a bone ascending morph index, file order among equal morphs (a stable sort of the entry list)
    std::vector<uint32_t> idx(n);
    for (uint32_t k = 0; k < n; ++k) idx[k] = k;
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return bone[a] != bone[b] ? bone[a] < bone[b] : morph[a] < morph[b]; });
    std::vector<uint32_t> off(c->B + 1, 0), mo(n);
    std::vector<float4> rot(n), tr(n);
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t e = idx[k];
        off[bone[e] + 1]++;
        mo[k] = morph[e];
        rot[k] = make_float4(rotation4[(size_t)e * 4], rotation4[(size_t)e * 4 + 1], rotation4[(size_t)e * 4 + 2], rotation4[(size_t)e * 4 + 3]);
        tr[k] = make_float4(translation3[(size_t)e * 3], translation3[(size_t)e * 3 + 1], translation3[(size_t)e * 3 + 2], 0.0f);
    }
    for (uint32_t b = 0; b < c->B; ++b) off[b + 1] += off[b];
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_bone_morphs(c);
    drop_graph(c);
    if (int r = to_device(&c->bm_off, off.data(), off.size())) return r;
    if (int r = to_device(&c->bm_morph, mo.data(), n)) return r;
    if (int r = to_device(&c->bm_rot, rot.data(), n)) return r;
    if (int r = to_device(&c->bm_tr, tr.data(), n)) return r;
    c->bm_count = n;
    return RZ_OK;
}

int rz_set_pose_local(rz_ctx *c, const float *local_rotations4, const float *local_translations3, const float *morph_weights)
{
    if (int r = use(c)) return r;
    if (!c->has_topology) return fail(RZ_ERR_INVALID, "rz_upload_skeleton_topology has not been called for this skeleton");
    if (!local_rotations4) return fail(RZ_ERR_INVALID, "null local rotations");
    if (int r = ensure_pose_buffers(c)) return r;
    const size_t nq = (size_t)c->I * c->B;
    c->pose_local_t = local_translations3 != nullptr;
    return upload_pose(c, local_rotations4, nq * sizeof(float4), local_translations3, local_translations3 ? nq * 3 * sizeof(float) : 0, true,
                       morph_weights);
}

int rz_upload_animation(rz_ctx *c, const rz_animation *a)
{
    if (int r = use(c)) return r;
    if (int r = static_unlocked(c, "rz_upload_animation")) return r;
    if (!a) return fail(RZ_ERR_INVALID, "null animation");
    if (c->B == 0) return fail(RZ_ERR_INVALID, "upload the skeleton before a motion");
    const uint32_t n = a->n_bone_tracks, mt = a->n_morph_tracks;
    if (n && (!a->track_bone || !a->key_off || !a->key_frame || !a->key_rot4 || !a->key_pos3)) return fail(RZ_ERR_INVALID, "null bone-track arrays");
    if (mt && (!a->mkey_off || !a->mkey_frame || !a->mkey_weight)) return fail(RZ_ERR_INVALID, "null morph-track arrays");
    if (c->M && mt && (!a->feed_off || (a->feed_off[c->M] && (!a->feed_track || !a->feed_ratio)))) return fail(RZ_ERR_INVALID, "null morph feeds");
    std::vector<int> bone_track(c->B, -1);
    const uint32_t K = n ? a->key_off[n] : 0;
    for (uint32_t t = 0; t < n; ++t) {
        if (a->key_off[t] > a->key_off[t + 1]) return fail(RZ_ERR_INVALID, "key offsets must be non-decreasing");
        const int32_t b = a->track_bone[t];
        if (b < 0 || (uint32_t)b >= c->B) continue;                      // a motion may key bones this model lacks
        if (bone_track[b] >= 0) return fail(RZ_ERR_INVALID, "bone %d is driven by two tracks", b);
        for (uint32_t k = a->key_off[t] + 1; k < a->key_off[t + 1]; ++k)
            // equal frames are legal (real VMD files carry duplicate keys; host/vmd-sampler.js keeps them too): the span search
            // lands on the last key <= frame and the first key > frame, so a zero-length span is never divided by
            if (!(a->key_frame[k] >= a->key_frame[k - 1])) return fail(RZ_ERR_INVALID, "track %u: key frames must not descend", t);
        bone_track[b] = (int)t;
    }
    const uint32_t Km = mt ? a->mkey_off[mt] : 0;
    for (uint32_t t = 0; t < mt; ++t) {
        if (a->mkey_off[t] > a->mkey_off[t + 1]) return fail(RZ_ERR_INVALID, "morph key offsets must be non-decreasing");
        for (uint32_t k = a->mkey_off[t] + 1; k < a->mkey_off[t + 1]; ++k)
            if (!(a->mkey_frame[k] >= a->mkey_frame[k - 1])) return fail(RZ_ERR_INVALID, "morph track %u: key frames must not descend", t);
    }
    std::vector<uint32_t> feed_off(c->M + 1, 0);
    uint32_t F = 0;
    if (c->M && mt) {
        for (uint32_t m = 0; m <= c->M; ++m) feed_off[m] = a->feed_off[m];
        F = feed_off[c->M];
        for (uint32_t m = 0; m < c->M; ++m)
            if (feed_off[m] > feed_off[m + 1]) return fail(RZ_ERR_INVALID, "feed offsets must be non-decreasing");
        for (uint32_t f = 0; f < F; ++f)
            if (a->feed_track[f] < 0 || (uint32_t)a->feed_track[f] >= mt) return fail(RZ_ERR_INVALID, "feed %u names morph track %d of %u", f, a->feed_track[f], mt);
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_animation(c);
    // per bone / per morph feed: the key range itself, so the sampler's chain of dependent loads starts one level lower
    std::vector<uint4> bone_range(c->B), feed_range(F);
    auto record = [](const uint32_t *off, const float *kf, int t) {
        uint4 r; r.x = r.y = r.z = r.w = 0;
        if (t < 0 || off[t + 1] == off[t]) return r;
        r.x = off[t]; r.y = off[t + 1];
        memcpy(&r.z, &kf[off[t]], 4);
        memcpy(&r.w, &kf[off[t + 1] - 1], 4);
        return r;
    };
    for (uint32_t b = 0; b < c->B; ++b) bone_range[b] = record(a->key_off, a->key_frame, bone_track[b]);
    for (uint32_t f = 0; f < F; ++f) feed_range[f] = record(a->mkey_off, a->mkey_frame, a->feed_track[f]);
    if (int r = to_device(&c->an_bone_range, bone_range.data(), c->B)) return r;
    if (int r = to_device(&c->an_key_frame, a->key_frame, K)) return r;
    if (int r = to_device(&c->an_key_rot, a->key_rot4, K)) return r;
    if (int r = to_device(&c->an_key_pos, a->key_pos3, (size_t)K * 3)) return r;
    if (a->key_interp16 && K)
        if (int r = to_device(&c->an_key_interp, a->key_interp16, K)) return r;
    if (int r = to_device(&c->an_mkey_frame, a->mkey_frame, Km)) return r;
    if (int r = to_device(&c->an_mkey_weight, a->mkey_weight, Km)) return r;
    if (int r = to_device(&c->an_feed_off, feed_off.data(), (size_t)c->M + 1)) return r;
    if (int r = to_device(&c->an_feed_range, feed_range.data(), F)) return r;
    if (int r = to_device(&c->an_feed_ratio, a->feed_ratio, F)) return r;
    c->an_M = c->M;
    c->has_animation = true;
    return RZ_OK;
}

int rz_set_pose_sampled(rz_ctx *c, const float *frames)
{
    if (int r = use(c)) return r;
    if (!c->has_animation) return fail(RZ_ERR_INVALID, "rz_upload_animation has not been called");
    if (!c->has_topology) return fail(RZ_ERR_INVALID, "rz_upload_skeleton_topology has not been called for this skeleton");
    if (c->an_M != c->M) return fail(RZ_ERR_INVALID, "the motion's morph feeds were built for %u vertex morphs, the context holds %u", c->an_M, c->M);
    if (!frames) return fail(RZ_ERR_INVALID, "null frames");
    if (int r = ensure_pose_buffers(c)) return r;
    if (c->I > c->an_frames_alloc) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        dfree(c->an_frames);
        HIP_TRY(hipMalloc(&c->an_frames, (size_t)c->I * sizeof(float)));
        c->an_frames_alloc = c->I;
    }
    c->pose_set = false;
    c->pose_sampled = true;
    c->pose_local = true;
    c->pose_local_t = true;
    Plan upl;
    if (int r = frame_plan(c, &upl)) return r;
    if (int r = set_overlap(c, want_overlap(c, upl))) return r;
    c->zc_cur = -1;                         // the pose is produced on the device: nothing of it sits in a pinned slot
    c->zc_seq_cur = 0;
    c->zc_epoch++;                          // its frame writes world matrices / weights into the pose block: tags staged before never match again
    c->world_resident = c->mw_resident = c->local_resident = true;
    c->frames_inline = c->I == 1 && !c->overlap_on && c->t_zerocopy != 0;
    if (c->frames_inline) {
        c->frame0 = frames[0];              // one character: the frame number rides in rz_fk_kernel's arguments
    } else {
        int slot = 0;
        if (int r = stage_acquire(c, std::max<size_t>((size_t)c->I * sizeof(float), 4096), &slot)) return r;
        memcpy(c->stage[slot], frames, (size_t)c->I * sizeof(float));
        hipStream_t us = front_stream(c);   // consumed by rz_fk_kernel, which runs on this stream
        HIP_TRY(hipMemcpyAsync(c->an_frames, c->stage[slot], (size_t)c->I * sizeof(float), hipMemcpyHostToDevice, us));
        HIP_TRY(hipEventRecord(c->stage_ev[slot], us));
        c->stage_used[slot] = true;
    }
    point_pose_slot(c, c->pose_slot);       // the sampled pose is written by rz_fk_kernel under the current counts
    memset(&c->ml, 0, sizeof c->ml);
    if (c->M > 0) c->ml.count = -1;          // the weights only exist on the device: the prep kernel compacts them
    c->pose_I = c->I;
    c->pose_set = true;
    return RZ_OK;
}

int rz_override_world(rz_ctx *c, uint32_t n, const uint32_t *instance, const uint32_t *bone, const float *world16)
{
    if (int r = use(c)) return r;
    if (n == 0) { c->ovr_count = 0; return RZ_OK; }
    if (!c->has_topology) return fail(RZ_ERR_INVALID, "rz_override_world applies to device-solved poses: call rz_upload_skeleton_topology first");
    if (!bone || !world16) return fail(RZ_ERR_INVALID, "null override arrays");
    // sort by (instance, bone); of several entries for one bone the LAST wins, like successive boneWorldMatrices.set() calls
    std::vector<uint32_t> order(n);
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t i = instance ? instance[k] : 0;
        if (i >= c->I || bone[k] >= c->B) return fail(RZ_ERR_INVALID, "override %u names instance %u bone %u (have %u x %u)", k, i, bone[k], c->I, c->B);
        for (int e = 0; e < 16; ++e) {
            const float x = world16[(size_t)k * 16 + e];
            if (!(x == x) || x - x != 0.0f) return fail(RZ_ERR_INVALID, "override %u is not finite", k);
        }
        order[k] = k;
    }
    auto key = [&](uint32_t k) { return (uint64_t)(instance ? instance[k] : 0) * c->B + bone[k]; };
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key(a) < key(b); });
    std::vector<int> off(c->I + 1, 0), bones;
    std::vector<float> mats;
    for (uint32_t q = 0; q < n; ++q) {
        if (q + 1 < n && key(order[q + 1]) == key(order[q])) continue;
        const uint32_t k = order[q];
        off[(instance ? instance[k] : 0) + 1]++;
        bones.push_back((int)bone[k]);
        mats.insert(mats.end(), world16 + (size_t)k * 16, world16 + (size_t)k * 16 + 16);
    }
    for (uint32_t i = 0; i < c->I; ++i) off[i + 1] += off[i];
    const size_t m = bones.size();
    if (m > c->ovr_alloc || off.size() > c->ovr_off_alloc) {
        HIP_TRY(hipStreamSynchronize(c->up_stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        drop_graph(c);
        dfree(c->ovr_off); dfree(c->ovr_bone); dfree(c->ovr_world);
        c->ovr_alloc = std::max<size_t>(m, 64); c->ovr_off_alloc = off.size();
        c->ovr_count = 0;
        HIP_TRY(hipMalloc(&c->ovr_off, c->ovr_off_alloc * sizeof(int)));
        HIP_TRY(hipMalloc(&c->ovr_bone, c->ovr_alloc * sizeof(int)));
        HIP_TRY(hipMalloc(&c->ovr_world, c->ovr_alloc * 16 * sizeof(float)));
    }
    // one pinned ring slot carries offsets | bones | matrices down the compute stream, in order with the frames
    const size_t b_off = off.size() * sizeof(int), b_bone = m * sizeof(int), b_mat = m * 16 * sizeof(float);
    int slot = 0;
    if (int r = stage_acquire(c, std::max<size_t>(b_off + b_bone + b_mat, 4096), &slot)) return r;
    char *st = static_cast<char *>(c->stage[slot]);
    memcpy(st, off.data(), b_off);
    memcpy(st + b_off, bones.data(), b_bone);
    memcpy(st + b_off + b_bone, mats.data(), b_mat);
    hipStream_t us = front_stream(c);       // consumed by rz_fk_kernel, which runs on this stream
    HIP_TRY(hipMemcpyAsync(c->ovr_off, st, b_off, hipMemcpyHostToDevice, us));
    HIP_TRY(hipMemcpyAsync(c->ovr_bone, st + b_off, b_bone, hipMemcpyHostToDevice, us));
    HIP_TRY(hipMemcpyAsync(c->ovr_world, st + b_off + b_bone, b_mat, hipMemcpyHostToDevice, us));
    HIP_TRY(hipEventRecord(c->stage_ev[slot], us));
    c->stage_used[slot] = true;
    c->ovr_count = (uint32_t)m;
    return RZ_OK;
}

int rz_read_world(rz_ctx *c, uint32_t instance, float *world16)
{
    if (int r = use(c)) return r;
    if (instance >= c->I || !world16 || !c->world) return fail(RZ_ERR_INVALID, "bad world read");
    if (c->zc_cur >= 0 && !c->zc_local && !c->world_resident) {     // a zero-copy pose no frame has consumed yet: still in its pinned slot
        memcpy(world16, c->zc_host[c->zc_cur], (size_t)c->B * 16 * sizeof(float));
        return RZ_OK;
    }
    HIP_TRY(hipStreamSynchronize(c->up_stream));        // rz_fk_kernel may have written them on the front stream
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(world16, c->world + (size_t)instance * c->B * 16, (size_t)c->B * 16 * sizeof(float), hipMemcpyDeviceToHost));
    return RZ_OK;
}

int rz_deform(rz_ctx *c)
{
    if (int r = use(c)) return r;
    if (int r = check_ready(c)) return r;
    if (int r = ensure_outputs(c)) return r;
    Plan pl;
    if (int r = frame_plan(c, &pl)) return r;
    if (int r = set_overlap(c, want_overlap(c, pl))) return r;
    return run_frame(c, pl);
}

// FNV-1a over the plain-data structs a frame's launches are built from: if none of them changed, a captured graph of
// those launches is still the same work.
static uint64_t fnv(uint64_t h, const void *p, size_t n)
{
    const unsigned char *b = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

static uint64_t frame_signature(rz_ctx *c, const Plan &pl)
{
    // every parameter block a frame's launches are built from, whole: device pointers included, so a buffer that was
    // freed and re-allocated (rz_upload_animation, rz_upload_skeleton_topology, a grown pose ring ...) changes the key,
    // and the bounding-box slot parity as it is RIGHT NOW (the captured frames alternate from it)
    uint64_t h = 1469598103934665603ull;
    const RzDeformParams dp = deform_params(c, pl);
    h = fnv(h, &dp, sizeof dp);
    h = fnv(h, &c->ml, sizeof c->ml);
    h = fnv(h, &pl, sizeof pl);
    const RzPrepParams pp = prep_params(c);
    h = fnv(h, &pp, sizeof pp);
    const RzFkParams fp = fk_params(c);
    h = fnv(h, &fp, sizeof fp);
    const uint64_t misc[6] = { c->I, c->pose_local, c->pose_local_t, c->pose_sampled, (uint64_t)c->morph_mode, (uint64_t)c->aabb_on };
    return fnv(h, misc, sizeof misc);
}

constexpr uint32_t kGraphFrames = 16;   // even: the bounding-box slot parity is the same after a replay as before it

int rz_deform_n(rz_ctx *c, uint32_t frames)
{
    if (int r = use(c)) return r;
    if (int r = check_ready(c)) return r;
    if (int r = ensure_outputs(c)) return r;
    Plan pl;
    if (int r = frame_plan(c, &pl)) return r;
    if (int r = set_overlap(c, want_overlap(c, pl))) return r;      // never on while the graph key is set
    uint32_t f = 0;
    if (c->t_graph && frames >= 2 * kGraphFrames) {
        // Launch-bound replay (a 30 k-vertex frame is 3-6 us of GPU time per ~3 us of launch work on the host): capture
        // kGraphFrames whole frames once into a hipGraph and replay that; one launch call per 16 frames.
        // The key is taken from the state the capture starts in. A cached graph whose key differs only in the bounding-box
        // slot parity is brought back in phase by one plain frame (which the graph's first build needs anyway: it sets the
        // kernel attributes and loads the modules); kGraphFrames is even, so a replay ends on the parity it began with.
        uint64_t sig = frame_signature(c, pl);
        if (!c->graph_exec || c->graph_sig != sig) {
            if (int r = run_frame(c, pl)) return r;
            ++f;
            sig = frame_signature(c, pl);
        }
        if (!c->graph_exec || c->graph_sig != sig) {
            drop_graph(c);
            hipGraph_t g = nullptr;
            HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            int rc = RZ_OK;
            for (uint32_t k = 0; k < kGraphFrames && rc == RZ_OK; ++k) {
                rc = launch_front(c, pl, c->stream);
                if (rc == RZ_OK) rc = launch_deform(c, pl);
            }
            hipError_t ce = hipStreamEndCapture(c->stream, &g);
            if (rc != RZ_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
            if (ce != hipSuccess || !g) return fail(RZ_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(ce));
            hipError_t ie = hipGraphInstantiate(&c->graph_exec, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (ie != hipSuccess) { c->graph_exec = nullptr; return fail(RZ_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ie)); }
            c->graph_sig = sig;
        }
        for (; f + kGraphFrames <= frames; f += kGraphFrames) HIP_TRY(hipGraphLaunch(c->graph_exec, c->stream));
    }
    for (; f < frames; ++f)
        if (int r = run_frame(c, pl)) return r;
    return RZ_OK;
}

int rz_deform_pair(rz_ctx *a, rz_ctx *b, uint32_t frames)
{
    if (!a || !b || a == b) return fail(RZ_ERR_INVALID, "rz_deform_pair needs two different contexts");
    if (a->device != b->device) return fail(RZ_ERR_INVALID, "rz_deform_pair: the contexts live on devices %d and %d", a->device, b->device);
    rz_ctx *cs[2] = { a, b };
    Plan pl[2];
    for (int k = 0; k < 2; ++k) {
        if (int r = use(cs[k])) return r;
        if (int r = check_ready(cs[k])) return r;
        if (int r = ensure_outputs(cs[k])) return r;
        if (int r = frame_plan(cs[k], &pl[k])) return r;
        if (int r = set_overlap(cs[k], want_overlap(cs[k], pl[k]))) return r;
    }
    for (uint32_t f = 0; f < frames; ++f)
        if (int r = run_frame(cs[f & 1], pl[f & 1])) return r;
    return RZ_OK;
}

int rz_sync(rz_ctx *c)
{
    if (int r = use(c)) return r;
    HIP_TRY(hipStreamSynchronize(c->up_stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RZ_OK;
}

int rz_read(rz_ctx *c, uint32_t instance, uint32_t v0, uint32_t n, float *pos3, float *nrm3)
{
    if (int r = use(c)) return r;
    if (instance >= c->I) return fail(RZ_ERR_INVALID, "instance %u out of range", instance);
    if ((uint64_t)v0 + n > c->V) return fail(RZ_ERR_INVALID, "vertex range [%u,%u) exceeds %u", v0, v0 + n, c->V);
    if (!c->out_pos) return fail(RZ_ERR_INVALID, "nothing deformed yet");
    HIP_TRY(hipStreamSynchronize(c->stream));
    const size_t off = ((size_t)instance * c->Vp + v0) * 3;
    const float *sp = c->ext_pos ? c->ext_pos : c->out_pos, *sn = c->ext_nrm ? c->ext_nrm : c->out_nrm;
    if (pos3 && n) HIP_TRY(hipMemcpy(pos3, sp + off, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (nrm3 && n) HIP_TRY(hipMemcpy(nrm3, sn + off, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return RZ_OK;
}

int rz_read_palette(rz_ctx *c, uint32_t instance, float *rows3x4)
{
    if (int r = use(c)) return r;
    if (instance >= c->I || !rows3x4 || !c->palette) return fail(RZ_ERR_INVALID, "bad palette read");
    if (c->palette_stale) {
        // the last frame was a bone-subset crowd frame: its workgroups formed the rows of their own bones in LDS and nobody
        // wrote the skinMatrixBuffer. Form it now from the resident world matrices — rz_prep_kernel's chain is the skin
        // kernel's chain, so these are the bits the frame used.
        if (int r = launch_prep(c, c->stream)) return r;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(rows3x4, c->palette + (size_t)instance * c->B * 3, (size_t)c->B * 12 * sizeof(float), hipMemcpyDeviceToHost));
    return RZ_OK;
}

int rz_upload_edge_scale(rz_ctx *c, uint32_t V, const float *edge_size)
{
    if (int r = use(c)) return r;
    if (int r = static_unlocked(c, "rz_upload_edge_scale")) return r;
    HIP_TRY(hipStreamSynchronize(c->stream));
    drop_graph(c);
    if (!edge_size) { dfree(c->edge); return RZ_OK; }
    if (V != c->V || V == 0) return fail(RZ_ERR_INVALID, "edge scale has %u entries but the mesh has %u vertices", V, c->V);
    dfree(c->edge);
    HIP_TRY(hipMalloc(&c->edge, (size_t)c->Vp * sizeof(float)));
    HIP_TRY(hipMemsetAsync(c->edge, 0, (size_t)c->Vp * sizeof(float), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(c->edge, edge_size, (size_t)V * sizeof(float), hipMemcpyHostToDevice));
    return ensure_outputs(c);
}

int rz_read_hull(rz_ctx *c, uint32_t instance, uint32_t v0, uint32_t n, float *pos3)
{
    if (int r = use(c)) return r;
    if (!c->edge || !c->out_hull) return fail(RZ_ERR_INVALID, "the outline hull is off (rz_upload_edge_scale)");
    if (instance >= c->I || (uint64_t)v0 + n > c->V || !pos3) return fail(RZ_ERR_INVALID, "bad hull read");
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (n) HIP_TRY(hipMemcpy(pos3, c->out_hull + ((size_t)instance * c->Vp + v0) * 3, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return RZ_OK;
}

int rz_enable_aabb(rz_ctx *c, int enable)
{
    if (int r = use(c)) return r;
    drop_graph(c);
    c->aabb_on = enable != 0;
    c->aabb_rearm = true;
    return ensure_outputs(c);
}

int rz_read_aabb(rz_ctx *c, uint32_t instance, float min_max6[6])
{
    if (int r = use(c)) return r;
    if (!c->aabb_on || !c->aabb) return fail(RZ_ERR_INVALID, "the bounding-box reduction is off (rz_enable_aabb)");
    if (instance >= c->I || !min_max6) return fail(RZ_ERR_INVALID, "bad aabb read");
    HIP_TRY(hipStreamSynchronize(c->stream));
    uint32_t keys[6];
    const int last = c->aabb_slot ^ 1;     // the slot the most recent frame accumulated into
    HIP_TRY(hipMemcpy(keys, c->aabb + ((size_t)instance * 2 + last) * 6, sizeof keys, hipMemcpyDeviceToHost));
    for (int k = 0; k < 6; ++k) {
        const uint32_t bits = (keys[k] & 0x80000000u) ? (keys[k] ^ 0x80000000u) : ~keys[k];
        memcpy(&min_max6[k], &bits, 4);
    }
    return RZ_OK;
}

int rz_time_frames(rz_ctx *c, uint32_t frames, rz_timing *out)
{
    if (int r = use(c)) return r;
    if (!out || frames == 0) return fail(RZ_ERR_INVALID, "rz_time_frames: bad arguments");
    if (int r = check_ready(c)) return r;
    if (int r = ensure_outputs(c)) return r;
    Plan pl;
    if (int r = frame_plan(c, &pl)) return r;
    if (int r = set_overlap(c, want_overlap(c, pl))) return r;
    memset(out, 0, sizeof *out);
    float ms = 0.f;
    // whole frames (front kernels when the plan has any + the deform / skin kernel), back to back exactly as rz_deform
    // issues them — for crowds that is the overlapped protocol: fronts on the upload stream, skin kernels on the
    // context's stream (the events below sit on the context's stream; the last skin kernel waits for the last front)
    if (int r = run_frame(c, pl)) return r;       // the deform-only loop below needs a palette
    HIP_TRY(hipStreamSynchronize(c->up_stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    for (uint32_t f = 0; f < frames; ++f)
        if (int r = run_frame(c, pl)) return r;
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    out->frame_ms = ms / frames;
    // the deform / skin kernel alone (reads the ring slot the last frame left current)
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    for (uint32_t f = 0; f < frames; ++f)
        if (int r = launch_deform(c, pl)) return r;
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    out->deform_kernel_ms = ms / frames;
    // the front kernels alone (only part of the frame when the plan is not the one-launch FAST form); everything has
    // drained at this point, so they may run on the context's stream whatever the protocol
    if ((pl.prep || c->pose_local) && !pl.fuse_fk) {
        HIP_TRY(hipStreamSynchronize(c->up_stream));
        HIP_TRY(hipEventRecord(c->ev0, c->stream));
        for (uint32_t f = 0; f < frames; ++f)
            if (int r = launch_front(c, pl, c->stream)) return r;
        HIP_TRY(hipEventRecord(c->ev1, c->stream));
        HIP_TRY(hipEventSynchronize(c->ev1));
        HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        out->prep_kernel_ms = ms / frames;
    }
    c->skin_recorded[0] = c->skin_recorded[1] = false;      // both streams are idle: no slot has a reader in flight
    out->verts_per_frame = (uint64_t)c->V * c->I;
    out->algorithmic_bytes_per_frame = algorithmic_bytes(c);
    out->frames = frames;
    return RZ_OK;
}

int rz_autotune_measure(rz_ctx *c, uint32_t frames, rz_tune_entry *table, int cap, int *count)
{
    if (int r = use(c)) return r;
    if (!table || cap < 1 || !count) return fail(RZ_ERR_INVALID, "rz_autotune_measure: bad table");
    *count = 0;
    if (int r = check_ready(c)) return r;
    if (int r = ensure_outputs(c)) return r;
    if (frames == 0) frames = 100;
    frames = std::min<uint32_t>(frames, 1000);      // a search, not a benchmark
    // Entry 0 = the heuristic plan. Then: morph split x workgroups per CU (single mesh), poses per workgroup x workgroups per
    // CU (instanced). Every candidate is a legal plan; the search only picks among the variants the parity tests already cover.
    std::vector<rz_tune_entry> cands;
    auto add = [&](int split, int capv, int loop) {
        rz_tune_entry e;
        memset(&e, 0, sizeof e);
        e.morph_split = split; e.grid_cap = capv; e.inst_loop = loop; e.same_as = -1;
        cands.push_back(e);
    };
    const int ncu = c->n_cu;
    const int keep_split = c->t_split, keep_cap = c->t_grid_cap, keep_loop = c->t_instloop;
    const bool keep_tuned = c->tuned_by_search;
    InstShape is;
    const bool instanced = inst_shape(c, &is);      // with the caller's inst_loop: 0 (crowd kernel off) and 9 (register form) are not searched over
    c->t_split = 0; c->t_grid_cap = 0; if (instanced) c->t_instloop = -1;
    add(0, 0, -1);
    if (instanced) {
        // total workgroups: one or two rounds of what the CUs hold at once (two 256-thread workgroups or one larger one);
        // with bone subsets the palettes of 16 poses still fit, which halves the number of workgroup fronts
        for (int loop : {8, 4, 16})
            for (int capv : {ncu, 2 * ncu, 3 * ncu, 4 * ncu}) add(0, capv, loop);
    } else {
        const int smax = c->morph_mode == 1 ? (int)std::min<uint32_t>(8, std::max<uint32_t>(1, c->M)) : 4;
        for (int sp = 1; sp <= smax; sp <<= 1)
            for (int capv : {ncu, 2 * ncu, 4 * ncu}) add(sp, (int)(capv * c->I), 0);
    }
    if ((int)cands.size() > cap) cands.resize(cap);
    const int n = (int)cands.size();
    auto restore = [&]() { c->t_split = keep_split; c->t_grid_cap = keep_cap; c->t_instloop = keep_loop; c->tuned_by_search = keep_tuned; };
    std::vector<Plan> plans(n);
    for (int i = 0; i < n; ++i) {
        c->t_split = cands[i].morph_split; c->t_grid_cap = cands[i].grid_cap; c->t_instloop = instanced ? cands[i].inst_loop : keep_loop;
        if (int r = frame_plan(c, &plans[i])) { restore(); return r; }
        const Plan &pl = plans[i];
        cands[i].eff_split = pl.v.S; cands[i].eff_grid = (int)pl.grid_x; cands[i].eff_inst_group = pl.inst_group;
        for (int k = 0; k < i && cands[i].same_as < 0; ++k)      // different requests often resolve to the same launch
            if (plans[k].grid_x == pl.grid_x && (pl.inst_group > 0 || plans[k].quads_per_wave == pl.quads_per_wave) && plans[k].v.S == pl.v.S &&
                plans[k].inst_group == pl.inst_group && plans[k].verts_per_wg == pl.verts_per_wg && plans[k].subsets == pl.subsets &&
                plans[k].inst_block == pl.inst_block)
                cands[i].same_as = cands[k].same_as >= 0 ? cands[k].same_as : k;
    }
    constexpr int kRounds = 5;
    std::vector<float> t((size_t)n * kRounds, 0.f);
    int rc = RZ_OK;
    {
        // the GPU reaches its sustained clocks only after a while of work (measured: the first timed round of the first
        // candidate came out 15-20 % slow on a cold device, which is enough to move a median of three): run the heuristic
        // plan for ~0.25 s first, untimed
        c->t_split = cands[0].morph_split; c->t_grid_cap = cands[0].grid_cap; c->t_instloop = instanced ? cands[0].inst_loop : keep_loop;
        Plan warm;
        rc = frame_plan(c, &warm);      // (a crowd's run lists belong to ONE launch shape: bring them back to entry 0's)
        if (rc == RZ_OK) rc = set_overlap(c, want_overlap(c, warm));
        const auto t0 = std::chrono::steady_clock::now();
        while (rc == RZ_OK && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(250)) {
            for (uint32_t f = 0; f < 64 && rc == RZ_OK; ++f) rc = run_frame(c, warm);
            if (rc == RZ_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(RZ_ERR_HIP, "rz_autotune_measure: warm-up failed");
        }
    }
    for (int round = -1; round < kRounds && rc == RZ_OK; ++round) {          // round -1 warms every variant up, untimed
        for (int i = 0; i < n && rc == RZ_OK; ++i) {
            if (cands[i].same_as >= 0) continue;
            c->t_split = cands[i].morph_split; c->t_grid_cap = cands[i].grid_cap; c->t_instloop = instanced ? cands[i].inst_loop : keep_loop;
            Plan pl;
            if ((rc = frame_plan(c, &pl)) != RZ_OK) break;       // crowd shapes alternate: the run lists follow (a rebuild, untimed)
            if ((rc = set_overlap(c, want_overlap(c, pl))) != RZ_OK) break;
            const uint32_t nf = round < 0 ? 8 : frames;
            for (uint32_t f = 0; f < 4 && rc == RZ_OK; ++f) rc = run_frame(c, pl);
            if (rc != RZ_OK) break;
            hipError_t he = hipEventRecord(c->ev0, c->stream);
            for (uint32_t f = 0; f < nf && rc == RZ_OK; ++f) rc = run_frame(c, pl);
            if (rc != RZ_OK) break;
            if (he == hipSuccess) he = hipEventRecord(c->ev1, c->stream);
            if (he == hipSuccess) he = hipEventSynchronize(c->ev1);
            float ms = 0.f;
            if (he == hipSuccess) he = hipEventElapsedTime(&ms, c->ev0, c->ev1);
            if (he != hipSuccess) { rc = fail(RZ_ERR_HIP, "rz_autotune_measure: %s", hipGetErrorString(he)); break; }
            if (round >= 0) t[(size_t)i * kRounds + round] = ms / nf;
        }
    }
    restore();
    if (rc != RZ_OK) return rc;
    for (int i = 0; i < n; ++i) {
        const int src = cands[i].same_as >= 0 ? cands[i].same_as : i;
        float v[kRounds];
        for (int k = 0; k < kRounds; ++k) v[k] = t[(size_t)src * kRounds + k];
        std::sort(v, v + kRounds);
        cands[i].ms = v[kRounds / 2]; cands[i].ms_min = v[0]; cands[i].ms_max = v[kRounds - 1];
        table[i] = cands[i];
    }
    *count = n;
    return RZ_OK;
}

int rz_autotune_pick(const rz_tune_entry *table, int count)
{
    if (!table || count < 1) return 0;
    // entry 0 (the heuristics) stays unless something is clearly faster: median >= 2 % lower AND, when the table carries the
    // per-round spread, its slowest round still under the heuristic's fastest one (overlapping ranges are box noise, and a pick that
    // follows noise differs from run to run); among the qualifying entries, the lowest median
    int best = 0;
    float best_ms = table[0].ms * 0.98f;
    const bool spread = table[0].ms_min > 0.f;
    for (int i = 1; i < count; ++i) {
        if (table[i].same_as == 0 || !(table[i].ms > 0.f) || !(table[i].ms < best_ms)) continue;
        if (spread && table[i].ms_max > 0.f && !(table[i].ms_max < table[0].ms_min)) continue;
        best = i; best_ms = table[i].ms;
    }
    return best;
}

int rz_autotune_apply(rz_ctx *c, const rz_tune_entry *e)
{
    if (int r = use(c)) return r;
    if (!e) return fail(RZ_ERR_INVALID, "rz_autotune_apply: null entry");
    const int sp = e->morph_split;
    if (sp != 0 && sp != 1 && sp != 2 && sp != 4 && sp != 8) return fail(RZ_ERR_INVALID, "rz_autotune_apply: morph_split %d", sp);
    if (e->grid_cap < 0 || e->inst_loop < -1 || e->inst_loop == 1 || e->inst_loop > 64) return fail(RZ_ERR_INVALID, "rz_autotune_apply: bad entry");
    if (e->inst_loop == 9 && !rz_has_all_variants())
        return fail(RZ_ERR_UNSUPPORTED, "rz_autotune_apply: inst_loop = 9 selects the register-resident crowd kernel, which the product library does not carry");
    // a caller who switched the crowd kernel off (inst_loop = 0) or chose the register form (9) keeps that choice: the entry's
    // pose-group size only applies where the search itself would have used one
    InstShape is;
    const bool instanced = inst_shape(c, &is);
    c->t_split = sp; c->t_grid_cap = e->grid_cap;
    if (instanced) c->t_instloop = e->inst_loop;
    c->tuned_by_search = true;
    return RZ_OK;
}

int rz_autotune(rz_ctx *c, uint32_t frames)
{
    rz_tune_entry table[32];
    int n = 0;
    if (int r = rz_autotune_measure(c, frames, table, 32, &n)) return r;
    if (n < 1) return RZ_OK;
    return rz_autotune_apply(c, &table[rz_autotune_pick(table, n)]);
}

int rz_set_tuning(rz_ctx *c, const char *key, int value)
{
    if (!c || !key) return fail(RZ_ERR_INVALID, "null argument");
    // variants that were measured slower everywhere are compiled into the tools-only build (make variants), not the product
    if (!rz_has_all_variants() && ((!strcmp(key, "unroll") && value == 4) || (!strcmp(key, "geo_lds") && value != 0) ||
                                   (!strcmp(key, "nontemporal") && value == 0) || (!strcmp(key, "inst_loop") && value == 9)))
        return fail(RZ_ERR_UNSUPPORTED, "%s = %d selects a kernel variant the product library does not carry (tools-only build: make -C reze-engine_amd/csrc variants)", key, value);
    if (!strcmp(key, "morph_split") || !strcmp(key, "grid_cap") || !strcmp(key, "inst_loop")) c->tuned_by_search = false;   // the caller owns the shape now
    if (!strcmp(key, "morph_split")) {
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8)
            return fail(RZ_ERR_INVALID, "morph_split must be 0 (auto),1,2,4,8");
        c->t_split = value;
    } else if (!strcmp(key, "unroll")) {
        if (value != 0 && value != 4 && value != 8) return fail(RZ_ERR_INVALID, "unroll must be 0 (auto), 4 or 8");
        c->t_unroll = value;
    } else if (!strcmp(key, "grid_cap")) {
        if (value < 0) return fail(RZ_ERR_INVALID, "grid_cap must be >= 0");
        c->t_grid_cap = value;
    } else if (!strcmp(key, "nontemporal")) {
        c->t_nt = value ? 1 : 0;
    } else if (!strcmp(key, "geo_lds")) {
        c->t_geo = value ? 1 : 0;
    } else if (!strcmp(key, "nt_store")) {
        c->t_nts = value < 0 ? -1 : (value ? 1 : 0);
    } else if (!strcmp(key, "out_cap")) {
        if (value < -1 || value > 2048) return fail(RZ_ERR_INVALID, "out_cap must be -1 (auto), 0 (off) or 64..2048 vertices per wave");
        c->t_outcap = value;
    } else if (!strcmp(key, "graph")) {
        if (value < 0 || value > 1) return fail(RZ_ERR_INVALID, "graph must be 0 or 1");
        c->t_graph = value;
    } else if (!strcmp(key, "dbg")) {
#ifdef RZ_ABLATE
        c->t_dbg = value;
#else
        // ablation modes (they make the kernels skip work, i.e. emit garbage) are compiled into the tools-only build only
        return fail(RZ_ERR_INVALID, "tuning key 'dbg' does not exist in the product library (tools-only build: make -C reze-engine_amd/csrc ablate)");
#endif
    } else if (!strcmp(key, "inst_loop")) {
        if (value < -1 || value == 1 || value > 64) return fail(RZ_ERR_INVALID, "inst_loop must be -1 (auto), 0 (off), 2..8 / 10..64 (poses per workgroup, LDS form) or 9 (register form)");
        c->t_instloop = value;
    } else if (!strcmp(key, "pose_prefetch")) {
        if (value < -1 || value > 1) return fail(RZ_ERR_INVALID, "pose_prefetch must be -1 (auto = on), 0 (a zero-copy frame never stages the next pose) or 1");
        c->t_prefetch = value;
    } else if (!strcmp(key, "inst_subsets")) {
        if (value < -1 || value > 1) return fail(RZ_ERR_INVALID, "inst_subsets must be -1 (auto = on), 0 (crowd frames always stage the whole palette) or 1");
        c->t_subsets = value;
    } else if (!strcmp(key, "fuse_fk")) {
        if (value < -1 || value > 1) return fail(RZ_ERR_INVALID, "fuse_fk must be -1 (auto), 0 (always rz_fk_kernel in front) or 1 (every device-animated single character)");
        c->t_fusefk = value;
    } else if (!strcmp(key, "zero_copy")) {
        if (value < -1 || value > 1) return fail(RZ_ERR_INVALID, "zero_copy must be -1 (auto = on for one character), 0 (every pose is copied to the device) or 1");
        c->t_zerocopy = value;
    } else if (!strcmp(key, "overlap")) {
        if (value < -1 || value > 1) return fail(RZ_ERR_INVALID, "overlap must be -1 (auto = off), 0 (off) or 1 (crowds: front kernels on the upload stream)");
        c->t_overlap = value;
    } else if (!strcmp(key, "inst_order")) {
        if (value != 0 && value != 1) return fail(RZ_ERR_INVALID, "inst_order must be 0 (an XCD takes one vertex run of every pose group) or 1 (every vertex run of its pose groups)");
        c->t_instorder = value;
    } else if (!strcmp(key, "inst_block")) {
        if (value != 0 && value != 256 && value != 512 && value != 1024) return fail(RZ_ERR_INVALID, "inst_block must be 0 (auto), 256, 512 or 1024 threads per workgroup");
        c->t_instblock = value;
        c->tuned_by_search = false;
    } else if (!strcmp(key, "fast")) {
        c->t_fast = value;        // -1 auto, 0 never (always prep kernel), 1 when possible
    } else {
        return fail(RZ_ERR_INVALID, "unknown tuning key '%s'", key);
    }
    return RZ_OK;
}

int rz_get_tuning(rz_ctx *c, const char *key, int *value)
{
    if (!c || !key || !value) return fail(RZ_ERR_INVALID, "null argument");
    if (!strncmp(key, "effective_", 10)) {
        // what the NEXT frame will launch: a crowd's plan depends on the run lists of its launch shape, so bring them up to
        // date first (as every entry point that launches frames does) instead of describing the whole-palette fallback
        if (int r = use(c)) return r;
        if (c->V && c->B) if (int r = ensure_run_subsets(c)) return r;
    }
    if (!strcmp(key, "morph_split")) *value = c->t_split;
    else if (!strcmp(key, "unroll")) *value = c->t_unroll;
    else if (!strcmp(key, "grid_cap")) *value = c->t_grid_cap;
    else if (!strcmp(key, "nontemporal")) *value = c->t_nt;
    else if (!strcmp(key, "geo_lds")) *value = c->t_geo;
    else if (!strcmp(key, "bones")) *value = (int)c->B;
    else if (!strcmp(key, "morphs")) *value = (int)c->M;
    else if (!strcmp(key, "instances")) *value = (int)c->I;
    else if (!strcmp(key, "verts")) *value = (int)c->V;
    else if (!strcmp(key, "nt_store")) *value = c->t_nts;
    else if (!strcmp(key, "fast")) *value = c->t_fast;
    else if (!strcmp(key, "morph_mode")) *value = c->morph_mode;
    else if (!strcmp(key, "effective_nt")) *value = make_plan(c).v.nt && c->morph_mode == 1 ? 1 : 0;
    else if (!strcmp(key, "effective_nt_store")) *value = make_plan(c).v.nts ? 1 : 0;
    else if (!strcmp(key, "effective_geo")) *value = make_plan(c).v.geo ? 1 : 0;
    else if (!strcmp(key, "effective_prep")) { const Plan pl = make_plan(c); *value = ((pl.prep || c->pose_local) && !pl.fuse_fk) ? 1 : 0; }
    else if (!strcmp(key, "effective_split")) *value = make_plan(c).v.S;
    else if (!strcmp(key, "effective_unroll")) *value = make_plan(c).v.U;
    else if (!strcmp(key, "effective_fast")) *value = make_plan(c).v.fast ? 1 : 0;
    else if (!strcmp(key, "inst_loop")) *value = c->t_instloop;
    else if (!strcmp(key, "inst_block")) *value = c->t_instblock;
    else if (!strcmp(key, "inst_order")) *value = c->t_instorder;
    else if (!strcmp(key, "overlap")) *value = c->t_overlap;
    else if (!strcmp(key, "zero_copy")) *value = c->t_zerocopy;
    else if (!strcmp(key, "fuse_fk")) *value = c->t_fusefk;
    else if (!strcmp(key, "effective_fuse_fk")) *value = make_plan(c).fuse_fk ? 1 : 0;
    else if (!strcmp(key, "pose_resident")) *value = (c->zc_cur < 0 || (c->world_resident && c->mw_resident && c->local_resident)) ? 1 : 0;
    else if (!strcmp(key, "effective_overlap")) *value = want_overlap(c, make_plan(c)) ? 1 : 0;
    else if (!strcmp(key, "effective_inst_block")) *value = make_plan(c).inst_block;
    else if (!strcmp(key, "out_cap")) *value = c->t_outcap;
    else if (!strcmp(key, "graph")) *value = c->t_graph;
    else if (!strcmp(key, "effective_out_cap")) *value = (int)make_plan(c).out_cap;
    else if (!strcmp(key, "effective_inst_group")) *value = make_plan(c).inst_group;
    else if (!strcmp(key, "effective_poses_per_wg")) *value = make_plan(c).poses_per_wg;
    else if (!strcmp(key, "effective_grid")) *value = (int)make_plan(c).grid_x;
    else if (!strcmp(key, "inst_subsets")) *value = c->t_subsets;
    else if (!strcmp(key, "all_variants")) *value = rz_has_all_variants() ? 1 : 0;
    else if (!strcmp(key, "pose_prefetch")) *value = c->t_prefetch;
    else if (!strcmp(key, "pose_staged")) {
        // did the helper of an earlier frame stage the CURRENT pose in device memory? (synchronises; for tests and tools)
        *value = 0;
        if (c->zc_tag && c->zc_seq_cur) {
            uint64_t tags[2] = {0, 0};
            HIP_TRY(hipSetDevice(c->device));
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(hipMemcpy(tags, c->zc_tag, sizeof tags, hipMemcpyDeviceToHost));
            *value = tags[c->pose_slot] == c->zc_seq_cur ? 1 : 0;
        }
    }
    else if (!strcmp(key, "effective_subsets")) *value = make_plan(c).subsets ? 1 : 0;
    else if (!strcmp(key, "effective_subset_bones")) *value = (int)make_plan(c).sub_bones;
    else if (!strcmp(key, "effective_inst_lds")) *value = (int)make_plan(c).inst_lds;
    else return fail(RZ_ERR_INVALID, "unknown tuning key '%s'", key);
    return RZ_OK;
}

#ifdef RZ_ALL_VARIANTS
// tools-only build (make variants), test hook — not part of the C ABI. close = 1: everything enqueued on the context's stream from
// here on waits behind a gate kernel; close = 0: the gate opens. (tests: frames are queued behind the gate, the NEXT pose is
// uploaded, the gate opens — the prefetch helper of the queued frame then finds that pose complete by construction.)
__attribute__((visibility("default"))) int rz_debug_gate(rz_ctx *c, int close)
{
    if (int r = use(c)) return r;
    if (!c->gate_host) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->gate_host), 64, hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&c->gate_dev), c->gate_host, 0));
        *c->gate_host = 1u;
    }
    if (close) {
        HIP_TRY(hipStreamSynchronize(c->stream));       // an earlier gate kernel has gone
        *reinterpret_cast<volatile uint32_t *>(c->gate_host) = 0u;
        std::atomic_thread_fence(std::memory_order_seq_cst);
        HIP_TRY(rz_launch_gate(c->gate_dev, c->stream));
    } else {
        std::atomic_thread_fence(std::memory_order_seq_cst);
        *reinterpret_cast<volatile uint32_t *>(c->gate_host) = 1u;
    }
    return RZ_OK;
}
#endif

#ifdef RZ_ABLATE
// tools-only build (make ablate): per-wave timeline of the frames that follow (tools/timeline.py). Not part of the C ABI.
__attribute__((visibility("default"))) int rz_debug_timeline_arm(rz_ctx *c, uint32_t *waves)
{
    if (int r = use(c)) return r;
    Plan pl;
    if (int r = frame_plan(c, &pl)) return r;
    const size_t wpw = pl.inst_group > 0 ? (size_t)pl.inst_block / 64 : 4;
    const size_t groups = pl.inst_group > 0 ? (c->I + pl.inst_group - 1) / pl.inst_group : c->I;
    const size_t n = ((size_t)pl.grid_x + 1) * groups * wpw;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (n > c->tl_waves) { dfree(c->tl); HIP_TRY(hipMalloc(&c->tl, n * 128)); c->tl_waves = n; }
    HIP_TRY(hipMemset(c->tl, 0, c->tl_waves * 128));
    c->t_dbg = 100;
    drop_graph(c);
    if (waves) *waves = (uint32_t)n;
    return RZ_OK;
}
__attribute__((visibility("default"))) int rz_debug_timeline_read(rz_ctx *c, unsigned long long *out, uint32_t waves)
{
    if (int r = use(c)) return r;
    if (!c->tl || waves > c->tl_waves || !out) return fail(RZ_ERR_INVALID, "no timeline armed");
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, c->tl, (size_t)waves * 128, hipMemcpyDeviceToHost));
    return RZ_OK;
}
#endif

int rz_output_ptrs(rz_ctx *c, void **pos, void **nrm, uint32_t *v_padded)
{
    if (int r = use(c)) return r;
    if (int r = ensure_outputs(c)) return r;
    if (pos) *pos = c->ext_pos ? c->ext_pos : c->out_pos;
    if (nrm) *nrm = c->ext_nrm ? c->ext_nrm : c->out_nrm;
    if (v_padded) *v_padded = c->Vp;
    return RZ_OK;
}

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
    if (c->I != 1) return fail(RZ_ERR_UNSUPPORTED, "instancing and vertex sharding are exclusive");
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
    uint32_t b0 = 0, n0 = 0;
    rz_shard_range(v_total, nranks, 0, &b0, &n0);
    c->chunk = round_up(n0, kShardGrain);
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
        if (c->I != 1) return fail(RZ_ERR_UNSUPPORTED, "instancing and vertex sharding are exclusive");
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
        if (c->I != 1) return fail(RZ_ERR_UNSUPPORTED, "instancing and vertex sharding are exclusive");
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
    uint32_t b0 = 0, n0 = 0;
    rz_shard_range(v_total, n, 0, &b0, &n0);
    const uint32_t chunk = round_up(n0, kShardGrain);
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
