// ctx.h — what the translation units of libreze_deform.so's host side share: the context (struct rz_ctx), the error / HIP / RCCL
// plumbing, the frame plan, and the internal entry points of each unit. Nothing here is part of the C ABI (include/reze_deform.h);
// everything internal lives in namespace rzi with hidden visibility.
//   core.cpp    context life cycle, buffers every unit sizes, error state            rz_create / rz_destroy / rz_fork / rz_sync ...
//   upload.cpp  static data: mesh, skeleton, topology, morph targets, motion            rz_upload_* / rz_set_instances / rz_shard_range
//   pose.cpp    per-frame inputs: pinned ring, zero-copy slots, copies                  rz_set_pose* / rz_override_world / rz_read_world
//   plan.cpp    launch shapes and the parameter blocks of the kernels                   (no exports)
//   frame.cpp   launching frames, replay, timing, readbacks                             rz_deform* / rz_time_frames / rz_read*
//   tune.cpp    launch-shape search and the tuning keys                                 rz_autotune* / rz_set_tuning / rz_get_tuning
//   comm.cpp    RCCL binding, all-gather, peer-direct gather                            rz_comm_* / rz_allgather* / rz_gather_*
#pragma once
#include "../../include/reze_deform.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "deform_kernels.h"

#pragma GCC visibility push(hidden)
namespace rzi {

int fail(int code, const char *fmt, ...);
const char *last_error();

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            (void)hipGetLastError();    /* reported here: a later launch's hipGetLastError() must not see it again */ \
            return fail(e_ == hipErrorOutOfMemory ? RZ_ERR_OOM : RZ_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                 \
        }                                                                                           \
    } while (0)


inline uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

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
extern Rccl g_rccl;
int rccl_bind();

#define NCCL_TRY(expr)                                                                              \
    do {                                                                                            \
        ncclResult_t r_ = (expr);                                                                   \
        if (r_ != ncclSuccess)                                                                      \
            return fail(RZ_ERR_RCCL, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));            \
    } while (0)

}  // namespace rzi
using rzi::kStageSlots;

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
    // the hierarchy's static block (kernels/fk.hip.h): [B][4] bone records (topology | bind | the motion's track | ancestors of the first
    // two doubling rounds) + [an_M][2] vertex-morph records; rebuilt by rebuild_fk_static when the topology or the motion changes
    uint4 *fk_rec = nullptr;
    uint2 *fk_anc_more = nullptr;       // [fk_rounds - 2][B] ancestor tables of the doubling rounds beyond the second (depth > 16)
    int fk_rounds = 0;                  // radix-4 doubling rounds = ceil(log4(depth))
    std::vector<uint4> fk_host;         // host copy of the bone records (w2 is filled in from an_host_range)
    std::vector<uint4> an_host_range, an_host_mrec;     // the motion's per-bone track records / per-morph records (host copies)
    bool pose_local_t = false;
    // device-side motion sampling (rz_upload_animation / rz_set_pose_sampled)
    bool has_animation = false, pose_sampled = false;
    uint4 *an_feed_range = nullptr;     // (first key, end, first frame, last frame) per morph feed
    uint32_t *an_feed_off = nullptr;
    float *an_key_frame = nullptr, *an_key_pos = nullptr, *an_mkey_frame = nullptr, *an_mkey_weight = nullptr, *an_feed_ratio = nullptr;
    float4 *an_key_rot = nullptr;
    uint4 *an_key_interp = nullptr;
    uint32_t an_M = 0;                  // vertex-morph count the feeds were built for
    float *an_frames = nullptr;         // [I]
    bool frames_inline = false;         // one character: the frame rides in the kernel arguments (frame0), nothing is uploaded
    float frame0 = 0.0f;
    size_t an_frames_alloc = 0;          // the current local pose carries translations (behind the rotations in its slot)
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
    // kernels of pose f: stage_ev[ring slot] = the pose has landed (the compute stream waits for it), ev_free[k] = everything
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
    // Poses of more than 256 KB (crowds) travel on the upload stream into a ring of kBigBlocks device blocks of their own (same
    // layout as pose_blk), so the upload of pose f+1 runs under the frame of pose f. A block is overwritten kBigBlocks uploads after
    // it was filled; that its readers are done is proven like a zero-copy slot's: ONE event per kBigBlocks / 2 uploads, recorded on
    // the compute stream at upload time and polled by the host before the block's next tenant is enqueued — no per-frame "slot is
    // free" hand-off between the streams (round 4 had one: a record on the compute stream + a wait on the upload stream per frame,
    // measured at 5 - 8 us per frame: tools/archive/overlapbench, profiles/r5_overlapbench.txt). 8 x 3.3 MB for C4.
    static constexpr int kBigBlocks = 8;
    float *big_blk[kBigBlocks] = {};
    size_t big_floats = 0;
    uint64_t big_uploads = 0;
    hipEvent_t big_ev[2] = {nullptr, nullptr};
    uint64_t big_ev_seq[2] = {~0ull, ~0ull};
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
    void *stage_dev[kStageSlots] = {};  // the slots' device addresses (device-mapped ring), or null: this ring can only be copied from
    size_t stage_bytes = 0;
    // Crowd poses (more than 256 KB) are PULLED out of the ring slot by rz_pull_pose_kernel on the upload stream instead of copied by
    // hipMemcpyAsync, world matrices as their upper three rows (pose.cpp: upload_pose_copy)
    int t_fkplain = -1;                 // "fuse_fk_plain": -1 / 1 = the fused frame of a plain pose runs the specialised kernel variant, 0 = always the generic one
    int t_pull = -1;                    // "pose_pull": -1 = world-matrix poses, 1 = every pose of more than 256 KB, 0 = hipMemcpyAsync of the pose as the host handed it over
    bool last_upload_pulled = false, last_upload_rows = false;      // what the most recent copy upload did (rz_get_tuning: pose_pulled / pose_rows)
    hipEvent_t stage_ev[kStageSlots] = {};
    bool stage_used[kStageSlots] = {};
    int stage_next = 0;

    // Zero-copy poses (one character, <= 256 KB): rz_set_pose* only writes the pose into a slot of this pinned,
    // device-mapped ring — no copy is enqueued at all. The first frame's kernels read it over the host link (rz_fk_kernel
    // the local pose; the one-launch deform kernel the world matrices, whose workgroup 0 also leaves them in the device
    // pose block for the frames that replay the pose); anything that needs a device-resident pose first (rz_prep_kernel)
    // gets it through make_resident(). Measured on MI355X (tools/archive/uploadbench): a 16.6 KB hipMemcpyAsync in front of a
    // frame costs 18 us, two of <= 16 KB 9.5 us, reading the pinned slot from the kernel 7 us with the loads fully exposed.
    // A slot is reused kZcSlots (32) uploads later; one event per kZcSlots / 2 uploads (recorded on the compute stream at upload time) proves
    // its readers are done, so there is no per-frame marker either.
    // Slots of the ring: one hipEventRecord per kZcSlots / 2 uploads guards slot reuse, and a record costs ~1.4 us of stream time: measured
    // on a 1/8 shard of C5 (tools/archive/live_shard.py, per-frame-pose loop over the resident replay): 8 slots +0.86 us per frame, 16 slots
    // +0.50 us, 32 slots +0.37 us. 32 x <= 256 KB of pinned memory per context.
    static constexpr int kZcSlots = 32;
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

    // rz_map_pose / rz_commit_pose: the slot handed out to the caller (map_layout < 0 = nothing mapped), and what it was sized for
    int map_layout = -1;
    bool map_zc = false;                // a slot of the zero-copy ring (one character) or of the staging ring (crowds)
    int map_slot = 0;
    uint64_t map_upload = 0;            // zero-copy: the upload index (1-based) zc_open gave the slot
    uint32_t map_I = 0, map_B = 0, map_M = 0;
    char *map_ptr = nullptr;

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
    // Crowd frames of device-animated poses with the hierarchy solved in the skin kernel's front (kernels/crowd.hip:
    // rz_skin_instances_fk_kernel): per vertex run the closure of its named bones under "parent of", one 80-byte record per closure
    // slot — derived from the run lists and the hierarchy's static data, rebuilt when either changes (plan.cpp: ensure_subfk)
    uint4 *subfk_rec = nullptr;
    uint32_t *subfk_count = nullptr;
    size_t subfk_rec_alloc = 0, subfk_count_alloc = 0;
    uint32_t subfk_stride = 0, subfk_rounds = 0;
    bool subfk_valid = false;
    uint64_t subfk_sub_gen = 0, subfk_fk_gen = 0;   // what it was built against
    uint64_t sub_gen = 0, fk_gen = 0;               // bumped when the run lists / the hierarchy's static block are rebuilt
    bool fk_stale = false;              // the last crowd frame solved its hierarchy in LDS only: rz_read_world / rz_read_palette run rz_fk_kernel on demand
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

namespace rzi {

// ---- core.cpp ----
int use(rz_ctx *c);
int static_unlocked(const rz_ctx *c, const char *what);       // static data shared between a context and its forks cannot be replaced
template <class T> void dfree(T *&p)
{
    if (p) { (void)hipFree(p); p = nullptr; }
}
int poll_event(hipEvent_t ev, const char *what);
void drop_graph(rz_ctx *c);
// upload-time scratch that is released on every exit path (HIP_TRY returns early on failure)
template <class T> struct Scratch {
    T *p = nullptr;
    ~Scratch() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count) { return hipMalloc(&p, count * sizeof(T)); }
};
int ensure_outputs(rz_ctx *c);
void set_ring(rz_ctx *c, int slot);
void point_pose_slot(rz_ctx *c, int k);
void point_pose_at(rz_ctx *c, float *block);     // the current pose lives in `block` (pose_blk[k] or a block of the big-pose ring)
void free_big_ring(rz_ctx *c);
int ensure_pose_buffers(rz_ctx *c);
void free_animation(rz_ctx *c);
int rebuild_fk_static(rz_ctx *c);      // upload.cpp: the device block behind fk_rec from the host copies
void free_bone_morphs(rz_ctx *c);
void forget_search(rz_ctx *c);
void free_morphs(rz_ctx *c);
template <typename T> int to_device(T **dst, const void *src, size_t count)
{
    *dst = nullptr;
    HIP_TRY(hipMalloc(dst, std::max<size_t>(count, 4) * sizeof(T)));      // (never fewer than four elements: the specialised sampler asks for key 0 unpredicated — three floats of a position)
    if (count) HIP_TRY(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    return RZ_OK;
}

// ---- plan.cpp ----
struct Plan { RzVariant v; uint32_t grid_x, n_quads, quads_per_wave; bool prep, dma; int inst_group; uint32_t verts_per_wg; int poses_per_wg; uint32_t out_cap; int inst_block; bool fuse_fk; bool subsets; uint32_t sub_bones; uint64_t inst_lds; bool pf; uint32_t sp_cap; bool subfk; };
// Launch shape of an instanced, morph-free crowd frame (rz_skin_instances_kernel): G poses per workgroup share one decode of each
// vertex; the grid is (vertex runs, pose groups), `total` workgroups in all.
struct InstShape { int G, blk, blk_full; bool want_in_kernel; uint32_t per, runs; };
bool inst_shape(const rz_ctx *c, InstShape *s);
Plan make_plan(const rz_ctx *c);
int ensure_run_subsets(rz_ctx *c);
int ensure_subfk(rz_ctx *c);
RzSubFk subfk_params(const rz_ctx *c);
int frame_plan(rz_ctx *c, Plan *pl);
RzDeformParams deform_params(const rz_ctx *c, const Plan &pl);
RzPrepParams prep_params(const rz_ctx *c);
RzFkParams fk_params(const rz_ctx *c);
uint64_t algorithmic_bytes(const rz_ctx *c);

// ---- pose.cpp ----
const float *src_world(const rz_ctx *c);
const float *src_morph_w(const rz_ctx *c);
const float4 *src_local_q(const rz_ctx *c);
int make_resident(rz_ctx *c);
uint64_t zc_seq(const rz_ctx *c, uint64_t upload_index_1, int kind);

// ---- frame.cpp ----
int check_ready(rz_ctx *c);
int launch_fk(rz_ctx *c, hipStream_t st);
bool solve_on_demand(const rz_ctx *c);
int launch_prep(rz_ctx *c, hipStream_t st);
int launch_front(rz_ctx *c, const Plan &pl, hipStream_t st);
int launch_deform(rz_ctx *c, const Plan &pl);
bool want_overlap(const rz_ctx *c, const Plan &pl);
int set_overlap(rz_ctx *c, bool on);
int run_frame(rz_ctx *c, const Plan &pl);
hipStream_t front_stream(const rz_ctx *c);       // the stream per-frame inputs travel on and front kernels run on

// ---- comm.cpp ----
void drop_direct_gather(rz_ctx *c);

}  // namespace rzi
#pragma GCC visibility pop
