// pose.cpp — per-frame inputs: the pinned staging ring, zero-copy slots and their prefetch protocol, copies (ctx.h).
#include "ctx.h"

using namespace rzi;

namespace rzi {

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

// Sequence number of a zero-copy pose, the value its slot header and — once staged — its device tag hold:
// ring epoch << 32 | pose kind << 30 | upload index (1-based, 30 bits). The KIND (0 world matrices, 1 local rotations, 2 local
// rotations + translations) is part of the number because a helper workgroup stages the next slot assuming the next pose has the
// layout and size of its own frame's: a pose of another kind can then never match what the helper expected. Sizes inside a kind
// only change with the skeleton / morph set / instance count, which start a new epoch.
uint64_t zc_seq(const rz_ctx *c, uint64_t upload_index_1, int kind)
{
    return ((uint64_t)c->zc_epoch << 32) | ((uint64_t)(kind & 3) << 30) | (upload_index_1 & 0x3fffffffull);
}

}  // namespace rzi

extern "C" {

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
    if (c->fk_stale) {                                  // a crowd frame that solved its hierarchy in LDS only: the solve as a kernel of its own, now
        if (int r = launch_fk(c, c->stream)) return r;
    }
    HIP_TRY(hipStreamSynchronize(c->up_stream));        // rz_fk_kernel may have written them on the front stream
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(world16, c->world + (size_t)instance * c->B * 16, (size_t)c->B * 16 * sizeof(float), hipMemcpyDeviceToHost));
    return RZ_OK;
}

}  // extern "C"
