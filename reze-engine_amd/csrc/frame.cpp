// frame.cpp — launching frames: front kernels + the deform / skin kernel, replay (hipGraph), two streams, timing, readbacks (ctx.h).
#include "ctx.h"

using namespace rzi;

namespace rzi {

int check_ready(rz_ctx *c)
{
    if (c->V == 0 || !c->geom) return fail(RZ_ERR_INVALID, "no mesh uploaded (rz_upload_mesh)");
    if (c->B == 0 || !c->inv_bind) return fail(RZ_ERR_INVALID, "no skeleton uploaded (rz_upload_skeleton)");
    if (!c->pose_set) return fail(RZ_ERR_INVALID, "no pose set (rz_set_pose)");
    return RZ_OK;
}

int launch_fk(rz_ctx *c, hipStream_t st)
{
    const RzFkParams fp = fk_params(c);
    const size_t lds = rz_fk_lds_bytes(fp);
    if (lds > 160 * 1024)
        return fail(RZ_ERR_UNSUPPORTED, "skeleton too large for the hierarchy solve on the device: %u bones need %zu B of LDS (116 B per bone + the pose's morph weights; the limit is 160 KB)", c->B, lds);
    HIP_TRY(rz_launch_fk(fp, c->I, st));
    c->palette_stale = false; c->fk_stale = false;
    return RZ_OK;
}

int launch_prep(rz_ctx *c, hipStream_t st)
{
    HIP_TRY(rz_launch_prep(prep_params(c), c->I, st));
    c->palette_stale = false;
    return RZ_OK;
}

// rz_read_world / rz_read_palette after a crowd frame that solved its hierarchy in LDS only (fk_stale): the solve runs as a kernel of its
// own — but only while the device-animated pose that frame deformed is still the current one. A new pose, skeleton or topology clears the
// flag where it goes away (upload_pose, rz_set_pose_sampled, rz_upload_skeleton*); this is the second lock on the same door: rz_fk_kernel
// on a pose block that holds world matrices, or on the records of a replaced skeleton, would overwrite the resident pose.
bool solve_on_demand(const rz_ctx *c) { return c->fk_stale && c->pose_local && c->has_topology && c->pose_set && c->fk_rec; }

// Everything a frame launches in front of the deform kernel: on-device FK (local-rotation poses) and/or the prep
// kernel. The FK kernel already writes the palette, so prep is only still needed for its morph compaction.
int launch_front(rz_ctx *c, const Plan &pl, hipStream_t st)
{
    if (pl.fuse_fk || pl.subfk) return RZ_OK;       // the deform / crowd kernel solves the hierarchy itself
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
    if (pl.inst_group > 0 && pl.subfk) {
        HIP_TRY(rz_launch_skin_instances_fk(p, subfk_params(c), pl.inst_group, (int)c->I, pl.verts_per_wg, pl.grid_x, pl.inst_block, pl.v.nts, (size_t)pl.inst_lds, c->stream));
        c->fk_stale = true;                 // neither world matrices nor palettes of this pose are in memory: formed on demand (rz_read_world / rz_read_palette)
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
    return c->t_overlap == 1 && c->I > 1 && c->morph_mode != 2 && !c->t_graph && (pl.prep || c->pose_local) && !pl.subfk;
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

}  // namespace rzi

extern "C" {

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
    const RzSubFk sf = subfk_params(c);
    h = fnv(h, &sf, sizeof sf);
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

// The K-step span of a benchmark's timed region, by a hipEvent pair ON the stream instead of the host's clock around
// enqueue + synchronize: at 20 steps of a 16.6 us shard frame the host's fixed cost per timed region (first launch, wake-up from the
// final wait: ~30 us, profiles/r6_bench_shard8_steps20.json) is 9 % of the region. `lead` untimed frames go in FRONT of the opening
// event, in the same call: the host enqueues the event and the first timed frames while the GPU is busy with them, so the K timed frames
// run back to back from the first one — a render loop never starts a frame on an idle GPU; without them the first timed frame waits for
// its own launch (~5 us behind the event, another 3 % of 20 shard frames).
int rz_time_span(rz_ctx *a, rz_ctx *b, uint32_t lead, uint32_t frames, double *span_ms)
{
    if (!span_ms || frames == 0) return fail(RZ_ERR_INVALID, "rz_time_span: bad arguments");
    *span_ms = 0.0;
    if (int r = use(a)) return r;
    float ms = 0.f;
    if (!b) {
        if (lead)
            if (int r = rz_deform_n(a, lead)) return r;
        HIP_TRY(hipEventRecord(a->ev0, a->stream));
        if (int r = rz_deform_n(a, frames)) return r;
        HIP_TRY(hipEventRecord(a->ev1, a->stream));
    } else {
        if (a == b || a->device != b->device) return fail(RZ_ERR_INVALID, "rz_time_span needs two different contexts on one device");
        // the span opens on a's stream — behind a's share of the lead frames AND b's (a waits for them) — and b's first timed frame waits
        // for it; it closes on a's stream behind b's last frame
        if (lead) {
            if (int r = rz_deform_pair(a, b, lead + (lead & 1u))) return r;         // (an even count: the timed frames start on a again)
            HIP_TRY(hipEventRecord(b->ev0, b->stream));
            HIP_TRY(hipStreamWaitEvent(a->stream, b->ev0, 0));
        }
        HIP_TRY(hipEventRecord(a->ev0, a->stream));
        HIP_TRY(hipStreamWaitEvent(b->stream, a->ev0, 0));
        if (int r = rz_deform_pair(a, b, frames)) return r;
        HIP_TRY(hipEventRecord(b->ev1, b->stream));
        HIP_TRY(hipStreamWaitEvent(a->stream, b->ev1, 0));
        HIP_TRY(hipEventRecord(a->ev1, a->stream));
    }
    HIP_TRY(hipEventSynchronize(a->ev1));
    HIP_TRY(hipEventElapsedTime(&ms, a->ev0, a->ev1));
    *span_ms = ms;
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
    if (solve_on_demand(c)) {
        // the last frame was a crowd frame that solved its hierarchy in the skin kernel's front: run the solve as a kernel of its own
        // now — the same functions on the same pose, the same bits (kernels/fk.hip.h)
        if (int r = launch_fk(c, c->stream)) return r;
    }
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

int rz_read_hull(rz_ctx *c, uint32_t instance, uint32_t v0, uint32_t n, float *pos3)
{
    if (int r = use(c)) return r;
    if (!c->edge || !c->out_hull) return fail(RZ_ERR_INVALID, "the outline hull is off (rz_upload_edge_scale)");
    if (instance >= c->I || (uint64_t)v0 + n > c->V || !pos3) return fail(RZ_ERR_INVALID, "bad hull read");
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (n) HIP_TRY(hipMemcpy(pos3, c->out_hull + ((size_t)instance * c->Vp + v0) * 3, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return RZ_OK;
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
    if ((pl.prep || c->pose_local) && !pl.fuse_fk && !pl.subfk) {
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
// tools-only build (make ablate): per-wave timeline of the frames that follow (tools/archive/timeline.py). Not part of the C ABI.
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

}  // extern "C"
