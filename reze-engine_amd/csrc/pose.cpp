// pose.cpp — per-frame inputs: the pinned staging ring, zero-copy slots and their prefetch protocol, copies (ctx.h).
#include "ctx.h"

#include <immintrin.h>

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
    if (need > c->stage_bytes || !c->stage[0]) {
        HIP_TRY(hipStreamSynchronize(c->up_stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        need = (need + 4095) / 4096 * 4096;
        for (int i = 0; i < kStageSlots; ++i) {
            if (c->stage[i]) { (void)hipHostFree(c->stage[i]); c->stage[i] = nullptr; c->stage_dev[i] = nullptr; }
            c->stage_used[i] = false;
        }
        c->stage_bytes = 0;
        for (int i = 0; i < kStageSlots; ++i) {
            // device-mapped, so that a crowd's pose can be pulled out of the slot by a kernel; a slot that cannot be mapped is still good for copies
            if (hipHostMalloc(&c->stage[i], need, hipHostMallocMapped) != hipSuccess) {
                (void)hipGetLastError();
                c->stage[i] = nullptr;
                HIP_TRY(hipHostMalloc(&c->stage[i], need, hipHostMallocDefault));
            } else if (hipHostGetDevicePointer(&c->stage_dev[i], c->stage[i], 0) != hipSuccess) {
                (void)hipGetLastError();
                c->stage_dev[i] = nullptr;
            }
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

// World matrices of a crowd travel as their upper three rows: 48 B per bone — the four columns' x y z, in the reference's
// column-major order (math.ts) — instead of 64. Only rows 0..2 of world * inverseBind ever reach a vertex (engine.ts:926-928 forms
// the product, vs() :262-272 keeps xyz), and they depend on rows 0..2 of the world matrix alone; the bottom row of an affine matrix
// is 0 0 0 1 and rz_pull_pose_kernel writes it back, so the device block holds what the host handed over — IF every bottom row is
// exactly that (bit patterns: -0 is not 0), which the packing loop checks in passing (returns false: the caller sends the pose as it
// is). Four bones per step: four 64-byte loads, three two-source permutes, three 64-byte NON-TEMPORAL stores — measured on the GPU
// box's EPYC 9575F (tools/archive/packbench, profiles/r5_packbench.txt) as fast as the memcpy it replaces while the ring is cache-resident
// (43 us for C4's 51 200 bones), and unlike cached stores it stays there when two contexts' rings (2 x 8 x 2.5 MB) no longer fit a
// CCD's L3: masked 48-byte stores then read every line for ownership first and the per-frame loop of a context and its fork went from
// 63 to 127-180 us per frame. (Needs AVX-512; without it the pose is not packed.)
__attribute__((target("avx512f"))) static bool pack_rows_avx512(const float *world, size_t bones, float *out)
{
    // out0 = m0[xyz of columns 0..3] m1[c0.xyz c1.x] | out1 = m1[c1.yz c2.xyz c3.xyz] m2[c0.xyz c1.xyz c2.xy] | out2 = m2[c2.z c3.xyz] m3[all twelve]
    const __m512i i0 = _mm512_setr_epi32(0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14, 16 + 0, 16 + 1, 16 + 2, 16 + 4);
    const __m512i i1 = _mm512_setr_epi32(5, 6, 8, 9, 10, 12, 13, 14, 16 + 0, 16 + 1, 16 + 2, 16 + 4, 16 + 5, 16 + 6, 16 + 8, 16 + 9);
    const __m512i i2 = _mm512_setr_epi32(10, 12, 13, 14, 16 + 0, 16 + 1, 16 + 2, 16 + 4, 16 + 5, 16 + 6, 16 + 8, 16 + 9, 16 + 10, 16 + 12, 16 + 13, 16 + 14);
    const __m512i bottom = _mm512_setr_epi32(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x3f800000);
    __m512i acc = _mm512_setzero_si512();
    size_t b = 0;
    if ((reinterpret_cast<uintptr_t>(out) & 63u) == 0) {
        for (; b + 4 <= bones; b += 4) {
            const float *w = world + b * 16;
            const __m512 m0 = _mm512_loadu_ps(w), m1 = _mm512_loadu_ps(w + 16), m2 = _mm512_loadu_ps(w + 32), m3 = _mm512_loadu_ps(w + 48);
            _mm512_stream_ps(out + b * 12, _mm512_permutex2var_ps(m0, i0, m1));
            _mm512_stream_ps(out + b * 12 + 16, _mm512_permutex2var_ps(m1, i1, m2));
            _mm512_stream_ps(out + b * 12 + 32, _mm512_permutex2var_ps(m2, i2, m3));
            // lanes 3, 7, 11, 15 of every matrix against 0 0 0 1
            const __m512i x01 = _mm512_or_si512(_mm512_xor_si512(_mm512_castps_si512(m0), bottom), _mm512_xor_si512(_mm512_castps_si512(m1), bottom));
            const __m512i x23 = _mm512_or_si512(_mm512_xor_si512(_mm512_castps_si512(m2), bottom), _mm512_xor_si512(_mm512_castps_si512(m3), bottom));
            acc = _mm512_or_si512(acc, _mm512_or_si512(x01, x23));
        }
    }
    __mmask16 bad = _mm512_mask_test_epi32_mask((__mmask16)0x8888, acc, acc);
    const __m512i pick = _mm512_setr_epi32(0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14, 3, 7, 11, 15);
    for (; b < bones; ++b) {                    // the last one to three bones
        const __m512 m = _mm512_loadu_ps(world + b * 16);
        _mm512_mask_storeu_ps(out + b * 12, (__mmask16)0x0fff, _mm512_permutexvar_ps(pick, m));
        bad |= _mm512_mask_cmpneq_epi32_mask((__mmask16)0x8888, _mm512_castps_si512(m), bottom);
    }
    _mm_sfence();                               // the streaming stores are visible before anything that follows (the pull's launch)
    return bad == 0;
}

static bool can_pack_rows()
{
    static const bool ok = __builtin_cpu_supports("avx512f");
    return ok;
}

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
// Header protocol of the pose prefetch: invalid while the pose is being written, its sequence number once it is complete (x86 stores
// retire in program order; the fences keep the compiler from moving them). World-matrix poses are prefetched by the one-launch frame's
// helper, local poses by the fused-hierarchy frame's (zc_seq: the kind is part of the number). zc_open takes the next slot and marks it
// invalid (RZ_ERR_UNSUPPORTED when no such memory can be had: the caller copies instead); the pose is written (lay_out_pose, or the
// caller of rz_map_pose); zc_publish writes the number and makes the slot the current pose.
static int zc_open(rz_ctx *c, const PoseParts &pp, int *zs)
{
    if (int r = zc_acquire(c, std::max<size_t>(std::max<size_t>((size_t)c->B * 64 + pp.mwb, pp.mwb + (size_t)c->B * 28), 4096), zs)) return r;
    char *slot = static_cast<char *>(c->zc_host[*zs]);
    *reinterpret_cast<volatile uint64_t *>(slot + c->zc_hdr_off) = 0;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    return RZ_OK;
}

static void zc_publish(rz_ctx *c, const PoseParts &pp, int zs, uint64_t upload_index_1)
{
    char *slot = static_cast<char *>(c->zc_host[zs]);
    volatile uint64_t *hdr = reinterpret_cast<volatile uint64_t *>(slot + c->zc_hdr_off);
    std::atomic_thread_fence(std::memory_order_seq_cst);
    const int kind = pp.local ? (pp.sbytes ? 2 : 1) : 0;
    const uint64_t seq = zc_seq(c, upload_index_1, kind);
    *hdr = seq;
    c->zc_seq_cur = seq;
    point_pose_slot(c, c->pose_slot ^ 1);   // where the pose will live once something makes it resident
    c->zc_cur = zs; c->zc_local = pp.local; c->zc_kind = kind; c->zc_total = pp.total;
    c->zc_mw_off = pp.local ? 0 : pp.pbytes; c->zc_lq_off = pp.mwb;
    c->world_resident = pp.local;           // a local pose has no world matrices to bring over: rz_fk_kernel writes them
    c->mw_resident = false;
    c->local_resident = !pp.local;
}

static int upload_pose_zero_copy(rz_ctx *c, const PoseParts &pp)
{
    int zs = 0;
    if (int r = zc_open(c, pp, &zs)) return r;
    lay_out_pose(c, pp, static_cast<char *>(c->zc_host[zs]));
    zc_publish(c, pp, zs, c->zc_uploads);       // zc_uploads is already this upload's index + 1
    return RZ_OK;
}

// A block of the big-pose ring for the next upload of more than 256 KB (ctx.h): (re)allocate the ring when the layout grew, record
// the one-in-four event, make sure the readers of the block's previous tenant (kBigBlocks uploads ago) are done.
static int big_acquire(rz_ctx *c, float **block)
{
    constexpr uint64_t N = rz_ctx::kBigBlocks, P = N / 2;
    const size_t Mq = std::max<uint32_t>(c->M, 1);
    const size_t need = (size_t)c->pose_alloc_I * c->pose_alloc_B * 16 + (((size_t)c->pose_alloc_I * std::max<size_t>(c->pose_alloc_M, Mq) + 3) / 4 * 4) +
                        (size_t)c->pose_alloc_I * c->pose_alloc_B * 7 + 4;
    if (!c->big_blk[0] || c->big_floats < need) {
        HIP_TRY(hipStreamSynchronize(c->up_stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        drop_graph(c);
        free_big_ring(c);
        for (uint64_t k = 0; k < N; ++k) {
            HIP_TRY(hipMalloc(&c->big_blk[k], need * sizeof(float)));
            HIP_TRY(hipMemsetAsync(c->big_blk[k], 0, need * sizeof(float), c->stream));
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->big_floats = need;
    }
    const uint64_t u = c->big_uploads;
    if (u % P == 0) {
        const int e = (int)((u / P) & 1);
        HIP_TRY(hipEventRecord(c->big_ev[e], c->stream));     // everything launched before upload u, i.e. every reader of uploads < u
        c->big_ev_seq[e] = u;
    }
    if (u >= N) {
        // previous tenant = upload u - N, read by frames launched before upload u - N + 1: covered by the event of the first multiple
        // of P that is >= u - N + 1 — it is <= u - P, the older of the two events kept (the arithmetic of zc_acquire)
        const uint64_t cand = (u - (N - 1) + (P - 1)) / P * P;
        const int e = (int)((cand / P) & 1);
        if (c->big_ev_seq[e] != cand) return fail(RZ_ERR_HIP, "big-pose ring bookkeeping is inconsistent (upload %llu)", (unsigned long long)u);
        if (int r = poll_event(c->big_ev[e], "big-pose ring")) return r;
    }
    *block = c->big_blk[u % N];
    c->big_uploads = u + 1;
    return RZ_OK;
}

// A pose that is copied (or pulled) to the device: it goes through a pinned ring slot into a device block the running frames do not read
// — on the upload stream for big poses, so the upload overlaps whatever the compute stream is still running; the compute stream then
// waits for it. Three steps, shared by rz_set_pose* (the library fills the slot: upload_pose_copy) and rz_map_pose / rz_commit_pose (the
// caller does): copy_begin takes the slot, the slot is filled, copy_send enqueues everything.
static size_t stage_need(const rz_ctx *c, const PoseParts &pp)
{
    return std::max<size_t>((size_t)c->I * c->B * 64 + pp.mwb, pp.mwb + (size_t)c->I * c->B * 28);
}

// Large poses (instanced crowds: MBs) take the upload stream and a block of the big-pose ring (ctx.h): nothing enqueued so far
// reads that block. Small ones that are copied at all (zero_copy = 0, small crowds) go down the compute stream itself, into the
// other of the two pose blocks — measured on C5, the two extra packets of a cross-stream hand-off (marker + barrier) cost 3 us
// more per frame than the copy they hide.
// Overlapped-front protocol (crowds, opt-in): EVERY per-frame input travels on the upload stream and is consumed there,
// by the front kernels — stream order is the only ordering needed, no event at all.
static bool pose_piped(const rz_ctx *c, const PoseParts &pp) { return !c->overlap_on && pp.total > (256u << 10); }

// How it crosses the host link (tools/archive/overlapbench, tools/pullbench: profiles/r5_overlapbench.txt, r5_pullbench.txt):
//  * world matrices of a crowd are PULLED out of the slot by rz_pull_pose_kernel (kernels/front.hip), three rows per bone: the
//    upload is what such a frame is bound by (3.28 MB: 84 us per hipMemcpyAsync back to back, 62 us pulled, 47 us pulled as rows),
//    and that the pull's 16 workgroups slow a concurrent skin kernel down (27 -> 35 us) hides under it;
//  * local rotations (a quarter of the bytes) are shorter than the frame they run under: the copy engine leaves that frame alone
//    (27.7 us per frame with the copy running, 35.5 us with the pull), so they stay with hipMemcpyAsync ("pose_pull" = 1 pulls them too).
static bool pose_pullable(const rz_ctx *c, const PoseParts &pp, int slot)
{
    return (pose_piped(c, pp) || c->overlap_on) && c->stage_dev[slot] && (pp.total & 3u) == 0;
}

static int copy_begin(rz_ctx *c, const PoseParts &pp, int *slot)
{
    return stage_acquire(c, stage_need(c, pp), slot);
}

// `rows`: the slot holds the world matrices as 48 B per bone (pack_rows_avx512's layout) followed by the weights; otherwise the pose in
// lay_out_pose's layout. `pull`: rz_pull_pose_kernel instead of hipMemcpyAsync (always with `rows`).
static int copy_send(rz_ctx *c, const PoseParts &pp, int slot, bool pull, bool rows)
{
    c->zc_cur = -1;
    c->zc_seq_cur = 0;
    c->zc_epoch++;        // this copy overwrites a pose block a helper may have staged and tagged: no later zero-copy pose may match that tag
    c->world_resident = c->mw_resident = c->local_resident = true;
    const bool piped = pose_piped(c, pp);
    hipStream_t us = (piped || c->overlap_on) ? c->up_stream : c->stream;
    float *block = nullptr;
    if (piped) {
        if (int r = big_acquire(c, &block)) return r;
    }
    // c->world / c->morph_w / c->local_q now name the pose's block under the current counts
    if (piped) point_pose_at(c, block);
    else point_pose_slot(c, (c->pose_slot & 1) ^ 1);
    void *dst = pp.local ? static_cast<void *>(c->morph_w) : static_cast<void *>(c->world);
    const size_t bones = (size_t)c->I * c->B;
    if (pull) HIP_TRY(rz_launch_pull_pose(c->stage_dev[slot], dst, rows ? (uint32_t)bones : 0u, rows ? pp.total - pp.pbytes : pp.total, us));
    else HIP_TRY(hipMemcpyAsync(dst, c->stage[slot], pp.total, hipMemcpyHostToDevice, us));
    c->last_upload_pulled = pull; c->last_upload_rows = rows;
    // one event says both "the ring slot may be written again" (the host polls it eight uploads later) and "the pose has landed" (the
    // compute stream waits for it): a record is ~1.4 us of stream time, and the upload stream is what a host-animated crowd is bound by
    HIP_TRY(hipEventRecord(c->stage_ev[slot], us));
    c->stage_used[slot] = true;
    if (piped) HIP_TRY(hipStreamWaitEvent(c->stream, c->stage_ev[slot], 0));
    return RZ_OK;
}

static int upload_pose_copy(rz_ctx *c, const PoseParts &pp)
{
    int slot = 0;
    if (int r = copy_begin(c, pp, &slot)) return r;
    char *st = static_cast<char *>(c->stage[slot]);
    const bool pull = pose_pullable(c, pp, slot) && (c->t_pull == 1 || (c->t_pull < 0 && !pp.local));
    const size_t bones = (size_t)c->I * c->B;
    bool rows = false;
    if (pull && !pp.local && can_pack_rows()) {
        rows = pack_rows_avx512(static_cast<const float *>(pp.primary), bones, reinterpret_cast<float *>(st));
        if (rows && c->M > 0) {
            char *st_mw = st + bones * 48;
            if (pp.morph_weights && pp.mb) memcpy(st_mw, pp.morph_weights, pp.mb); else memset(st_mw, 0, pp.mb);
            if (pp.mwb > pp.mb) memset(st_mw + pp.mb, 0, pp.mwb - pp.mb);
        }
    }
    if (!rows) lay_out_pose(c, pp, st);     // (a matrix that is not affine: the whole pose as it was handed over)
    return copy_send(c, pp, slot, pull, rows);
}

// What a new pose does to the context BEFORE its bytes move: the pose kind decides the plan, the plan decides which stream protocol the
// frame (and therefore this upload) follows.
static int pose_kind_changes(rz_ctx *c, bool local)
{
    c->pose_set = false;
    c->pose_local = local;
    c->pose_sampled = false;
    c->fk_stale = false;                // (it spoke of the pose this one replaces: rz_read_world / rz_read_palette must not solve THAT on demand any more)
    Plan upl;
    if (int r = frame_plan(c, &upl)) return r;          // the plan frames will use (run lists first), not the whole-palette fallback
    return set_overlap(c, want_overlap(c, upl));
}

static PoseParts pose_parts(const rz_ctx *c, const void *primary, size_t pbytes, const void *secondary, size_t sbytes, bool local, const float *morph_weights)
{
    PoseParts pp;
    pp.primary = primary; pp.pbytes = pbytes; pp.secondary = secondary; pp.sbytes = sbytes; pp.morph_weights = morph_weights; pp.local = local;
    pp.mb = (size_t)c->I * c->M * sizeof(float);
    pp.mwb = ((size_t)c->I * std::max<uint32_t>(c->M, 1) + 3) / 4 * 4 * sizeof(float);
    pp.total = local ? pp.mwb + pbytes + sbytes : pbytes + (c->M > 0 ? pp.mwb : 0);
    return pp;
}

static bool pose_zero_copy(const rz_ctx *c, const PoseParts &pp) { return !c->overlap_on && c->I == 1 && pp.total <= (256u << 10) && c->t_zerocopy != 0; }

// ordered compaction of the non-zero weights for the one-launch path (instance 0)
static void compact_morph_list(rz_ctx *c, const float *morph_weights)
{
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
}

// Shared tail of rz_set_pose / rz_set_pose_local.
static int upload_pose(rz_ctx *c, const void *primary, size_t pbytes, const void *secondary, size_t sbytes, bool local,
                       const float *morph_weights)
{
    c->map_layout = -1;                 // a pose handed over whole cancels a mapping that was never committed
    if (int r = pose_kind_changes(c, local)) return r;
    const PoseParts pp = pose_parts(c, primary, pbytes, secondary, sbytes, local, morph_weights);
    int rc = RZ_ERR_UNSUPPORTED;
    if (pose_zero_copy(c, pp)) {
        rc = upload_pose_zero_copy(c, pp);
        if (rc == RZ_ERR_UNSUPPORTED) c->t_zerocopy = 0;      // no pinned device-mapped memory: from now on every pose is copied
    }
    if (rc == RZ_ERR_UNSUPPORTED) rc = upload_pose_copy(c, pp);
    if (rc) return rc;
    c->pose_I = c->I;
    compact_morph_list(c, morph_weights);
    c->pose_set = true;
    return RZ_OK;
}

#ifdef RZ_ALL_VARIANTS
// tools-only build (make variants), test hook — not part of the C ABI: the host-side packing loop of a crowd's world matrices on its own
// (no GPU involved), so that the CPU suite can hold it to its contract. Returns 1 = packed, 0 = some bottom row is not 0 0 0 1 (the pose
// would travel whole), -1 = this host cannot pack (no AVX-512).
__attribute__((visibility("default"))) int rz_debug_pack_rows(const float *world16, uint32_t bones, float *rows12)
{
    if (!can_pack_rows()) return -1;
    return pack_rows_avx512(world16, bones, rows12) ? 1 : 0;
}
#endif

int rz_set_pose(rz_ctx *c, const float *world, const float *morph_weights)
{
    if (int r = use(c)) return r;
    if (c->B == 0) return fail(RZ_ERR_INVALID, "no skeleton uploaded");
    if (!world) return fail(RZ_ERR_INVALID, "null world matrices");
    if (int r = ensure_pose_buffers(c)) return r;
    return upload_pose(c, world, (size_t)c->I * c->B * 16 * sizeof(float), nullptr, 0, false, morph_weights);
}

// ---- caller-written poses (ABI 7) ----
// rz_map_pose hands out the very memory the next pose upload would have copied the caller's matrices INTO — the next slot of the pinned
// ring a crowd's pull kernel / the copy engine reads (poses of more than 256 KB), or of the zero-copy ring one character's frame reads
// in place — and rz_commit_pose does what is left of rz_set_pose: nothing is packed, nothing is copied on the host. Which ring, and
// whether the slot can hold the 48-byte rows form, is decided at map time from the state the context is in; if that state changed by
// commit time (another instance count, tuning keys) the commit falls back to a plain rz_set_pose FROM the mapped memory (WORLD16) or
// refuses (ROWS12: there is nothing on the host side that could expand the rows).
int rz_map_pose(rz_ctx *c, int layout, float **matrices, float **morph_weights)
{
    if (int r = use(c)) return r;
    if (matrices) *matrices = nullptr;
    if (morph_weights) *morph_weights = nullptr;
    if (!matrices) return fail(RZ_ERR_INVALID, "rz_map_pose: null out pointer");
    if (layout != RZ_POSE_WORLD16 && layout != RZ_POSE_ROWS12) return fail(RZ_ERR_INVALID, "rz_map_pose: layout must be RZ_POSE_WORLD16 or RZ_POSE_ROWS12");
    if (c->B == 0) return fail(RZ_ERR_INVALID, "no skeleton uploaded");
    if (int r = ensure_pose_buffers(c)) return r;
    c->map_layout = -1;
    const size_t bones = (size_t)c->I * c->B;
    // (the opt-in overlapped-front protocol moves every per-frame input onto the upload stream and decides that per pose KIND at upload
    // time: such a context hands its poses over with rz_set_pose)
    if (c->t_overlap == 1 || c->overlap_on) return fail(RZ_ERR_UNSUPPORTED, "rz_map_pose: not with the overlapped-front protocol (\"overlap\" = 1); hand the pose over with rz_set_pose");
    const PoseParts pp = pose_parts(c, nullptr, bones * 64, nullptr, 0, false, nullptr);
    char *st = nullptr;
    if (pose_zero_copy(c, pp)) {
        if (layout == RZ_POSE_ROWS12)
            return fail(RZ_ERR_UNSUPPORTED, "rz_map_pose: a pose of %zu bytes is read in place by its frame, which wants whole 4 x 4 matrices: map it as RZ_POSE_WORLD16 (rows are for poses of more than 256 KB)", pp.total);
        // Frames of the RESIDENT pose may be launched between this call and the commit. The ring's reuse proof (zc_acquire) counts every
        // reader of an earlier upload as launched before this upload's slot was taken — so such a frame must not be the first one of a
        // zero-copy pose, which reads its pinned slot: bring that pose into its device block now (one stream-ordered copy, rare).
        if (c->pose_set)
            if (int r = make_resident(c)) return r;
        int zs = 0;
        const int rc = zc_open(c, pp, &zs);
        if (rc == RZ_OK) { c->map_zc = true; c->map_slot = zs; c->map_upload = c->zc_uploads; st = static_cast<char *>(c->zc_host[zs]); }
        else if (rc == RZ_ERR_UNSUPPORTED) c->t_zerocopy = 0;       // no pinned device-mapped memory: from now on every pose is copied
        else return rc;
    }
    if (!st) {
        int slot = 0;
        if (int r = copy_begin(c, pp, &slot)) return r;
        if (layout == RZ_POSE_ROWS12 && !(pose_pullable(c, pp, slot) && c->t_pull != 0))
            return fail(RZ_ERR_UNSUPPORTED, "rz_map_pose: rows need the pull kernel (a pose of more than 256 KB, a device-mapped ring, \"pose_pull\" != 0): map this pose as RZ_POSE_WORLD16");
        c->map_zc = false; c->map_slot = slot;
        st = static_cast<char *>(c->stage[slot]);
    }
    const size_t mat_bytes = bones * (layout == RZ_POSE_ROWS12 ? 48 : 64);
    if (c->M > 0) {
        memset(st + mat_bytes, 0, pp.mwb);          // weights the caller does not write are zero
        if (morph_weights) *morph_weights = reinterpret_cast<float *>(st + mat_bytes);
    }
    *matrices = reinterpret_cast<float *>(st);
    c->map_layout = layout; c->map_I = c->I; c->map_B = c->B; c->map_M = c->M; c->map_ptr = st;
    return RZ_OK;
}

int rz_commit_pose(rz_ctx *c)
{
    if (int r = use(c)) return r;
    if (c->map_layout < 0) return fail(RZ_ERR_INVALID, "rz_commit_pose without rz_map_pose (a pose call in between cancels a mapping)");
    const int layout = c->map_layout;
    const bool rows = layout == RZ_POSE_ROWS12;
    c->map_layout = -1;
    if (c->map_I != c->I || c->map_B != c->B || c->map_M != c->M) return fail(RZ_ERR_INVALID, "rz_commit_pose: the crowd, skeleton or morph set changed since rz_map_pose");
    const size_t bones = (size_t)c->I * c->B;
    char *st = c->map_ptr;
    const float *mw = c->M > 0 ? reinterpret_cast<const float *>(st + bones * (rows ? 48 : 64)) : nullptr;
    const PoseParts pp = pose_parts(c, nullptr, bones * 64, nullptr, 0, false, nullptr);
    // the same decisions as at map time, on the state as it is NOW (a context under the overlapped-front protocol was refused at map time
    // and a world-matrix pose never switches it on by itself) — all of them BEFORE anything is changed: a refused commit leaves the
    // resident pose as it was
    if (c->t_overlap == 1 || c->overlap_on) return fail(RZ_ERR_UNSUPPORTED, "rz_commit_pose: the overlapped-front protocol was switched on since rz_map_pose; hand the pose over with rz_set_pose");
    const bool zc_now = pose_zero_copy(c, pp);
    bool same = c->map_zc ? (zc_now && c->zc_host[c->map_slot] == static_cast<void *>(st) && c->zc_uploads == c->map_upload)
                          : (!zc_now && c->stage[c->map_slot] == static_cast<void *>(st));
    const bool pull = !c->map_zc && same && pose_pullable(c, pp, c->map_slot) && c->t_pull != 0;       // ("pose_pull" = 0 since the map: rows are refused below)
    if (rows && !pull) same = false;
    if (!same) {
        if (rows) return fail(RZ_ERR_UNSUPPORTED, "rz_commit_pose: the context changed since rz_map_pose and this pose can no longer be pulled as rows: map it again");
        // a plain rz_set_pose out of the mapped memory (the slot it takes is the NEXT one of its ring, never the source)
        return upload_pose(c, st, bones * 64, nullptr, 0, false, mw);
    }
    if (int r = pose_kind_changes(c, false)) return r;
    if (c->map_zc) zc_publish(c, pp, c->map_slot, c->map_upload);
    else if (int r = copy_send(c, pp, c->map_slot, pull, rows)) return r;
    c->pose_I = c->I;
    compact_morph_list(c, mw);
    c->pose_set = true;
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
    c->fk_stale = false;
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
    if (solve_on_demand(c)) {                           // a crowd frame that solved its hierarchy in LDS only: the solve as a kernel of its own, now
        if (int r = launch_fk(c, c->stream)) return r;
    }
    HIP_TRY(hipStreamSynchronize(c->up_stream));        // rz_fk_kernel may have written them on the front stream
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(world16, c->world + (size_t)instance * c->B * 16, (size_t)c->B * 16 * sizeof(float), hipMemcpyDeviceToHost));
    return RZ_OK;
}

}  // extern "C"
