// kernels/deform_dense.hip — fused morph + skin of ONE mesh with DENSE morph targets (BASELINE configs C5, C3, the 1/8 shards): a frame
// that is one long read stream (828 B per vertex at 64 targets, 93 % of it the morph planes) and is bound by HBM reads.
//   S     morph-split: lanes per quad (1,2,4,8). S lanes of a wave cooperate on one quad; lane-slice s accumulates the active morphs
//         a = s, s+S, ... and the partial sums are combined with a __shfl_xor butterfly. S > 1 multiplies the number of waves in flight
//         when the per-GPU shard is small (8-GPU strong scaling); even the 1 M-vertex mesh streams 3 % faster with two lanes per quad.
//   U     morphs in flight per lane-slice iteration (3*U 16-byte loads issued back to back)
//   NT    nontemporal loads on the morph stream (read once per frame: do not keep it in cache)
//   NTS   nontemporal stores of the outputs
//   GEO   1 = rest geometry is read with 16-byte loads by the quad owner and transposed through the wave's LDS scratch (tools-only
//             build); 0 = only the morphed position goes through LDS and the vertex-per-lane phase reads normal / joints / weights
//             with 4-byte loads
//   FAST  single-instance frame in ONE launch: the palette (world * inverseBind, engine.ts:926-928) is formed by every workgroup
//         itself — each thread requests its bone's two matrices before anything else and does the 36 FMAs while the first group of
//         morph loads is in flight, one barrier before the first skin phase publishes it — and the active-morph list comes in the
//         kernel arguments (compacted on the host by rz_set_pose). No prep kernel, no launch boundary, no load in front of the morph
//         stream. !FAST reads the palette / list produced by rz_prep_kernel (instanced frames, > 128 active morphs) or, for a
//         device-animated character (fk_on), solves the hierarchy in its prologue.
// grid = (tiles capped, instances); block = 256. dynamic LDS = palette | active-morph list (!FAST) | per-wave transpose scratch | parked outputs.
//
// Two phases per tile, both fully coalesced:
//   phase 1 (lane = quad of 4 vertices, slice s of S): stream the active morph planes with 16-byte loads, FMA into 12 partial sums,
//           __shfl_xor butterfly across the S slices, add to the rest position, park the quad in the wave's LDS scratch (ds_write_b128).
//   phase 2 (lane = one vertex): read it back (conflict-free ds_read_b32), gather the four bones' 3x4 rows from the LDS palette, LBS,
//           normalize, 12 B + 12 B per lane — parked in LDS and written as 16-byte stores at the end of the wave's run when it fits.
// Tried and removed (NOTEBOOK.md): the skin phase's attributes by LDS-DMA at the top of the step (R4.3: C5 123.2 -> 128.5 us),
// dynamically claimed steps (R4.6), matrices behind the first morph group for every frame (R4.10).
#include "deform_parts.hip.h"

namespace {

// __launch_bounds__(256, 2): the persistent grid is two workgroups per CU (2 waves per SIMD), so the register allocator may use up
// to 256 VGPRs but not one more (a 257th would halve residency).
template <int S, int U, bool NT, bool NTS, bool GEO, bool FAST, int FKV = 0>
__global__ void __launch_bounds__(kBlock, 2) rz_deform_dense_kernel(const float *k_geom, const float *k_world, const float *k_inv_bind, const uint32_t k_bf,
                                                                    const uint32_t k_Vp, const uint32_t k_nq, const uint32_t k_qpw, const uint32_t *k_j01,
                                                                    const uint32_t *k_j23, const uint32_t *k_wq, const RzDeformParams p, const RzMorphList ml)
{
    // Where the wave's first loads take their addresses from. The small-frame kernel reads the PRELOADED leading arguments (SGPRs at
    // wave start: deform_parts.hip.h). For a frame that is one long stream that only pays at S = 8 (C3: 30 k vertices, 7.3 vs 7.6 us);
    // with fewer, longer-lived waves the kernel is FASTER reading everything out of `p` as in round 3 — measured in one process,
    // alternating builds (profiles/r5_ab_dense_entry.txt): 1/8 shard of C5 (S = 4) 17.52 -> 16.63 us, C5 (S = 2) 127.2 -> 126.0 us.
    constexpr bool LEAD = S == 8;
    const float *a_geom = LEAD ? k_geom : p.geom;
    const float *a_world = LEAD ? k_world : (FAST ? (p.st_tag ? p.st_world : p.world) : (p.fk_on ? reinterpret_cast<const float *>(p.fk.bone_rec) : nullptr));
    const float *a_inv_bind = LEAD ? k_inv_bind : (FAST ? p.inv_bind : reinterpret_cast<const float *>((uintptr_t)(p.fk_on ? p.fk.sample.M : 0)));
    const uint32_t a_Vp = LEAD ? k_Vp : p.Vp, a_nq = LEAD ? k_nq : p.n_quads, a_qpw = LEAD ? k_qpw : p.quads_per_wave;
    const uint32_t *a_j01 = LEAD ? k_j01 : p.joints01, *a_j23 = LEAD ? k_j23 : p.joints23, *a_wq = LEAD ? k_wq : p.weights;
    const uint32_t a_bf = LEAD ? k_bf : ((uint32_t)p.B | (p.pf_src ? 1u << 16 : 0u) | (p.st_tag ? 1u << 17 : 0u) | (p.world_copy ? 1u << 18 : 0u));
    constexpr int QPW = 64 / S;              // quads per wave
    constexpr int VW = 4 * QPW;              // vertices per wave per tile
    constexpr int NPL = GEO ? 9 : 3;         // scratch planes per wave
    constexpr int ROUNDS = (VW + 63) / 64;
    constexpr bool LDS_LIST = !FAST;

    const int tid = threadIdx.x;
    const int inst = blockIdx.y;
    const int lane = tid & 63, wave = tid >> 6;
    const int kB = (int)(a_bf & 0xffffu);            // == p.B
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *pal = reinterpret_cast<float4 *>(smem);                       // B*3 float4
    RZ_TL_DECL;
    RZ_STAMP(0);                 // entry

    // Zero-copy pose prefetch (see RzDeformParams): workgroup 0 of such a launch is the helper, the workers shift by one.
    const bool pf_on = (a_bf >> 16) & 1u;            // == p.pf_src != nullptr (only one-launch and fused-hierarchy frames ever carry one)
    if (pf_on && blockIdx.x == 0) {
        pose_prefetch_helper(p, tid);
        return;
    }
    const uint32_t wid = blockIdx.x - (pf_on ? 1u : 0u);                 // worker index of this workgroup
    // fused-hierarchy frame (!FAST, fk_on): the static records of this thread's bones and morph are asked for before anything else
    FkEarly fke;
    if (!FAST && a_world) fke = fk_issue_static(reinterpret_cast<const uint4 *>(a_world), kB, (int)(uintptr_t)a_inv_bind, tid);
    // THIS frame's pose: staged in device memory by the previous frame's helper, or still in its pinned slot. The answer is one
    // tag away, and waiting for it before asking for the matrices would put two memory latencies in a row in front of the
    // palette. So a frame that MAY find its pose staged (spec) asks for the tag and, at once, for the matrices of the staged
    // copy (the device pose block: valid memory whatever it holds); the tag is looked at when the palette is formed, and
    // only a miss then fetches the matrices from the pinned slot.
    const bool spec = FAST && ((a_bf >> 17) & 1u);                       // == p.st_tag != nullptr
    const float *world_in = a_world;                                     // == spec ? p.st_world : p.world (re-pointed at the pinned slot on a miss)
    const bool from_host = ((a_bf >> 18) & 1u) && !spec;                 // (p.world_copy != nullptr) the matrices are asked for over the host link up front

    // FAST: this thread's bone (tid < B covers the first 256 bones) — its world and inverse-bind matrices are
    // requested FIRST, as plain loads into registers, so they are the oldest entries of the vmcnt queue: the palette
    // math below only has to wait for them (a counted wait) while the morph loads issued after them stay in flight.
    float4 ew0, ew1, ew2, ew3, ei0, ei1, ei2, ei3;
    const bool early = FAST && tid < kB && RZ_DBG(p) != 3;      // dbg 3: ablation — no palette staging (output is garbage)
    // Zero-copy frame (world_copy != null: `world` is pinned HOST memory, a few microseconds away): vmcnt retires in order, so host
    // loads at the head of the queue would hold back the first morph FMAs; there the world matrices are requested BEHIND the first
    // morph group instead, and the palette is formed after the last group.
    const bool late_world = FAST && from_host;
    bool world_pending = early && late_world;
    // (the early loads stay predicated on `early` here — the latency-bound kernel loads unpredicated, deform_small.hip — because the
    // dense kernels have no register to spare: unpredicated they spill, 255 VGPRs)
    const int eb = tid;
    auto load_world = [&]() {
        const float4 *gw = reinterpret_cast<const float4 *>(world_in) + eb * 4;
        ew0 = gw[0]; ew1 = gw[1]; ew2 = gw[2]; ew3 = gw[3];
        world_pending = false;
    };
    if (FAST && early) {
        const float4 *gi = reinterpret_cast<const float4 *>(a_inv_bind) + eb * 4;
        if (!late_world) load_world();
        ei0 = gi[0]; ei1 = gi[1]; ei2 = gi[2]; ei3 = gi[3];
    }
    const int s = lane / QPW;                // morph slice of this lane
    const int qi = lane % QPW;
    const size_t Vp = a_Vp;
    const size_t plane4 = Vp / 4;            // float4 per plane
    // persistent, evenly balanced partition: every wave of the grid owns one contiguous run of quads
    // (a multiple of 8 quads = 128 B per plane) and walks it QPW quads at a time; the last step is masked.
    const uint32_t wave_global = wid * (kBlock / 64) + wave;
    const size_t q_begin = (size_t)wave_global * a_qpw;
    const size_t q_end = min((size_t)a_nq, q_begin + a_qpw);

    // rest geometry of the quad (slice 0 only), issued at the top of a step so it overlaps the morph stream. The skin phase's
    // attribute loads stay behind the morph phase: the kernel lives at 245 VGPRs.
    float4 gx, gy, gz, gnx, gny, gnz;
    uint4 gj01, gj23, gw;
    auto issue = [&](const size_t qw) {
        const size_t q = qw + qi;
        if (s == 0 && q < q_end) {
            const float4 *G = reinterpret_cast<const float4 *>(a_geom) + q;
            gx = G[0]; gy = G[plane4]; gz = G[2 * plane4];
            if (GEO) {
                gnx = G[3 * plane4]; gny = G[4 * plane4]; gnz = G[5 * plane4];
                gj01 = reinterpret_cast<const uint4 *>(a_j01)[q];
                gj23 = reinterpret_cast<const uint4 *>(a_j23)[q];
                gw = reinterpret_cast<const uint4 *>(a_wq)[q];
            }
        }
    };

    // ---- from here on the kernel reads `p` (scalar loads of the kernel arguments: the loads above are in flight under them) ----
    uint32_t *s_idx = reinterpret_cast<uint32_t *>(smem + (size_t)p.B * 48);   // Mpad   (LDS_LIST)
    float *s_w = reinterpret_cast<float *>(s_idx + (LDS_LIST ? p.Mpad : 0));
    float *scratch_all = s_w + (LDS_LIST ? p.Mpad : 0);                   // 16-B aligned: Mpad % 4 == 0
    const uint64_t st_tagv = spec ? *p.st_tag : 0ull;                   // requested here, compared later (workgroup-uniform)
    int fused_count = 0;
    if (!FAST && (FKV == 1 || FKV == 2 || p.fk_on)) {
        // FUSED single-character frame: hierarchy solve (and motion sampling) as this workgroup's prologue (FKV 1 / 2: specialised for a
        // plain uploaded / sampled pose, kernels/fk.hip.h), then the ordered compaction of the pose's morph weights into the LDS list
        __shared__ int fz_cnt[kBlock / 64];
        float *lds_mw = fused_hierarchy_prologue<true, (FKV == 1 || FKV == 2) ? FKV : 0>(p.fk, fke, p.st_tag, p.st_expect, p.st_morph_w, p.morph_w, p.morph_w_copy, p.M, pal, scratch_all, wid, RZ_TL_FK);
        fused_count = compact_active(lds_mw, p.M, p.Mpad, s_idx, s_w, fz_cnt);
        __syncthreads();
    } else if (!FAST && FKV != 1 && FKV != 2) {
        const float4 *gpal = p.palette + (size_t)inst * p.B * 3;
        for (int i = tid; i < p.B * 3; i += kBlock) pal[i] = gpal[i];
        const uint32_t *gi = p.act_idx + (size_t)inst * p.Mpad;
        const float *gw = p.act_w + (size_t)inst * p.Mpad;
        for (int i = tid; i < p.Mpad; i += kBlock) { s_idx[i] = gi[i]; s_w[i] = gw[i]; }
        __syncthreads();
    }

    RZ_STAMP(1);                 // prologue done (hierarchy solve / staged palette + morph list)
    const int count = FAST ? ml.count : ((FKV == 1 || FKV == 2 || p.fk_on) ? fused_count : p.act_count[inst]);
    float *scr = scratch_all + (size_t)wave * NPL * VW;
    const uint32_t bmax = (uint32_t)(p.B - 1);
    float *opos = p.out_pos + (size_t)inst * Vp * 3;
    float *onrm = p.out_nrm + (size_t)inst * Vp * 3;
    bool need_palette = FAST && RZ_DBG(p) != 3;
    float bb[6] = { __builtin_inff(), __builtin_inff(), __builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff() };

    auto palette_rows = [&](int b, const float4 &a0, const float4 &a1, const float4 &a2, const float4 &a3, const float4 &b0,
                            const float4 &b1, const float4 &b2, const float4 &b3) {
        palette_rows_to(pal, (wid == 0 && p.palette) ? p.palette : nullptr, b, a0, a1, a2, a3, b0, b1, b2, b3);
    };
    // executed once per wave, wherever the first step has its loads in flight; no barrier here
    // zero-copy first frame: `world` is pinned host memory; workgroup 0 leaves the matrices in device memory for the replays
    bool keep_world = FAST && wid == 0 && from_host;     // (a staged pose already sits where world_copy points)
    auto form_palette = [&]() {
        if (world_pending) load_world();          // no morph group ran in front of us
        if (spec && st_tagv != p.st_expect) {
            // miss: the previous frame's helper did not stage this pose (the host was not ahead): it is in its pinned slot
            world_in = p.world;
            keep_world = wid == 0 && p.world_copy != nullptr;
            if (early) load_world();
        }
        if (early) {
            palette_rows(tid, ew0, ew1, ew2, ew3, ei0, ei1, ei2, ei3);
            if (keep_world) { float4 *d = reinterpret_cast<float4 *>(p.world_copy) + tid * 4; d[0] = ew0; d[1] = ew1; d[2] = ew2; d[3] = ew3; }
        }
        for (int b = tid + kBlock; b < p.B; b += kBlock) {      // bones beyond the first 256: plain loads, late (they hide under the morph stream)
            const float4 *gw = reinterpret_cast<const float4 *>(world_in) + b * 4;
            const float4 *gi = reinterpret_cast<const float4 *>(p.inv_bind) + b * 4;
            const float4 w0 = gw[0], w1 = gw[1], w2 = gw[2], w3 = gw[3];
            palette_rows(b, w0, w1, w2, w3, gi[0], gi[1], gi[2], gi[3]);
            if (keep_world) { float4 *d = reinterpret_cast<float4 *>(p.world_copy) + b * 4; d[0] = w0; d[1] = w1; d[2] = w2; d[3] = w3; }
        }
        need_palette = false;
    };
    bool need_sync = FAST && RZ_DBG(p) != 3;          // one workgroup barrier publishes the palette before the first phase 2

    // write batching (deform_parts.hip.h: flush_parked)
    const uint32_t cap = p.out_cap;
    float *ob_pos = scratch_all + (size_t)(kBlock / 64) * NPL * VW + (size_t)wave * cap * 6;
    float *ob_nrm = ob_pos + (size_t)cap * 3;
    uint32_t ob_fill = 0;               // vertices parked
    size_t ob_v0 = q_begin * 4;         // global vertex index of the first parked vertex
    auto flush_out = [&]() {
        flush_parked<NTS>(opos, onrm, ob_pos, ob_nrm, ob_v0, ob_fill, lane);
        ob_v0 += ob_fill;
        ob_fill = 0;
    };

    // One step = QPW quads. The body is instantiated twice: FIRST (the run's first step, which also forms the
    // palette from the early-loaded matrices) and the steady-state form, where those 32 registers are dead.
    auto step = [&](const size_t qw, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const size_t q = qw + qi;                                    // this lane's quad
        const bool live = q < q_end;
        float4 ax = make_float4(0.f, 0.f, 0.f, 0.f), ay = ax, az = ax;

        issue(qw);

        if (live) {
            const float4 *D = reinterpret_cast<const float4 *>(p.dense) + q;
            // slice s of S accumulates the active morphs a = s, s+S, s+2S, ... in ascending order.
            // The loop counter a0 is wave-uniform, so on the FAST path the list entries are fetched with
            // scalar loads straight from the kernel arguments and each lane picks its slice's entry with
            // v_cndmask — no vector-memory load sits in front of the morph stream.
            auto entry = [&](int base, uint32_t &m, float &w) {
                if (FAST) {
                    // (The compiler folds the select chain below into ONE per-lane indexed load ml.idx[base + s] from the kernel-argument
                    // segment. Keeping the entries in SGPRs instead — readfirstlane + v_cndmask — was measured twice under fresh-context
                    // alternation: -0.05 % / +3.5 % in round 5's two sessions, +0.6 % on C5 and +1.9 % on the N = 2 shard in 10 of 10
                    // rounds in round 6 (profiles/r6_ab_pin.txt): not adopted, the switch is gone.)
                    m = (uint32_t)ml.idx[base]; w = ml.w[base];
#pragma unroll
                    for (int k = 1; k < S; ++k) {
                        const uint32_t mk = (uint32_t)ml.idx[base + k];
                        const float wk = ml.w[base + k];
                        m = (s == k) ? mk : m; w = (s == k) ? wk : w;
                    }
                } else {
                    m = s_idx[base + s]; w = s_w[base + s];
                }
            };
            int a0 = 0;
            // full groups of U morphs per slice: 3*U independent 16-byte loads in flight per lane
            for (; a0 + U * S <= count; a0 += U * S) {
                float4 dx[U], dy[U], dz[U];
                float w[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    uint32_t m;
                    entry(a0 + u * S, m, w[u]);
                    const float4 *d = D + (size_t)m * 3 * plane4;
                    dx[u] = ld_stream(d, NT);
                    dy[u] = ld_stream(d + plane4, NT);
                    dz[u] = ld_stream(d + 2 * plane4, NT);
                }
                if (FAST && FIRST && world_pending) load_world();     // zero-copy frame: behind the first group's loads
                // first group of the first step: the palette math overlaps the 3*U loads just issued. On a zero-copy frame the
                // matrices come over the host link (a few microseconds): there the palette waits until the LAST group has
                // issued its loads, so the whole morph stream of the step is in flight under that latency.
                if (FAST && FIRST && need_palette && (!from_host || a0 + 2 * U * S > count)) form_palette();
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    ax.x = fmaf(w[u], dx[u].x, ax.x); ax.y = fmaf(w[u], dx[u].y, ax.y);
                    ax.z = fmaf(w[u], dx[u].z, ax.z); ax.w = fmaf(w[u], dx[u].w, ax.w);
                    ay.x = fmaf(w[u], dy[u].x, ay.x); ay.y = fmaf(w[u], dy[u].y, ay.y);
                    ay.z = fmaf(w[u], dy[u].z, ay.z); ay.w = fmaf(w[u], dy[u].w, ay.w);
                    az.x = fmaf(w[u], dz[u].x, az.x); az.y = fmaf(w[u], dz[u].y, az.y);
                    az.z = fmaf(w[u], dz[u].z, az.z); az.w = fmaf(w[u], dz[u].w, az.w);
                }
            }
            for (; a0 < count; a0 += S) {     // remainder, one morph per slice at a time (list is zero-padded)
                uint32_t m;
                float w;
                entry(a0, m, w);
                if (a0 + s < count) {
                    const float4 *d = D + (size_t)m * 3 * plane4;
                    float4 dx = ld_stream(d, NT), dy = ld_stream(d + plane4, NT), dz = ld_stream(d + 2 * plane4, NT);
                    ax.x = fmaf(w, dx.x, ax.x); ax.y = fmaf(w, dx.y, ax.y); ax.z = fmaf(w, dx.z, ax.z); ax.w = fmaf(w, dx.w, ax.w);
                    ay.x = fmaf(w, dy.x, ay.x); ay.y = fmaf(w, dy.y, ay.y); ay.z = fmaf(w, dy.z, ay.z); ay.w = fmaf(w, dy.w, ay.w);
                    az.x = fmaf(w, dz.x, az.x); az.y = fmaf(w, dz.y, az.y); az.z = fmaf(w, dz.z, az.z); az.w = fmaf(w, dz.w, az.w);
                }
            }
        }
        if (S > 1) {
            // combine the S partial sums of each quad: __shfl_xor butterfly over the slice bits of the
            // lane id (every lane takes part; lanes past the end of the run carry zeros)
#pragma unroll
            for (int off = QPW; off < 64; off <<= 1) {
                ax.x += __shfl_xor(ax.x, off); ax.y += __shfl_xor(ax.y, off);
                ax.z += __shfl_xor(ax.z, off); ax.w += __shfl_xor(ax.w, off);
                ay.x += __shfl_xor(ay.x, off); ay.y += __shfl_xor(ay.y, off);
                ay.z += __shfl_xor(ay.z, off); ay.w += __shfl_xor(ay.w, off);
                az.x += __shfl_xor(az.x, off); az.y += __shfl_xor(az.y, off);
                az.z += __shfl_xor(az.z, off); az.w += __shfl_xor(az.w, off);
            }
        }

        if (FIRST) RZ_STAMP(2);       // first step: morph phase done (its loads have landed)
        if (FAST && FIRST && need_palette) form_palette();            // no morph group ran (nothing active)
        if (FAST && FIRST && need_sync) { __syncthreads(); need_sync = false; }   // the palette of every wave is in LDS
        if (FIRST) RZ_STAMP(3);       // first step: palette published

        // ---- park the quad in the wave's scratch: plane-major [NPL][VW] dwords ----
        if (s == 0 && live) {
            float4 *sc4 = reinterpret_cast<float4 *>(scr) + qi;
            sc4[0 * QPW] = make_float4(gx.x + ax.x, gx.y + ax.y, gx.z + ax.z, gx.w + ax.w);
            sc4[1 * QPW] = make_float4(gy.x + ay.x, gy.y + ay.y, gy.z + ay.z, gy.w + ay.w);
            sc4[2 * QPW] = make_float4(gz.x + az.x, gz.y + az.y, gz.z + az.z, gz.w + az.w);
            if (GEO) {
                sc4[3 * QPW] = gnx; sc4[4 * QPW] = gny; sc4[5 * QPW] = gnz;
                uint4 *su4 = reinterpret_cast<uint4 *>(scr) + qi;
                su4[6 * QPW] = gj01; su4[7 * QPW] = gj23; su4[8 * QPW] = gw;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        const size_t vw0 = qw * 4;     // first vertex of this wave's step
        const int v_live = RZ_DBG(p) == 4 ? 0 : (int)min((size_t)VW, (q_end - qw) * 4);   // dbg 4: ablation — morph phase only

        // ---- phase 2: one vertex per lane ----
        auto skin_round = [&](const int r) {
            const int vl = r * 64 + lane;
            if (vl < v_live) {
                const size_t v = vw0 + vl;
                const float x = scr[0 * VW + vl], y = scr[1 * VW + vl], z = scr[2 * VW + vl];
                float nx, ny, nz;
                uint32_t j01, j23, wq;
                if (GEO) {
                    nx = scr[3 * VW + vl]; ny = scr[4 * VW + vl]; nz = scr[5 * VW + vl];
                    const uint32_t *su = reinterpret_cast<const uint32_t *>(scr);
                    j01 = su[6 * VW + vl]; j23 = su[7 * VW + vl]; wq = su[8 * VW + vl];
                } else {
                    nx = p.geom[3 * Vp + v]; ny = p.geom[4 * Vp + v]; nz = p.geom[5 * Vp + v];
                    j01 = p.joints01[v]; j23 = p.joints23[v]; wq = p.weights[v];
                }
                const Skinned o = skin_vertex(pal, x, y, z, nx, ny, nz, j01, j23, wq, bmax);
                emit_vertex<NTS, FKV == 0>(p, o, v, inst, Vp, cap, ob_pos, ob_nrm, (ob_fill + vl) * 3, opos, onrm, bb);
            }
        };
#pragma unroll 1
        for (int r = 0; r < ROUNDS; ++r) skin_round(r);
        __builtin_amdgcn_wave_barrier();
        if (FIRST) RZ_STAMP(4);       // first step: skin phase issued
        if (cap) {
            ob_fill += (uint32_t)v_live;
            if (ob_fill + VW > cap) flush_out();      // the next step might not fit
        }
    };
    {
        size_t qw = q_begin;
        if (qw < q_end) { step(qw, std::true_type{}); qw += QPW; }
        for (; qw < q_end; qw += QPW) step(qw, std::false_type{});
    }
    RZ_STAMP(5);                 // last step done
    if (cap && ob_fill) flush_out();
    if (FAST && need_palette) form_palette();    // a wave with an empty run still owes the workgroup its bones ...
    if (FAST && need_sync) __syncthreads();      // ... and its barrier
    if (FKV == 0 && p.aabb) aabb_commit(p, inst, bb, lane, tid, wid, q_begin < q_end);
    RZ_TL_FLUSH(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (kBlock / 64) + wave);
}

}  // namespace

// ---- launch: which instantiations the library carries ----
// The PRODUCT instantiates what a plan can select by default or through rz_autotune: rest geometry by 4-byte loads (GEO = false),
// nontemporal morph loads (NT = true), 8 morphs in flight (U = 8) — 16 kernels. The variants measured slower everywhere (GEO = true,
// NT = false, U = 4: profiles/archive/r1_*sweep*) live in the tools-only build (-DRZ_ALL_VARIANTS, `make variants`), where the parity tests
// still cover every one of them; in the product rz_set_tuning refuses the keys that would select them.
#ifdef RZ_ALL_VARIANTS
constexpr bool kAllVariants = true;
#else
constexpr bool kAllVariants = false;
#endif

template <int S, int U, bool NT, bool NTS, bool GEO, bool FAST, int FKV = 0>
static hipError_t launch_one(const RzDeformParams &p, const RzMorphList &ml, dim3 grid, size_t lds, hipStream_t st)
{
    // FKV — which variant of the kernel (the shapes a plan selects only): 0 = everything compiled in (the fused consumers: outline hull,
    // bounding box); 3 = without them; 1 / 2 = without them AND the hierarchy solve specialised for a plain uploaded / sampled pose
    if constexpr (!GEO && NT && U == 8 && FKV == 0) {
        if (!p.edge && !p.aabb) {
            if constexpr (!FAST) {
                if (p.fk_on && p.fk_kind == 1) return launch_one<S, U, NT, NTS, GEO, FAST, 1>(p, ml, grid, lds, st);
                if (p.fk_on && p.fk_kind == 2) return launch_one<S, U, NT, NTS, GEO, FAST, 2>(p, ml, grid, lds, st);
            }
            return launch_one<S, U, NT, NTS, GEO, FAST, 3>(p, ml, grid, lds, st);
        }
    }
    auto k = rz_deform_dense_kernel<S, U, NT, NTS, GEO, FAST, FKV>;
    if (p.B > 0xffff) return hipErrorInvalidValue;      // k_bf carries the bone count in 16 bits (the 48 B per bone LDS palette keeps it far below today)
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    // the leading arguments (kernel-argument preload: deform_parts.hip.h) repeat fields of `p`; k_world is the pose the kernel asks for
    // FIRST — the copy a helper may have staged when the frame looks for one, else p.world
    // (!FAST: the hierarchy's static block and the motion's morph count ride in the two matrix slots, deform_parts.hip.h)
    const float *k_world = FAST ? (p.st_tag ? p.st_world : p.world) : (p.fk_on ? reinterpret_cast<const float *>(p.fk.bone_rec) : nullptr);
    const float *k_inv_bind = FAST ? p.inv_bind : reinterpret_cast<const float *>((uintptr_t)(p.fk_on ? p.fk.sample.M : 0));
    hipLaunchKernelGGL(k, grid, dim3(kBlock), lds, st, p.geom, k_world, k_inv_bind, rz_deform_k_bf(p, grid.x), p.Vp, p.n_quads, p.quads_per_wave, p.joints01, p.joints23, p.weights, p, ml);
    return hipGetLastError();
}

template <int S, int U, bool NT, bool NTS>
static hipError_t launch_gf(const RzDeformParams &p, const RzMorphList &ml, const RzVariant &v, dim3 grid, size_t lds, hipStream_t st)
{
    if constexpr (kAllVariants) {
        if (v.geo) return v.fast ? launch_one<S, U, NT, NTS, true, true>(p, ml, grid, lds, st)
                                 : launch_one<S, U, NT, NTS, true, false>(p, ml, grid, lds, st);
    } else if (v.geo) return hipErrorInvalidValue;
    return v.fast ? launch_one<S, U, NT, NTS, false, true>(p, ml, grid, lds, st)
                  : launch_one<S, U, NT, NTS, false, false>(p, ml, grid, lds, st);
}

template <int S, int U>
static hipError_t launch_nt(const RzDeformParams &p, const RzMorphList &ml, const RzVariant &v, dim3 grid, size_t lds, hipStream_t st)
{
    if (v.nt || !kAllVariants) {
        if (!v.nt) return hipErrorInvalidValue;
        return v.nts ? launch_gf<S, U, true, true>(p, ml, v, grid, lds, st) : launch_gf<S, U, true, false>(p, ml, v, grid, lds, st);
    }
    if constexpr (kAllVariants)
        return v.nts ? launch_gf<S, U, false, true>(p, ml, v, grid, lds, st) : launch_gf<S, U, false, false>(p, ml, v, grid, lds, st);
    return hipErrorInvalidValue;
}

template <int S>
static hipError_t launch_split(const RzDeformParams &p, const RzMorphList &ml, const RzVariant &v, dim3 grid, size_t lds, hipStream_t st)
{
    if (v.U >= 8) return launch_nt<S, 8>(p, ml, v, grid, lds, st);
    if constexpr (kAllVariants) return launch_nt<S, 4>(p, ml, v, grid, lds, st);
    return hipErrorInvalidValue;
}

bool rz_has_all_variants() { return kAllVariants; }

hipError_t rz_launch_deform_dense(const RzDeformParams &p, const RzMorphList &ml, const RzVariant &v, dim3 grid, size_t lds, hipStream_t st)
{
    switch (v.S) {
    case 2: return launch_split<2>(p, ml, v, grid, lds, st);
    case 4: return launch_split<4>(p, ml, v, grid, lds, st);
    case 8: return launch_split<8>(p, ml, v, grid, lds, st);
    default: return launch_split<1>(p, ml, v, grid, lds, st);
    }
}
