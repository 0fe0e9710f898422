// kernels/deform_small.hip — skin (+ sparse vertex morphs) of ONE mesh WITHOUT a dense morph stream: one character (BASELINE config C2),
// small crowds through the generic path, PMX-style sparse targets (the demo model). These frames are latency chains, not streams: a
// 30 k-vertex frame is 2.7 us inside the GPU (profiles/r4_timeline_c2.txt) — 0.9 us until the kernel arguments are there, then
// dependent memory round trips of ~0.8 us each, then 1.3 us until the next kernel starts. So this kernel asks for EVERYTHING a step
// reads at its top — rest position, the skin phase's normal / joints / weights, the sparse rows' bounds — and the first step asks in
// front of the workgroup's prologue (hierarchy solve, morph weights, palette), so that the mesh arrives under it; the thread's second
// bone (skeletons of 257-512 bones) is requested up front too; the matrix loads are unpredicated (inside `if (tid < B)` the compiler
// ended the divergent block by shuffling the loaded registers, i.e. with a wait for them in front of everything else: NOTEBOOK R4.9);
// the sparse frame's morph weights are parked in LDS in front of the barrier that publishes the palette instead of behind one of
// their own; and the four waves of a workgroup take runs a quarter of the mesh apart (a face region's heavy steps land on 28 CUs,
// not 7).
//   S     wave step: 1 -> 256 vertices (4 rounds of the skin phase), 4 -> 64 vertices (a small mesh reaches four times as many CUs)
//   MODE  0 = no morphs, 2 = per-vertex sparse CSR
//   NTS   nontemporal stores of the outputs
//   GEO   rest geometry through LDS (tools-only build)
//   FAST  single-instance frame in ONE launch (palette formed by every workgroup, see deform_dense.hip); !FAST: palette from
//         rz_prep_kernel / rz_fk_kernel, or (fk_on) the hierarchy solved in the workgroup's prologue — one launch per device-animated frame
// Sparse targets (MODE 2): per-vertex CSR, entry = (dx, dy, dz, bits(morph)), a vertex's entries ascending by morph, the entries of a
// step's vertices one contiguous range [E0, E1). The wave copies that range into its LDS buffer by LDS-DMA — consecutive lanes,
// 16-byte entries: every instruction is one 1 KiB burst, no register is held and ALL of them are in flight at once — and only then
// does each lane (= one vertex) walk its own row, out of LDS: acc = fma(w, d, acc) over ascending entries — the order of the CPU
// oracle's sparse accumulate, whatever the launch shape. Ranges larger than the buffer go through it in pieces; a row that straddles
// two pieces keeps its running sum. A 64-vertex step asks for its first piece at the TOP of the step, as soon as its row bounds are
// there: it lands while the palette is formed and the quad is parked. (Rounds 1-3 let every lane walk its row in global memory, 4
// entries at a time: the face of the demo model — 60 expression morphs on the same ~1 800 vertices — kept its waves 4.3 us in that
// loop: NOTEBOOK.md R4.1; now: profiles/r4_timeline_demo.txt.) LDS slots are XOR-swizzled (bits 0..3 with bits 4..7 of the entry's
// index in the piece): lanes read rows whose starts are a row length apart, and with rows of 16 / 32 / 48 entries — or the demo shape's
// 20 — plain slots put a whole wave on the same few banks. The DMA cannot scatter, so the swizzle is applied on the way IN: lane L of a
// burst fetches the entry whose slot L is (an involution inside aligned 16-entry groups: the burst still reads the same 256-byte
// segments). Four bursts share one LDS base (M0); the instruction offset — added to the global AND the LDS address — steps through
// them (rewriting M0 for every burst doubled the time a wave needs to issue a 20 KB piece: tools/archive/dmabench, 2 155 vs 995 cycles).
#include "deform_parts.hip.h"

namespace {

template <int S, int MODE, bool NTS, bool GEO, bool FAST, int FKV = 0>
__global__ void __launch_bounds__(kBlock, 2) rz_deform_small_kernel(const float *k_geom, const float *k_world, const float *k_inv_bind, const uint32_t k_bf,
                                                                    const uint32_t k_Vp, const uint32_t k_nq, const uint32_t k_qpw, const uint32_t *k_j01,
                                                                    const uint32_t *k_j23, const uint32_t *k_wq, const RzDeformParams p)
{
    static_assert(MODE == 0 || MODE == 2, "the dense morph stream lives in deform_dense.hip");
    constexpr int QPW = 64 / S;              // quads per wave
    constexpr int VW = 4 * QPW;              // vertices per wave per tile
    constexpr int NPL = GEO ? 9 : 3;         // scratch planes per wave
    constexpr int ROUNDS = (VW + 63) / 64;
    constexpr bool LDS_LIST = MODE == 2;     // sparse targets keep all M weights in LDS on both paths

    const int tid = threadIdx.x;
    const int inst = blockIdx.y;
    const int lane = tid & 63, wave = tid >> 6;
    const int kB = (int)(k_bf & 0xffffu);            // == p.B
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *pal = reinterpret_cast<float4 *>(smem);                       // B*3 float4
    RZ_TL_DECL;
    RZ_STAMP(0);                 // entry

    // Zero-copy pose prefetch (see RzDeformParams): workgroup 0 of such a launch is the helper, the workers shift by one.
    const bool pf_on = (k_bf >> 16) & 1u;            // == p.pf_src != nullptr (only one-launch and fused-hierarchy frames ever carry one)
    if (pf_on && blockIdx.x == 0) {
        pose_prefetch_helper(p, tid);
        return;
    }
    const uint32_t wid = blockIdx.x - (pf_on ? 1u : 0u);                 // worker index of this workgroup
    // fused-hierarchy frame (!FAST, fk_on): the static records of this thread's bones and morph are asked for before anything else
    FkEarly fke;
    if (!FAST && k_world) fke = fk_issue_static(reinterpret_cast<const uint4 *>(k_world), kB, (int)(uintptr_t)k_inv_bind, tid);
    // THIS frame's pose: staged in device memory by the previous frame's helper, or still in its pinned slot (see deform_dense.hip).
    // Sparse weights are needed at once: that mode waits for the tag.
    const bool spec = FAST && ((k_bf >> 17) & 1u);                       // == p.st_tag != nullptr
    const float *world_in = k_world;                                     // == spec ? p.st_world : p.world (re-pointed at the pinned slot on a miss)
    const bool from_host = ((k_bf >> 18) & 1u) && !spec;                 // (p.world_copy != nullptr) the matrices are asked for over the host link up front

    // FAST: this thread's bone(s) — world and inverse-bind matrices are requested FIRST, unpredicated (threads past the last bone
    // re-read bone B - 1: a dead load is cheaper than the wait the compiler puts behind a divergent block), and the thread's SECOND
    // bone (skeletons of 257..512 bones: the demo model has 349) with them — left to form_palette()'s late loop it was one more
    // memory round trip in front of the first skin phase of every workgroup.
    float4 ew0, ew1, ew2, ew3, ei0, ei1, ei2, ei3;
    const bool early = FAST && tid < kB && RZ_DBG(p) != 3;      // dbg 3: ablation — no palette staging (output is garbage)
    const int eb = min(tid, kB - 1);
    auto load_world = [&]() {
        const float4 *gw = reinterpret_cast<const float4 *>(world_in) + eb * 4;
        ew0 = gw[0]; ew1 = gw[1]; ew2 = gw[2]; ew3 = gw[3];
    };
    if (FAST) {
        const float4 *gi = reinterpret_cast<const float4 *>(k_inv_bind) + eb * 4;
        load_world();
        ei0 = gi[0]; ei1 = gi[1]; ei2 = gi[2]; ei3 = gi[3];
    }
    float4 fw0, fw1, fw2, fw3, fi0, fi1, fi2, fi3;
    const bool early2 = FAST && early && tid + kBlock < kB;
    const int eb2 = min(tid + kBlock, kB - 1);
    auto load_world2 = [&]() {
        const float4 *gw = reinterpret_cast<const float4 *>(world_in) + eb2 * 4;
        fw0 = gw[0]; fw1 = gw[1]; fw2 = gw[2]; fw3 = gw[3];
    };
    if constexpr (FAST) {                      // (unpredicated like the first bone's: with <= 256 bones every thread re-reads bone B - 1)
        const float4 *gi = reinterpret_cast<const float4 *>(k_inv_bind) + eb2 * 4;
        load_world2();
        fi0 = gi[0]; fi1 = gi[1]; fi2 = gi[2]; fi3 = gi[3];
    }
    const int s = lane / QPW;                // (S lanes share a quad: only slice 0 carries it — S only sets the size of a wave step here)
    const int qi = lane % QPW;
    const size_t Vp = k_Vp;
    const size_t plane4 = Vp / 4;            // float4 per plane
    // persistent, evenly balanced partition: every wave of the grid owns one contiguous run of quads (a multiple of 8 quads = 128 B
    // per plane) and walks it QPW quads at a time; the last step is masked. The four waves of a workgroup take runs that lie a quarter
    // of the mesh apart instead of next to each other: work on such frames is uneven — the demo model's 60 expression morphs all sit on
    // one 1 800-vertex face region, 28 consecutive 64-vertex steps — and four neighbouring heavy steps on ONE CU share its LDS and its
    // texture path (2.7 us in the row walk with four face waves per CU: NOTEBOOK.md R4.1).
    // (the worker count rides in k_bf's upper bits — gridDim.x is a hidden kernel argument, i.e. one more scalar load; a grid too large
    // for the 13 bits keeps neighbouring runs)
    const uint32_t n_workers = k_bf >> 19;
    const uint32_t wave_global = n_workers ? (uint32_t)wave * n_workers + wid : wid * (kBlock / 64) + wave;
    const size_t q_begin = (size_t)wave_global * k_qpw;
    const size_t q_end = min((size_t)k_nq, q_begin + k_qpw);

    // Everything a step reads from the static mesh, asked for at once: the quad's rest position, the skin phase's normal / joints /
    // weights (vertex per lane) and, for sparse targets, the bounds of the vertex's row; and the FIRST step asks right here, in front
    // of whatever the workgroup does first (the hierarchy solve, the staging of the morph weights, the palette), so the mesh arrives
    // under that prologue instead of behind it (NOTEBOOK.md R4.1: one round trip is ~1.1 us of a 4-7 us frame).
    constexpr bool PRE = !GEO;
    constexpr bool PRE_SP = PRE && MODE == 2 && ROUNDS == 1;      // (S = 1 steps are 4 rounds: their bounds are loaded round by round)
    float4 gx, gy, gz, gnx, gny, gnz;
    uint4 gj01, gj23, gw;
    float pnx[PRE ? ROUNDS : 1], pny[PRE ? ROUNDS : 1], pnz[PRE ? ROUNDS : 1];
    uint32_t pj01[PRE ? ROUNDS : 1], pj23[PRE ? ROUNDS : 1], pwq[PRE ? ROUNDS : 1];
    uint32_t sb0[1], sb1[1];                    // PRE_SP: row bounds of this lane's vertex
    auto issue = [&](const size_t qw) {
        const size_t q = qw + qi;
        if (s == 0 && q < q_end) {
            const float4 *G = reinterpret_cast<const float4 *>(k_geom) + q;
            gx = G[0]; gy = G[plane4]; gz = G[2 * plane4];
            if (GEO) {
                gnx = G[3 * plane4]; gny = G[4 * plane4]; gnz = G[5 * plane4];
                gj01 = reinterpret_cast<const uint4 *>(k_j01)[q];
                gj23 = reinterpret_cast<const uint4 *>(k_j23)[q];
                gw = reinterpret_cast<const uint4 *>(k_wq)[q];
            }
        }
        if constexpr (PRE) {
            const int v_live = (int)min((size_t)VW, (q_end - qw) * 4);
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const int vl = r * 64 + lane;
                if (vl < v_live) {
                    const size_t v = qw * 4 + vl;
                    pnx[r] = k_geom[3 * Vp + v]; pny[r] = k_geom[4 * Vp + v]; pnz[r] = k_geom[5 * Vp + v];
                    pj01[r] = k_j01[v]; pj23[r] = k_j23[v];
                }
            }
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r)         // (the weights plane's pointer is not among the preloaded arguments: asked for last)
                if (r * 64 + lane < v_live) pwq[r] = k_wq[qw * 4 + r * 64 + lane];
            if constexpr (PRE_SP) {
                sb0[0] = 0u; sb1[0] = 0u;
                if (lane < v_live) { sb0[0] = p.sp_ptr[qw * 4 + lane]; sb1[0] = p.sp_ptr[qw * 4 + lane + 1]; }
            }
        }
    };
    if (PRE && q_begin < q_end) issue(q_begin);

    // ---- from here on the kernel reads `p` (scalar loads of the kernel arguments: the loads above are in flight under them) ----
    uint32_t *s_idx = reinterpret_cast<uint32_t *>(smem + (size_t)p.B * 48);   // Mpad   (LDS_LIST)
    float *s_w = reinterpret_cast<float *>(s_idx + (LDS_LIST ? p.Mpad : 0));
    float *scratch_all = s_w + (LDS_LIST ? p.Mpad : 0);                   // 16-B aligned: Mpad % 4 == 0
    const uint64_t st_tagv = spec ? *p.st_tag : 0ull;                   // requested here, compared later (workgroup-uniform)
    const bool staged_now = MODE == 2 && spec && st_tagv == p.st_expect;
    const float *morph_w_in = (staged_now && p.st_morph_w) ? p.st_morph_w : p.morph_w;
    if (!FAST && (FKV == 1 || FKV == 2 || p.fk_on)) {
        // FUSED single-character frame: hierarchy solve (and motion sampling) as this workgroup's prologue (FKV 1 / 2: the variant
        // specialised for a plain uploaded / sampled pose — the launcher only picks it for fused frames)
        float *lds_mw = fused_hierarchy_prologue<MODE != 0, (FKV == 1 || FKV == 2) ? FKV : 0>(p.fk, fke, p.st_tag, p.st_expect, p.st_morph_w, p.morph_w, p.morph_w_copy, p.M, pal, scratch_all, wid, RZ_TL_FK);
        if (MODE == 2)
            for (int i = tid; i < p.M; i += kBlock) s_w[i] = lds_mw[i];
        __syncthreads();
    } else if (!FAST && FKV != 1 && FKV != 2) {
        const float4 *gpal = p.palette + (size_t)inst * p.B * 3;
        for (int i = tid; i < p.B * 3; i += kBlock) pal[i] = gpal[i];
        if (MODE == 2) {
            const float *gw = p.morph_w + (size_t)inst * p.M;
            for (int i = tid; i < p.M; i += kBlock) s_w[i] = gw[i];
        }
        __syncthreads();
    }

    // Sparse targets, one-launch frame: the pose's morph weights go to LDS. Up to 256 morphs (one per thread) the weight is only
    // REQUESTED here — it travels with the matrices and the first step's mesh loads — and is parked in LDS in front of the barrier
    // that publishes the palette (publish_weights): one wait and one barrier for everything the first skin phase needs, where
    // rounds 1-3 had a load -> LDS -> barrier sequence of their own in front of the palette (0.8 us of every such frame).
    const bool keep_w = FAST && MODE == 2 && wid == 0 && p.morph_w_copy != nullptr && !(staged_now && p.st_morph_w);     // zero-copy first frame, as for `world`
    bool w_pending = FAST && MODE == 2 && p.M <= kBlock;
    float w_early = 0.0f;
    if (w_pending && tid < p.M) w_early = morph_w_in[tid];
    auto publish_weights = [&]() {
        if (w_pending) {
            if (tid < p.M) { s_w[tid] = w_early; if (keep_w) p.morph_w_copy[tid] = w_early; }
            w_pending = false;
        }
    };
    if (FAST && MODE == 2 && !w_pending) {
        for (int i = tid; i < p.M; i += kBlock) {
            const float w = morph_w_in[i];
            s_w[i] = w;
            if (keep_w) p.morph_w_copy[i] = w;
        }
        __syncthreads();
    }

    RZ_STAMP(1);                 // prologue done (hierarchy solve / staged palette / sparse weights)
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
        if (spec && st_tagv != p.st_expect) {
            // miss: the previous frame's helper did not stage this pose (the host was not ahead): it is in its pinned slot
            world_in = p.world;
            keep_world = wid == 0 && p.world_copy != nullptr;
            if (early) load_world();
            if (early2) load_world2();
        }
        if (early) {
            palette_rows(tid, ew0, ew1, ew2, ew3, ei0, ei1, ei2, ei3);
            if (keep_world) { float4 *d = reinterpret_cast<float4 *>(p.world_copy) + tid * 4; d[0] = ew0; d[1] = ew1; d[2] = ew2; d[3] = ew3; }
        }
        if (early2) {
            palette_rows(tid + kBlock, fw0, fw1, fw2, fw3, fi0, fi1, fi2, fi3);
            if (keep_world) { float4 *d = reinterpret_cast<float4 *>(p.world_copy) + (tid + kBlock) * 4; d[0] = fw0; d[1] = fw1; d[2] = fw2; d[3] = fw3; }
        }
        for (int b = tid + 2 * kBlock; b < p.B; b += kBlock) {      // bones beyond what was asked for up front: plain loads, late
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
    float *sp_all = scratch_all + (size_t)(kBlock / 64) * NPL * VW + (size_t)(kBlock / 64) * cap * 6;      // MODE 2: 4 x sp_cap staged CSR entries (16-byte aligned: every term is a multiple of 4 floats)
    uint32_t ob_fill = 0;               // vertices parked
    size_t ob_v0 = q_begin * 4;         // global vertex index of the first parked vertex
    auto flush_out = [&]() {
        flush_parked<NTS>(opos, onrm, ob_pos, ob_nrm, ob_v0, ob_fill, lane);
        ob_v0 += ob_fill;
        ob_fill = 0;
    };

    // ---- sparse morph targets (MODE 2): the step's piece of the vertex-ordered CSR goes through LDS (see the top of the file) ----
    float4 *sp_buf = reinterpret_cast<float4 *>(sp_all) + (size_t)wave * p.sp_cap;
    const uint32_t sp_cap = p.sp_cap;
    auto sp_slot = [](uint32_t i) { return i ^ ((i >> 4) & 15u); };
    auto sp_stage = [&](const uint32_t c0, const uint32_t c1) {
        typedef const __attribute__((address_space(1))) void *gptr_t;
        typedef __attribute__((address_space(3))) void *lptr_t;
        for (uint32_t i = 0; c0 + i < c1; i += 256) {
            const lptr_t l = (lptr_t)(uint32_t)(uintptr_t)(sp_buf + i);
            const uint32_t e0 = c0 + sp_slot(i + (uint32_t)lane), e1 = c0 + sp_slot(i + 64 + (uint32_t)lane);
            const uint32_t e2 = c0 + sp_slot(i + 128 + (uint32_t)lane), e3 = c0 + sp_slot(i + 192 + (uint32_t)lane);
            if (e0 < c1) __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)(p.sp_entries + e0), l, 16, 0, 0);
            if (e1 < c1) __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)(p.sp_entries + e1 - 64), l, 16, 1024, 0);
            if (e2 < c1) __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)(p.sp_entries + e2 - 128), l, 16, 2048, 0);
            if (e3 < c1) __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)(p.sp_entries + e3 - 192), l, 16, 3072, 0);
        }
    };
    uint32_t spE0 = 0u, spE1 = 0u;          // PRE_SP: the step's entry range

    // One step = QPW quads. The body is instantiated twice: FIRST (the run's first step, which also forms the
    // palette from the early-loaded matrices) and the steady-state form, where those registers are dead.
    auto step = [&](const size_t qw, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const size_t q = qw + qi;                                    // this lane's quad
        const bool live = q < q_end;

        // the whole step's loads (the run's first step has asked at the top of the kernel already)
        if (!PRE || !FIRST) issue(qw);
        if constexpr (PRE_SP) {
            // the row bounds are the oldest loads in flight (the first step's were asked for at the top of the kernel): the first
            // piece of the step's entries is requested before anything else the step does
            const int n_live = (int)min((size_t)64, (q_end - qw) * 4);
            spE0 = __builtin_amdgcn_readfirstlane(sb0[0]);
            spE1 = __builtin_amdgcn_readlane(sb1[0], n_live - 1);
            if (spE0 < spE1) sp_stage(spE0, min(spE1, spE0 + sp_cap));
        }

        if (FIRST) RZ_STAMP(2);       // first step: (no morph stream here)
        if (FAST && FIRST && need_palette) form_palette();
        if (FAST && MODE == 2 && FIRST) publish_weights();
        if (FAST && FIRST && need_sync) { __syncthreads(); need_sync = false; }   // palette (and sparse weights) of every wave are in LDS
        if (FIRST) RZ_STAMP(3);       // first step: palette published

        // ---- park the quad in the wave's scratch: plane-major [NPL][VW] dwords ----
        if (s == 0 && live) {
            float4 *sc4 = reinterpret_cast<float4 *>(scr) + qi;
            sc4[0 * QPW] = gx;
            sc4[1 * QPW] = gy;
            sc4[2 * QPW] = gz;
            if (GEO) {
                sc4[3 * QPW] = gnx; sc4[4 * QPW] = gny; sc4[5 * QPW] = gnz;
                uint4 *su4 = reinterpret_cast<uint4 *>(scr) + qi;
                su4[6 * QPW] = gj01; su4[7 * QPW] = gj23; su4[8 * QPW] = gw;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        const size_t vw0 = qw * 4;     // first vertex of this wave's step
        const int v_live = RZ_DBG(p) == 4 ? 0 : (int)min((size_t)VW, (q_end - qw) * 4);   // dbg 4: ablation — no skin phase

        float spx[ROUNDS], spy[ROUNDS], spz[ROUNDS];        // MODE 2: this lane's vertex's morph offset, per round
        if constexpr (MODE == 2) {
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const int vl = r * 64 + lane;
                const int n_live = min(64, v_live - r * 64);        // lanes of this round that own a vertex (wave-uniform)
                spx[r] = 0.0f; spy[r] = 0.0f; spz[r] = 0.0f;
                if (n_live <= 0) continue;
                uint32_t b0, b1, E0, E1;
                if constexpr (PRE_SP) { b0 = sb0[0]; b1 = sb1[0]; E0 = spE0; E1 = spE1; }        // (the first piece is on its way)
                else {
                    b0 = 0u; b1 = 0u;
                    if (vl < v_live) { b0 = p.sp_ptr[vw0 + vl]; b1 = p.sp_ptr[vw0 + vl + 1]; }
                    E0 = __builtin_amdgcn_readfirstlane(b0);
                    E1 = __builtin_amdgcn_readlane(b1, n_live - 1);
                }
                for (uint32_t c0 = E0; c0 < E1; c0 += sp_cap) {             // wave-uniform; no iteration at all for a step without offsets
                    const uint32_t c1 = min(E1, c0 + sp_cap);
                    if (!PRE_SP || c0 != E0) sp_stage(c0, c1);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t lo = max(b0, c0), hi = min(b1, c1);
                    constexpr int SU = 8;                                   // LDS reads in flight per lane (entry, then its weight)
                    for (uint32_t j = lo; j < hi; j += SU) {
                        float4 ent[SU];
                        float w[SU];
#pragma unroll
                        for (int u = 0; u < SU; ++u) ent[u] = sp_buf[sp_slot(min(j + u, hi - 1) - c0)];
#pragma unroll
                        for (int u = 0; u < SU; ++u) w[u] = s_w[__float_as_uint(ent[u].w)];
#pragma unroll
                        for (int u = 0; u < SU; ++u) {
                            const float wu = j + u < hi ? w[u] : 0.0f;       // (past the row's end the clamped index re-read its last entry)
                            spx[r] = fmaf(wu, ent[u].x, spx[r]); spy[r] = fmaf(wu, ent[u].y, spy[r]); spz[r] = fmaf(wu, ent[u].z, spz[r]);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();                        // every lane is done with this piece before the next one lands
                }
            }
        }

        // ---- phase 2: one vertex per lane ----
        auto skin_round = [&](const int r, float nx, float ny, float nz, uint32_t j01, uint32_t j23, uint32_t wq) {
            const int vl = r * 64 + lane;
            if (vl < v_live) {
                const size_t v = vw0 + vl;
                float x = scr[0 * VW + vl], y = scr[1 * VW + vl], z = scr[2 * VW + vl];
                if constexpr (MODE == 2) { x += spx[r]; y += spy[r]; z += spz[r]; }
                if constexpr (PRE) {
                    // (asked for at the top of the step)
                } else {
                    nx = scr[3 * VW + vl]; ny = scr[4 * VW + vl]; nz = scr[5 * VW + vl];
                    const uint32_t *su = reinterpret_cast<const uint32_t *>(scr);
                    j01 = su[6 * VW + vl]; j23 = su[7 * VW + vl]; wq = su[8 * VW + vl];
                }
                const Skinned o = skin_vertex(pal, x, y, z, nx, ny, nz, j01, j23, wq, bmax);
                emit_vertex<NTS, FKV == 0>(p, o, v, inst, Vp, cap, ob_pos, ob_nrm, (ob_fill + vl) * 3, opos, onrm, bb);
            }
        };
        if constexpr (PRE) {
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) skin_round(r, pnx[r], pny[r], pnz[r], pj01[r], pj23[r], pwq[r]);
        } else if constexpr (MODE == 2) {      // (GEO form of the sparse kernel, tools-only build: spx[r] wants a constant index)
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) skin_round(r, 0.0f, 0.0f, 0.0f, 0u, 0u, 0u);
        } else {
#pragma unroll 1
            for (int r = 0; r < ROUNDS; ++r) skin_round(r, 0.0f, 0.0f, 0.0f, 0u, 0u, 0u);
        }
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
    if (FAST && MODE == 2) publish_weights();    // ... its morph weights ...
    if (FAST && need_sync) __syncthreads();      // ... and its barrier
    if (FKV == 0 && p.aabb) aabb_commit(p, inst, bb, lane, tid, wid, q_begin < q_end);
    RZ_TL_FLUSH(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (kBlock / 64) + wave);
}

}  // namespace

#ifdef RZ_ALL_VARIANTS
constexpr bool kAllVariants = true;
#else
constexpr bool kAllVariants = false;
#endif

template <int S, int MODE, bool NTS, bool GEO, bool FAST, int FKV = 0>
static hipError_t launch_one(const RzDeformParams &p, dim3 grid, size_t lds, hipStream_t st)
{
    // FKV — which variant of the kernel: 0 = everything compiled in (the fused consumers: outline hull, bounding box); 3 = without them;
    // 1 / 2 = without them AND the hierarchy solve specialised for a plain uploaded / sampled pose (fk_solve<true, KIND>)
    if constexpr (!GEO && FKV == 0) {
        if (!p.edge && !p.aabb) {
            if constexpr (!FAST) {
                if (p.fk_on && p.fk_kind == 1) return launch_one<S, MODE, NTS, GEO, FAST, 1>(p, grid, lds, st);
                if (p.fk_on && p.fk_kind == 2) return launch_one<S, MODE, NTS, GEO, FAST, 2>(p, grid, lds, st);
            }
            return launch_one<S, MODE, NTS, GEO, FAST, 3>(p, grid, lds, st);
        }
    }
    auto k = rz_deform_small_kernel<S, MODE, NTS, GEO, FAST, FKV>;
    if (p.B > 0xffff) return hipErrorInvalidValue;      // k_bf carries the bone count in 16 bits
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    // (!FAST: the hierarchy's static block and the motion's morph count ride in the two matrix slots, deform_parts.hip.h)
    const float *k_world = FAST ? (p.st_tag ? p.st_world : p.world) : (p.fk_on ? reinterpret_cast<const float *>(p.fk.bone_rec) : nullptr);
    const float *k_inv_bind = FAST ? p.inv_bind : reinterpret_cast<const float *>((uintptr_t)(p.fk_on ? p.fk.sample.M : 0));
    hipLaunchKernelGGL(k, grid, dim3(kBlock), lds, st, p.geom, k_world, k_inv_bind, rz_deform_k_bf(p, grid.x), p.Vp, p.n_quads, p.quads_per_wave, p.joints01, p.joints23, p.weights, p);
    return hipGetLastError();
}

template <int S, int MODE>
static hipError_t launch_mode(const RzDeformParams &p, const RzVariant &v, dim3 grid, size_t lds, hipStream_t st)
{
    if (v.geo) {
        if constexpr (kAllVariants) {
            if (v.nts) return v.fast ? launch_one<S, MODE, true, true, true>(p, grid, lds, st) : launch_one<S, MODE, true, true, false>(p, grid, lds, st);
            return v.fast ? launch_one<S, MODE, false, true, true>(p, grid, lds, st) : launch_one<S, MODE, false, true, false>(p, grid, lds, st);
        }
        return hipErrorInvalidValue;      // rest geometry through LDS is a tools-only variant
    }
    if (v.nts) return v.fast ? launch_one<S, MODE, true, false, true>(p, grid, lds, st) : launch_one<S, MODE, true, false, false>(p, grid, lds, st);
    return v.fast ? launch_one<S, MODE, false, false, true>(p, grid, lds, st) : launch_one<S, MODE, false, false, false>(p, grid, lds, st);
}

hipError_t rz_launch_deform_small(const RzDeformParams &p, const RzVariant &v, dim3 grid, size_t lds, hipStream_t st)
{
    if (v.mode == 0) return v.S == 4 ? launch_mode<4, 0>(p, v, grid, lds, st) : launch_mode<1, 0>(p, v, grid, lds, st);
    return v.S == 4 ? launch_mode<4, 2>(p, v, grid, lds, st) : launch_mode<1, 2>(p, v, grid, lds, st);
}
