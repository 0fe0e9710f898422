// kernels/crowd.hip — instanced skin (BASELINE config C4: many poses of one static mesh) and its plan-time bone-subset pass.
#include "fk.hip.h"

namespace {

// ------------------------------------------------------------------------------------------------
// instanced skin (MODE 0, I > 1 — BASELINE config C4: many poses of one static mesh).
// A workgroup owns a run of vertices and a GROUP of G instances whose palettes it stages together in
// LDS (LDS-DMA from the prep kernel's palette). Each lane decodes a vertex ONCE — rest position/normal,
// the four joints as palette row offsets, the four weights as floats — and then loops over the G poses:
// 12 ds_read_b128 + blend + transform + 24 B store per pose. The static mesh is read I/G times instead
// of I times, and the per-pose body has no global load in front of it (the generic kernel was latency-
// bound here: 43 % of wave time in s_waitcnt, VALU 30 %, LDS 31 % — profiles/archive/r1_sq_counters.txt).
// grid = (vertex runs, instance groups); block = 256; dynamic LDS = G * B * 48 bytes.
// ------------------------------------------------------------------------------------------------

// BLOCK threads per workgroup (256: two workgroups per CU; 512 / 1024: one, with 8 / 16 waves sharing one staged palette
// group — half / a quarter of the palette traffic per CU). NB = how many influences the pose loop gathers: the host picks
// nothing here, every wave decides per vertex step from a ballot over its lanes' weights (wave-uniform, so no divergence):
// a step whose 64 vertices are all BDEF1 / BDEF2 (real PMX models cluster them by mesh part) reads 3 / 6 palette rows per
// pose instead of 12. Skipped terms are w = 0, i.e. fma(0, row, m) = m: the result bits do not depend on the path taken.
//
// SUB = bone-subset form. A vertex run names only a few of the skeleton's bones (PMX meshes are bone-local: the synthetic C4
// mesh's 3 750-vertex runs touch ~34 of 200), and rz_run_subsets_kernel has listed them per run and rewritten the joints as
// slots of that list. The workgroup stages ONLY those bones of its G poses: 8 x 34 matrices instead of 8 x 200 — the front of
// every workgroup (LDS-DMA + palette product, during which the CU stores nothing) shrinks from 2.9 us to 1.3 us (profiles/archive/r3_c4_front.txt), and the
// group's LDS footprint from 102 KB to 30 KB. World matrices are staged behind the palette region, so the product needs no
// in-place rounds. The palette rows a vertex gathers hold the same bits wherever they sit in LDS: outputs do not change.
// (Tried and measured slower or without effect, then removed again — NOTEBOOK.md R3.1 / R3.4 / R3.9: forcing three workgroups per CU
// (80 VGPRs spill), a branch-free pose loop with clamped tail lanes, four poses unrolled, raised wave priority around the stores.)
template <int BLOCK, bool NTS, bool SUB>
__global__ void __launch_bounds__(BLOCK) rz_skin_instances_kernel(const uint32_t *k_sub_count, const uint16_t *k_sub_list, const float4 *k_src,
                                                                  const float *k_inv_bind, const int G, const int n_inst, const uint32_t verts_per_wg,
                                                                  const uint32_t k_grid, const uint32_t k_bf, const uint32_t k_Vp, const RzDeformParams p)
{
    // The leading arguments are preloaded into SGPRs at wave start (14 dwords; see rz_deform_kernel): what the FRONT of a workgroup
    // needs — the run's bone list, the matrices it stages (k_src = the poses' world matrices, or their finished palettes behind
    // rz_prep_kernel / rz_fk_kernel), the inverse bind matrices, the launch shape (k_grid = gridDim.x | gridDim.y << 16: the hidden
    // arguments would be one more scalar load) and k_bf = bone count | inst_order << 16 | dma << 17 | padded length of the run lists << 18.
    const int kB = (int)(k_bf & 0xffffu);
    const bool k_order = (k_bf >> 16) & 1u, k_dma = (k_bf >> 17) & 1u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *pal = reinterpret_cast<float4 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    RZ_TL_DECL;
    RZ_STAMP(0);                 // entry
    // Workgroups are dealt to the 8 XCDs round-robin in launch order (x fastest). inst_order 0: x = vertex run — an XCD sees every
    // pose group and one eighth of the mesh; 1: consecutive workgroups take consecutive pose groups — an XCD sees one eighth of
    // the poses' matrices and the whole mesh.
    const uint32_t n_groups = k_grid >> 16, lin = blockIdx.x + (k_grid & 0xffffu) * blockIdx.y;
    const uint32_t wg_group = k_order ? lin % n_groups : blockIdx.y, wg_run = k_order ? lin / n_groups : blockIdx.x;
    const int inst0 = (int)wg_group * G;
    const int ng = min(G, n_inst - inst0);
    const int rows = kB * 3;                       // float4 per palette, in global memory and (finished) in LDS
    constexpr int rstride = 3;                      // float4 per bone of a finished palette
    // SUB: this run's bone list. Its LENGTH is not read: every list is padded with bone 0 up to the plan's longest (k_bf >> 18), and a
    // workgroup stages and converts that many slots — a few matrices nobody gathers, in exchange for one dependent scalar load less at
    // the head of the front (count -> list -> matrices was three round trips in front of the first store in round 3's form, which read
    // k_sub_count[wg_run]: -0.8 % in alternation, profiles/r5_timeline_c4.txt)
    const int ns = SUB ? (int)(k_bf >> 18) : 0;
    const uint16_t *sub = SUB ? k_sub_list + (size_t)wg_run * kB : nullptr;      // (the lists' stride is the bone count)
    const int lrows = SUB ? ns * 3 : rows;          // float4 per pose of the finished LDS palettes
    float4 *stage = pal + (size_t)G * ns * 3;       // SUB, one-launch frame: staged world matrices sit behind the palette region
    if constexpr (SUB) {
        // prep-kernel / device-FK path (dma): the listed bones' finished rows, 3 float4 per bone, straight to their place.
        // one-launch frame: the listed bones' world matrices, 4 float4 per bone, into the staging region.
        // Either way element e of the linear LDS image is (pose g, slot s, cell k): a per-lane global address, a linear LDS one.
        const int epb = k_dma ? 3 : 4;
        const int n = RZ_DBG(p) == 8 ? 0 : ng * ns * epb;           // dbg 8 (tools-only build): neither staging nor product
        const float4 *src = k_src + (size_t)inst0 * kB * (k_dma ? 3 : 4);
        float4 *dst = k_dma ? pal : stage;
        for (int c = wave * 64; c < n; c += BLOCK) {
            const int e = c + lane;
            if (e < n) {
                const int gs = k_dma ? e / 3 : e >> 2, k = e - gs * epb;
                const int g = gs / ns, sl = gs - g * ns;
                const float4 *a = src + ((size_t)g * kB + sub[sl]) * epb + k;
                typedef const __attribute__((address_space(1))) void *gptr_t;
                typedef __attribute__((address_space(3))) void *lptr_t;
                __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)a, (lptr_t)(uint32_t)(uintptr_t)(dst + c), 16, 0, 0);
            }
        }
    } else {
        // prep-kernel path (dma): the group's finished palettes are contiguous in global memory -> one linear LDS-DMA copy.
        // one-launch frame (!dma): the group's WORLD matrices are staged instead, whole (64-byte slots, the same linear
        // 16-byte LDS-DMA); the conversion pass below multiplies by the inverse bind matrix and re-packs the rows to the
        // same 48-byte stride. (Leaving them in the 64-byte slots made every fourth bone share its LDS banks: 52 % of the
        // skin loop's LDS cycles were bank conflicts, against 11 % at 48 bytes — profiles/archive/r2_sq_counters_c4.txt.)
        const float4 *src = k_src + (size_t)inst0 * kB * (k_dma ? 3 : 4);
        const int n = (RZ_DBG(p) == 6 || RZ_DBG(p) == 7 || RZ_DBG(p) == 8) ? 0 : (k_dma ? ng * rows : ng * kB * 4);   // dbg 6 / 7 (tools-only build): no palette staging
        for (int c = wave * 64; c < n; c += BLOCK) {
            const int e = c + lane;
            if (e < n) {
                typedef const __attribute__((address_space(1))) void *gptr_t;
                typedef __attribute__((address_space(3))) void *lptr_t;
                __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)(src + e), (lptr_t)(uint32_t)(uintptr_t)(pal + c), 16, 0, 0);
            }
        }
    }
    // One-launch frame (the host only plans it for B <= BLOCK): which (bone, pose stripe) this thread converts. With
    // B <= BLOCK / 2 the spare threads take a second, third ... stripe of the group's poses (stripe s converts poses s,
    // s + stripes, ...): 200 bones on 512 threads = 2 stripes.
    // SUB: the same mapping over the run's ns listed bones (the host plans the form only for ns <= BLOCK).
    const int cvB = SUB ? max(ns, 1) : kB;
    const int stripes = !k_dma ? max(1, min(ng, BLOCK / cvB)) : 1;
    const int cv_b0 = tid % cvB;
    const int cv_g0 = tid / cvB;
    const bool cv_on = !k_dma && cv_g0 < stripes && (!SUB || ns > 0);
    float4 ib0 = {0, 0, 0, 0}, ib1 = ib0, ib2 = ib0, ib3 = ib0;
    if (cv_on) {                                    // requested first: lands while the staging copy is in flight
        const float4 *gi = reinterpret_cast<const float4 *>(k_inv_bind) + (SUB ? (int)sub[cv_b0] : cv_b0) * 4;
        ib0 = gi[0]; ib1 = gi[1]; ib2 = gi[2]; ib3 = gi[3];
    }
    const size_t Vp = k_Vp;
    const uint32_t v_begin = wg_run * verts_per_wg;
    const uint32_t v_end = min(p.n_quads * 4u, v_begin + verts_per_wg);
    const uint32_t bmax = (uint32_t)(p.B - 1);
    // software-pipelined vertex loop: the next vertex's nine attribute loads are issued before the current
    // vertex's pose loop, so their L2 latency hides behind the poses' LDS gathers + FMA
    // (Round 4 tried two ways of sharing the run's last, partial step (a 3 750-vertex run is 7 steps of 512 and one of 166) among
    // all waves — its vertices dealt out in quarter-wave pieces: 33.5 -> 36.3 us; its poses split between the waves that hold the
    // same piece: 33.0 -> 33.7 us. The waves of a workgroup end up to 5.7 us apart (profiles/r4_timeline_c4.txt), but not because
    // of that step: NOTEBOOK.md R4.5. Both removed.)
    auto vert_of = [&](const uint32_t vb) { return vb + (uint32_t)tid; };
    uint32_t v = vert_of(v_begin);
    float x = 0, y = 0, z = 0, nx = 0, ny = 0, nz = 0;
    uint32_t j01 = 0, j23 = 0, wq = 0;
    const uint32_t *jp01 = SUB ? p.rj01 : p.joints01, *jp23 = SUB ? p.rj23 : p.joints23;     // SUB: joints as slots of the run's list
    if (v < v_end) {
        x = p.geom[0 * Vp + v]; y = p.geom[1 * Vp + v]; z = p.geom[2 * Vp + v];
        nx = p.geom[3 * Vp + v]; ny = p.geom[4 * Vp + v]; nz = p.geom[5 * Vp + v];
        j01 = jp01[v]; j23 = jp23[v]; wq = p.weights[v];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // palettes / world matrices (and the first vertex) have landed
    RZ_STAMP(1);                 // staged matrices have landed
    __syncthreads();
    if constexpr (SUB) {
        if (!p.dma && RZ_DBG(p) != 8) {
            // palette rows of the listed bones: slot (g, s) = rows 0..2 of world * inverseBind (engine.ts:926-928), the same
            // packed FMA chain as below and as rz_prep_kernel — out of the staging region, into the palette region: no hazard,
            // one barrier. With ns ~ 34 and 512 threads every (pose, bone) pair has a thread of its own.
            const f2 px01 = {ib0.x, ib1.x}, px23 = {ib2.x, ib3.x}, py01 = {ib0.y, ib1.y}, py23 = {ib2.y, ib3.y};
            const f2 pz01 = {ib0.z, ib1.z}, pz23 = {ib2.z, ib3.z}, pw01 = {ib0.w, ib1.w}, pw23 = {ib2.w, ib3.w};
            auto row = [&](float e0, float e1, float e2, float e3, const f2 &bx, const f2 &by, const f2 &bz, const f2 &bw) {
                return pk_fma(f2{e3, e3}, bw, pk_fma(f2{e2, e2}, bz, pk_fma(f2{e1, e1}, by, f2{e0, e0} * bx)));
            };
            if (cv_on)
                for (int g = cv_g0; g < ng; g += stripes) {
                    const float4 *slot = stage + ((size_t)g * ns + cv_b0) * 4;
                    const float4 a0 = slot[0], a1 = slot[1], a2 = slot[2], a3 = slot[3];      // the world matrix's columns
                    const f2 r0 = row(a0.x, a1.x, a2.x, a3.x, px01, py01, pz01, pw01), r1 = row(a0.x, a1.x, a2.x, a3.x, px23, py23, pz23, pw23);
                    const f2 r2 = row(a0.y, a1.y, a2.y, a3.y, px01, py01, pz01, pw01), r3 = row(a0.y, a1.y, a2.y, a3.y, px23, py23, pz23, pw23);
                    const f2 r4 = row(a0.z, a1.z, a2.z, a3.z, px01, py01, pz01, pw01), r5 = row(a0.z, a1.z, a2.z, a3.z, px23, py23, pz23, pw23);
                    float4 *dst = pal + ((size_t)g * ns + cv_b0) * 3;
                    dst[0] = make_float4(r0.x, r0.y, r1.x, r1.y);
                    dst[1] = make_float4(r2.x, r2.y, r3.x, r3.y);
                    dst[2] = make_float4(r4.x, r4.y, r5.x, r5.y);
                }
            __syncthreads();
            // (the skinMatrixBuffer is not written here: a workgroup only holds its run's bones. rz_read_palette forms it on
            // demand with rz_prep_kernel — the same chain, the same bits.)
        }
    } else
    if (!p.dma && RZ_DBG(p) != 8) {                  // dbg 8 (tools-only build): neither staging nor conversion
        // in-place conversion: slot (g, b) = rows 0..2 of world * inverseBind (engine.ts:926-928). Packed math: the
        // inverse bind matrix is held as column PAIRS (c0,c1),(c2,c3) per k, so each result row is two v_pk_fma
        // chains (the same chain as rz_prep_kernel: ((a0*b0 + a1*b1) + a2*b2) + a3*b3); the next pose's cells are read
        // before the current product is formed.
        // Rounds of up to four poses per thread: read the staged world matrices (64-byte slots) and form the palette rows
        // in registers (engine.ts:926-928, packed math, the same FMA chain as rz_prep_kernel: ((a0*b0 + a1*b1) + a2*b2) +
        // a3*b3) -> barrier -> write them back at the 48-byte stride. Poses ascend, and the compact rows of pose g only
        // ever land on staged matrices of poses <= g, which every thread has read by then.
        const f2 px01 = {ib0.x, ib1.x}, px23 = {ib2.x, ib3.x}, py01 = {ib0.y, ib1.y}, py23 = {ib2.y, ib3.y};
        const f2 pz01 = {ib0.z, ib1.z}, pz23 = {ib2.z, ib3.z}, pw01 = {ib0.w, ib1.w}, pw23 = {ib2.w, ib3.w};
        auto row = [&](float e0, float e1, float e2, float e3, const f2 &bx, const f2 &by, const f2 &bz, const f2 &bw) {
            return pk_fma(f2{e3, e3}, bw, pk_fma(f2{e2, e2}, bz, pk_fma(f2{e1, e1}, by, f2{e0, e0} * bx)));
        };
        constexpr int PR = 4;                               // poses per thread per round
        for (int g_base = 0; g_base < ng; g_base += PR * stripes) {       // workgroup-uniform trip count
            f2 res[PR][6];
#pragma unroll
            for (int i = 0; i < PR; ++i) {
                const int g = g_base + cv_g0 + i * stripes;
                if (cv_on && g < ng) {
                    const float4 *slot = pal + ((size_t)g * p.B + cv_b0) * 4;
                    const float4 a0 = slot[0], a1 = slot[1], a2 = slot[2], a3 = slot[3];      // the world matrix's columns
                    res[i][0] = row(a0.x, a1.x, a2.x, a3.x, px01, py01, pz01, pw01); res[i][1] = row(a0.x, a1.x, a2.x, a3.x, px23, py23, pz23, pw23);
                    res[i][2] = row(a0.y, a1.y, a2.y, a3.y, px01, py01, pz01, pw01); res[i][3] = row(a0.y, a1.y, a2.y, a3.y, px23, py23, pz23, pw23);
                    res[i][4] = row(a0.z, a1.z, a2.z, a3.z, px01, py01, pz01, pw01); res[i][5] = row(a0.z, a1.z, a2.z, a3.z, px23, py23, pz23, pw23);
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < PR; ++i) {
                const int g = g_base + cv_g0 + i * stripes;
                if (cv_on && g < ng) {
                    float4 *dst = pal + ((size_t)g * p.B + cv_b0) * 3;
                    dst[0] = make_float4(res[i][0].x, res[i][0].y, res[i][1].x, res[i][1].y);
                    dst[1] = make_float4(res[i][2].x, res[i][2].y, res[i][3].x, res[i][3].y);
                    dst[2] = make_float4(res[i][4].x, res[i][4].y, res[i][5].x, res[i][5].y);
                }
            }
        }
        __syncthreads();
        if (p.palette) {
            // keep the skinMatrixBuffer observable (rz_read_palette): the vertex runs of a pose group each copy one
            // slice of the finished palettes out of LDS, coalesced
            const int n = ng * rows, per = (n + (int)gridDim.x - 1) / (int)gridDim.x;
            const int lo = (int)wg_run * per, hi = min(n, lo + per);
            float4 *gp = p.palette + (size_t)inst0 * rows;
            for (int i = lo + tid; i < hi; i += BLOCK) gp[i] = pal[i];
        }
    }
    RZ_STAMP(2);                 // palettes formed and published: the front is over
#include "crowd_pose_loop.inc.h"
    RZ_STAMP(5);                 // last vertex step issued
    RZ_TL_FLUSH(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (BLOCK / 64) + wave);
}

// ------------------------------------------------------------------------------------------------
// Crowd frame of DEVICE-ANIMATED poses in ONE launch (round 5): the front of every workgroup solves the part of the hierarchy its
// vertex run needs, straight into its LDS palettes — no rz_fk_kernel in front (6.4-6.6 us of a 38 us frame), no world matrices or
// palettes through memory. A run names ns bones (the subset lists); the plan (plan.cpp: ensure_subfk) has closed that list under
// "parent of" — the CLOSURE, nc bones: on the synthetic C4 mesh 36 named, ~75 with their ancestors, of 200 — and written one 80-byte
// record per closure slot: (bone | palette slot << 16, append parent, bits(append ratio), flags) (bind x y z) (ancestor SLOTS of the
// radix-4 doubling rounds 0, 1) (round 2) (the motion's track of the bone). Work item (pose g of the group, closure slot c), at most two
// per thread: record -> the pose's rotation / translation of that bone (or the keys of its track's guessed span) + inverse bind matrix
// -> local matrix into LDS -> doubling rounds over the closure (kernels/fk.hip.h: fk_round; a bone's world matrix depends on its own
// chain only, so it has the bits rz_fk_kernel gives it) -> palette rows of the named bones. Then the pose loop of rz_skin_instances_kernel.
// Leading (preloaded) arguments: k_cnt = closure slots per run, k_rec = the records, k_src = the poses' local rotations [I][B] (sampled
// poses: the per-instance frame numbers), k_inv_bind; k_g = poses per workgroup | sampled << 16 | has translations << 17 | padded length of the run lists << 18;
// k_bf = bone count | inst_order << 16 | doubling rounds << 18 | record stride (closure slots) << 20.
// LDS: two matrix buffers of G x nc x 48 B, then the palettes G x ns x 48 B.
// ------------------------------------------------------------------------------------------------
constexpr int kSubFkWords = 5;          // uint4 per closure record

template <int BLOCK, bool NTS>
__global__ void __launch_bounds__(BLOCK) rz_skin_instances_fk_kernel(const uint32_t *k_cnt, const uint4 *k_rec, const float4 *k_src, const float *k_inv_bind,
                                                                     const uint32_t k_g, const int n_inst, const uint32_t verts_per_wg, const uint32_t k_grid,
                                                                     const uint32_t k_bf, const uint32_t k_Vp, const RzDeformParams p)
{
    constexpr bool SUB = true;
    constexpr int rstride = 3;
    const int kB = (int)(k_bf & 0xffffu), G = (int)(k_g & 0xffffu);
    const bool k_order = (k_bf >> 16) & 1u, sampled = (k_g >> 16) & 1u, has_lt = (k_g >> 17) & 1u;
    const int n_rounds = (int)((k_bf >> 18) & 3u), rec_stride = (int)(k_bf >> 20);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    (void)lane; (void)wave;
    RZ_TL_DECL;
    RZ_STAMP(0);                 // entry
    const uint32_t n_groups = k_grid >> 16, lin = blockIdx.x + (k_grid & 0xffffu) * blockIdx.y;
    const uint32_t wg_group = k_order ? lin % n_groups : blockIdx.y, wg_run = k_order ? lin / n_groups : blockIdx.x;
    const int inst0 = (int)wg_group * G;
    const int ng = min(G, n_inst - inst0);
    // (neither the run's closure length nor its list length is read: records and lists are padded to the plan's longest, as in
    // rz_skin_instances_kernel — which takes two dependent scalar loads out of the front)
    const int nc = rec_stride;
    const uint4 *recs = k_rec + (size_t)wg_run * rec_stride * kSubFkWords;
    float4 *mA = reinterpret_cast<float4 *>(smem), *mB = mA + (size_t)G * nc * 3, *pal = mB + (size_t)G * nc * 3;
    // this thread's work items e = tid, tid + BLOCK: (pose g, closure slot c); past the end the LAST item is re-read (unpredicated loads)
    constexpr int NIT = 2;
    const int n_items = ng * nc;
    int ig[NIT], ic[NIT];
    bool on[NIT];
    uint4 w0[NIT], w1[NIT], w2[NIT], w3[NIT], w4[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int e = tid + k * BLOCK;
        on[k] = e < n_items;
        const int ec = min(e, max(n_items - 1, 0));
        ig[k] = ec / max(nc, 1); ic[k] = ec - ig[k] * nc;
        const uint4 *r = recs + (size_t)ic[k] * kSubFkWords;
        w0[k] = r[0]; w1[k] = r[1]; w2[k] = r[2]; w3[k] = r[3];
        w4[k] = sampled ? r[4] : make_uint4(0, 0, 0, 0);
    }
    // the run's first vertex (its loads land under the solve)
    const size_t Vp = k_Vp;
    const uint32_t v_begin = wg_run * verts_per_wg;
    const uint32_t v_end = min(p.n_quads * 4u, v_begin + verts_per_wg);
    const uint32_t bmax = (uint32_t)(p.B - 1);
    auto vert_of = [&](const uint32_t vb) { return vb + (uint32_t)tid; };
    uint32_t v = vert_of(v_begin);
    float x = 0, y = 0, z = 0, nx = 0, ny = 0, nz = 0;
    uint32_t j01 = 0, j23 = 0, wq = 0;
    const uint32_t *jp01 = p.rj01, *jp23 = p.rj23;
    if (v < v_end) {
        x = p.geom[0 * Vp + v]; y = p.geom[1 * Vp + v]; z = p.geom[2 * Vp + v];
        nx = p.geom[3 * Vp + v]; ny = p.geom[4 * Vp + v]; nz = p.geom[5 * Vp + v];
        j01 = jp01[v]; j23 = jp23[v]; wq = p.weights[v];
    }
    const int ns = (int)(k_g >> 18);                   // named bones = palette slots of a run (the longest list's)
    const int lrows = ns * 3;
    // ---- the pose of every item: uploaded rotations (+ translations), or the motion sampled at the instance's frame ----
    float4 q[NIT], apq[NIT], ib0[NIT], ib1[NIT], ib2[NIT], ib3[NIT];
    float ltx[NIT], lty[NIT], ltz[NIT], apx[NIT], apy[NIT], apz[NIT];
    const float *glt = p.fk.local_t;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const uint32_t bone = w0[k].x & 0xffffu, pslot = w0[k].x >> 16;
        const int ap = (int)w0[k].y;
        const size_t ib_row = (size_t)(inst0 + ig[k]) * kB;
        ltx[k] = lty[k] = ltz[k] = apx[k] = apy[k] = apz[k] = 0.0f;
        apq[k] = make_float4(0.f, 0.f, 0.f, 1.f);
        if (sampled) {
            const float frame = reinterpret_cast<const float *>(k_src)[inst0 + ig[k]];
            BoneKeys bk = bone_issue(p.fk.sample, frame, w4[k]);
            BoneKeys ak;
            uint4 apr = make_uint4(0, 0, 0, 0);
            if (ap >= 0) apr = p.fk.bone_rec[4 * (size_t)ap + 2];                 // the append parent's track: its LOCAL pose is all that is needed
            ak = bone_issue(p.fk.sample, frame, apr);
            bone_finish(p.fk.sample, frame, bk, q[k], ltx[k], lty[k], ltz[k]);
            if (ap >= 0) bone_finish(p.fk.sample, frame, ak, apq[k], apx[k], apy[k], apz[k]);
        } else {
            q[k] = k_src[ib_row + bone];
            if (has_lt) { const float *t = glt + (ib_row + bone) * 3; ltx[k] = t[0]; lty[k] = t[1]; ltz[k] = t[2]; }
            if (ap >= 0) {
                apq[k] = k_src[ib_row + (size_t)ap];
                if (has_lt) { const float *t = glt + (ib_row + (size_t)ap) * 3; apx[k] = t[0]; apy[k] = t[1]; apz[k] = t[2]; }
            }
        }
        ib0[k] = ib1[k] = ib2[k] = ib3[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pslot != kNoAnc) {
            const float4 *gi = reinterpret_cast<const float4 *>(k_inv_bind) + (size_t)bone * 4;
            ib0[k] = gi[0]; ib1[k] = gi[1]; ib2[k] = gi[2]; ib3[k] = gi[3];
        }
    }
    // ---- local matrices into buffer A ----
    float4 rm[NIT][3];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        fk_local_matrix(q[k], w0[k], __uint_as_float(w1[k].x), __uint_as_float(w1[k].y), __uint_as_float(w1[k].z), sampled || has_lt, ltx[k], lty[k], ltz[k],
                        apq[k], apx[k], apy[k], apz[k], rm[k][0], rm[k][1], rm[k][2]);
        if (on[k]) { float4 *d = mA + ((size_t)ig[k] * nc + ic[k]) * 3; d[0] = rm[k][0]; d[1] = rm[k][1]; d[2] = rm[k][2]; }
    }
    RZ_STAMP(1);                 // local matrices formed
    __syncthreads();
    // ---- doubling rounds over the closure (ancestor SLOTS; a pose's matrices sit nc x 3 float4 apart) ----
    float4 *src = mA, *dst = mB;
    for (int r = 0; r < n_rounds; ++r) {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const uint32_t lo = r == 0 ? w2[k].x : r == 1 ? w2[k].z : w3[k].x, hi = r == 0 ? w2[k].y : r == 1 ? w2[k].w : w3[k].y;
            fk_round(src + (size_t)ig[k] * nc * 3, lo & 0xffffu, lo >> 16, hi & 0xffffu, rm[k][0], rm[k][1], rm[k][2]);
            if (on[k]) { float4 *d = dst + ((size_t)ig[k] * nc + ic[k]) * 3; d[0] = rm[k][0]; d[1] = rm[k][1]; d[2] = rm[k][2]; }
        }
        __syncthreads();
        float4 *t4 = src; src = dst; dst = t4;
    }
    // ---- palette rows of the named bones ----
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const uint32_t pslot = w0[k].x >> 16;
        if (on[k] && pslot != kNoAnc) {
            float4 q0, q1, q2;
            fk_palette_rows(rm[k][0], rm[k][1], rm[k][2], ib0[k], ib1[k], ib2[k], ib3[k], q0, q1, q2);
            float4 *d = pal + ((size_t)ig[k] * ns + pslot) * 3;
            d[0] = q0; d[1] = q1; d[2] = q2;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the first vertex has landed)
    __syncthreads();
    RZ_STAMP(2);                 // palettes formed and published: the front is over
#include "crowd_pose_loop.inc.h"
    RZ_STAMP(5);                 // last vertex step issued
    RZ_TL_FLUSH(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (BLOCK / 64) + wave);
}

#ifdef RZ_ALL_VARIANTS
// ------------------------------------------------------------------------------------------------
// instanced skin, register-resident form: a workgroup owns a run of KV*256 vertices and a RANGE of poses.
// Each lane loads and decodes its KV vertices ONCE into registers (the static mesh is read once per
// (run, pose range) instead of once per pose), then walks the poses: the palette of pose g+1 streams into
// the other half of a 2-deep LDS ring by LDS-DMA while pose g is skinned, and every pose is written as one
// contiguous KV*256*12-byte block per output array. LDS is only 2 palettes (19 KB at 200 bones).
// grid = (vertex runs, pose ranges); block = 256.
// ------------------------------------------------------------------------------------------------
template <int KV, bool NTS>
__global__ void __launch_bounds__(kBlock) rz_skin_instances_reg_kernel(const RzDeformParams p, int n_inst, int poses_per_wg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *ring = reinterpret_cast<float4 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rows = p.B * 3;
    const int inst0 = blockIdx.y * poses_per_wg;
    const int ng = min(poses_per_wg, n_inst - inst0);
    const size_t Vp = p.Vp;
    const uint32_t v_lim = p.n_quads * 4u;
    const uint32_t v0 = blockIdx.x * (KV * kBlock) + tid;
    const uint32_t bmax = (uint32_t)(p.B - 1);

    auto dma_palette = [&](int g) {
        const float4 *src = p.palette + (size_t)(inst0 + g) * rows;
        float4 *dst = ring + (size_t)(g & 1) * rows;
        for (int c = wave * 64; c < rows; c += kBlock) {
            const int e = c + lane;
            if (e < rows) {
                typedef const __attribute__((address_space(1))) void *gptr_t;
                typedef __attribute__((address_space(3))) void *lptr_t;
                __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)(src + e), (lptr_t)(uint32_t)(uintptr_t)(dst + c), 16, 0, 0);
            }
        }
    };
    dma_palette(0);

    // ---- decode KV vertices per lane, once ----
    f2 vx[KV], vy[KV], vz[KV];
    float w0[KV], w1[KV], w2[KV], w3[KV];
    uint32_t j01[KV], j23[KV];
    {
        float x[KV], y[KV], z[KV], nx[KV], ny[KV], nz[KV];
        uint32_t wq[KV];
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const uint32_t v = v0 + k * kBlock;
            const bool live = v < v_lim;
            const size_t vs = live ? v : 0;
            x[k] = p.geom[0 * Vp + vs]; y[k] = p.geom[1 * Vp + vs]; z[k] = p.geom[2 * Vp + vs];
            nx[k] = p.geom[3 * Vp + vs]; ny[k] = p.geom[4 * Vp + vs]; nz[k] = p.geom[5 * Vp + vs];
            j01[k] = p.joints01[vs]; j23[k] = p.joints23[vs]; wq[k] = p.weights[vs];
        }
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const uint32_t b0 = wq[k] & 255u, b1 = (wq[k] >> 8) & 255u, b2 = (wq[k] >> 16) & 255u, b3 = wq[k] >> 24;
            const uint32_t isum = b0 + b1 + b2 + b3;
            const bool ok = isum != 0u;
            const float inv = __builtin_amdgcn_rcpf((float)(ok ? isum : 1u));
            w0[k] = ok ? (float)b0 * inv : 1.0f; w1[k] = (float)b1 * inv; w2[k] = (float)b2 * inv; w3[k] = (float)b3 * inv;
            // joints -> clamped palette row offsets, two 16-bit fields per register (B*3 <= 65535 is checked on the host)
            const uint32_t o0 = min(j01[k] & 0xffffu, bmax) * 3u, o1 = min(j01[k] >> 16, bmax) * 3u;
            const uint32_t o2 = min(j23[k] & 0xffffu, bmax) * 3u, o3 = min(j23[k] >> 16, bmax) * 3u;
            j01[k] = o0 | (o1 << 16); j23[k] = o2 | (o3 << 16);
            vx[k] = f2{x[k], nx[k]}; vy[k] = f2{y[k], ny[k]}; vz[k] = f2{z[k], nz[k]};
        }
    }
    const f2 vw = {1.0f, 0.0f};

    for (int g = 0; g < ng; ++g) {
        // pose g's palette has landed (own DMA drained, barrier publishes everyone's part and also retires
        // every wave's reads of the other ring slot, which pose g+1 may now overwrite)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (g + 1 < ng) dma_palette(g + 1);
        const float4 *pg = ring + (size_t)(g & 1) * rows;
        float *op = p.out_pos + (size_t)(inst0 + g) * Vp * 3;
        float *on = p.out_nrm + (size_t)(inst0 + g) * Vp * 3;
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const uint32_t v = v0 + k * kBlock;
            const uint32_t o0 = j01[k] & 0xffffu, o1 = j01[k] >> 16, o2 = j23[k] & 0xffffu, o3 = j23[k] >> 16;
            f2 r[3][2];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float4 a = pg[o0 + q], c = pg[o1 + q], d = pg[o2 + q], e = pg[o3 + q];
                const f2 axy = {a.x, a.y}, azw = {a.z, a.w}, cxy = {c.x, c.y}, czw = {c.z, c.w};
                const f2 dxy = {d.x, d.y}, dzw = {d.z, d.w}, exy = {e.x, e.y}, ezw = {e.z, e.w};
                r[q][0] = w3[k] * exy + (w2[k] * dxy + (w1[k] * cxy + w0[k] * axy));
                r[q][1] = w3[k] * ezw + (w2[k] * dzw + (w1[k] * czw + w0[k] * azw));
            }
            const f2 q0 = r[0][0].x * vx[k] + (r[0][0].y * vy[k] + (r[0][1].x * vz[k] + r[0][1].y * vw));
            const f2 q1 = r[1][0].x * vx[k] + (r[1][0].y * vy[k] + (r[1][1].x * vz[k] + r[1][1].y * vw));
            const f2 q2 = r[2][0].x * vx[k] + (r[2][0].y * vy[k] + (r[2][1].x * vz[k] + r[2][1].y * vw));
            const float tx = q0.y, ty = q1.y, tz = q2.y;
            const float l2 = fmaf(tz, tz, fmaf(ty, ty, tx * tx));
            const bool good = (l2 > 0.0f) && (l2 < __builtin_inff());
            const float rl = __builtin_amdgcn_rsqf(good ? l2 : 1.0f);
            if (v < v_lim) {
                st3<NTS>(op + (size_t)v * 3, q0.x, q1.x, q2.x);
                st3<NTS>(on + (size_t)v * 3, good ? tx * rl : vx[k].y, good ? ty * rl : vy[k].y, good ? tz * rl : vz[k].y);
            }
            __builtin_amdgcn_sched_barrier(0);   // one vertex at a time: bounds the live palette rows (12 x float4)
        }
    }
}
#endif  // RZ_ALL_VARIANTS

// zero-weight influence still gathers its bone's rows, so it stays the SAME bone: fma(0, row, m) keeps m's bits only while the
// row is the one the full-palette form would have read), ranks them ascending, writes the list and the joints as slots of it.
__global__ void __launch_bounds__(kBlock) rz_run_subsets_kernel(const uint32_t *j01, const uint32_t *j23, uint32_t v_lim, uint32_t per,
                                                                uint32_t B, uint16_t *list, uint32_t *count, uint32_t *rj01, uint32_t *rj23)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t nw = (B + 31) / 32;
    uint32_t *bits = reinterpret_cast<uint32_t *>(smem);            // [nw] bone bitmap
    uint32_t *before = bits + nw;                                    // [nw + 1] listed bones in front of each word
    uint16_t *slot = reinterpret_cast<uint16_t *>(before + nw + 1);  // [B]
    const uint32_t tid = threadIdx.x, run = blockIdx.x;
    const uint32_t v0 = run * per, v1 = min(v_lim, v0 + per), bmax = B - 1;
    for (uint32_t i = tid; i < nw; i += kBlock) bits[i] = 0;
    __syncthreads();
    for (uint32_t v = v0 + tid; v < v1; v += kBlock) {
        const uint32_t a = j01[v], b = j23[v];
        const uint32_t j[4] = { min(a & 0xffffu, bmax), min(a >> 16, bmax), min(b & 0xffffu, bmax), min(b >> 16, bmax) };
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicOr(&bits[j[k] >> 5], 1u << (j[k] & 31u));
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0;
        for (uint32_t i = 0; i < nw; ++i) { before[i] = acc; acc += __popc(bits[i]); }
        before[nw] = acc;
        count[run] = acc;
    }
    __syncthreads();
    for (uint32_t b = tid; b < B; b += kBlock) {
        const uint32_t w = bits[b >> 5], m = 1u << (b & 31u);
        if (w & m) {
            const uint32_t sl = before[b >> 5] + __popc(w & (m - 1u));
            slot[b] = (uint16_t)sl;
            list[(size_t)run * B + sl] = (uint16_t)b;
        }
    }
    for (uint32_t sl = before[nw] + tid; sl < B; sl += kBlock) list[(size_t)run * B + sl] = 0;      // padding: a valid bone (see the crowd kernel)
    __syncthreads();
    for (uint32_t v = v0 + tid; v < v1; v += kBlock) {
        const uint32_t a = j01[v], b = j23[v];
        rj01[v] = (uint32_t)slot[min(a & 0xffffu, bmax)] | ((uint32_t)slot[min(a >> 16, bmax)] << 16);
        rj23[v] = (uint32_t)slot[min(b & 0xffffu, bmax)] | ((uint32_t)slot[min(b >> 16, bmax)] << 16);
    }
}

}  // namespace

size_t rz_skin_instances_lds_bytes(int G, uint32_t bones, bool dma, bool subsets)
{
    // finished palettes are 48 B per (pose, bone). One-launch frame: the whole-palette form stages the world matrices in
    // the palette region's place (64-byte slots, re-packed in place); the subset form stages them behind it (48 + 64).
    const size_t per = dma ? 48 : (subsets ? 112 : 64);
    return (size_t)G * bones * per;
}

template <int BLOCK>
static hipError_t launch_skin_instances(const RzDeformParams &p, int G, int n_inst, uint32_t verts_per_wg, uint32_t grid_x,
                                        bool nts, size_t lds, hipStream_t st)
{
    const bool sub = p.sub_list != nullptr;
    auto k = sub ? (nts ? rz_skin_instances_kernel<BLOCK, true, true> : rz_skin_instances_kernel<BLOCK, false, true>)
                 : (nts ? rz_skin_instances_kernel<BLOCK, true, false> : rz_skin_instances_kernel<BLOCK, false, false>);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    dim3 grid(grid_x, (n_inst + G - 1) / G);
    if (grid.x > 0xffffu || grid.y > 0xffffu || p.B > 0xffff || (sub && (p.sub_stride != p.B || p.sub_max <= 0 || p.sub_max > 0x3fff))) return hipErrorInvalidValue;      // (k_grid packs both; the lists' stride is the bone count; k_bf carries their padded length in 14 bits)
    // leading arguments = what the front of a workgroup needs, preloaded into SGPRs (see the kernel)
    const float4 *k_src = p.dma ? p.palette : reinterpret_cast<const float4 *>(p.world);
    const uint32_t k_grid = grid.x | (grid.y << 16), k_bf = (uint32_t)p.B | (p.inst_order ? 1u << 16 : 0u) | (p.dma ? 1u << 17 : 0u) | (sub ? (uint32_t)p.sub_max << 18 : 0u);
    hipLaunchKernelGGL(k, grid, dim3(BLOCK), lds, st, p.sub_count, p.sub_list, k_src, p.inv_bind, G, n_inst, verts_per_wg, k_grid, k_bf, p.Vp, p);
    return hipGetLastError();
}

size_t rz_skin_instances_fk_lds_bytes(int G, uint32_t closure, uint32_t named) { return (size_t)G * (2 * (size_t)closure + named) * 48; }

template <int BLOCK>
static hipError_t launch_skin_instances_fk(const RzDeformParams &p, const RzSubFk &f, int G, int n_inst, uint32_t verts_per_wg, uint32_t grid_x, bool nts,
                                           size_t lds, hipStream_t st)
{
    auto k = nts ? rz_skin_instances_fk_kernel<BLOCK, true> : rz_skin_instances_fk_kernel<BLOCK, false>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    dim3 grid(grid_x, (n_inst + G - 1) / G);
    const bool sampled = p.fk.sample.frames != nullptr;
    // what the kernel can take: two 16-bit grid fields, 16 bits of bone count, 12 bits of record stride, 2 bits of rounds, two work items per thread
    if (grid.x > 0xffffu || grid.y > 0xffffu || p.B > 0xffff || f.stride > 0xfffu || f.rounds > 3 || (size_t)G * f.stride > 2 * (size_t)BLOCK || !p.sub_count || !p.rj01)
        return hipErrorInvalidValue;
    const float4 *k_src = sampled ? reinterpret_cast<const float4 *>(p.fk.sample.frames) : p.fk.local_q;
    if (p.sub_max <= 0 || p.sub_max > 0x3fff) return hipErrorInvalidValue;
    const uint32_t k_g = (uint32_t)G | (sampled ? 1u << 16 : 0u) | (p.fk.local_t ? 1u << 17 : 0u) | ((uint32_t)p.sub_max << 18);
    const uint32_t k_grid = grid.x | (grid.y << 16), k_bf = (uint32_t)p.B | (p.inst_order ? 1u << 16 : 0u) | (f.rounds << 18) | (f.stride << 20);
    hipLaunchKernelGGL(k, grid, dim3(BLOCK), lds, st, f.count, f.rec, k_src, p.inv_bind, k_g, n_inst, verts_per_wg, k_grid, k_bf, p.Vp, p);
    return hipGetLastError();
}

hipError_t rz_launch_skin_instances_fk(const RzDeformParams &p, const RzSubFk &f, int G, int n_inst, uint32_t verts_per_wg, uint32_t grid_x,
                                       int block, bool nts, size_t lds_bytes, hipStream_t st)
{
    if (block == 1024) return launch_skin_instances_fk<1024>(p, f, G, n_inst, verts_per_wg, grid_x, nts, lds_bytes, st);
    if (block == 512) return launch_skin_instances_fk<512>(p, f, G, n_inst, verts_per_wg, grid_x, nts, lds_bytes, st);
    return launch_skin_instances_fk<256>(p, f, G, n_inst, verts_per_wg, grid_x, nts, lds_bytes, st);
}

hipError_t rz_launch_skin_instances(const RzDeformParams &p, int G, int n_inst, uint32_t verts_per_wg, uint32_t grid_x,
                                    int block, bool nts, size_t lds_bytes, hipStream_t st)
{
    if (block == 1024) return launch_skin_instances<1024>(p, G, n_inst, verts_per_wg, grid_x, nts, lds_bytes, st);
    if (block == 512) return launch_skin_instances<512>(p, G, n_inst, verts_per_wg, grid_x, nts, lds_bytes, st);
    return launch_skin_instances<256>(p, G, n_inst, verts_per_wg, grid_x, nts, lds_bytes, st);
}

hipError_t rz_launch_run_subsets(const uint32_t *j01, const uint32_t *j23, uint32_t v_lim, uint32_t per, uint32_t runs, uint32_t B,
                                 uint16_t *list, uint32_t *count, uint32_t *rj01, uint32_t *rj23, hipStream_t st)
{
    if (runs == 0) return hipSuccess;
    const uint32_t nw = (B + 31) / 32;
    const size_t lds = (size_t)(2 * nw + 1) * 4 + (size_t)B * 2;
    hipLaunchKernelGGL(rz_run_subsets_kernel, dim3(runs), dim3(kBlock), lds, st, j01, j23, v_lim, per, B, list, count, rj01, rj23);
    return hipGetLastError();
}

hipError_t rz_launch_skin_instances_reg(const RzDeformParams &p, int n_inst, int poses_per_wg, uint32_t grid_x, bool nts,
                                        hipStream_t st)
{
#ifdef RZ_ALL_VARIANTS
    constexpr int KV = 8;
    const size_t lds = (size_t)2 * p.B * 48;
    auto k = nts ? rz_skin_instances_reg_kernel<KV, true> : rz_skin_instances_reg_kernel<KV, false>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    dim3 grid(grid_x, (n_inst + poses_per_wg - 1) / poses_per_wg);
    hipLaunchKernelGGL(k, grid, dim3(kBlock), lds, st, p, n_inst, poses_per_wg);
    return hipGetLastError();
#else
    (void)p; (void)n_inst; (void)poses_per_wg; (void)grid_x; (void)nts; (void)st;
    return hipErrorInvalidValue;       // the register-resident crowd kernel (measured slower, inst_loop = 9) is a tools-only variant
#endif
}