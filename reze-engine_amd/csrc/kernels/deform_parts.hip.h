// kernels/deform_parts.hip.h — the pieces the two single-mesh frame kernels share (deform_dense.hip: the dense morph STREAM, bound by HBM
// reads; deform_small.hip: frames without one — one character, sparse targets — bound by the LATENCY of a few dependent round trips):
// the pose-prefetch helper workgroup, the palette product, write batching, the per-vertex outputs and the bounding-box epilogue.
// Both kernels take the same arguments (the leading ones are preloaded into SGPRs, see below) and the same RzDeformParams.
//
// Kernel-argument preload (round 4): the files are compiled with -amdgpu-kernarg-preload-count=16, so the command processor hands the
// first 14 dwords of scalar arguments over in SGPRs when a wave starts (16 user SGPRs less the kernel-argument pointer: everything up
// to k_j23; k_wq follows by scalar load like `p`). The leading arguments therefore repeat what the FIRST loads of a wave need — the
// matrices, the partition, the mesh planes — and
//   k_bf = bone count | helper-workgroup flag << 16 | may-be-staged flag << 17 | pose-in-pinned-memory flag << 18 | worker workgroups << 19.
// Launches that are NOT the one-launch form (!FAST) do not read the matrices themselves, so their k_world / k_inv_bind slots carry what
// a fused-hierarchy frame (fk_on) needs first instead: k_world = the hierarchy's static block (RzFkParams::bone_rec; null = no fused
// solve), k_inv_bind = the number of vertex morphs the motion samples, as an integer — the records of the thread's bones and morph are
// then on their way before `p` has arrived (kernels/fk.hip.h: fk_issue_static).
// Everything else comes out of `p` by scalar loads, which take ~0.9 us to arrive (profiles/r4_timeline_c2.txt: "entry -> prologue
// done"): a 3-17 us frame no longer waits for them before asking for its matrices and its mesh.
#pragma once
#include "fk.hip.h"

namespace {

// Zero-copy pose prefetch (RzDeformParams::pf_*): workgroup 0 of a launch that carries a helper. Seqlock read of the next upload's
// pinned slot: header == the expected sequence number -> the host has finished writing that pose (it writes the header last); copy;
// header again; only then the tag. The ring protocol already keeps the host from re-using the slot while this kernel runs, the second
// look is belt and braces.
__device__ __forceinline__ void pose_prefetch_helper(const RzDeformParams &p, const int tid)
{
    const uint64_t h1 = __builtin_nontemporal_load(p.pf_src_seq);
    if (h1 != p.pf_expect) return;                                   // workgroup-uniform
    const float4 *src = reinterpret_cast<const float4 *>(p.pf_src);
    float4 *dst = reinterpret_cast<float4 *>(p.pf_dst);
    const uint32_t n4 = p.pf_bytes / 16;
    // eight independent host loads in flight per thread (a 16.6 KB pose is one pass); indices past the end are clamped,
    // so the tail threads re-copy the last cell instead of branching
    const uint32_t last = n4 - 1;
    for (uint32_t i = tid; i < n4; i += kBlock * 8) {
        const uint32_t i0 = min(i, last), i1 = min(i + kBlock, last), i2 = min(i + 2 * kBlock, last), i3 = min(i + 3 * kBlock, last);
        const uint32_t i4 = min(i + 4 * kBlock, last), i5 = min(i + 5 * kBlock, last), i6 = min(i + 6 * kBlock, last), i7 = min(i + 7 * kBlock, last);
        const float4 a0 = src[i0], a1 = src[i1], a2 = src[i2], a3 = src[i3], a4 = src[i4], a5 = src[i5], a6 = src[i6], a7 = src[i7];
        dst[i0] = a0; dst[i1] = a1; dst[i2] = a2; dst[i3] = a3; dst[i4] = a4; dst[i5] = a5; dst[i6] = a6; dst[i7] = a7;
    }
    __syncthreads();
    if (tid == 0) {
        const uint64_t h2 = __builtin_nontemporal_load(p.pf_src_seq);
        if (h2 == p.pf_expect) *p.pf_tag = p.pf_expect;               // consumed by the NEXT kernel on this stream
    }
}

// palette rows 0..2 of world * inverseBind for bone b, out[c*4+r] = ((a0[r]*b0 + a1[r]*b1) + a2[r]*b2) + a3[r]*b3 (engine.ts:928):
// into the LDS palette and, when `gp` is not null (workgroup 0: keep the skinMatrixBuffer observable, rz_read_palette), into memory
__device__ __forceinline__ void palette_rows_to(float4 *pal, float4 *gp, const int b, const float4 &a0, const float4 &a1, const float4 &a2,
                                                const float4 &a3, const float4 &b0, const float4 &b1, const float4 &b2, const float4 &b3)
{
    const float4 bc[4] = { b0, b1, b2, b3 };
    float r0[4], r1[4], r2[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        r0[c] = fmaf(a3.x, bc[c].w, fmaf(a2.x, bc[c].z, fmaf(a1.x, bc[c].y, a0.x * bc[c].x)));
        r1[c] = fmaf(a3.y, bc[c].w, fmaf(a2.y, bc[c].z, fmaf(a1.y, bc[c].y, a0.y * bc[c].x)));
        r2[c] = fmaf(a3.z, bc[c].w, fmaf(a2.z, bc[c].z, fmaf(a1.z, bc[c].y, a0.z * bc[c].x)));
    }
    const float4 q0 = make_float4(r0[0], r0[1], r0[2], r0[3]), q1 = make_float4(r1[0], r1[1], r1[2], r1[3]),
                 q2 = make_float4(r2[0], r2[1], r2[2], r2[3]);
    pal[b * 3 + 0] = q0; pal[b * 3 + 1] = q1; pal[b * 3 + 2] = q2;
    if (gp) {
        gp += (size_t)b * 3;
        gp[0] = q0; gp[1] = q1; gp[2] = q2;
    }
}

// Write batching (RzDeformParams::out_cap > 0): deformed vertices are parked in a per-wave LDS buffer and written as 16-byte-per-lane
// stores when it fills and at the end of a wave's run. Interleaving 24 B of stores per vertex with the read stream cost 10.7 us of a
// 130 us C5 frame (ablation dbg 5) although the bytes are only 3 % of the traffic; batched, the HBM write bursts are long and rare.
// Writes `fill` parked vertices (a multiple of 4) to vertex v0 of the output arrays.
template <bool NTS>
__device__ __forceinline__ void flush_parked(float *opos, float *onrm, const float *ob_pos, const float *ob_nrm, const size_t v0, const uint32_t fill, const int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t n4 = fill * 3 / 4;                    // float4 per array (runs are multiples of 4 vertices)
    float4 *gp = reinterpret_cast<float4 *>(opos + v0 * 3);
    float4 *gn = reinterpret_cast<float4 *>(onrm + v0 * 3);
    const float4 *lp = reinterpret_cast<const float4 *>(ob_pos);
    const float4 *ln = reinterpret_cast<const float4 *>(ob_nrm);
    for (uint32_t i = lane; i < n4; i += 64) {
        if (NTS) {
            typedef float f4v __attribute__((ext_vector_type(4)));
            const float4 a = lp[i], b = ln[i];
            __builtin_nontemporal_store(f4v{a.x, a.y, a.z, a.w}, reinterpret_cast<f4v *>(gp + i));
            __builtin_nontemporal_store(f4v{b.x, b.y, b.z, b.w}, reinterpret_cast<f4v *>(gn + i));
        } else {
            gp[i] = lp[i]; gn[i] = ln[i];
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// One deformed vertex leaves the skin phase: parked (write batching) or stored, plus the fused consumers — the outline pass's
// inverted hull (engine.ts:458-461: expandedPos = worldPos + worldNormal * edgeSize * 0.01) and the running
// bounding box (padding vertices of the last quad stay out of it).
// EPI = false: a kernel variant without the fused consumers (frames of a context that has neither an edge scale nor the bounding box
// switched on — nearly all of them): the two uniform branches and the pointers behind them cost the latency-bound frames 2-3 %
// (profiles/r5_ab_epi.txt).
template <bool NTS, bool EPI = true>
__device__ __forceinline__ void emit_vertex(const RzDeformParams &p, const Skinned &o, const size_t v, const int inst, const size_t Vp, const uint32_t cap,
                                            float *ob_pos, float *ob_nrm, const uint32_t li, float *opos, float *onrm, float (&bb)[6])
{
    if (cap) {
        ob_pos[li] = o.px; ob_pos[li + 1] = o.py; ob_pos[li + 2] = o.pz;
        ob_nrm[li] = o.nx; ob_nrm[li + 1] = o.ny; ob_nrm[li + 2] = o.nz;
    } else if (RZ_DBG(p) != 5 || o.px == 1234.5f) {   // dbg 5: ablation — skin phase without its output stream
        st3<NTS>(opos + v * 3, o.px, o.py, o.pz);
        st3<NTS>(onrm + v * 3, o.nx, o.ny, o.nz);
    }
    if (EPI && p.edge) {
        const float e = p.edge[v];
        st3<NTS>(p.out_hull + ((size_t)inst * Vp + v) * 3, o.px + (o.nx * e) * 0.01f, o.py + (o.ny * e) * 0.01f,
                 o.pz + (o.nz * e) * 0.01f);
    }
    if (EPI && p.aabb && v < p.n_verts) {
        bb[0] = fminf(bb[0], o.px); bb[1] = fminf(bb[1], o.py); bb[2] = fminf(bb[2], o.pz);
        bb[3] = fmaxf(bb[3], o.px); bb[4] = fmaxf(bb[4], o.py); bb[5] = fmaxf(bb[5], o.pz);
    }
}

// Fused per-frame bounding box: per-lane running min/max -> wave butterfly -> one atomic per wave and component on order-preserving
// integer keys. The kernel also re-arms the OTHER slot for the next frame, so no memset launch sits between frames.
__device__ __forceinline__ void aabb_commit(const RzDeformParams &p, const int inst, float (&bb)[6], const int lane, const int tid, const uint32_t wid, const bool has_run)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            bb[k] = fminf(bb[k], __shfl_xor(bb[k], off));
            bb[3 + k] = fmaxf(bb[3 + k], __shfl_xor(bb[3 + k], off));
        }
    }
    uint32_t *slot = p.aabb + ((size_t)inst * 2 + (p.aabb_slot & 1)) * 6;
    if (lane < 6 && has_run) {
        const float sel = lane == 0 ? bb[0] : lane == 1 ? bb[1] : lane == 2 ? bb[2] : lane == 3 ? bb[3] : lane == 4 ? bb[4] : bb[5];
        const uint32_t bits = __float_as_uint(sel);
        const uint32_t key = bits ^ ((bits >> 31) ? 0xffffffffu : 0x80000000u);
        if (lane < 3) atomicMin(slot + lane, key); else atomicMax(slot + lane, key);
    }
    if (wid == 0 && tid < 6) {
        uint32_t *next = p.aabb + ((size_t)inst * 2 + ((p.aabb_slot + 1) & 1)) * 6;
        next[tid] = tid < 3 ? 0xffffffffu : 0u;
    }
}

// FUSED single-character frame (RzDeformParams::fk_on), the workgroup's prologue: hierarchy solve (and motion sampling) straight into
// the LDS palette `pal`; the solve's scratch and the pose's morph weights alias the wave scratch behind it (`work`), which nothing uses
// yet. A zero-copy local pose is staged in the device block by the previous frame's helper when the tag says so (requested here,
// compared after the speculative loads of the staged copy have been issued), else still in its pinned slot; on a miss workgroup 0
// leaves the pose in the device block for the frames that replay it. `stage_weights`: the uploaded (not sampled) morph weights are
// parked in LDS first — the morph modes and bone morphs need them. Returns where the pose's morph weights sit in LDS. Ends with a barrier.
template <bool STAGE_WEIGHTS_ALWAYS, int KIND = 0>
__device__ __forceinline__ float *fused_hierarchy_prologue(const RzFkParams &fk, const FkEarly &early, const uint64_t *st_tag, const uint64_t st_expect, const float *st_morph_w,
                                                           const float *morph_w, float *morph_w_copy, const int M, float4 *pal, float *work, const uint32_t wid,
                                                           unsigned long long *tl_f)
{
    const int tid = threadIdx.x;
    unsigned char *fscr = reinterpret_cast<unsigned char *>(work);
    float *lds_mw = reinterpret_cast<float *>(fscr + rz_fk_scratch_bytes(fk.B));
    const bool sampled = KIND == 2 || (KIND == 0 && (fk.sample.frames != nullptr || fk.sample.frames_inline));
    const bool fspec = KIND != 2 && st_tag != nullptr;
    const uint64_t ftag = fspec ? *st_tag : 0ull;
    if ((STAGE_WEIGHTS_ALWAYS || fk.bm_off) && !sampled) {
        const float *mw0 = fspec ? st_morph_w : morph_w;               // uploaded weights (staged copy, pinned slot or device block)
        for (int i = tid; i < M; i += kBlock) {
            float w = mw0[i];
            const bool miss = fspec && ftag != st_expect;
            if (miss) w = morph_w[i];
            lds_mw[i] = w;
            if (wid == 0 && morph_w_copy && (!fspec || miss)) morph_w_copy[i] = w;
        }
    }
    fk_solve<true, KIND>(fk, early, 0, pal, fscr, lds_mw, wid == 0, ftag, tl_f);       // ends with a barrier: pal and lds_mw are complete
    return lds_mw;
}

// k_bf of a launch of `grid_x` workgroups (see the top of the file)
inline uint32_t rz_deform_k_bf(const RzDeformParams &p, const uint32_t grid_x)
{
    const uint32_t workers = grid_x - (p.pf_src ? 1u : 0u);
    return (uint32_t)p.B | (p.pf_src ? 1u << 16 : 0u) | (p.st_tag ? 1u << 17 : 0u) | (p.world_copy ? 1u << 18 : 0u) | (workers < 8192u ? workers << 19 : 0u);
}

}  // namespace
