// kernels/common.hip.h — what every kernel file of libreze_deform.so shares: block size, the tools-only instrumentation macros,
// the streaming load / store forms, the ordered active-morph compaction and the per-vertex skinning arithmetic.
//
// The kernels (hand-written CDNA4 / gfx950, memory-bound gather-transforms, no MFMA) replace, in the reference (paths relative to the
// reference repo root):
//   rz_prep_kernel (front.hip)        the WGSL compute shader engine/src/engine.ts:906-930 (skinMatrices[b] = worldMatrices[b] *
//                                     inverseBindMatrices[b]) — plus the active-morph compaction, which has no reference counterpart
//   rz_deform_dense_kernel (deform_dense.hip) / rz_deform_small_kernel (deform_small.hip) / rz_skin_instances_kernel (crowd.hip)
//                                     the skinning body of the WGSL vertex shader vs() engine/src/engine.ts:253-272 (and its copies
//                                     :440-443, :700-703), run once per frame instead of once per draw pass, fused with vertex-morph
//                                     accumulation (new capability; the reference skips PMX morphs, engine/src/pmx-loader.ts:450-553)
//   rz_fk_kernel (front.hip, fk.hip.h) Model.computeWorldMatrices engine/src/model.ts:330-420 on the device
// Shared design:
//   * static mesh is planar SoA in HBM: x[],y[],z[],nx[],ny[],nz[] float planes, joints as two u32 planes (j0|j1<<16, j2|j3<<16),
//     weights as one u32 plane, dense morph targets as D[m][3][Vp] planes. A lane owns a QUAD of 4 consecutive vertices, so every
//     stream is read with one 16-byte load per lane = 1 KiB contiguous per wave-instruction.
//   * the per-instance bone palette (3x4 affine rows, 48 B/bone) is staged once per workgroup into LDS; bones are gathered from LDS
//     with ds_read_b128.
//   * outputs are packed float3 arrays (the vertex-buffer layout a renderer binds).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "../deform_kernels.h"

#pragma clang diagnostic ignored "-Wint-to-void-pointer-cast"   // 32-bit LDS pointers built from integers

namespace {

constexpr int kBlock = 256;

// Ablation switches for profiling experiments exist only in the tools-only build (make ablate ->
// tools/ablate/libreze_deform_ablate.so, -DRZ_ABLATE). In the shipped library RZ_DBG is the constant 0, the branches
// fold away, and rz_set_tuning("dbg", ...) is rejected: no key can make rz_deform emit anything but the deformed mesh.
#ifdef RZ_ABLATE
#define RZ_DBG(p) ((p).dbg)
// Per-wave TIMELINE (tools-only build, rz_set_tuning dbg = 100; tools/archive/timeline.py): a wave keeps up to seven readings of the
// chip-wide 100 MHz counter (s_memrealtime: 10 ns steps, the same clock on every XCD) in scalar registers and writes them out
// when it ends, with the XCC / CU / SIMD it ran on (16 x u64 per wave: 0..6 stamps, 7 where, 8..12 stamps inside the fused
// hierarchy solve). Slot 6 is taken after every store of the wave has been acknowledged.
#define RZ_TL_DECL unsigned long long tl_t[7] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull}, tl_f[5] = {0ull, 0ull, 0ull, 0ull, 0ull}
#define RZ_STAMP(k) do { if (p.tl) tl_t[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define RZ_TL_FLUSH(wave_index) do { if (p.tl) { \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
        tl_t[6] = __builtin_amdgcn_s_memrealtime(); \
        if ((threadIdx.x & 63) == 0) { \
            unsigned long long *o_ = p.tl + (size_t)(wave_index) * 16; \
            for (int k_ = 0; k_ < 7; ++k_) o_[k_] = tl_t[k_]; \
            for (int k_ = 0; k_ < 5; ++k_) o_[8 + k_] = tl_f[k_]; \
            o_[7] = (unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) | \
                    ((unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) << 8); \
        } } } while (0)
#define RZ_TL_FK (p.tl ? tl_f : nullptr)
#else
#define RZ_DBG(p) 0
#define RZ_TL_DECL do {} while (0)
#define RZ_STAMP(k) do {} while (0)
#define RZ_TL_FLUSH(wave_index) do {} while (0)
#define RZ_TL_FK nullptr
#endif

__device__ __forceinline__ float4 ld_stream(const float4 *p, bool nt)
{
    // morph planes are read exactly once per frame: optionally bypass-hint the load
    if (nt) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *p;
}

// Ordered compaction of the non-zero morph weights of one pose (whole workgroup of kBlock threads; `mw` may be LDS or global,
// `aidx` / `aw` likewise): entries keep ascending morph order = the oracle's accumulation order; the tail up to Mpad is
// zero-padded so unrolled readers may over-read harmlessly. Returns the number of active morphs (workgroup-uniform).
__device__ __forceinline__ int compact_active(const float *mw, const int M, const int Mpad, uint32_t *aidx, float *aw, int *wave_cnt)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int base = 0;
    for (int m0 = 0; m0 < M; m0 += kBlock) {
        const int m = m0 + tid;
        const float w = (m < M) ? mw[m] : 0.0f;
        const bool on = (w != 0.0f);
        const unsigned long long bal = __ballot(on);
        const int rank = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < kBlock / 64; ++k) {
            const int c = wave_cnt[k];
            before += (k < wave) ? c : 0;
            total += c;
        }
        if (on) {
            aidx[base + before + rank] = (uint32_t)m;
            aw[base + before + rank] = w;
        }
        base += total;
        __syncthreads();
    }
    for (int k = base + tid; k < Mpad; k += kBlock) { aidx[k] = 0; aw[k] = 0.0f; }
    return base;
}

// ------------------------------------------------------------------------------------------------
// helpers for the skin phase
// ------------------------------------------------------------------------------------------------
struct Skinned { float px, py, pz, nx, ny, nz; };

// vs() lines engine.ts:255-272 for one vertex. `pal` = LDS palette (3 float4 rows per bone).
//   weights: w_i = (u8_i/255) / sum_k(u8_k/255)  (engine.ts:255-257)  ==  u8_i / isum  up to rounding;
//            isum == 0 takes the select((1,0,0,0)) branch (a sum of unorm8 values is either 0 or >= 1/255 > 1e-4).
//   blend:   because the map is linear, M = sum_i w_i * S[j_i] is formed once (12 FMA per bone) and applied
//            to the position and the normal, instead of transforming both by every bone (24 FMA per bone).
//            Rounding differs from the oracle's evaluation order by a few ulp (tolerance 1e-4).
__device__ __forceinline__ Skinned skin_vertex(const float4 *pal, float x, float y, float z, float nx, float ny,
                                               float nz, uint32_t j01, uint32_t j23, uint32_t wq, uint32_t bmax)
{
    const uint32_t b0 = wq & 255u, b1 = (wq >> 8) & 255u, b2 = (wq >> 16) & 255u, b3 = wq >> 24;
    const uint32_t isum = b0 + b1 + b2 + b3;
    const bool ok = isum != 0u;
    const float inv = __builtin_amdgcn_rcpf((float)(ok ? isum : 1u));
    const float w[4] = { ok ? (float)b0 * inv : 1.0f, (float)b1 * inv, (float)b2 * inv, (float)b3 * inv };
    // joints are < B by construction (pmx-loader.ts:861-880); clamp so bad input cannot read past the palette
    const uint32_t j[4] = { min(j01 & 0xffffu, bmax), min(j01 >> 16, bmax), min(j23 & 0xffffu, bmax),
                            min(j23 >> 16, bmax) };
    float4 m0 = make_float4(0.f, 0.f, 0.f, 0.f), m1 = m0, m2 = m0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 r0 = pal[j[i] * 3 + 0], r1 = pal[j[i] * 3 + 1], r2 = pal[j[i] * 3 + 2];
        m0.x = fmaf(w[i], r0.x, m0.x); m0.y = fmaf(w[i], r0.y, m0.y); m0.z = fmaf(w[i], r0.z, m0.z); m0.w = fmaf(w[i], r0.w, m0.w);
        m1.x = fmaf(w[i], r1.x, m1.x); m1.y = fmaf(w[i], r1.y, m1.y); m1.z = fmaf(w[i], r1.z, m1.z); m1.w = fmaf(w[i], r1.w, m1.w);
        m2.x = fmaf(w[i], r2.x, m2.x); m2.y = fmaf(w[i], r2.y, m2.y); m2.z = fmaf(w[i], r2.z, m2.z); m2.w = fmaf(w[i], r2.w, m2.w);
    }
    Skinned o;
    o.px = fmaf(m0.z, z, fmaf(m0.y, y, fmaf(m0.x, x, m0.w)));
    o.py = fmaf(m1.z, z, fmaf(m1.y, y, fmaf(m1.x, x, m1.w)));
    o.pz = fmaf(m2.z, z, fmaf(m2.y, y, fmaf(m2.x, x, m2.w)));
    const float tx = fmaf(m0.z, nz, fmaf(m0.y, ny, m0.x * nx));
    const float ty = fmaf(m1.z, nz, fmaf(m1.y, ny, m1.x * nx));
    const float tz = fmaf(m2.z, nz, fmaf(m2.y, ny, m2.x * nx));
    const float l2 = fmaf(tz, tz, fmaf(ty, ty, tx * tx));
    // normalize() of a zero / non-finite vector is undefined in WGSL: the rest normal is returned (build-defined)
    const bool good = (l2 > 0.0f) && (l2 < __builtin_inff());
    const float rl = __builtin_amdgcn_rsqf(good ? l2 : 1.0f);
    o.nx = good ? tx * rl : nx;
    o.ny = good ? ty * rl : ny;
    o.nz = good ? tz * rl : nz;
    return o;
}

template <bool NTS> __device__ __forceinline__ void st3(float *d, float a, float b, float c)
{
    if (NTS) {
        __builtin_nontemporal_store(a, d); __builtin_nontemporal_store(b, d + 1); __builtin_nontemporal_store(c, d + 2);
    } else {
        d[0] = a; d[1] = b; d[2] = c;
    }
}

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }   // v_pk_fma_f32

}  // namespace
