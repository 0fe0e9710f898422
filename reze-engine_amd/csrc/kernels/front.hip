// kernels/front.hip — what a frame may launch in FRONT of its deform / skin kernel (rz_prep_kernel: palette + active-morph list;
// rz_fk_kernel: hierarchy solve + motion sampling), the one-off upload re-layout kernels, and the host-side launch dispatch.
#include "fk.hip.h"

namespace {

// ------------------------------------------------------------------------------------------------
// prep: palette rows + ordered compaction of the non-zero morph weights. One workgroup per instance.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) rz_prep_kernel(RzPrepParams p)
{
    const int inst = blockIdx.x;
    const int tid = threadIdx.x;
    const float *world = p.world + (size_t)inst * p.B * 16;
    float4 *pal = p.palette + (size_t)inst * p.B * 3;

    for (int b = tid; b < p.B; b += kBlock) {
        const float4 *Wm = reinterpret_cast<const float4 *>(world + (size_t)b * 16);
        const float4 *Im = reinterpret_cast<const float4 *>(p.inv_bind + (size_t)b * 16);
        // column-major: a_k = column k of W (x,y,z = rows 0..2)
        float4 a0 = Wm[0], a1 = Wm[1], a2 = Wm[2], a3 = Wm[3];
        float r0[4], r1[4], r2[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4 bc = Im[c];
            // out[c*4+r] = ((a0[r]*b0 + a1[r]*b1) + a2[r]*b2) + a3[r]*b3   (engine.ts:928)
            r0[c] = fmaf(a3.x, bc.w, fmaf(a2.x, bc.z, fmaf(a1.x, bc.y, a0.x * bc.x)));
            r1[c] = fmaf(a3.y, bc.w, fmaf(a2.y, bc.z, fmaf(a1.y, bc.y, a0.y * bc.x)));
            r2[c] = fmaf(a3.z, bc.w, fmaf(a2.z, bc.z, fmaf(a1.z, bc.y, a0.z * bc.x)));
        }
        pal[b * 3 + 0] = make_float4(r0[0], r0[1], r0[2], r0[3]);
        pal[b * 3 + 1] = make_float4(r1[0], r1[1], r1[2], r1[3]);
        pal[b * 3 + 2] = make_float4(r2[0], r2[1], r2[2], r2[3]);
    }

    if (p.M > 0) {
        __shared__ int wave_cnt[kBlock / 64];
        const int n = compact_active(p.morph_w + (size_t)inst * p.M, p.M, p.Mpad, p.act_idx + (size_t)inst * p.Mpad, p.act_w + (size_t)inst * p.Mpad, wave_cnt);
        if (tid == 0) p.act_count[inst] = n;
    }
}

// One workgroup per pose. The leading arguments are preloaded into SGPRs (kernels/deform_parts.hip.h): the static block and the two
// counts a thread needs to ask for its records before `p` has arrived.
__global__ void __launch_bounds__(kBlock) rz_fk_kernel(const uint4 *k_rec, const uint32_t k_B, const uint32_t k_M, const RzFkParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const FkEarly early = fk_issue_static(k_rec, (int)k_B, (int)k_M, (int)threadIdx.x);
    unsigned char *scr = smem + (size_t)p.B * 48;
    fk_solve<false>(p, early, (int)blockIdx.x, reinterpret_cast<float4 *>(smem), scr, reinterpret_cast<float *>(scr + rz_fk_scratch_bytes(p.B)), true);
}

#ifdef RZ_ALL_VARIANTS

// Test hook of the tools-only build (tests/conftest.py: rzv): a one-thread kernel that holds its stream until the host opens
// the gate (a word in pinned memory) — so a test can put frames BEHIND it, write the next pose, and only then let them run:
// the pose-prefetch helper then finds the next pose complete by construction, not because the host happened to be ahead.
// Gives up after two seconds of the 100 MHz counter: a test that dies with the gate closed must not take the GPU with it.
__global__ void rz_gate_kernel(const uint32_t *flag)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_nontemporal_load(flag) == 0u) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) break;
        __builtin_amdgcn_s_sleep(64);
    }
}

#endif  // RZ_ALL_VARIANTS

// ------------------------------------------------------------------------------------------------
// upload-time re-layout kernels (one-off, not on the per-frame path)
// ------------------------------------------------------------------------------------------------
// packed [n][stride] floats -> planes; `stride` = 3 (packed xyz) or 8 (reference interleaved vertex)
__global__ void rz_deinterleave_kernel(const float *src, int stride, int offset, uint32_t n, float *px,
                                       float *py, float *pz)
{
    uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const float *s = src + (size_t)v * stride + offset;
    px[v] = s[0]; py[v] = s[1]; pz[v] = s[2];
}

__global__ void rz_pack_skinning_kernel(const uint16_t *joints4, const uint8_t *weights4, uint32_t n,
                                        uint32_t *j01, uint32_t *j23, uint32_t *wq)
{
    uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const uint2 j = reinterpret_cast<const uint2 *>(joints4)[v];
    j01[v] = j.x; j23[v] = j.y;
    wq[v] = reinterpret_cast<const uint32_t *>(weights4)[v];
}

// Plan-time pass of the bone-subset crowd frame (one-off per launch shape, not per frame): one workgroup per vertex run.

}  // namespace

// ------------------------------------------------------------------------------------------------
// host-side launchers (C++ linkage, used by reze_deform.cpp)
// ------------------------------------------------------------------------------------------------
hipError_t rz_launch_prep(const RzPrepParams &p, uint32_t instances, hipStream_t st)
{
    hipLaunchKernelGGL(rz_prep_kernel, dim3(instances), dim3(kBlock), 0, st, p);
    return hipGetLastError();
}

size_t rz_fk_lds_bytes(const RzFkParams &p)
{
    size_t lds = (size_t)p.B * 48 + rz_fk_scratch_bytes(p.B);
    if (p.bm_off) lds += (size_t)std::max(std::max(p.bm_M, p.sample.M), 1) * 4;      // the pose's morph weights, for the bone morphs
    return lds;
}

hipError_t rz_launch_fk(const RzFkParams &p, uint32_t instances, hipStream_t st)
{
    const size_t lds = rz_fk_lds_bytes(p);
    if (lds > 160 * 1024) return hipErrorInvalidValue;      // (the host checks first and says why: launch_fk in frame.cpp)
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(rz_fk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(rz_fk_kernel, dim3(instances), dim3(kBlock), lds, st, p.bone_rec, (uint32_t)p.B, (uint32_t)p.sample.M, p);
    return hipGetLastError();
}


#ifdef RZ_ALL_VARIANTS
hipError_t rz_launch_gate(const uint32_t *flag, hipStream_t st)
{
    hipLaunchKernelGGL(rz_gate_kernel, dim3(1), dim3(1), 0, st, flag);
    return hipGetLastError();
}
#endif

hipError_t rz_launch_deinterleave(const float *src, int stride, int offset, uint32_t n, float *px, float *py,
                                  float *pz, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(rz_deinterleave_kernel, dim3((n + 255) / 256), dim3(256), 0, st, src, stride, offset, n,
                       px, py, pz);
    return hipGetLastError();
}

hipError_t rz_launch_pack_skinning(const uint16_t *joints4, const uint8_t *weights4, uint32_t n, uint32_t *j01,
                                   uint32_t *j23, uint32_t *wq, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(rz_pack_skinning_kernel, dim3((n + 255) / 256), dim3(256), 0, st, joints4, weights4, n,
                       j01, j23, wq);
    return hipGetLastError();
}

size_t rz_deform_lds_bytes(const RzDeformParams &p, const RzVariant &v)
{
    const size_t vw = 256 / v.S;   // vertices per wave per tile
    size_t scratch = (size_t)(kBlock / 64) * (v.geo ? 9 : 3) * vw * 4;
    const size_t list = (v.mode == 2 || (!v.fast && v.mode == 1)) ? (size_t)p.Mpad * 8 : 0;
    size_t work = scratch + (size_t)(kBlock / 64) * p.out_cap * 24 + (v.mode == 2 ? (size_t)(kBlock / 64) * p.sp_cap * 16 : 0);     // sparse: staged CSR pieces
    if (p.fk_on) work = std::max(work, rz_fk_scratch_bytes(p.B) + (size_t)std::max(p.M, 1) * 4 + 16);   // the fused solve's scratch aliases it
    return (size_t)p.B * 48 + list + work;
}

uint32_t rz_quads_per_tile(int S) { return (kBlock / 64) * (64 / S); }

// One mesh, one launch: the dense morph stream (deform_dense.hip) or the latency-bound frame without one (deform_small.hip)
hipError_t rz_launch_deform(const RzDeformParams &p, const RzMorphList &ml, const RzVariant &v, uint32_t grid_x,
                            uint32_t instances, hipStream_t st)
{
    const size_t lds = rz_deform_lds_bytes(p, v);
    dim3 grid(grid_x, instances);
    if (v.mode == 1) return rz_launch_deform_dense(p, ml, v, grid, lds, st);
    return rz_launch_deform_small(p, v, grid, lds, st);
}
