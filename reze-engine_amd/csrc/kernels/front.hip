// kernels/front.hip — what a frame may launch in FRONT of its deform / skin kernel (rz_prep_kernel: palette + active-morph list;
// rz_fk_kernel: hierarchy solve + motion sampling; rz_pull_pose_kernel: a crowd's pose over the host link), the one-off upload re-layout kernels, and the host-side launch dispatch.
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


// ------------------------------------------------------------------------------------------------
// rz_pull_pose_kernel: a crowd's per-frame pose (0.8 - 3.3 MB for 256 characters) comes down by a PULL. The host lays the pose out in
// a pinned, device-mapped ring slot; 16 workgroups read it over the host link with 16-byte loads (1 KiB contiguous per wave
// instruction, three in flight per lane) and store it into the device pose block. Measured against hipMemcpyAsync from the same
// pinned memory, back to back on one stream (tools/pullbench, profiles/r5_pullbench.txt): 3.28 MB 84 -> 62 us, 0.82 MB 24 -> 18 us —
// the copy engine path costs ~24 us per upload whatever the size; more workgroups or more loads in flight per lane pull SLOWER.
// World matrices travel as their upper three rows (the bottom row of an affine matrix is 0 0 0 1; the host checks that while it
// packs): 48 B per bone as the four columns' x y z, i.e. exactly three float4. A wave takes 64 bones per step — three coalesced
// loads, an exchange through its own 3 KB of LDS, then every lane writes its bone's 4 x 4 matrix (engine.ts:2383-2389 uploads all
// sixteen floats; every reader of the world matrices keeps the reference's layout).
// ------------------------------------------------------------------------------------------------
constexpr int kPullBlock = 512, kPullGrid = 16;

__global__ void __launch_bounds__(kPullBlock) rz_pull_pose_kernel(const float4 *src, float4 *dst, const uint32_t bones, const uint32_t raw4,
                                                                  const uint32_t raw_words)
{
    __shared__ float4 sh[kPullBlock / 64][192];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (bones) {
        const uint32_t gw = blockIdx.x * (kPullBlock / 64) + wave, nw = gridDim.x * (kPullBlock / 64), last = bones * 3u - 1u;
        float4 *mine = sh[wave];
        for (uint32_t b0 = gw * 64u; b0 < bones; b0 += nw * 64u) {
            const uint32_t base = b0 * 3u + lane;
            const float4 v0 = src[min(base, last)], v1 = src[min(base + 64u, last)], v2 = src[min(base + 128u, last)];
            mine[lane] = v0; mine[64u + lane] = v1; mine[128u + lane] = v2;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const float4 a = mine[3u * lane], b = mine[3u * lane + 1u], c = mine[3u * lane + 2u];
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();            // every lane has read its bone before the next step's columns land
            if (b0 + lane < bones) {
                float4 *d = dst + (size_t)(b0 + lane) * 4;
                d[0] = make_float4(a.x, a.y, a.z, 0.0f);
                d[1] = make_float4(a.w, b.x, b.y, 0.0f);
                d[2] = make_float4(b.z, b.w, c.x, 0.0f);
                d[3] = make_float4(c.y, c.z, c.w, 1.0f);
            }
        }
    }
    // what travels as it is: morph weights behind the matrices; local rotations / translations
    const float4 *rs = src + (size_t)bones * 3;
    float4 *rd = dst + (size_t)bones * 4;
    if (raw4) {
        const uint32_t stride = gridDim.x * kPullBlock, last = raw4 - 1u;
        for (uint32_t i = blockIdx.x * kPullBlock + threadIdx.x; i < raw4; i += stride * 4u) {
            const uint32_t i1 = min(i + stride, last), i2 = min(i + 2u * stride, last), i3 = min(i + 3u * stride, last);
            const float4 v0 = rs[i], v1 = rs[i1], v2 = rs[i2], v3 = rs[i3];
            rd[i] = v0; rd[i1] = v1; rd[i2] = v2; rd[i3] = v3;        // (beyond the end: the last element again, the same value)
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < raw_words)
        reinterpret_cast<float *>(rd + raw4)[threadIdx.x] = reinterpret_cast<const float *>(rs + raw4)[threadIdx.x];
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


// `src` = device address of the pinned slot: [bones x 48 B of packed matrix columns | raw_bytes carried as they are]; `dst` = the
// device pose block, where the matrices arrive as 4 x 4 and the rest behind them. raw_bytes is a multiple of 4.
hipError_t rz_launch_pull_pose(const void *src, void *dst, uint32_t bones, size_t raw_bytes, hipStream_t st)
{
    if ((raw_bytes & 3u) || raw_bytes / 16 > 0xffffffffull || (uint64_t)bones * 3u > 0xffffffffull) return hipErrorInvalidValue;
    if (bones == 0 && raw_bytes == 0) return hipSuccess;
    hipLaunchKernelGGL(rz_pull_pose_kernel, dim3(kPullGrid), dim3(kPullBlock), 0, st, static_cast<const float4 *>(src), static_cast<float4 *>(dst),
                       bones, (uint32_t)(raw_bytes / 16), (uint32_t)((raw_bytes & 15u) / 4));
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
