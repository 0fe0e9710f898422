// LDS-gather and permuted-store micro-benchmarks behind the instanced (C4) kernel's design.
//   hipcc --offload-arch=gfx950 -O3 tools/archive/ldsbench.hip -o tools/archive/ldsbench
// Q1  Does a ds_read_b128 with only part of the wave active cost fewer LDS cycles? (exec-masked palette gathers for
//     vertices with fewer than four influences.) Same instruction count, different active-lane patterns.
// Q2  What does a ds_bpermute_b32 cost next to it (un-permuting results inside a wave)?
// Q3  Do 12-byte-stride dword stores cost more when the lane -> vertex map inside a wave's 768-byte window is a
//     permutation instead of the identity (184 MB output stream of C4)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// every lane gathers 3 rows (3 x ds_read_b128) of `bones` bone slots per iteration; lanes with (active_mask >> lane) & 1 == 0
// skip the gathers of slots >= 1 (slot 0 is always read), i.e. the mask models "this lane has more than one influence".
template <int SLOTS> __global__ void __launch_bounds__(256) k_gather(const uint32_t *joint_tab, unsigned long long mask, int iters, float *out, int pal_bones)
{
    extern __shared__ float4 pal[];
    for (int i = threadIdx.x; i < pal_bones * 3; i += 256) pal[i] = make_float4(i, 1.f, 2.f, 3.f);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const bool on = (mask >> lane) & 1ull;
    uint32_t j[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) j[s] = joint_tab[(blockIdx.x * 256 + threadIdx.x) * 4 + s] * 3;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};                 // packed adds: 2 VALU ops per 16-byte read, so LDS is the bound
    for (int it = 0; it < iters; ++it) {
        const float4 *pg = pal + (it & 7) * 3 * 8;          // move around a little (8 "poses" x 8-bone offset), stays in range
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float4 a = pg[j[0] + k];
            acc0 += f2{a.x, a.y}; acc1 += f2{a.z, a.w};
        }
        if (on) {
#pragma unroll
            for (int s = 1; s < SLOTS; ++s)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    float4 a = pg[j[s] + k];
                    acc0 += f2{a.x, a.y}; acc1 += f2{a.z, a.w};
                }
        }
    }
    if (acc0.x == 1234.5f) out[threadIdx.x] = acc0.x + acc0.y + acc1.x + acc1.y;
}

__global__ void __launch_bounds__(256) k_bperm(int iters, float *out)
{
    const int lane = threadIdx.x & 63;
    float v = (float)threadIdx.x;
    const int src = ((lane * 37) & 63) << 2;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 6; ++k) v = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v)) + k);
    }
    if (v == 1234.5f) out[threadIdx.x] = v;
}

__device__ __forceinline__ void st3(float *d, float a) { d[0] = a; d[1] = a; d[2] = a; }
// C4's output stream: vertex-major, G poses per workgroup. PERM: lane handles vertex base + perm(lane) of its 64-vertex group.
template <int PERM> __global__ void __launch_bounds__(256) k_store(float *pos, float *nrm, int V, int Vp, int G, int per)
{
    const int inst0 = blockIdx.y * G, v0 = blockIdx.x * per, v1 = min(V, v0 + per), tid = threadIdx.x;
    const int lane = tid & 63;
    const int pl = PERM == 0 ? lane : PERM == 1 ? ((lane * 37 + 11) & 63) : (int)(__brev((unsigned)lane) >> 26);
    const size_t S = (size_t)Vp * 3;
    for (int vb = v0 + (tid & ~63); vb < v1; vb += 256) {
        const int v = vb + pl;
        if (v < v1)
            for (int g = 0; g < G; ++g) { st3(pos + (inst0 + g) * S + (size_t)v * 3, 1.f); st3(nrm + (inst0 + g) * S + (size_t)v * 3, 2.f); }
    }
}

template <class F> double timeit(F f, int reps = 20)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize();
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        for (int i = 0; i < reps; ++i) f();
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms / reps < best) best = ms / reps;
    }
    return best * 1e3;
}

int main()
{
    const int B = 200, WGS = 512, ITERS = 2000;
    std::vector<uint32_t> jt((size_t)WGS * 256 * 4);
    srand(5);
    for (size_t t = 0; t < jt.size() / 4; ++t) {                 // 9-bone window per 64 lanes, like the synthetic mesh
        const uint32_t base = (uint32_t)((t / 64) * 7 % (B - 80));
        for (int s = 0; s < 4; ++s) jt[t * 4 + s] = base + (uint32_t)(rand() % 9);
    }
    uint32_t *djt; float *dout;
    CK(hipMalloc(&djt, jt.size() * 4)); CK(hipMalloc(&dout, 4096));
    CK(hipMemcpy(djt, jt.data(), jt.size() * 4, hipMemcpyHostToDevice));
    const int lds = 77 * 1024;
    CK(hipFuncSetAttribute((const void *)k_gather<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void *)k_gather<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void *)k_gather<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    struct { const char *name; unsigned long long mask; } masks[] = {
        {"all 64 lanes", ~0ull}, {"lanes 0..39 (60 %)", (1ull << 40) - 1}, {"lanes 0..31", (1ull << 32) - 1}, {"lanes 0..15", 0xffffull},
        {"lanes 0..7", 0xffull}, {"lanes 0..4 (8 %)", 0x1full}, {"every 8th lane", 0x0101010101010101ull}, {"every 2nd lane", 0x5555555555555555ull},
        {"none", 0ull}};
    printf("== Q1: 4-slot gather, slots 1..3 exec-masked; %d WGs x 256 thr, %d iterations, 77 KB LDS (2 WG/CU)\n", WGS, ITERS);
    double base1 = timeit([&] { k_gather<1><<<WGS, 256, lds>>>(djt, ~0ull, ITERS, dout, B); });
    printf("   1 slot  (3 ds_read_b128/iter)               : %8.1f us\n", base1);
    for (auto &m : masks) {
        double t = timeit([&] { k_gather<4><<<WGS, 256, lds>>>(djt, m.mask, ITERS, dout, B); });
        printf("   4 slots, slots 1..3 active on %-20s: %8.1f us   (extra over 1 slot: %.1f us)\n", m.name, t, t - base1);
    }
    double t2 = timeit([&] { k_gather<2><<<WGS, 256, lds>>>(djt, ~0ull, ITERS, dout, B); });
    printf("   2 slots all lanes                            : %8.1f us\n", t2);
    printf("== Q2: 6 x ds_bpermute_b32 per iteration\n");
    double tb = timeit([&] { k_bperm<<<WGS, 256>>>(ITERS, dout); });
    printf("   %8.1f us  (vs %.1f us for 3 ds_read_b128 per iteration)\n", tb, base1);
    printf("== Q3: C4 output stream, 184 MB, G = 8, 16 runs: identity vs permuted lane -> vertex inside each 64-vertex group\n");
    const int V = 30000, Vp = 30720, I = 256, G = 8, runs = 16;
    float *pos, *nrm;
    const size_t bytes = (size_t)I * Vp * 3 * 4;
    CK(hipMalloc(&pos, bytes)); CK(hipMalloc(&nrm, bytes));
    const int per = ((V + runs - 1) / runs + 63) / 64 * 64;
    dim3 grid((V + per - 1) / per, I / G);
    const double mb = 2.0 * I * V * 12 / 1e6;
    double s0 = timeit([&] { k_store<0><<<grid, 256>>>(pos, nrm, V, Vp, G, per); }, 50);
    double s1 = timeit([&] { k_store<1><<<grid, 256>>>(pos, nrm, V, Vp, G, per); }, 50);
    double s2 = timeit([&] { k_store<2><<<grid, 256>>>(pos, nrm, V, Vp, G, per); }, 50);
    printf("   identity %.1f us (%.0f GB/s)   (lane*37+11)&63 %.1f us (%.0f GB/s)   bit-reversed %.1f us (%.0f GB/s)\n", s0, mb / s0 * 1e3, s1, mb / s1 * 1e3, s2, mb / s2 * 1e3);
    return 0;
}
