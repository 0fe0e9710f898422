// Output-stream experiment for the instanced (C4) kernel: 256 instances x 30 000 vertices x (position, normal) x 12 B
// = 184 MB written per frame. Which ORDER of the same stores is fastest?  hipcc --offload-arch=gfx950 -O3 storebench.hip
//   A  vertex-major, 8 poses per workgroup (the kernel's order): for v (stride 256): for g < G: st3 pos, st3 nrm
//   B  as A but four vertices per lane per pose step (3 KB contiguous per wave per stream)
//   C  pose-major inside the workgroup: for g < G: for v
//   D  one pose per workgroup (G = 1), a plain linear run
//   E  one linear 184 MB fill3
// `lds` bytes of dynamic LDS are requested to hold the residency at the real kernel's 2 workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__device__ __forceinline__ void st3(float *d, float a) { d[0] = a; d[1] = a; d[2] = a; }
template <int MODE> __global__ void __launch_bounds__(256) k(float *pos, float *nrm, int V, int Vp, int G, int per)
{
    extern __shared__ float smem[];
    if (threadIdx.x == 0) smem[0] = 0.f;
    const int inst0 = blockIdx.y * G, v0 = blockIdx.x * per, v1 = min(V, v0 + per), tid = threadIdx.x;
    const size_t S = (size_t)Vp * 3;
    if (MODE == 0) {
        for (int v = v0 + tid; v < v1; v += 256)
            for (int g = 0; g < G; ++g) { st3(pos + (inst0 + g) * S + (size_t)v * 3, 1.f); st3(nrm + (inst0 + g) * S + (size_t)v * 3, 2.f); }
    } else if (MODE == 1) {
        const int lane = tid & 63, wave = tid >> 6;
        for (int vb = v0 + wave * 256; vb < v1; vb += 1024)
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const int v = vb + k4 * 64 + lane;
                    if (v < v1) { st3(pos + (inst0 + g) * S + (size_t)v * 3, 1.f); st3(nrm + (inst0 + g) * S + (size_t)v * 3, 2.f); }
                }
    } else if (MODE == 3) {
        // vertex-major like A, but a wave writes its 64 vertices x 12 B = 768 B per array as 48 lanes x 16 B (hipMemset's shape)
        const int lane = tid & 63;
        for (int vb = v0 + (tid & ~63); vb < v1; vb += 256)
            for (int g = 0; g < G; ++g)
                if (lane < 48 && vb + 64 <= v1 + 63) {
                    float4 *dp = reinterpret_cast<float4 *>(pos + (inst0 + g) * S + (size_t)vb * 3) + lane;
                    float4 *dn = reinterpret_cast<float4 *>(nrm + (inst0 + g) * S + (size_t)vb * 3) + lane;
                    *dp = make_float4(1.f, 1.f, 1.f, 1.f); *dn = make_float4(2.f, 2.f, 2.f, 2.f);
                }
    } else {
        for (int g = 0; g < G; ++g)
            for (int v = v0 + tid; v < v1; v += 256) { st3(pos + (inst0 + g) * S + (size_t)v * 3, 1.f); st3(nrm + (inst0 + g) * S + (size_t)v * 3, 2.f); }
    }
}
__global__ void __launch_bounds__(256) k_lin(float *dst, size_t nvert)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvert; i += (size_t)gridDim.x * 256) st3(dst + i * 3, 1.f);
}
template <class F> double timeit(F f)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) f();
    hipDeviceSynchronize();
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        for (int i = 0; i < 50; ++i) f();
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms / 50 < best) best = ms / 50;
    }
    return best * 1e3;
}
int main()
{
    const int V = 30000, I = 256;
    float *pos, *nrm;
    const size_t bytes = (size_t)I * 32768 * 3 * 4;
    // instance stride (vertices): 30720 puts every pose's line for the same vertex on the same L2 channel (368 640 B = 1440 x 256 B,
    // 1440 % 16 == 0); the others rotate the channel from pose to pose
    for (int Vp : {30720, 30784, 30848, 30976, 31744}) {
    printf("---- instance stride %d vertices = %d B (%d x 256 B, mod 16 = %d)\n", Vp, Vp * 12, Vp * 12 / 256, (Vp * 12 / 256) % 16);
    CK(hipMalloc(&pos, bytes)); CK(hipMalloc(&nrm, bytes));
    CK(hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CK(hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CK(hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CK(hipFuncSetAttribute((const void *)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    const double mb = 2.0 * I * V * 12 / 1e6;
    for (int lds : {0, 77 * 1024})
        for (int G : {8})
            for (int runs : {8, 16}) {
                const int per = ((V + runs - 1) / runs + 63) / 64 * 64;
                dim3 grid((V + per - 1) / per, I / G);
                double a = timeit([&] { k<0><<<grid, 256, lds>>>(pos, nrm, V, Vp, G, per); });
                double b = timeit([&] { k<1><<<grid, 256, lds>>>(pos, nrm, V, Vp, G, per); });
                double c = timeit([&] { k<2><<<grid, 256, lds>>>(pos, nrm, V, Vp, G, per); });
                double f = timeit([&] { k<3><<<grid, 256, lds>>>(pos, nrm, V, Vp, G, per); });
                printf("lds=%dK G=%d runs=%d wgs=%d : A %.1f us (%.0f GB/s)  B %.1f us  C %.1f us  F(48 lanes x 16 B) %.1f us (%.0f GB/s)\n", lds >> 10, G, runs, grid.x * grid.y, a, mb / a * 1e3, b, c, f, mb / f * 1e3);
            }
    }
    const int Vp = 30720;
    for (int grid : {512, 1024, 4096, 8192}) {
        double e = timeit([&] { k_lin<<<grid, 256>>>(pos, (size_t)I * Vp); k_lin<<<grid, 256>>>(nrm, (size_t)I * Vp); });
        printf("linear fill3 x2 grid=%d : %.1f us (%.0f GB/s)\n", grid, e, 2.0 * I * Vp * 12 / 1e6 / e * 1e3);
    }
    return 0;
}
