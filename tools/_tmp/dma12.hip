#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#pragma clang diagnostic ignored "-Wint-to-void-pointer-cast"
__global__ void k(const float4 *src, float *out)
{
    __shared__ __attribute__((aligned(16))) float lds[64 * 4 * 2];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = -1.0f;
    __syncthreads();
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)(src + threadIdx.x), (lptr_t)(uint32_t)(uintptr_t)(lds), 12, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main()
{
    float4 *src; float *out;
    hipMalloc(&src, 64 * 16); hipMalloc(&out, 512 * 4);
    float h[256]; for (int i = 0; i < 256; ++i) h[i] = (float)((i / 4) * 10 + (i % 4));   // lane*10 + component
    hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out);
    float o[512]; hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost);
    for (int i = 0; i < 256; ++i) { printf("%g ", o[i]); if (i % 16 == 15) printf("\n"); }
    return 0;
}
