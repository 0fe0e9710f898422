// How fast can ONE wave bring a 20 KB piece of an L2-resident array into LDS?  (round 4: staging a step's sparse-morph rows)
//   hipcc --offload-arch=gfx950 -O3 tools/archive/dmabench.hip -o tools/archive/dmabench
// forms: 0 = LDS-DMA, 16 B per lane, M0 rewritten for every 1 KiB burst          (what rz_deform_kernel MODE 2 did first)
//        1 = LDS-DMA, M0 rewritten once per 4 bursts, instruction offsets in between
//        2 = plain 16-byte loads into registers, 10 in flight, then ds_write_b128
//        3 = LDS-DMA, 4 B per lane (four times the instructions)
// Each wave reports, from s_memtime: cycles until its last request was ISSUED, and until the data was usable.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <int FORM> __global__ void __launch_bounds__(256) k_stage(const float4 *src, int bursts, unsigned long long *out, float *sink)
{
    extern __shared__ float4 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 *buf = lds + (size_t)wave * bursts * 64;
    const float4 *g = src + ((size_t)blockIdx.x * 4 + wave) * bursts * 64;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (FORM == 0) {
        for (int i = 0; i < bursts; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)(g + i * 64 + lane), (lptr_t)(uint32_t)(uintptr_t)(buf + i * 64), 16, 0, 0);
    } else if (FORM == 1) {
        for (int i = 0; i + 4 <= bursts; i += 4) {
            __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)(g + i * 64 + lane), (lptr_t)(uint32_t)(uintptr_t)(buf + i * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)(g + i * 64 + lane), (lptr_t)(uint32_t)(uintptr_t)(buf + i * 64), 16, 1024, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)(g + i * 64 + lane), (lptr_t)(uint32_t)(uintptr_t)(buf + i * 64), 16, 2048, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)(g + i * 64 + lane), (lptr_t)(uint32_t)(uintptr_t)(buf + i * 64), 16, 3072, 0);
        }
    } else if (FORM == 2) {
        for (int i0 = 0; i0 < bursts; i0 += 10) {
            float4 t[10];
#pragma unroll
            for (int u = 0; u < 10; ++u) t[u] = i0 + u < bursts ? g[(i0 + u) * 64 + lane] : make_float4(0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 10; ++u) if (i0 + u < bursts) buf[(i0 + u) * 64 + lane] = t[u];
        }
    } else {
        const float *gf = reinterpret_cast<const float *>(g);
        float *bf = reinterpret_cast<float *>(buf);
        for (int i = 0; i < bursts * 4; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)(gf + i * 64 + lane), (lptr_t)(uint32_t)(uintptr_t)(bf + i * 64), 4, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const unsigned long long t2 = __builtin_readcyclecounter();
    float acc = 0.f;
    for (int i = 0; i < bursts; ++i) acc += buf[i * 64 + ((lane * 7) & 63)].x;
    if (acc == 1234.5f) sink[threadIdx.x] = acc;
    if (lane == 0) { out[((size_t)blockIdx.x * 4 + wave) * 2] = t1 - t0; out[((size_t)blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0; }
}

int main()
{
    const int bursts = 20, wgs = 64;
    float4 *src; unsigned long long *out; float *sink;
    const size_t n = (size_t)wgs * 4 * bursts * 64;
    CK(hipMalloc(&src, n * 16)); CK(hipMemset(src, 0, n * 16)); CK(hipMalloc(&out, wgs * 4 * 16)); CK(hipMalloc(&sink, 1024));
    std::vector<unsigned long long> h(wgs * 8);
    for (int active = 4; active >= 1; active -= 3)
    for (int form = 0; form < 4; ++form) {
        for (int rep = 0; rep < 3; ++rep) {
            const size_t lds = (size_t)4 * bursts * 64 * 16;
            const dim3 blk(64 * active);
            auto launch = [&](auto k) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k, dim3(wgs), blk, lds, 0, src, bursts, out, sink); };
            if (form == 0) launch(k_stage<0>); else if (form == 1) launch(k_stage<1>); else if (form == 2) launch(k_stage<2>); else launch(k_stage<3>);
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h.data(), out, wgs * 4 * 16, hipMemcpyDeviceToHost));
        double si = 0, sd = 0; int cnt = 0;
        for (int w = 0; w < wgs * 4; ++w) if ((w & 3) < active) { si += h[w * 2]; sd += h[w * 2 + 1]; ++cnt; }
        printf("form %d, %d wave(s) per workgroup: %d x 1 KiB per wave: issued after %.0f cycles, usable after %.0f cycles (s_memtime, 100 MHz ticks x ?: raw counter units; mean over %d waves, third launch)\n", form, active, bursts, si / cnt, sd / cnt, cnt);
    }
    return 0;
}
