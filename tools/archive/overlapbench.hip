// overlapbench — does a kernel that pulls a crowd's pose over the host link overlap a store-bound frame kernel running on ANOTHER stream?
//   hipcc --offload-arch=gfx950 -O3 tools/archive/overlapbench.hip -o tools/archive/overlapbench
// Frame stand-in: 256 workgroups x 512 threads filling 184 MB (C4's output stream), `fill_lds` bytes of dynamic LDS per workgroup.
// Upload stand-in: the pull kernel of tools/pullbench (16 x 512, 4 loads in flight per lane) or hipMemcpyAsync from the same pinned memory.
//  (a) each alone, back to back on its stream                      (b) both streams free-running, no dependency: N of each, wall time
//  (c) the per-frame protocol of rz_set_pose + rz_deform: upload(u) on the upload stream waits for frame(u-2), frame(u) waits for upload(u)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(512) pull_kernel(const float4 *src, float4 *dst, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * 512, last = n4 - 1;
    for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n4; i += stride * 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = src[min(i + u * stride, last)];
#pragma unroll
        for (int u = 0; u < 4; ++u) dst[min(i + u * stride, last)] = v[u];
    }
}

__global__ void __launch_bounds__(512) fill_kernel(float4 *fill, size_t fill_n4)
{
    extern __shared__ float4 sh[];
    if (fill_n4 == 1) sh[threadIdx.x] = make_float4(0, 0, 0, 0);
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)blockIdx.x);
    const size_t per = (fill_n4 + gridDim.x - 1) / gridDim.x, b = (size_t)blockIdx.x * per, e = min(fill_n4, b + per);
    for (size_t i = b + threadIdx.x; i < e; i += 512) fill[i] = v;
}

int main(int argc, char **argv)
{
    const int N = 400;
    hipStream_t sf, su;
    CK(hipStreamCreateWithFlags(&sf, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&su, hipStreamNonBlocking));
    const size_t fill_bytes = 184320000;
    float4 *fill;
    CK(hipMalloc(&fill, fill_bytes));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us_since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::micro>(now() - t0).count(); };
    for (size_t fill_lds : {(size_t)32768, (size_t)73728}) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fill_lds));
        auto frame = [&](hipStream_t s) { hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(512), fill_lds, s, fill, fill_bytes / 16); };
        for (size_t bytes : {(size_t)819200, (size_t)2457600}) {
            void *h = nullptr, *hd = nullptr;
            float4 *d[2];
            CK(hipHostMalloc(&h, bytes, hipHostMallocMapped));
            CK(hipHostGetDevicePointer(&hd, h, 0));
            CK(hipMalloc(&d[0], bytes)); CK(hipMalloc(&d[1], bytes));
            memset(h, 1, bytes);
            for (int mode = 0; mode < 2; ++mode) {         // 0 = pull kernel, 1 = hipMemcpyAsync
                auto upload = [&](hipStream_t s, int k) {
                    if (mode == 0) hipLaunchKernelGGL(pull_kernel, dim3(16), dim3(512), 0, s, (const float4 *)hd, d[k], bytes / 16);
                    else hipMemcpyAsync(d[k], h, bytes, hipMemcpyHostToDevice, s);
                };
                for (int i = 0; i < 300; ++i) { frame(sf); upload(su, i & 1); }
                CK(hipStreamSynchronize(sf)); CK(hipStreamSynchronize(su));
                auto t0 = now();
                for (int i = 0; i < N; ++i) frame(sf);
                CK(hipStreamSynchronize(sf));
                const double t_frame = us_since(t0) / N;
                t0 = now();
                for (int i = 0; i < N; ++i) upload(su, i & 1);
                CK(hipStreamSynchronize(su));
                const double t_up = us_since(t0) / N;
                t0 = now();
                for (int i = 0; i < N; ++i) { upload(su, i & 1); frame(sf); }
                CK(hipStreamSynchronize(sf)); CK(hipStreamSynchronize(su));
                const double t_free = us_since(t0) / N;
                // the protocol: ev_free[k] (frame stream) = everything that reads device slot k has been enqueued; ev_up = upload landed
                hipEvent_t ev_free[2], ev_up[8];
                for (auto &e : ev_free) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                for (auto &e : ev_up) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                bool rec[2] = {false, false};
                int cur = 0;
                auto step = [&](int i) {
                    const int k = cur ^ 1;
                    hipEventRecord(ev_free[cur], sf); rec[cur] = true;
                    if (!rec[k]) hipEventRecord(ev_free[k], sf);
                    hipStreamWaitEvent(su, ev_free[k], 0);
                    upload(su, k);
                    hipEventRecord(ev_up[i & 7], su);
                    hipStreamWaitEvent(sf, ev_up[i & 7], 0);
                    rec[k] = false; cur = k;
                    frame(sf);
                };
                for (int i = 0; i < 100; ++i) step(i);
                CK(hipStreamSynchronize(sf)); CK(hipStreamSynchronize(su));
                t0 = now();
                for (int i = 0; i < N; ++i) step(i);
                CK(hipStreamSynchronize(sf)); CK(hipStreamSynchronize(su));
                const double t_proto = us_since(t0) / N;
                // the same without the "slot free" edge (three device slots would make it unnecessary): frame(u) waits for upload(u) only
                t0 = now();
                for (int i = 0; i < N; ++i) { upload(su, i & 1); hipEventRecord(ev_up[i & 7], su); hipStreamWaitEvent(sf, ev_up[i & 7], 0); frame(sf); }
                CK(hipStreamSynchronize(sf)); CK(hipStreamSynchronize(su));
                const double t_half = us_since(t0) / N;
                // (d) two contexts (a context and its fork: four streams) taking turns, each with its own upload stream, device blocks
                // and pinned source — frame(u) of a context waits for that context's upload(u) only
                hipStream_t sf2, su2;
                CK(hipStreamCreateWithFlags(&sf2, hipStreamNonBlocking));
                CK(hipStreamCreateWithFlags(&su2, hipStreamNonBlocking));
                void *h2 = nullptr, *hd2 = nullptr;
                float4 *d2;
                CK(hipHostMalloc(&h2, bytes, hipHostMallocMapped));
                CK(hipHostGetDevicePointer(&hd2, h2, 0));
                CK(hipMalloc(&d2, bytes));
                memset(h2, 1, bytes);
                auto pair_step = [&](int i) {
                    const bool b = i & 1;
                    hipStream_t s_u = b ? su2 : su, s_f = b ? sf2 : sf;
                    if (mode == 0) hipLaunchKernelGGL(pull_kernel, dim3(16), dim3(512), 0, s_u, (const float4 *)(b ? hd2 : hd), b ? d2 : d[0], bytes / 16);
                    else hipMemcpyAsync(b ? d2 : d[0], b ? h2 : h, bytes, hipMemcpyHostToDevice, s_u);
                    hipEventRecord(ev_up[i & 7], s_u); hipStreamWaitEvent(s_f, ev_up[i & 7], 0);
                    frame(s_f);
                };
                for (int i = 0; i < 100; ++i) pair_step(i);
                CK(hipStreamSynchronize(sf)); CK(hipStreamSynchronize(su)); CK(hipStreamSynchronize(sf2)); CK(hipStreamSynchronize(su2));
                t0 = now();
                for (int i = 0; i < N; ++i) pair_step(i);
                CK(hipStreamSynchronize(sf)); CK(hipStreamSynchronize(su)); CK(hipStreamSynchronize(sf2)); CK(hipStreamSynchronize(su2));
                const double t_pair = us_since(t0) / N;
                printf("    two contexts taking turns (four streams): %.1f us per frame\n", t_pair);
                hipFree(d2); hipHostFree(h2); hipStreamDestroy(sf2); hipStreamDestroy(su2);
                printf("frame LDS %3zu KB, %.2f MB by %s: frame alone %.1f us | upload alone %.1f us | both free-running %.1f us | rz_set_pose protocol %.1f us | only frame-waits-for-upload %.1f us\n",
                       fill_lds / 1024, bytes / 1e6, mode == 0 ? "pull kernel   " : "hipMemcpyAsync", t_frame, t_up, t_free, t_proto, t_half);
                fflush(stdout);
                for (auto &e : ev_free) hipEventDestroy(e);
                for (auto &e : ev_up) hipEventDestroy(e);
            }
            hipFree(d[0]); hipFree(d[1]); hipHostFree(h);
        }
    }
    return 0;
}
