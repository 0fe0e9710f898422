// Does hipExtLaunchKernel(..., hipExtAnyOrderLaunch) clear the AQL barrier bit on gfx950, i.e. can a small kernel launched
// right behind a long one IN THE SAME STREAM start while the long one is still running? (hip_ext.h carries an old note
// saying the flag is not supported on GFX9xx.)   hipcc --offload-arch=gfx950 -O3 tools/archive/anyorder.hip -o tools/archive/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_long(unsigned long long *t, int spin)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) t[0] = wall_clock64();
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    if (a == 1234.5f) t[3] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) t[1] = wall_clock64();
}
__global__ void k_small(unsigned long long *t, int slot)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) t[slot] = wall_clock64();
}
int main()
{
    unsigned long long *t, h[8];
    CK(hipMalloc(&t, 64));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(t, 0, 64));
            CK(hipDeviceSynchronize());
            int spin = 20000, slot = 2;
            void *a1[] = {&t, &spin}, *a2[] = {&t, &slot};
            CK(hipExtLaunchKernel((const void *)k_long, dim3(256), dim3(256), a1, 0, s, nullptr, nullptr, 0));
            CK(hipExtLaunchKernel((const void *)k_small, dim3(64), dim3(256), a2, 0, s, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0));
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(h, t, 64, hipMemcpyDeviceToHost));
            const double tick_ns = 10.0;     // wall_clock64 runs at 100 MHz
            printf("%s: long kernel %.1f us; small kernel started %.1f us after the long one STARTED, %.1f us relative to its END -> %s\n",
                   mode ? "any-order" : "ordered  ", (h[1] - h[0]) * tick_ns / 1e3, ((double)h[2] - (double)h[0]) * tick_ns / 1e3,
                   ((double)h[2] - (double)h[1]) * tick_ns / 1e3, h[2] < h[1] ? "OVERLAPPED" : "serialized");
        }
    }
    // throughput: N pairs (small, long) back to back, ordered vs the small one any-order
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        int spin = 3000, slot = 2;
        void *a1[] = {&t, &spin}, *a2[] = {&t, &slot};
        for (int w = 0; w < 2; ++w) {
            hipEventRecord(e0, s);
            for (int i = 0; i < 200; ++i) {
                hipExtLaunchKernel((const void *)k_small, dim3(256), dim3(256), a2, 0, s, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0);
                hipExtLaunchKernel((const void *)k_long, dim3(512), dim3(256), a1, 0, s, nullptr, nullptr, 0);
            }
            hipEventRecord(e1, s); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s pairs: %.2f us per (small + long) pair\n", mode ? "any-order" : "ordered  ", ms / 200 * 1e3);
    }
    return 0;
}
