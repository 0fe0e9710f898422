// pullbench — how should a crowd's per-frame pose (0.8 - 3.3 MB) reach the GPU?   hipcc --offload-arch=gfx950 -O3 tools/pullbench.hip -o tools/pullbench
//  (a) hipMemcpyAsync from pinned memory, back to back on one stream (what rz_set_pose's copy path pays per upload)
//  (b) a PULL kernel: workgroups read the pinned, device-mapped buffer over the host link with 16-byte loads and store to HBM
//  (c) the same pull done by HELPER workgroups appended to a store-bound kernel (184 MB of fill = the C4 skin kernel's output stream):
//      does the frame get longer, and by how much?
//  (d) the shader clock small back-to-back kernels actually run at (s_memtime cycles per s_memrealtime tick)
//  (e) round 6: the pull on ITS OWN stream while a store-bound kernel runs back to back on another (what a crowd's per-frame loop looks
//      like once the host no longer packs): per-pull time by workgroups x loads in flight x stream priority, against the copy engine
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int U>
__device__ __forceinline__ void pull_range(const float4 *src, float4 *dst, size_t n4, size_t first, size_t stride)
{
    const size_t last = n4 - 1;
    for (size_t i = first; i < n4; i += stride * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[min(i + u * stride, last)];
#pragma unroll
        for (int u = 0; u < U; ++u) dst[min(i + u * stride, last)] = v[u];
    }
}

template <int U>
__global__ void __launch_bounds__(512) pull_kernel(const float4 *src, float4 *dst, size_t n4)
{
    pull_range<U>(src, dst, n4, (size_t)blockIdx.x * 512 + threadIdx.x, (size_t)gridDim.x * 512);
}

// fill workgroups 0 .. n_fill-1 write `fill_n4` float4 (12-byte-per-lane stores would be closer to the skin kernel; 16-byte is the
// friendlier case), helper workgroups n_fill .. pull. One launch.
template <int U>
__global__ void __launch_bounds__(512) fill_pull_kernel(float4 *fill, size_t fill_n4, unsigned n_fill, const float4 *src, float4 *dst, size_t n4)
{
    if (blockIdx.x >= n_fill) {
        const unsigned h = blockIdx.x - n_fill, nh = gridDim.x - n_fill;
        pull_range<U>(src, dst, n4, (size_t)h * 512 + threadIdx.x, (size_t)nh * 512);
        return;
    }
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)blockIdx.x);
    const size_t per = (fill_n4 + n_fill - 1) / n_fill, b = (size_t)blockIdx.x * per, e = min(fill_n4, b + per);
    for (size_t i = b + threadIdx.x; i < e; i += 512) fill[i] = v;
}

// a kernel that touches no memory at all and lasts `ticks` x 10 ns
__global__ void __launch_bounds__(512) spin_kernel(unsigned long long ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

__global__ void clock_kernel(unsigned long long *out, int spin)
{
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = (a == 1234.5f); }
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t fill_bytes = 184320000;        // C4's output
    float4 *fill;
    CK(hipMalloc(&fill, fill_bytes));
    auto timed = [&](auto &&fn, int n) -> float {
        for (int w = 0; w < 20; ++w) fn();
        hipStreamSynchronize(st);
        std::vector<float> r;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, st);
            for (int i = 0; i < n; ++i) fn();
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            r.push_back(ms / n * 1e3f);
        }
        std::sort(r.begin(), r.end());
        return r[2];
    };
    // warm the clocks
    for (int i = 0; i < 3000; ++i) hipLaunchKernelGGL(fill_pull_kernel<8>, dim3(256), dim3(512), 0, st, fill, fill_bytes / 16, 256u, (const float4 *)nullptr, (float4 *)nullptr, (size_t)0);
    CK(hipStreamSynchronize(st));
    const float t_fill = timed([&] { hipLaunchKernelGGL(fill_pull_kernel<8>, dim3(256), dim3(512), 0, st, fill, fill_bytes / 16, 256u, (const float4 *)nullptr, (float4 *)nullptr, (size_t)0); }, 200);
    printf("fill alone (184 MB, 256 workgroups x 512 threads, 16-byte stores): %.2f us\n", t_fill);
    for (size_t bytes : {(size_t)819200, (size_t)2457600, (size_t)3276800}) {
        void *h = nullptr, *hd = nullptr;
        float4 *d;
        CK(hipHostMalloc(&h, bytes, hipHostMallocMapped));
        CK(hipHostGetDevicePointer(&hd, h, 0));
        CK(hipMalloc(&d, bytes));
        memset(h, 1, bytes);
        const float t_copy = timed([&] { hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st); }, 200);
        printf("%.2f MB: hipMemcpyAsync back to back %.2f us (%.1f GB/s)\n", bytes / 1e6, t_copy, bytes / t_copy / 1e3);
        for (int g : {4, 8, 16, 32, 64, 128, 256}) {
            const float t4 = timed([&] { hipLaunchKernelGGL(pull_kernel<4>, dim3(g), dim3(512), 0, st, (const float4 *)hd, d, bytes / 16); }, 200);
            const float t8 = timed([&] { hipLaunchKernelGGL(pull_kernel<8>, dim3(g), dim3(512), 0, st, (const float4 *)hd, d, bytes / 16); }, 200);
            const float t16 = timed([&] { hipLaunchKernelGGL(pull_kernel<16>, dim3(g), dim3(512), 0, st, (const float4 *)hd, d, bytes / 16); }, 200);
            printf("%.2f MB: pull kernel %3d workgroups: U=4 %.2f us (%.1f GB/s) | U=8 %.2f us (%.1f GB/s) | U=16 %.2f us (%.1f GB/s)\n", bytes / 1e6, g, t4, bytes / t4 / 1e3, t8, bytes / t8 / 1e3, t16, bytes / t16 / 1e3);
        }
        for (int nh : {8, 16, 32, 64}) {
            const float t = timed([&] { hipLaunchKernelGGL(fill_pull_kernel<8>, dim3(256 + nh), dim3(512), 0, st, fill, fill_bytes / 16, 256u, (const float4 *)hd, d, bytes / 16); }, 200);
            printf("%.2f MB: fill + %2d helper workgroups pulling in the same launch: %.2f us (fill alone %.2f)\n", bytes / 1e6, nh, t, t_fill);
        }
        hipFree(d); hipHostFree(h);
    }
    // (e) the pull under a concurrent store-bound stream
    {
        const size_t bytes = 2457600;
        void *h = nullptr, *hd = nullptr;
        float4 *d;
        CK(hipHostMalloc(&h, bytes, hipHostMallocMapped));
        CK(hipHostGetDevicePointer(&hd, h, 0));
        CK(hipMalloc(&d, bytes));
        memset(h, 1, bytes);
        int lo = 0, hi = 0;
        CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        hipStream_t sp[2];
        CK(hipStreamCreateWithPriority(&sp[0], hipStreamNonBlocking, lo));      // (lo = least urgent, hi = most urgent, numerically lower)
        CK(hipStreamCreateWithPriority(&sp[1], hipStreamNonBlocking, hi));
        printf("(e) stream priorities: least %d, greatest %d\n", lo, hi);
        // per pull: n pulls on `ps` between two events, while `fills` fill kernels (27 us each) run on st; the fills outlast the pulls
        auto under_fill = [&](hipStream_t ps, auto &&pull, int n) -> float {
            std::vector<float> r;
            for (int rep = 0; rep < 5; ++rep) {
                hipStreamSynchronize(st); hipStreamSynchronize(ps);
                for (int i = 0; i < 6 * n; ++i) hipLaunchKernelGGL(fill_pull_kernel<8>, dim3(256), dim3(512), 0, st, fill, fill_bytes / 16, 256u, (const float4 *)nullptr, (float4 *)nullptr, (size_t)0);
                hipEventRecord(e0, ps);
                for (int i = 0; i < n; ++i) pull(ps);
                hipEventRecord(e1, ps);
                hipEventSynchronize(e1);
                float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                r.push_back(ms / n * 1e3f);
            }
            hipStreamSynchronize(st);
            std::sort(r.begin(), r.end());
            return r[2];
        };
        for (int pr = 0; pr < 2; ++pr) {
            printf("(e) 2.46 MB per pull on a %s-priority stream, fills running back to back on another stream:\n", pr ? "GREATEST" : "least");
            const float tc = under_fill(sp[pr], [&](hipStream_t s_) { hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s_); }, 40);
            printf("    hipMemcpyAsync %.2f us (%.1f GB/s)\n", tc, bytes / tc / 1e3);
            for (int g : {16, 32, 64, 128}) {
                const float t4 = under_fill(sp[pr], [&](hipStream_t s_) { hipLaunchKernelGGL(pull_kernel<4>, dim3(g), dim3(512), 0, s_, (const float4 *)hd, d, bytes / 16); }, 40);
                const float t8 = under_fill(sp[pr], [&](hipStream_t s_) { hipLaunchKernelGGL(pull_kernel<8>, dim3(g), dim3(512), 0, s_, (const float4 *)hd, d, bytes / 16); }, 40);
                const float t16 = under_fill(sp[pr], [&](hipStream_t s_) { hipLaunchKernelGGL(pull_kernel<16>, dim3(g), dim3(512), 0, s_, (const float4 *)hd, d, bytes / 16); }, 40);
                printf("    pull kernel %3d workgroups: U=4 %.2f us (%.1f GB/s) | U=8 %.2f us (%.1f GB/s) | U=16 %.2f us (%.1f GB/s)\n", g, t4, bytes / t4 / 1e3, t8, bytes / t8 / 1e3, t16, bytes / t16 / 1e3);
            }
        }
        // ... and what the fills pay for it: 40 fills with pulls (or copies) running alongside on the other stream
        auto fill_under = [&](const char *what, auto &&pull) {
            hipStreamSynchronize(st); hipStreamSynchronize(sp[0]);
            for (int i = 0; i < 80; ++i) pull(sp[0]);
            hipEventRecord(e0, st);
            for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(fill_pull_kernel<8>, dim3(256), dim3(512), 0, st, fill, fill_bytes / 16, 256u, (const float4 *)nullptr, (float4 *)nullptr, (size_t)0);
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const bool still = hipStreamQuery(sp[0]) == hipErrorNotReady;
            hipStreamSynchronize(sp[0]);
            printf("(e) fill while %s: %.2f us per fill (alone %.2f)%s\n", what, ms / 40 * 1e3f, t_fill, still ? "" : "  [the other stream ran dry before the fills ended]");
        };
        fill_under("the copy engine copies 2.46 MB back to back", [&](hipStream_t s_) { hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s_); });
        fill_under("1 workgroup pulls (U=4)", [&](hipStream_t s_) { hipLaunchKernelGGL(pull_kernel<4>, dim3(1), dim3(512), 0, s_, (const float4 *)hd, d, bytes / 16); });
        fill_under("2 workgroups pull (U=4)", [&](hipStream_t s_) { hipLaunchKernelGGL(pull_kernel<4>, dim3(2), dim3(512), 0, s_, (const float4 *)hd, d, bytes / 16); });
        fill_under("4 workgroups pull (U=4)", [&](hipStream_t s_) { hipLaunchKernelGGL(pull_kernel<4>, dim3(4), dim3(512), 0, s_, (const float4 *)hd, d, bytes / 16); });
        fill_under("8 workgroups pull (U=4)", [&](hipStream_t s_) { hipLaunchKernelGGL(pull_kernel<4>, dim3(8), dim3(512), 0, s_, (const float4 *)hd, d, bytes / 16); });
        fill_under("16 workgroups pull (U=4)", [&](hipStream_t s_) { hipLaunchKernelGGL(pull_kernel<4>, dim3(16), dim3(512), 0, s_, (const float4 *)hd, d, bytes / 16); });
        fill_under("16 workgroups pull (U=1)", [&](hipStream_t s_) { hipLaunchKernelGGL(pull_kernel<1>, dim3(16), dim3(512), 0, s_, (const float4 *)hd, d, bytes / 16); });
        fill_under("64 workgroups pull (U=1)", [&](hipStream_t s_) { hipLaunchKernelGGL(pull_kernel<1>, dim3(64), dim3(512), 0, s_, (const float4 *)hd, d, bytes / 16); });
        // a kernel that does nothing but last 50 / 200 us: is it memory at all, or do two queues complete in lock-step?
        fill_under("16 workgroups spin for 50 us, touching no memory", [&](hipStream_t s_) { hipLaunchKernelGGL(spin_kernel, dim3(16), dim3(512), 0, s_, 5000ull); });
        fill_under("16 workgroups spin for 200 us, touching no memory", [&](hipStream_t s_) { hipLaunchKernelGGL(spin_kernel, dim3(16), dim3(512), 0, s_, 20000ull); });
        {
            // the same on streams created later (other hardware queues / pipes?)
            hipStream_t more[6];
            for (auto &m : more) CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking));
            for (int k = 0; k < 6; ++k) {
                hipStream_t keep = sp[0];
                sp[0] = more[k];
                char what[96];
                snprintf(what, sizeof what, "16 workgroups pull (U=4) on extra stream #%d", k);
                fill_under(what, [&](hipStream_t s_) { hipLaunchKernelGGL(pull_kernel<4>, dim3(16), dim3(512), 0, s_, (const float4 *)hd, d, bytes / 16); });
                sp[0] = keep;
            }
            for (auto &m : more) hipStreamDestroy(m);
        }
        {
            // ONE long pull (24.6 MB out of a larger pinned buffer, ~500 us) against 27 us fills: do fills complete DURING it?
            const size_t big = bytes * 10;
            void *hb = nullptr, *hbd = nullptr;
            float4 *db;
            CK(hipHostMalloc(&hb, big, hipHostMallocMapped));
            CK(hipHostGetDevicePointer(&hbd, hb, 0));
            CK(hipMalloc(&db, big));
            memset(hb, 1, big);
            fill_under("ONE pull kernel of 24.6 MB per launch runs (16 workgroups, U=4)", [&](hipStream_t s_) { hipLaunchKernelGGL(pull_kernel<4>, dim3(16), dim3(512), 0, s_, (const float4 *)hbd, db, big / 16); });
            // ... and the other way round: many SHORT pulls (154 KB each, 16 per 2.46 MB)
            fill_under("short pulls of 154 KB each run back to back (16 workgroups, U=4)", [&](hipStream_t s_) {
                for (int k = 0; k < 16; ++k) hipLaunchKernelGGL(pull_kernel<4>, dim3(16), dim3(512), 0, s_, (const float4 *)hbd + (size_t)k * (bytes / 256), db + (size_t)k * (bytes / 256), bytes / 256); });
            // how many launches should ONE pose's pull be cut into? per pose: the pull under fills, and the fill under pulls
            for (int chunks : {1, 2, 3, 4, 6, 8, 16}) {
                const size_t per4 = (bytes / 16 + chunks - 1) / chunks;
                auto pose = [&](hipStream_t s_) {
                    for (int k = 0; k < chunks; ++k) {
                        const size_t b4 = (size_t)k * per4, n4 = std::min(per4, bytes / 16 - b4);
                        hipLaunchKernelGGL(pull_kernel<4>, dim3(16), dim3(512), 0, s_, (const float4 *)hbd + b4, db + b4, n4);
                    }
                };
                const float ta = timed([&] { pose(st); }, 100);
                const float tp = under_fill(sp[0], pose, 40);
                char what[96];
                snprintf(what, sizeof what, "one pose = %d pull launches (alone %.2f us, under fills %.2f us per pose)", chunks, ta, tp);
                fill_under(what, pose);
            }
            hipFree(db); hipHostFree(hb);
        }
        // a pull out of DEVICE memory of the same size: is it the host link, or just a second kernel alongside?
        float4 *d2;
        CK(hipMalloc(&d2, bytes));
        fill_under("16 workgroups copy 2.46 MB device -> device (U=4)", [&](hipStream_t s_) { hipLaunchKernelGGL(pull_kernel<4>, dim3(16), dim3(512), 0, s_, (const float4 *)d2, d, bytes / 16); });
        hipFree(d2);
        hipFree(d); hipHostFree(h);
    }
    // (d) clock of small kernels launched back to back
    unsigned long long *co, hc[3];
    CK(hipMalloc(&co, 64));
    for (int spin : {200, 2000, 20000}) {
        for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(clock_kernel, dim3(236), dim3(256), 0, st, co, spin);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(hc, co, 24, hipMemcpyDeviceToHost));
        printf("clock: %d-iteration kernels back to back: %llu s_memtime ticks in %llu x 10 ns -> %.0f MHz if s_memtime counts shader cycles\n", spin, hc[0], hc[1], hc[1] ? hc[0] / (hc[1] * 0.01) : 0.0);
    }
    return 0;
}
