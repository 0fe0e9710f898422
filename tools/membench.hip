// membench.hip — calibrates the streaming ceilings of THIS MI355X box so the roofline fraction of
// rz_deform_kernel can be read against a measured number, not only the 8 TB/s datasheet peak.
// read-only (sum), write-only (fill) and copy, 16 B/lane, plain vs nontemporal, HIP-event timed.
// Build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool NT, int U> __global__ void __launch_bounds__(256) k_read(const f4v *__restrict__ src, size_t n, float *out)
{
    f4v acc = {0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (; i + (U - 1) * 256 < n; i += stride) {
        f4v v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * 256) : src[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
template <bool NT> __global__ void __launch_bounds__(256) k_fill(f4v *dst, size_t n, float val)
{
    f4v v = {val, val, val, val};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
    }
}
// 12 bytes per lane, lanes contiguous: the store pattern of rz_deform_kernel's outputs (global_store_dwordx3)
template <bool NT> __global__ void __launch_bounds__(256) k_fill3(float *dst, size_t nvert, float val)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvert; i += (size_t)gridDim.x * 256) {
        float *d = dst + i * 3;
        if (NT) { __builtin_nontemporal_store(val, d); __builtin_nontemporal_store(val, d + 1); __builtin_nontemporal_store(val, d + 2); }
        else { d[0] = val; d[1] = val; d[2] = val; }
    }
}
template <bool NT> __global__ void __launch_bounds__(256) k_copy(const f4v *__restrict__ src, f4v *dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        f4v v = NT ? __builtin_nontemporal_load(src + i) : src[i];
        if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
    }
}
template <class F> double timeit(F f, int iters)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        for (int i = 0; i < iters; ++i) f();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms / iters < best) best = ms / iters;
    }
    return best;
}
int main(int argc, char **argv)
{
    const bool quick = argc > 1;   // PMC calibration: one size, one launch geometry

    const size_t sizes_full[] = {184u << 20, 828u << 20, 2048u << 20};
    const size_t sizes_quick[] = {828u << 20};
    const size_t *sizes_p = quick ? sizes_quick : sizes_full;
    const int nsizes = quick ? 1 : 3;
    float *out; CK(hipMalloc(&out, 16));
    for (int si = 0; si < nsizes; ++si) {
        const size_t bytes = sizes_p[si];
        f4v *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
        CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
        const size_t n = bytes / 16;
        for (int grid : {1024, 2048, 4096, 8192}) {
            if (quick && grid != 2048) continue;
            double t;
            t = timeit([&] { k_read<false, 4><<<grid, 256>>>(a, n, out); }, 20);
            printf("{\"op\":\"read\",\"nt\":0,\"U\":4,\"MB\":%zu,\"grid\":%d,\"us\":%.2f,\"GBps\":%.1f}\n", bytes >> 20, grid, t * 1e3, bytes / t / 1e6);
            t = timeit([&] { k_read<true, 4><<<grid, 256>>>(a, n, out); }, 20);
            printf("{\"op\":\"read\",\"nt\":1,\"U\":4,\"MB\":%zu,\"grid\":%d,\"us\":%.2f,\"GBps\":%.1f}\n", bytes >> 20, grid, t * 1e3, bytes / t / 1e6);
            t = timeit([&] { k_read<true, 8><<<grid, 256>>>(a, n, out); }, 20);
            printf("{\"op\":\"read\",\"nt\":1,\"U\":8,\"MB\":%zu,\"grid\":%d,\"us\":%.2f,\"GBps\":%.1f}\n", bytes >> 20, grid, t * 1e3, bytes / t / 1e6);
            t = timeit([&] { k_fill<false><<<grid, 256>>>(b, n, 1.f); }, 20);
            printf("{\"op\":\"fill\",\"nt\":0,\"MB\":%zu,\"grid\":%d,\"us\":%.2f,\"GBps\":%.1f}\n", bytes >> 20, grid, t * 1e3, bytes / t / 1e6);
            t = timeit([&] { k_fill<true><<<grid, 256>>>(b, n, 1.f); }, 20);
            printf("{\"op\":\"fill\",\"nt\":1,\"MB\":%zu,\"grid\":%d,\"us\":%.2f,\"GBps\":%.1f}\n", bytes >> 20, grid, t * 1e3, bytes / t / 1e6);
            t = timeit([&] { k_fill3<false><<<grid, 256>>>((float *)b, bytes / 12, 1.f); }, 20);
            printf("{\"op\":\"fill3\",\"nt\":0,\"MB\":%zu,\"grid\":%d,\"us\":%.2f,\"GBps\":%.1f}\n", bytes >> 20, grid, t * 1e3, (bytes / 12 * 12) / t / 1e6);
            t = timeit([&] { k_fill3<true><<<grid, 256>>>((float *)b, bytes / 12, 1.f); }, 20);
            printf("{\"op\":\"fill3\",\"nt\":1,\"MB\":%zu,\"grid\":%d,\"us\":%.2f,\"GBps\":%.1f}\n", bytes >> 20, grid, t * 1e3, (bytes / 12 * 12) / t / 1e6);
            t = timeit([&] { k_copy<false><<<grid, 256>>>(a, b, n); }, 20);
            printf("{\"op\":\"copy\",\"nt\":0,\"MB\":%zu,\"grid\":%d,\"us\":%.2f,\"GBps\":%.1f}\n", bytes >> 20, grid, t * 1e3, 2.0 * bytes / t / 1e6);
            t = timeit([&] { k_copy<true><<<grid, 256>>>(a, b, n); }, 20);
            printf("{\"op\":\"copy\",\"nt\":1,\"MB\":%zu,\"grid\":%d,\"us\":%.2f,\"GBps\":%.1f}\n", bytes >> 20, grid, t * 1e3, 2.0 * bytes / t / 1e6);
        }
        double t = timeit([&] { CK(hipMemsetAsync(b, 0, bytes, 0)); }, 20);
        printf("{\"op\":\"hipMemset\",\"MB\":%zu,\"us\":%.2f,\"GBps\":%.1f}\n", bytes >> 20, t * 1e3, bytes / t / 1e6);
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
