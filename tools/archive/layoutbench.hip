// layoutbench.hip — does the HBM LAYOUT of the dense morph targets cap the fused kernel's read stream?
// rz_deform_kernel reads, per wave step of 64 quads, 3 planes x 64 morphs: 192 pieces of 1 KiB that sit 4 MB apart (plane-major
// layout D[m][3][Vp]). A tile-major layout D[tile][m][3][TQ quads] makes the same 192 pieces one contiguous block. Both patterns
// are emulated here with the kernel's own shape (persistent grid, 2 workgroups per CU, 4 waves, 24 nontemporal 16-byte loads in
// flight per lane) next to the plain grid-stride stream of tools/membench; V = 1 M vertices, M = 64 -> 768 MB.
// Build: hipcc --offload-arch=gfx950 -O3 tools/archive/layoutbench.hip -o tools/archive/layoutbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// MODE 0: plane-major (the product's layout). MODE 1: tile-major, tile = TQ quads. S = lanes per quad is 1 here (the 1 M mesh's
// S = 2 halves the quads per step and doubles the morph stride; same bytes).
template <int MODE, int U>
__global__ void __launch_bounds__(256, 2) k_morph(const f4v *__restrict__ D, size_t plane4, int M, uint32_t n_quads, uint32_t quads_per_wave, uint32_t TQ, float *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t wave_global = blockIdx.x * 4 + wave;
    const size_t q_begin = (size_t)wave_global * quads_per_wave;
    const size_t q_end = min((size_t)n_quads, q_begin + quads_per_wave);
    f4v ax = {0, 0, 0, 0}, ay = ax, az = ax;
    for (size_t qw = q_begin; qw < q_end; qw += 64) {
        const size_t q = qw + lane;
        if (q >= q_end) continue;
        for (int a0 = 0; a0 + U <= M; a0 += U) {
            f4v dx[U], dy[U], dz[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int m = a0 + u;
                const f4v *d;
                size_t step;
                if (MODE == 0) { d = D + (size_t)m * 3 * plane4 + q; step = plane4; }
                else { const size_t tile = q / TQ, qi = q % TQ; d = D + ((tile * M + m) * 3) * (size_t)TQ + qi; step = TQ; }
                dx[u] = __builtin_nontemporal_load(d);
                dy[u] = __builtin_nontemporal_load(d + step);
                dz[u] = __builtin_nontemporal_load(d + 2 * step);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { ax += dx[u]; ay += dy[u]; az += dz[u]; }
        }
    }
    const f4v s = ax + ay + az;
    if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = s.x;
}

template <class F> double timeit(F f, int iters)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3 / iters;
}

int main()
{
    const uint32_t V = 1000000, Vp = 1000448, M = 64;           // Vp: multiple of 1024
    const size_t plane4 = Vp / 4, total4 = (size_t)M * 3 * (plane4 + 1024);      // room for the last (partial) tile of every tile size
    f4v *D;
    float *out;
    CK(hipMalloc(&D, total4 * sizeof(f4v)));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(D, 0, total4 * sizeof(f4v)));
    const uint32_t n_quads = (V + 3) / 4;
    const double bytes = (double)M * 3 * n_quads * 16;
    for (int rep = 0; rep < 2; ++rep)
        for (uint32_t grid : {512u, 1024u}) {
            uint32_t per_wave = (n_quads + grid * 4 - 1) / (grid * 4);
            per_wave = (per_wave + 63) / 64 * 64;
            const uint32_t g = (n_quads + per_wave * 4 - 1) / (per_wave * 4);
            double t;
            t = timeit([&] { k_morph<0, 8><<<g, 256>>>(D, plane4, M, n_quads, per_wave, 256, out); }, 30);
            printf("{\"layout\":\"plane-major\",\"U\":8,\"grid\":%u,\"us\":%.2f,\"GBps\":%.1f}\n", g, t, bytes / t / 1e3);
            t = timeit([&] { k_morph<0, 4><<<g, 256>>>(D, plane4, M, n_quads, per_wave, 256, out); }, 30);
            printf("{\"layout\":\"plane-major\",\"U\":4,\"grid\":%u,\"us\":%.2f,\"GBps\":%.1f}\n", g, t, bytes / t / 1e3);
            for (uint32_t TQ : {64u, 256u, 1024u}) {
                t = timeit([&] { k_morph<1, 8><<<g, 256>>>(D, plane4, M, n_quads, per_wave, TQ, out); }, 30);
                printf("{\"layout\":\"tile-major\",\"TQ\":%u,\"U\":8,\"grid\":%u,\"us\":%.2f,\"GBps\":%.1f}\n", TQ, g, t, bytes / t / 1e3);
                t = timeit([&] { k_morph<1, 4><<<g, 256>>>(D, plane4, M, n_quads, per_wave, TQ, out); }, 30);
                printf("{\"layout\":\"tile-major\",\"TQ\":%u,\"U\":4,\"grid\":%u,\"us\":%.2f,\"GBps\":%.1f}\n", TQ, g, t, bytes / t / 1e3);
            }
        }
    return 0;
}
