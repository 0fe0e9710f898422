// What does a frame pay for its 16 KB of per-frame input? Per-iteration time (HIP events over 2000 iterations, one stream) of
//   A  K                              K = a ~16 us streaming kernel (reads 104 MB), standing in for a 1/8-shard frame
//   B  hipMemcpyAsync(16.6 KB pinned -> device) ; K
//   C  B + hipEventRecord (the staging-slot marker)
//   D  fetch kernel (1 workgroup copies the 16.6 KB out of mapped pinned host memory) ; K
//   E  K reads the 16.6 KB straight from mapped pinned host memory in EVERY workgroup
//   F  D with the fetch done by the first workgroup of K itself, the others spin on a flag  (not built: needs co-residency)
// hipcc --offload-arch=gfx950 -O3 tools/archive/uploadbench.hip -o tools/archive/uploadbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_stream(const f4v *src, size_t n, const float4 *pose, int pose_n, float *out)
{
    __shared__ float4 sp[1100];
    for (int i = threadIdx.x; i < pose_n; i += 256) sp[i] = pose[i];
    __syncthreads();
    f4v acc = {0, 0, 0, 0};
    const size_t per = (n + gridDim.x - 1) / gridDim.x, b = blockIdx.x * per, e = min(n, b + per);
    for (size_t i = b + threadIdx.x; i < e; i += 256) acc += __builtin_nontemporal_load(src + i);
    acc.x += sp[threadIdx.x % pose_n].x;
    if (acc.x == 1234.5f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
__global__ void __launch_bounds__(256) k_fetch(const float4 *host, float4 *dev, int n)
{
    for (int i = threadIdx.x; i < n; i += 256) dev[i] = host[i];
}
int main()
{
    const size_t bytes = 104u << 20, n = bytes / 16;
    const int PN = 1040;                    // 16.6 KB of float4
    f4v *src; float4 *dpose, *hpose, *hmapped; float *out;
    CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 0, bytes)); CK(hipMalloc(&dpose, PN * 16)); CK(hipMalloc(&out, 4096));
    CK(hipHostMalloc(&hpose, PN * 16, hipHostMallocDefault));
    CK(hipHostMalloc(&hmapped, PN * 16, hipHostMallocMapped));
    memset(hpose, 0, PN * 16); memset(hmapped, 0, PN * 16);
    float4 *hdev = nullptr;
    CK(hipHostGetDevicePointer((void **)&hdev, hmapped, 0));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1, mark[8]; hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto &m : mark) hipEventCreateWithFlags(&m, hipEventDisableTiming);
    const int N = 2000, GRID = 492;
    auto run = [&](const char *name, auto body) {
        for (int i = 0; i < 300; ++i) body(i);
        hipStreamSynchronize(s);
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, s);
            for (int i = 0; i < N; ++i) body(i);
            hipEventRecord(e1, s); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-78s %.2f us per iteration\n", name, best / N * 1e3);
    };
    run("A  K", [&](int) { k_stream<<<GRID, 256, 0, s>>>(src, n, dpose, PN, out); });
    run("B  hipMemcpyAsync(16.6 KB pinned -> device) ; K", [&](int) { hipMemcpyAsync(dpose, hpose, PN * 16, hipMemcpyHostToDevice, s); k_stream<<<GRID, 256, 0, s>>>(src, n, dpose, PN, out); });
    run("C  hipMemcpyAsync ; hipEventRecord ; K", [&](int i) { hipMemcpyAsync(dpose, hpose, PN * 16, hipMemcpyHostToDevice, s); hipEventRecord(mark[i & 7], s); k_stream<<<GRID, 256, 0, s>>>(src, n, dpose, PN, out); });
    run("D  fetch kernel (1 WG reads mapped host memory) ; K", [&](int) { k_fetch<<<1, 256, 0, s>>>(hdev, dpose, PN); k_stream<<<GRID, 256, 0, s>>>(src, n, dpose, PN, out); });
    run("D' fetch kernel ; hipEventRecord ; K", [&](int i) { k_fetch<<<1, 256, 0, s>>>(hdev, dpose, PN); hipEventRecord(mark[i & 7], s); k_stream<<<GRID, 256, 0, s>>>(src, n, dpose, PN, out); });
    run("E  K reads the pose from mapped host memory in every workgroup", [&](int) { k_stream<<<GRID, 256, 0, s>>>(src, n, hdev, PN, out); });
    run("G  2 x hipMemcpyAsync (16.4 KB + 256 B) ; hipEventRecord ; K  (round 1's frame)", [&](int i) { hipMemcpyAsync(dpose, hpose, 1024 * 16, hipMemcpyHostToDevice, s); hipMemcpyAsync(dpose + 1024, hpose + 1024, 256, hipMemcpyHostToDevice, s); hipEventRecord(mark[i & 7], s); k_stream<<<GRID, 256, 0, s>>>(src, n, dpose, PN, out); });
    return 0;
}
