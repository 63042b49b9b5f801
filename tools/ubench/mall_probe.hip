// mall_probe.hip — what the 256 MiB Infinity Cache does for a produce -> consume scratch buffer.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mall_probe.hip -o tools/ubench/mall_probe
// For buffer sizes 16 MiB .. 4 GiB: (W) a kernel that only stores the buffer, (R) a kernel that only loads it,
// (WR) store then load, alternating — each timed over many repetitions with HIP events.  If stores of a buffer
// that fits the cache run well above the HBM copy rate, the cache absorbs them (write-back); if the loads do, the
// freshly stored lines are served from it.  Also prints the plain device-to-device copy ceiling (2 GiB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_store(float4 *p, size_t n4, float v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) p[i] = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
}
__global__ void __launch_bounds__(256) k_load(const float4 *p, size_t n4, float *sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 1.2345e-30f) *sink = acc;
}
__global__ void __launch_bounds__(256) k_store_nt(float4 *p, size_t n4, float v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) { v4f x = {v, v + 1.f, v + 2.f, v + 3.f}; __builtin_nontemporal_store(x, (v4f *)(p + i)); }
}
__global__ void __launch_bounds__(256) k_load_nt(const float4 *p, size_t n4, float *sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) { const v4f v = __builtin_nontemporal_load((const v4f *)(p + i)); acc += v.x + v.y + v.z + v.w; }
    if (acc == 1.2345e-30f) *sink = acc;
}
// 8-byte accesses (the spectra are complex64): store / load float2
__global__ void __launch_bounds__(256) k_store8(float2 *p, size_t n2, float v, int nt) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    if (nt) for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) { v2f x = {v, v + 1.f}; __builtin_nontemporal_store(x, (v2f *)(p + i)); }
    else for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) p[i] = make_float2(v, v + 1.f);
}
__global__ void __launch_bounds__(256) k_copy_nt(const float4 *a, float4 *b, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load((const v4f *)(a + i)), (v4f *)(b + i));
}
__global__ void __launch_bounds__(256) k_copy(const float4 *a, float4 *b, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) b[i] = a[i];
}

int main() {
    const size_t max_bytes = (size_t)4 << 30;
    float4 *buf = nullptr, *buf2 = nullptr; float *sink = nullptr;
    CK(hipMalloc((void **)&buf, max_bytes));
    CK(hipMalloc((void **)&buf2, (size_t)2 << 30));
    CK(hipMalloc((void **)&sink, 4));
    CK(hipMemset(buf, 0, max_bytes));
    CK(hipMemset(buf2, 0, (size_t)2 << 30));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    auto timeit = [&](auto &&fn, int reps) { fn(); fn(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps; };
    {
        const size_t n4 = ((size_t)2 << 30) / 16;
        const float ms = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, buf, buf2, n4); }, 10);
        printf("copy 2 GiB: %.3f ms  %.2f TB/s read+write\n", ms, 2.0 * (2ull << 30) / ms / 1e9);
    }
    {
        const size_t n4 = ((size_t)2 << 30) / 16;
        float ms = timeit([&] { hipLaunchKernelGGL(k_copy_nt, dim3(grid), dim3(256), 0, 0, buf, buf2, n4); }, 10);
        printf("copy 2 GiB nt/nt: %.3f ms  %.2f TB/s read+write\n", ms, 2.0 * (2ull << 30) / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL(k_store, dim3(grid), dim3(256), 0, 0, buf, n4, 1.f); }, 10);
        printf("store 2 GiB 16B plain: %.2f TB/s\n", (double)(2ull << 30) / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL(k_store_nt, dim3(grid), dim3(256), 0, 0, buf, n4, 1.f); }, 10);
        printf("store 2 GiB 16B nt   : %.2f TB/s\n", (double)(2ull << 30) / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL(k_store8, dim3(grid), dim3(256), 0, 0, (float2 *)buf, 2 * n4, 1.f, 0); }, 10);
        printf("store 2 GiB 8B plain : %.2f TB/s\n", (double)(2ull << 30) / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL(k_store8, dim3(grid), dim3(256), 0, 0, (float2 *)buf, 2 * n4, 1.f, 1); }, 10);
        printf("store 2 GiB 8B nt    : %.2f TB/s\n", (double)(2ull << 30) / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL(k_load, dim3(grid), dim3(256), 0, 0, buf, n4, sink); }, 10);
        printf("load 2 GiB 16B plain : %.2f TB/s\n", (double)(2ull << 30) / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL(k_load_nt, dim3(grid), dim3(256), 0, 0, buf, n4, sink); }, 10);
        printf("load 2 GiB 16B nt    : %.2f TB/s\n", (double)(2ull << 30) / ms / 1e9);
        // concurrent store-only and load-only kernels on two streams (does the fabric serve both directions at once?)
        hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
        ms = timeit([&] { hipLaunchKernelGGL(k_store, dim3(grid / 2), dim3(256), 0, s1, buf, n4, 1.f);
                          hipLaunchKernelGGL(k_load, dim3(grid / 2), dim3(256), 0, s2, buf2, n4 / 1, sink);
                          CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); }, 10);
        printf("store 2 GiB || load 2 GiB on two streams: %.3f ms  %.2f TB/s total\n", ms, 2.0 * (2ull << 30) / ms / 1e9);
    }
    const size_t sizes_mb[] = {16, 32, 64, 96, 128, 160, 192, 224, 256, 320, 512, 1024, 4096};
    printf("%8s %12s %12s %12s %12s\n", "MiB", "W TB/s", "R TB/s", "W+R TB/s", "WR pair ms");
    for (size_t mb : sizes_mb) {
        const size_t bytes = mb << 20, n4 = bytes / 16;
        const int reps = mb <= 256 ? 200 : 20;
        const float w = timeit([&] { hipLaunchKernelGGL(k_store, dim3(grid), dim3(256), 0, 0, buf, n4, 1.0f); }, reps);
        const float r = timeit([&] { hipLaunchKernelGGL(k_load, dim3(grid), dim3(256), 0, 0, buf, n4, sink); }, reps);
        const float wr = timeit([&] { hipLaunchKernelGGL(k_store, dim3(grid), dim3(256), 0, 0, buf, n4, 2.0f);
                                      hipLaunchKernelGGL(k_load, dim3(grid), dim3(256), 0, 0, buf, n4, sink); }, reps);
        printf("%8zu %12.2f %12.2f %12.2f %12.4f\n", mb, bytes / w / 1e9, bytes / r / 1e9, 2.0 * bytes / wr / 1e9, wr);
    }
    // ring: a stream of "chunks": store chunk i (size c) into slot i % slots of a ring, then load it; the ring
    // (slots * c) either fits the cache or not, the total traffic is far larger than the cache either way.
    printf("ring of 8 slots, store slot then load slot:\n");
    for (size_t mb : {8, 16, 24, 32, 64, 128}) {
        const size_t bytes = mb << 20, n4 = bytes / 16;
        int i = 0;
        const float ms = timeit([&] { float4 *s = buf + (size_t)(i++ % 8) * n4;
                                      hipLaunchKernelGGL(k_store, dim3(grid), dim3(256), 0, 0, s, n4, 3.0f);
                                      hipLaunchKernelGGL(k_load, dim3(grid), dim3(256), 0, 0, s, n4, sink); }, 200);
        printf("  slot %4zu MiB (ring %5zu MiB): %.2f TB/s store+load\n", mb, 8 * mb, 2.0 * bytes / ms / 1e9);
    }
    return 0;
}
