// Micro-benchmark: one wave-private LDS exchange (write 128 B per lane, read it back transposed, consume) as the
// tile kernel's sub-FFT does it, at the kernel's occupancy (512 threads = 2 waves/SIMD, one workgroup per CU):
//   mode 0: 16 ds_write_b64 + 16 ds_read_b64   (today's kernel: one complex value per access)
//   mode 1:  8 ds_write_b128 + 8 ds_read_b128  (two complex values per access, same bytes)
// Prints cycles per exchange per wave.  MI355X_MICROARCH.md §LDS: 8-byte accesses need ~4 waves/SIMD to reach their
// rate when drained every 16 operations, 16-byte ones reach theirs with one.
#include <hip/hip_runtime.h>
#include <cstdio>

struct alignas(8) f2 { float x, y; };
struct alignas(16) f4 { float x, y, z, w; };

template <int MODE>
__global__ void __launch_bounds__(512) k_exch(float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    unsigned char *mine = smem + wave * 18432;                 // 2 rows x 576 x 16 B per wave
    float acc = 0.f;
    f2 a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i].x = t + i; a[i].y = i; }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            f2 *s = reinterpret_cast<f2 *>(mine);
#pragma unroll
            for (int k = 0; k < 16; ++k) s[(k & 7) * 72 + lane + (k >> 3) * 576] = a[k];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int l0 = lane & 7, kap = lane >> 3;
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = s[kap * 72 + l0 + 8 * (k & 7) + (k >> 3) * 576];
        } else {
            f4 *s = reinterpret_cast<f4 *>(mine);
#pragma unroll
            for (int k = 0; k < 8; ++k) { f4 v; v.x = a[k].x; v.y = a[k].y; v.z = a[k + 8].x; v.w = a[k + 8].y; s[k * 72 + lane] = v; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int l0 = lane & 7, kap = lane >> 3;
#pragma unroll
            for (int k = 0; k < 8; ++k) { const f4 v = s[kap * 72 + l0 + 8 * k]; a[k].x = v.x; a[k].y = v.y; a[k + 8].x = v.z; a[k + 8].y = v.w; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // a radix-8-sized dose of arithmetic that consumes every value (keeps the loads honest)
#pragma unroll
        for (int k = 0; k < 16; ++k) { a[k].x = a[k].x * 0.999f + a[(k + 1) & 15].y; a[k].y = a[k].y * 1.001f - a[(k + 3) & 15].x * 0.5f; }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += a[k].x + a[k].y;
    out[blockIdx.x * 512 + t] = acc;
}

template <int MODE> void run(float *d, const char *name) {
    const int iters = 20000;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_exch<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 18432);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_exch<MODE>, dim3(256), dim3(512), 8 * 18432, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_exch<MODE>, dim3(256), dim3(512), 8 * 18432, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.3f ms -> %.0f cycles per exchange+butterfly step per wave (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / iters);
}

int main() {
    float *d; hipMalloc(&d, sizeof(float) * 256 * 512);
    run<0>(d, "16 x b64 write + 16 x b64 read");
    run<1>(d, " 8 x b128 write + 8 x b128 read");
    run<0>(d, "16 x b64 write + 16 x b64 read");
    run<1>(d, " 8 x b128 write + 8 x b128 read");
    return 0;
}
