// Micro-benchmark: FP64 FMA issue rate and dependent-issue latency on gfx950 (what bounds the EQ cascade kernel).
//   chains = independent FMA chains per thread (1 = fully dependent: latency; 16 = throughput)
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CH>
__global__ void __launch_bounds__(256) k_f64(double *out, int iters) {
    double a[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) a[i] = threadIdx.x * 0.001 + i;
    const double b = 1.0000001, c = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64 / CH; ++r)
#pragma unroll
            for (int i = 0; i < CH; ++i) a[i] = __builtin_fma(a[i], b, c);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CH> void run(int waves_per_simd, double *d) {
    const int iters = 4000;
    const int threads = 256;                       // 4 waves = one per SIMD
    const int blocks = 256 * waves_per_simd;       // waves_per_simd workgroups per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_f64<CH>, dim3(blocks), dim3(threads), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_f64<CH>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fmas_per_wave = (double)iters * 64;
    // cycles per wave-instruction on one SIMD, assuming 2.4 GHz and waves_per_simd waves sharing it
    const double cyc = ms * 1e-3 * 2.4e9 / (fmas_per_wave * waves_per_simd);
    printf("chains %2d, %d wave(s)/SIMD: %.3f ms -> %.2f cycles per wave-FMA per SIMD (%.1f TFLOP/s)\n", CH, waves_per_simd, ms, cyc,
           2.0 * 64 * fmas_per_wave * blocks * 4 / (ms * 1e-3) / 1e12);
}

int main() {
    double *d; hipMalloc(&d, sizeof(double) * 256 * 256 * 8);
    for (int w : {1, 2, 4}) { run<1>(w, d); run<2>(w, d); run<4>(w, d); run<16>(w, d); }
    return 0;
}
