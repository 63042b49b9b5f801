// Micro-benchmark: VALU issue rate vs waves per SIMD on gfx950 (v_fma_f32, v_add_f32, v_pk_fma_f32, ds_write/ds_read).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(1024) k_valu(float *out, int iters) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    float b = 1.0001f, c = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) a[i] = __builtin_fmaf(a[i], b, c);
                else if (MODE == 1) a[i] = a[i] + c;
                else if (MODE == 2) a[i] = a[i] * b;
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef float float2v __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(1024) k_pk(float *out, int iters) {
    float2v a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = float2v{threadIdx.x * 0.001f + i, 1.0f * i};
    float2v b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], b, c);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// LDS: each lane writes then reads 8-byte elements, conflict-free contiguous
template <int W>   // bytes per lane: 4, 8, 16
__global__ void __launch_bounds__(1024) k_lds(float *out, int iters, int do_write, int do_read) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x;
    float acc = 0;
    if (W == 8) {
        float2 *p = reinterpret_cast<float2 *>(smem);
        float2 v = {t * 1.0f, 2.0f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (do_write) p[t + r * blockDim.x] = v;
                if (do_read) { float2 q = p[t + r * blockDim.x]; acc += q.x; v.y += q.y * 1e-9f; }
            }
        }
    } else if (W == 4) {
        float *p = reinterpret_cast<float *>(smem);
        float v = t;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (do_write) p[t + r * blockDim.x] = v;
                if (do_read) { float q = p[t + r * blockDim.x]; acc += q; v += q * 1e-9f; }
            }
        }
    } else {
        float4 *p = reinterpret_cast<float4 *>(smem);
        float4 v = {t * 1.0f, 2.0f, 3.f, 4.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (do_write) p[t + r * blockDim.x] = v;
                if (do_read) { float4 q = p[t + r * blockDim.x]; acc += q.x; v.y += q.w * 1e-9f; }
            }
        }
    }
    out[blockIdx.x * blockDim.x + t] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    float *out; CK(hipMalloc(&out, 256 * 1024 * sizeof(float) * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const double clk = prop.clockRate * 1e3;   // Hz
    printf("device %s, CUs %d, clock %.0f MHz\n", prop.name, prop.multiProcessorCount, clk / 1e6);
    for (int mode = 0; mode < 4; ++mode) {
        for (int wps : {1, 2, 4}) {            // waves per SIMD
            const int threads = 64 * 4 * wps;  // one workgroup per CU, 4 SIMDs
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(k_valu<0>, dim3(256), dim3(threads), 0, 0, out, iters);
                else if (mode == 1) hipLaunchKernelGGL(k_valu<1>, dim3(256), dim3(threads), 0, 0, out, iters);
                else if (mode == 2) hipLaunchKernelGGL(k_valu<2>, dim3(256), dim3(threads), 0, 0, out, iters);
                else hipLaunchKernelGGL(k_pk, dim3(256), dim3(threads), 0, 0, out, iters);
            };
            launch(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double instr_per_wave = (double)iters * 8 * (mode == 3 ? 8 : 16);
            const double cyc = ms * 1e-3 * clk;
            printf("mode %d (%s) waves/SIMD %d: %.3f ms, cycles per wave-instruction per SIMD = %.2f (nominal clock)\n", mode,
                   mode == 0 ? "v_fma_f32" : mode == 1 ? "v_add_f32" : mode == 2 ? "v_mul_f32" : "v_pk_fma_f32", wps, ms,
                   cyc / (instr_per_wave * wps));
        }
    }
    // LDS
    for (int W : {4, 8, 16}) for (int rw = 1; rw <= 3; ++rw) for (int wps : {1, 2, 4}) {
        const int threads = 64 * 4 * wps;
        const int elems = W == 16 ? 8 : 16;
        const size_t lds = (size_t)threads * elems * W;
        if (lds > 160 * 1024) continue;
        const int dw = rw & 1, dr = (rw >> 1) & 1;
        auto launch = [&]() {
            if (W == 4) hipLaunchKernelGGL(k_lds<4>, dim3(256), dim3(threads), lds, 0, out, iters, dw, dr);
            else if (W == 8) hipLaunchKernelGGL(k_lds<8>, dim3(256), dim3(threads), lds, 0, out, iters, dw, dr);
            else hipLaunchKernelGGL(k_lds<16>, dim3(256), dim3(threads), lds, 0, out, iters, dw, dr);
        };
        if (W == 4) CK(hipFuncSetAttribute((const void *)k_lds<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (W == 8) CK(hipFuncSetAttribute((const void *)k_lds<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (W == 16) CK(hipFuncSetAttribute((const void *)k_lds<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double ops_per_wave = (double)iters * elems * (dw + dr);
        const double cyc = ms * 1e-3 * clk;
        const double waves = 4.0 * wps;
        printf("LDS %2d B/lane %s waves/SIMD %d: %.3f ms, CU cycles per wave-instruction = %.2f, B/clk/CU = %.1f\n", W,
               rw == 1 ? "write" : rw == 2 ? "read " : "w+r  ", wps, ms, cyc / (ops_per_wave * waves), (ops_per_wave * waves * 64 * W) / cyc);
    }
    return 0;
}
