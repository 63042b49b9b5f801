// stream_probe.hip — read-only probe of the rows kernel's access shape: 256-thread workgroups, each step one 32-KB block (16 x 8-byte
// loads per thread, 512-byte runs per wave) with the next block's loads in flight, blocks handed to workgroups in different orders:
//   mode 0: block = step * G + g            (workgroups that run together read neighbouring blocks: a streaming read)
//   mode 1: block = g * steps + step        (every workgroup walks its own contiguous region)
//   mode 2: the rows kernel's order         (row pair = xcd + 8 q fixed, stream-window varies fastest over the slots; 7 rows of a tile 4 MB apart)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/stream_probe.hip -o tools/ubench/stream_probe ; run: stream_probe [wgs per CU] [random data 0|1] [sustain seconds]
//   sustain > 0: each mode is launched back to back for that long first; the in-kernel clock of the last launch is reported like rows_bench does
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256, 3) k_probe(const v2f *buf, v2f *sink, long long n_blocks, int mode, long long n_sw, unsigned long long *clk) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const long long G = gridDim.x, g = blockIdx.x;
    const long long steps = n_blocks / G;
    const int tid = threadIdx.x;
    v2f acc = {0.f, 0.f};
    v2f raw[16];
    auto block_of = [&](long long s) -> long long {
        if (mode == 0) return s * G + g;
        if (mode == 1) return g * steps + s;
        // mode 2: tiles (rp, sw) with 7 rows each; xcd = g % 8 handles rp = xcd + 8 q; slot = g / 8 strides over sw
        const long long tiles_step = s / 7; const int row = (int)(s % 7);
        const long long per_xcd = G / 8, slot = g / 8, xcd = g % 8;
        const long long vid = slot + tiles_step * per_xcd;             // virtual tile of this XCD
        const long long q = vid / n_sw, sw = vid % n_sw;
        const long long rp = xcd + 8 * q;
        const int pair = row >> 1; const long long r = (row & 1) ? 127 - rp : rp;
        // layout [sw][pair][row r][4096]: 3.5 pairs x 128 rows per sw
        return (sw * 448 + (long long)pair * 128 + r);
    };
    auto load = [&](long long b) {
        const v2f *p = buf + b * 4096 + tid;
#pragma unroll
        for (int j = 0; j < 16; ++j) raw[j] = __builtin_nontemporal_load(p + 256 * j);
    };
    load(block_of(0));
    for (long long s = 0; s < steps; ++s) {
        v2f v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = raw[j];
        load(block_of(s + 1 < steps ? s + 1 : s));
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += v[j];
    }
    if (acc.x == 1.2345e-30f) sink[g * 256 + tid] = acc;
    if (clk && tid == 0) { clk[2 * g] = __builtin_amdgcn_s_memtime() - c0; clk[2 * g + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
}
__global__ void k_fill(float *d, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        d[i] = (float)(x & 0xffff) / 65536.f - 0.5f;
    }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char **argv) {
    const int per_cu = argc > 1 ? atoi(argv[1]) : 3;
    const long long n_sw = 1024, n_blocks_total = n_sw * 448;            // 448 rows of 32 KB per stream-window = 15 GB
    v2f *buf, *sink;
    CK(hipMalloc((void **)&buf, (size_t)n_blocks_total * 4096 * 8));
    CK(hipMalloc((void **)&sink, 4096 * 256 * 8));
    CK(hipMemset(buf, 0, (size_t)n_blocks_total * 4096 * 8));
    if (argc > 2 && atoi(argv[2]) == 1) {                   // random contents instead of zeros
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (float *)buf, (size_t)n_blocks_total * 4096 * 2, 1u);
        CK(hipDeviceSynchronize());
    }
    const unsigned G = 256u * per_cu;
    const double sustain = argc > 3 ? atof(argv[3]) : 0.0;
    unsigned long long *clk; CK(hipMalloc((void **)&clk, (size_t)G * 16)); CK(hipMemset(clk, 0, (size_t)G * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
        const long long n_blocks = mode == 2 ? (n_sw * 64 * 7 / G) * G : n_blocks_total / G * G;
        float best = 1e9f;
        for (double spent = 0; spent < sustain;) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_probe, dim3(G), dim3(256), 0, 0, buf, sink, n_blocks, mode, n_sw, clk);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            spent += ms * 1e-3;
        }
        for (int it = 0; it < 5; ++it) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_probe, dim3(G), dim3(256), 0, 0, buf, sink, n_blocks, mode, n_sw, clk);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 1) best = std::min(best, ms);
        }
        CK(hipGetLastError());
        printf("mode %d, %u workgroups (%d per CU): %.3f ms for %.2f GB -> %.2f TB/s\n", mode, G, per_cu, best, n_blocks * 32768.0 / 1e9, n_blocks * 32768.0 / best / 1e9);
        {
            std::vector<unsigned long long> hc((size_t)G * 2);
            CK(hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> ghz;
            for (unsigned g = 0; g < G; ++g) if (hc[2 * g + 1] > 0) ghz.push_back((double)hc[2 * g] / (double)hc[2 * g + 1] * 0.1);
            std::sort(ghz.begin(), ghz.end());
            if (!ghz.empty()) printf("  in-kernel clock: median %.3f GHz, p10 %.3f, p90 %.3f\n", ghz[ghz.size() / 2], ghz[ghz.size() / 10], ghz[ghz.size() * 9 / 10]);
        }
    }
    return 0;
}
