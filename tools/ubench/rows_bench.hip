// rows_bench.hip — the long-window rows kernel alone on cfg-3-shaped synthetic scratch (1024 stream-windows x 64 row pairs,
// 3.5 channel pairs, random rows and tables): a seconds-long A/B loop for variants of tile_lw16.hpp / tile_lw.hpp (timing only).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -fno-slp-vectorize -DAW_XA_REG=1 -DAW_LDS_ATOMIC_READS=1 -Iairwave_amd/csrc -Iinclude tools/ubench/rows_bench.hip -o tools/ubench/rows_bench
//   -DRB_FORM=8: the 8-point kernel (rows1);  -DAW_STAMPS=1: per-phase s_memtime stamps of wave 0 (diagnostic; perturbs the timing)
//   run: rows_bench [workgroups per CU] [streams] [sustain seconds]
//   sustain > 0: launches back to back for that long first, then reports the IN-KERNEL clock of the last launches — every workgroup stamps
//   s_memtime (shader cycles) and s_memrealtime (100 MHz) once before and once after its tile loop (outside the timed code; the stamps go to
//   a buffer nothing else reads): clock = d(memtime) / d(memrealtime) x 100 MHz, median over workgroups (MI355X_MICROARCH.md, DVFS item 6)
#include "device/tile_ols.hpp"
#include "device/gpu_ctx.hpp"
#include "device/tile_lw16.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>
#ifndef RB_FORM
#define RB_FORM 16
#endif
#ifndef RB_NP
#define RB_NP 4
#define RB_REAL true
#endif
#ifndef AW_R16_MIN_WAVES
#define AW_R16_MIN_WAVES 3
#endif
namespace awk {
#if RB_FORM == 16
__global__ void __launch_bounds__(kR16Threads, AW_R16_MIN_WAVES) k_rows(LwParams p, long long n_sw, unsigned long long *dbg, unsigned long long *clk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    GpuCtx ctx{reinterpret_cast<cf *>(smem), dbg ? dbg + (long long)blockIdx.x * kStamps : nullptr};
    const int g = (int)gridDim.x, b = (int)blockIdx.x;
    const int xcd = b % 8, slot = b / 8;
    const int per_xcd_wg = (g - xcd + 7) / 8;
    lw_rows16_tiles<GpuCtx, RB_NP, RB_REAL>(ctx, p, (long long)slot, (long long)per_xcd_wg, n_sw, xcd, 8);
    ctx.flush_stamps();
    if (clk && threadIdx.x == 0) { clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0; clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
}
constexpr int kThr = kR16Threads, kLds = kR16LdsBytes;
#else
__global__ void __launch_bounds__(kThreads, 4) k_rows(LwParams p, long long n_sw, unsigned long long *dbg, unsigned long long *clk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    GpuCtx ctx{reinterpret_cast<cf *>(smem), dbg ? dbg + (long long)blockIdx.x * kStamps : nullptr};
    const int g = (int)gridDim.x, b = (int)blockIdx.x;
    const int xcd = b % 8, slot = b / 8;
    const int per_xcd_wg = (g - xcd + 7) / 8;
    lw_rows_tiles<GpuCtx, RB_NP, RB_REAL, 1>(ctx, p, (long long)slot, (long long)per_xcd_wg, n_sw, xcd, 8);
    ctx.flush_stamps();
    if (clk && threadIdx.x == 0) { clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0; clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
}
constexpr int kThr = kThreads, kLds = lw_rows_lds_elems<1>() * 8;
#endif
__global__ void k_fill(float *d, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        d[i] = (float)(x & 0xffff) / 65536.f - 0.5f;
    }
}
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char **argv) {
    using namespace awk;
    const int per_cu = argc > 1 ? atoi(argv[1]) : (RB_FORM == 16 ? 3 : 2);
    const int S = argc > 2 ? atoi(argv[2]) : 1024;
    const double sustain = argc > 3 ? atof(argv[3]) : 0.0;
    const int R = 128; const long long N = (long long)R * kLwM;
    const int real_last = RB_REAL ? 1 : 0;
    LwParams p{};
    p.R = R; p.N = (int)N; p.n_pairs = RB_NP; p.real_last = real_last; p.n_windows = 1;
    p.spec_per_sw = (long long)(RB_NP - real_last) * N + (real_last ? N / 2 : 0);
    const long long n_sw = S;
    cf *spec, *wrows; float *tab; cf *tw;
    const size_t n_spec = (size_t)n_sw * p.spec_per_sw, n_w = (size_t)n_sw * N, n_tab = (size_t)(R / 2) * RB_NP * kLwM * 8;
    CK(hipMalloc((void **)&spec, n_spec * 8)); CK(hipMalloc((void **)&wrows, n_w * 8)); CK(hipMalloc((void **)&tab, n_tab * 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (float *)spec, n_spec * 2, 1u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, tab, n_tab, 7u);
    std::vector<cf> h_tw(4096 + 512 + 64);
    for (int t = 0; t < 512; ++t) h_tw[t] = mk(cosf(-2 * 3.14159265f * t / 4096), sinf(-2 * 3.14159265f * t / 4096));                 // tw1m
    for (int m0 = 0; m0 < 16; ++m0) for (int a = 0; a < 16; ++a) h_tw[512 + m0 * 16 + a] = mk(cosf(-2 * 3.14159265f * a * m0 / 256), sinf(-2 * 3.14159265f * a * m0 / 256));
    for (int ka = 0; ka < 8; ++ka) for (int l = 0; l < 64; ++l) h_tw[1024 + ka * 64 + l] = mk(cosf(-2 * 3.14159265f * l * ka / 512), sinf(-2 * 3.14159265f * l * ka / 512));
    for (int kb = 0; kb < 8; ++kb) for (int l = 0; l < 8; ++l) h_tw[2048 + kb * 8 + l] = mk(cosf(-2 * 3.14159265f * l * kb / 64), sinf(-2 * 3.14159265f * l * kb / 64));
    CK(hipMalloc((void **)&tw, h_tw.size() * 8)); CK(hipMemcpy(tw, h_tw.data(), h_tw.size() * 8, hipMemcpyHostToDevice));
    p.spec = spec; p.wrows = wrows; p.tab = reinterpret_cast<const LwTab *>(tab); p.tab16 = reinterpret_cast<const LwTab2 *>(tab);
    p.tw1m = tw; p.tw2 = tw + 512; p.twa = tw + 1024; p.twb = tw + 2048; p.rows_form = RB_FORM;
    const long long n_tiles = n_sw * (R / 2);
    const unsigned grid = (unsigned)std::min<long long>(n_tiles, 256LL * per_cu) / 8 * 8;
    unsigned long long *dbg = nullptr;
#if AW_STAMPS
    CK(hipMalloc((void **)&dbg, (size_t)grid * kStamps * 8)); CK(hipMemset(dbg, 0, (size_t)grid * kStamps * 8));
#endif
    CK(hipFuncSetAttribute((const void *)k_rows, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned long long *clk = nullptr;
    CK(hipMalloc((void **)&clk, (size_t)grid * 16)); CK(hipMemset(clk, 0, (size_t)grid * 16));
    if (sustain > 0) {          // the chip's clock governor settles over tens of milliseconds of a load: keep it loaded first
        hipLaunchKernelGGL(k_rows, dim3(grid), dim3(kThr), kLds, 0, p, n_sw, dbg, clk);
        CK(hipEventRecord(e0)); CK(hipEventSynchronize(e0));
        const int per = 50;
        double spent = 0;
        while (spent < sustain) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < per; ++i) hipLaunchKernelGGL(k_rows, dim3(grid), dim3(kThr), kLds, 0, p, n_sw, dbg, clk);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            spent += ms * 1e-3;
            printf("  sustained: %d launches, %.3f ms each\n", per, ms / per);
        }
    }
    float best = 1e9f, sum = 0; int cnt = 0;
    for (int it = 0; it < 7; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_rows, dim3(grid), dim3(kThr), kLds, 0, p, n_sw, dbg, clk);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 2) { best = std::min(best, ms); sum += ms; ++cnt; }
    }
    CK(hipGetLastError());
    const double bytes = (double)n_spec * 8 + (double)n_w * 8;
    printf("rows kernel form %d <%d,%d> grid %u (%d per CU), %d stream-windows: best %.3f ms, mean %.3f ms, %.2f TB/s, %.2f us per tile per CU\n", RB_FORM, RB_NP, (int)RB_REAL,
           grid, per_cu, S, best, sum / cnt, bytes / best / 1e9, best * 1e3 / (n_tiles / 256.0));
    {
        std::vector<unsigned long long> hc((size_t)grid * 2);
        CK(hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> ghz;
        for (unsigned g = 0; g < grid; ++g) if (hc[2 * g + 1] > 0) ghz.push_back((double)hc[2 * g] / (double)hc[2 * g + 1] * 0.1);
        std::sort(ghz.begin(), ghz.end());
        if (!ghz.empty()) printf("in-kernel clock (last launch; d s_memtime / d s_memrealtime x 100 MHz over each workgroup's whole tile loop): median %.3f GHz, p10 %.3f, p90 %.3f (%zu workgroups)\n",
                                 ghz[ghz.size() / 2], ghz[ghz.size() / 10], ghz[ghz.size() * 9 / 10], ghz.size());
    }
#if AW_STAMPS
    std::vector<unsigned long long> h((size_t)grid * kStamps);
    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    printf("stamps (median over workgroups of stamp[i+1] - stamp[i], shader cycles; last tile of each workgroup):\n");
    for (int i = 0; i + 1 < kStamps; ++i) {
        std::vector<long long> d;
        int nx = i + 1;
        while (nx < kStamps - 1 && h[nx] == 0) ++nx;            // next stamp that is recorded (workgroup 0)
        for (unsigned g = 0; g < grid; ++g) { const auto a = h[(size_t)g * kStamps + i], b = h[(size_t)g * kStamps + nx]; if (a && b && b > a) d.push_back((long long)(b - a)); }
        if (d.empty()) continue;
        std::sort(d.begin(), d.end());
        printf("  %2d -> %2d : median %7lld  p10 %7lld  p90 %7lld  (n %zu)\n", i, nx, d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10], d.size());
    }
#endif
    return 0;
}
