// one_rows16.hip — register allocation of the 16-point rows kernel <4, true> (cfg 3's instantiation) in seconds:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -fno-slp-vectorize -Iairwave_amd/csrc -Iinclude -c tools/ubench/one_rows16.hip -Rpass-analysis=kernel-resource-usage -o /dev/null
#include "device/tile_ols.hpp"
#include "device/gpu_ctx.hpp"
#include "device/tile_lw16.hpp"
#ifndef AW_R16_MIN_WAVES
#define AW_R16_MIN_WAVES 3
#endif
#ifndef PROBE_NP
#define PROBE_NP 4
#define PROBE_REAL true
#endif
namespace awk {
__global__ void __launch_bounds__(kR16Threads, AW_R16_MIN_WAVES) k_rows16(LwParams p, long long n_sw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GpuCtx ctx{reinterpret_cast<cf *>(smem), nullptr};
    const int g = (int)gridDim.x, b = (int)blockIdx.x;
    const int xcd = b % 8, slot = b / 8;
    const int per_xcd_wg = (g - xcd + 7) / 8;
    lw_rows16_tiles<GpuCtx, PROBE_NP, PROBE_REAL>(ctx, p, (long long)slot, (long long)per_xcd_wg, n_sw, xcd, 8);
}
}
