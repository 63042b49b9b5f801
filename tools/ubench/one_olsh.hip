// Probe build: only the sibling tile code for <8, 4, true> (seconds instead of minutes), to read its register allocation.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iairwave_amd/csrc -Iinclude -S --cuda-device-only -o /tmp/one.s tools/ubench/one_olsh.hip
#include "device/tile_olsh.hpp"
#include "device/gpu_ctx.hpp"
#ifndef AW_OLSH_WAVES
#define AW_OLSH_WAVES 4
#endif
#ifndef ONE_CS
#define ONE_CS 8
#define ONE_NP 4
#endif
namespace awk {
__global__ void __launch_bounds__(kThreads, AW_OLSH_WAVES) one_olsh(TileParams p, long long n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GpuCtx ctx{reinterpret_cast<cf *>(smem), nullptr};
    tiles_fused_olsq<GpuCtx, ONE_CS, ONE_NP, true>(ctx, p, blockIdx.x >> 1, gridDim.x >> 1, n_tiles, blockIdx.x & 1);
}
}
