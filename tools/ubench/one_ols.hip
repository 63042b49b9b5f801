// Probe build: only the 8192-window tile code for one layout (seconds instead of minutes), to read its register allocation.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -Iairwave_amd/csrc -Iinclude -S --cuda-device-only -o /tmp/one1.s tools/ubench/one_ols.hip
#include "device/tile_ols.hpp"
#include "device/gpu_ctx.hpp"
#ifndef ONE_CS
#define ONE_CS 8
#define ONE_NP 4
#endif
namespace awk {
__global__ void __launch_bounds__(kThreads) one_ols(TileParams p, long long n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GpuCtx ctx{reinterpret_cast<cf *>(smem), nullptr};
    tiles_fused_ols<GpuCtx, ONE_CS, ONE_NP, true>(ctx, p, blockIdx.x, gridDim.x, n_tiles);
}
}
