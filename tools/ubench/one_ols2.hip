// Probe build: only the 16384-window tile code for one layout (seconds instead of minutes), to read its register allocation.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -Iairwave_amd/csrc -Iinclude -S --cuda-device-only -o /tmp/one2.s tools/ubench/one_ols2.hip
#include "device/tile_ols2.hpp"
#include "device/gpu_ctx.hpp"
#ifndef ONE_CS
#define ONE_CS 8
#define ONE_NB 4
#endif
namespace awk {
__global__ void __launch_bounds__(kThreads) one_ols2(TileParams p, long long n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GpuCtx ctx{reinterpret_cast<cf *>(smem), nullptr};
    tiles_fused_ols2<GpuCtx, ONE_CS, ONE_NB, true>(ctx, p, blockIdx.x, gridDim.x, n_tiles);
}
}
