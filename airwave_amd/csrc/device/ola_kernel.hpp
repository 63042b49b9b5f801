// ola_kernel.hpp — the overlap-add tile kernel (tile_ola.hpp) as a __global__ template; included by ola_kernels.hip (the product's
// instantiations) and by one-kernel probe builds under tools/ubench/.
#pragma once
#include "kernels.hpp"
#include "gpu_ctx.hpp"
#include "tile_ola.hpp"

namespace awk {

// Persistent: workgroup b owns the contiguous run [n b / g, n (b + 1) / g) of the stream-major block list — blocks of one stream are
// consecutive, so the carry stays in the workgroup's registers; a run that starts mid-stream rebuilds it (tile_ola.hpp).  No input
// line is shared between workgroups any more (blocks do not overlap), so the workgroup -> run map needs no XCD pairing; the filter
// tables (128 KB per pair) are what every XCD's L2 keeps.
template <int CS, int NP, int H>
__global__ void __launch_bounds__(kThreads) aw_fused_ola_kernel(TileParams p, long long n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GpuCtx ctx{reinterpret_cast<cf *>(smem), nullptr};
    const long long g = gridDim.x, b = blockIdx.x;
    tiles_fused_ola<GpuCtx, CS, NP, H>(ctx, p, n_tiles * b / g, n_tiles * (b + 1) / g);
}

}  // namespace awk
