// ols2_kernel.hpp — the 16384-frame-window tile kernel template, shared by the two translation units that instantiate it:
// kernels.hip (1, 2, 3, 5, 7 channels, built with -fno-slp-vectorize) and ols2_even_kernels.hip (4, 6, 8 channels, built with
// the SLP vectoriser: without it those layouts spill 212 instead of 96 VGPRs — tools/ubench/one_ols2.hip — and run 25-30 %
// slower (8 channels, 4320 taps: 21.6 against 24.9 G frames/s), while the others drop from 36 spills to 3 and run 10-20 %
// faster (stereo 6146 taps: 121 against 99)).
#pragma once
#include "kernels.hpp"
#include "gpu_ctx.hpp"

namespace awk {

// The 16384-frame window path (tile_ols2.hpp).  CS = real channels, NB = batches of four pseudo-channels.
template <int CS, int NB, bool INTERIOR>
__global__ void __launch_bounds__(kThreads) aw_fused_ols2_kernel(TileParams p, long long n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GpuCtx ctx{reinterpret_cast<cf *>(smem), p.dbg ? p.dbg + (long long)blockIdx.x * kStamps : nullptr};
    ctx.stamp_thread_ = p.stagger;
    const long long g = gridDim.x, b = blockIdx.x;
    const long long xcd = b % 8, slot = b / 8;
    const long long per_xcd_wg = (g - xcd + 7) / 8;
    const long long q = n_tiles / 8, r = n_tiles % 8;
    const long long lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const long long hi = lo + (xcd < r ? q + 1 : q);
    tiles_fused_ols2<GpuCtx, CS, NB, INTERIOR>(ctx, p, lo + slot, per_xcd_wg, hi);
}

// even channel counts (ols2_even_kernels.hip)
hipError_t prepare_ols2_even();
void launch_ols2_even(const TileParams &p, bool interior, long long n_tiles, dim3 grid, hipStream_t stream);

}  // namespace awk
