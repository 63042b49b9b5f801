// ols2_kernel.hpp — the 16384-frame-window tile kernel template, shared by the two translation units that instantiate it:
// kernels.hip (1, 2, 3, 5, 7 channels) and ols2_even_kernels.hip (4, 6, 8 channels), both built with -fno-slp-vectorize.
// History: on the 8 x 8 x 8 row transforms of rounds 1-3 the even layouts wanted the SLP vectoriser (without it they spilled 212
// instead of 96 VGPRs — tools/ubench/one_ols2.hip — and ran 25-30 % slower) while the others dropped from 36 spills to 3 without it;
// on the half-wave row transform of round 4 (tables in parts of four bins) nothing spills and every layout is faster without SLP.
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
