// lw_kernels.hip — the three kernels of the long-window path (tile_lw.hpp): split, rows, merge.
// Its own translation unit (built with -fno-slp-vectorize like the other tile kernels, airwave_amd/build.py).
#include "kernels.hpp"
#include "gpu_ctx.hpp"
#include "tile_lw.hpp"
#include "tile_lw16.hpp"
#include "lw_split_inst.hpp"

namespace awk {

// Row pairs are pinned to XCDs (blockIdx % 8 labels the XCD): XCD x walks the row pairs x, x + 8, ... one after the other
// and, within a row pair, every (stream, window); its 32 workgroups therefore share one 512 KB table slice at a time.
template <int NP, bool REAL_LAST>
__global__ void __launch_bounds__(kThreads) aw_lw_rows_kernel(LwParams p, long long n_sw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GpuCtx ctx{reinterpret_cast<cf *>(smem), nullptr};
    const int g = (int)gridDim.x, b = (int)blockIdx.x;
    const int xcd = b % 8, slot = b / 8;
    const int per_xcd_wg = (g - xcd + 7) / 8;
    lw_rows_tiles<GpuCtx, NP, REAL_LAST>(ctx, p, (long long)slot, (long long)per_xcd_wg, n_sw, xcd, 8);
}

// PB = 1: one exchange buffer, two workgroups per CU
template <int NP, bool REAL_LAST>
__global__ void __launch_bounds__(kThreads, 4) aw_lw_rows1_kernel(LwParams p, long long n_sw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GpuCtx ctx{reinterpret_cast<cf *>(smem), nullptr};
    const int g = (int)gridDim.x, b = (int)blockIdx.x;
    const int xcd = b % 8, slot = b / 8;
    const int per_xcd_wg = (g - xcd + 7) / 8;
    lw_rows_tiles<GpuCtx, NP, REAL_LAST, 1>(ctx, p, (long long)slot, (long long)per_xcd_wg, n_sw, xcd, 8);
}

// 16 points of one row per thread, 256-thread workgroups (tile_lw16.hpp); the same XCD pinning of row pairs
#ifndef AW_R16_MIN_WAVES
#define AW_R16_MIN_WAVES 3
#endif
template <int NP, bool REAL_LAST>
__global__ void __launch_bounds__(kR16Threads, AW_R16_MIN_WAVES) aw_lw_rows16_kernel(LwParams p, long long n_sw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GpuCtx ctx{reinterpret_cast<cf *>(smem), nullptr};
    const int g = (int)gridDim.x, b = (int)blockIdx.x;
    const int xcd = b % 8, slot = b / 8;
    const int per_xcd_wg = (g - xcd + 7) / 8;
    lw_rows16_tiles<GpuCtx, NP, REAL_LAST>(ctx, p, (long long)slot, (long long)per_xcd_wg, n_sw, xcd, 8);
}

template <int RA>
__global__ void __launch_bounds__(kThreads, 4) aw_lw_merge_kernel(LwParams p, long long n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GpuCtx ctx{reinterpret_cast<cf *>(smem), nullptr};
    lw_merge_tiles<GpuCtx, RA>(ctx, p, (long long)blockIdx.x, (long long)gridDim.x, n_tiles);
}

template <int RA> constexpr int lw_merge_lds_bytes() { return lw_merge_lds_elems<RA>() * (int)sizeof(cf); }
#define AW_LW_FOR_RA(X) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(12) X(14) X(15) X(16)

constexpr int kLwRows1LdsBytes = lw_rows_lds_elems<1>() * (int)sizeof(cf);

hipError_t prepare_lw_kernels() {
    hipError_t e = hipSuccess;
#define AW_SET(RA) if (e == hipSuccess) e = lw_split_prepare<RA>();
    AW_LW_FOR_RA(AW_SET)
#undef AW_SET
#define AW_SET(NP)                                                                                     \
    if (e == hipSuccess)                                                                               \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_lw_rows_kernel<NP, false>),         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);                \
    if (e == hipSuccess)                                                                               \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_lw_rows_kernel<NP, true>),          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    AW_SET(1) AW_SET(2) AW_SET(3) AW_SET(4)
#undef AW_SET
#define AW_SET(NP)                                                                                     \
    if (e == hipSuccess)                                                                               \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_lw_rows1_kernel<NP, false>),        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLwRows1LdsBytes);         \
    if (e == hipSuccess)                                                                               \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_lw_rows1_kernel<NP, true>),         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLwRows1LdsBytes);
    AW_SET(1) AW_SET(2) AW_SET(3) AW_SET(4) AW_SET(5) AW_SET(6) AW_SET(7) AW_SET(8)
#undef AW_SET
#define AW_SET(NP)                                                                                     \
    if (e == hipSuccess)                                                                               \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_lw_rows16_kernel<NP, false>),       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kR16LdsBytes);             \
    if (e == hipSuccess)                                                                               \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_lw_rows16_kernel<NP, true>),        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kR16LdsBytes);
    AW_SET(1) AW_SET(2) AW_SET(3) AW_SET(4) AW_SET(5) AW_SET(6) AW_SET(7) AW_SET(8)
#undef AW_SET
#define AW_SET(RA) if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_lw_merge_kernel<RA>), hipFuncAttributeMaxDynamicSharedMemorySize, lw_merge_lds_bytes<RA>());
    AW_LW_FOR_RA(AW_SET)
#undef AW_SET
    return e;
}

static unsigned lw_grid(long long n_tiles, const LwParams &p, int wgs_per_cu) {
    long long wgs = (long long)(p.persistent_wgs >= 8 ? p.persistent_wgs : 256) * wgs_per_cu;
    return (unsigned)(n_tiles < wgs ? n_tiles : wgs);
}

hipError_t launch_lw_split(const LwParams &p, int n_streams, hipStream_t stream, StageTimer *tm) {
    if (p.n_channels < 1 || p.n_channels > 16) return hipErrorInvalidValue;
    const int ra = p.R / 8;
    if (p.R % 8 != 0 || !lw_ra_ok(ra)) return hipErrorInvalidValue;
    const bool wide = p.n_channels > 8;              // 9-16 channels: both channel halves of a frame in one wave, 32 frames per tile
    const long long n_tiles = (long long)n_streams * p.n_windows * (wide ? kLwChunksW : kLwChunks);
    if (n_tiles <= 0) return hipSuccess;
    if (n_tiles > 0x7fffffffLL || (wide && !p.tail)) return hipErrorInvalidValue;
    const int cs = wide ? p.n_channels - 8 : p.n_channels;
    const dim3 grid(lw_grid(n_tiles, p, ra > 8 ? 1 : 2));
    if (tm) tm->begin();
    hipError_t e = hipErrorInvalidValue;
    switch (ra) {
#define AW_CASE(RA) case RA: e = lw_split_launch<RA>(p, wide, cs, grid, stream, n_tiles); break;
        AW_LW_FOR_RA(AW_CASE)
#undef AW_CASE
        default: break;
    }
    if (tm) tm->end(wide ? "aw_lw_split_wide_kernel" : "aw_lw_split_kernel");
    return e;
}

hipError_t launch_lw_rows(const LwParams &p, int n_streams, hipStream_t stream, StageTimer *tm) {
    const long long n_sw = (long long)n_streams * p.n_windows;
    const long long n_tiles = n_sw * (p.R / 2);
    if (n_tiles <= 0) return hipSuccess;
    if (n_tiles > 0x7fffffffLL) return hipErrorInvalidValue;
    if (p.rows_form == 16) {
        if (!p.tab16 || !p.tw2) return hipErrorInvalidValue;
        const int per_cu = p.rows16_wgs >= 1 && p.rows16_wgs <= 4 ? p.rows16_wgs : 3;
        unsigned grid16 = lw_grid((n_tiles + 7) / 8 * 8, p, per_cu) / 8 * 8;
        if (grid16 < 8) grid16 = 8;
        if (tm) tm->begin();
        const bool real16 = p.real_last != 0;
        switch (p.n_pairs) {
#define AW_CASE(NP)                                                                                                       \
        case NP:                                                                                                          \
            if (real16) hipLaunchKernelGGL((aw_lw_rows16_kernel<NP, true>), dim3(grid16), dim3(kR16Threads), kR16LdsBytes, stream, p, n_sw);   \
            else hipLaunchKernelGGL((aw_lw_rows16_kernel<NP, false>), dim3(grid16), dim3(kR16Threads), kR16LdsBytes, stream, p, n_sw);         \
            break;
        AW_CASE(1) AW_CASE(2) AW_CASE(3) AW_CASE(4) AW_CASE(5) AW_CASE(6) AW_CASE(7) AW_CASE(8)
#undef AW_CASE
        default: return hipErrorInvalidValue;
        }
        if (tm) tm->end("aw_lw_rows_kernel");
        return hipGetLastError();
    }
    // 8 XCD groups: a grid that is a multiple of 8 (every group has the same number of workgroups), at least 8
    const bool one = p.rows_pairs_per_batch == 1 || p.n_pairs > 4;          // the two-pairs-per-batch form exists for up to four pairs
    unsigned grid = lw_grid((n_tiles + 7) / 8 * 8, p, one ? 2 : 1) / 8 * 8;
    if (grid < 8) grid = 8;
    if (tm) tm->begin();
    const bool real = p.real_last != 0;
    if (one) switch (p.n_pairs) {
#define AW_CASE(NP)                                                                                                       \
        case NP:                                                                                                          \
            if (real) hipLaunchKernelGGL((aw_lw_rows1_kernel<NP, true>), dim3(grid), dim3(kThreads), kLwRows1LdsBytes, stream, p, n_sw);   \
            else hipLaunchKernelGGL((aw_lw_rows1_kernel<NP, false>), dim3(grid), dim3(kThreads), kLwRows1LdsBytes, stream, p, n_sw);       \
            break;
        AW_CASE(1) AW_CASE(2) AW_CASE(3) AW_CASE(4) AW_CASE(5) AW_CASE(6) AW_CASE(7) AW_CASE(8)
#undef AW_CASE
        default: return hipErrorInvalidValue;
    }
    else switch (p.n_pairs) {
#define AW_CASE(NP)                                                                                                       \
        case NP:                                                                                                          \
            if (real) hipLaunchKernelGGL((aw_lw_rows_kernel<NP, true>), dim3(grid), dim3(kThreads), kLdsBytes, stream, p, n_sw);   \
            else hipLaunchKernelGGL((aw_lw_rows_kernel<NP, false>), dim3(grid), dim3(kThreads), kLdsBytes, stream, p, n_sw);       \
            break;
        AW_CASE(1) AW_CASE(2) AW_CASE(3) AW_CASE(4)
#undef AW_CASE
        default: return hipErrorInvalidValue;
    }
    if (tm) tm->end("aw_lw_rows_kernel");
    return hipGetLastError();
}

hipError_t launch_lw_merge(const LwParams &p, int n_streams, hipStream_t stream, StageTimer *tm) {
    const long long n_tiles = (long long)n_streams * p.n_windows * kLwChunks;
    if (n_tiles <= 0) return hipSuccess;
    if (n_tiles > 0x7fffffffLL) return hipErrorInvalidValue;
    if (tm) tm->begin();
    switch (p.R / 8) {
#define AW_CASE(RA) case RA: hipLaunchKernelGGL((aw_lw_merge_kernel<RA>), dim3(lw_grid(n_tiles, p, 2)), dim3(kThreads), lw_merge_lds_bytes<RA>(), stream, p, n_tiles); break;
        AW_LW_FOR_RA(AW_CASE)
#undef AW_CASE
        default: return hipErrorInvalidValue;
    }
    if (tm) tm->end("aw_lw_merge_kernel");
    return hipGetLastError();
}

}  // namespace awk
