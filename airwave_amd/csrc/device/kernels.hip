// kernels.hip — gfx950 kernels of the batch HRIR spatializer.
//
// The tile body lives in tile_ols.hpp (shared with the CPU emulation harness); this file adds the
// GPU execution context (LDS, barriers), the XCD-aware workgroup -> tile mapping and the small
// utility kernels (history carry, synthetic fill, planar<->interleaved).
#include <cstring>
#include "kernels.hpp"
#include "gpu_ctx.hpp"
#include "ols2_kernel.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace awk {

// Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the XCD group).  Give
// each XCD a contiguous run of tiles so that consecutive tiles of a stream, whose input windows
// overlap by N - hop frames, share one L2 (MI355X_MICROARCH "Workgroup dispatch"; speed only).
__device__ __forceinline__ long long xcd_remap(long long bid, long long nwg) {
    const long long q = nwg / 8, r = nwg % 8;
    const long long xcd = bid % 8, idx = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// INTERIOR = true : tiles [tile_lo, tile_hi) of every stream (window inside the call's input)
// INTERIOR = false: the remaining boundary tiles (history at the start, zero fill at the end)
template <int CS, int NP, bool INTERIOR, bool ACC = false>
__global__ void __launch_bounds__(kThreads) aw_fused_ols_kernel(TileParams p, long long n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GpuCtx ctx{reinterpret_cast<cf *>(smem), p.dbg ? p.dbg + (long long)blockIdx.x * kStamps : nullptr};
    ctx.stamp_thread_ = p.stagger;          // diagnostic builds: which thread's wave is recorded
    // Persistent workgroups, XCD-aware: workgroups are dealt round-robin over the 8 XCDs, so
    // blockIdx % 8 labels the XCD group.  Each group owns a contiguous eighth of the tile list and
    // its workgroups walk it interleaved, so the ~32 tiles in flight on one XCD are consecutive
    // tiles of a stream: their overlapping input windows meet in that XCD's L2 (speed only).
    const long long g = gridDim.x, b = blockIdx.x;
    const long long xcd = b % 8, slot = b / 8;
    const long long per_xcd_wg = (g - xcd + 7) / 8;                       // workgroups in this group
    const long long q = n_tiles / 8, r = n_tiles % 8;
    const long long lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const long long hi = lo + (xcd < r ? q + 1 : q);
    tiles_fused_ols<GpuCtx, CS, NP, INTERIOR, ACC>(ctx, p, lo + slot, per_xcd_wg, hi);
}


// Windows [tile_lo, tile_hi) of every stream lie inside the call's input (INTERIOR), the others touch the
// history or the zero page.  p.tile_lo/hi carry the window range here.
#ifndef AW_FWD_RUNS
#define AW_FWD_RUNS 0
#endif
// MODE 1: windows [tile_lo, tile_hi) (interior); MODE 2: windows [head_lo, tile_lo) (head: history + input);
// MODE 0: the rest, [0, head_lo) and [tile_hi, n_windows).  Persistent, XCD-aware like the fused kernels: each XCD group
// walks a contiguous eighth of the window list, so windows that overlap by half meet in one L2.
template <int CS, int MODE>
__global__ void __launch_bounds__(kThreads) aw_part_forward_kernel(TileParams p, long long n_ids, int head_lo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GpuCtx ctx{reinterpret_cast<cf *>(smem), nullptr};
    const int n_windows = p.n_blocks + p.partitions - 1;
    const int per = MODE == 1 ? p.tile_hi - p.tile_lo : MODE == 2 ? p.tile_lo - head_lo : n_windows - (p.tile_hi - head_lo);
    const int w0 = MODE == 1 ? p.tile_lo : head_lo, skip = p.tile_hi - head_lo;
    const long long g = gridDim.x, b = blockIdx.x;
    const long long xcd = b % 8, slot = b / 8;
    const long long per_xcd_wg = (g - xcd + 7) / 8;
    const long long q = n_ids / 8, r = n_ids % 8;
    const long long lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const long long hi = lo + (xcd < r ? q + 1 : q);
#if AW_FWD_RUNS
    // each workgroup walks a contiguous run of windows: consecutive windows of a stream overlap by half, and the half a
    // workgroup has just read is the likeliest to still be in its XCD's L2
    const long long run = (hi - lo + per_xcd_wg - 1) / per_xcd_wg;
    const long long first = lo + slot * run;
    tiles_part_forward<GpuCtx, CS, MODE>(ctx, p, first, 1, first + run < hi ? first + run : hi, per, w0, skip);
#else
    tiles_part_forward<GpuCtx, CS, MODE>(ctx, p, lo + slot, per_xcd_wg, hi, per, w0, skip);
#endif
}

// One-pair form: workgroup id -> (stream, window, pair), pairs innermost; kInvLdsBytes of LDS, two workgroups per CU.
template <int CS, int MODE>
__global__ void __launch_bounds__(kThreads, 4) aw_part_forward1_kernel(TileParams p, long long nwg, int head_lo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GpuCtx ctx{reinterpret_cast<cf *>(smem), nullptr};
    const long long id = xcd_remap((long long)blockIdx.x, nwg);
    const long long wid = id / p.n_pairs;
    const int pair = (int)(id - wid * p.n_pairs);
    const int n_windows = p.n_blocks + p.partitions - 1;
    const int per = MODE == 1 ? p.tile_hi - p.tile_lo : MODE == 2 ? p.tile_lo - head_lo : n_windows - (p.tile_hi - head_lo);
    const long long stream = wid / per;
    int w = (int)(wid - stream * per);
    if (MODE == 1) w += p.tile_lo;
    else if (MODE == 2) w += head_lo;
    else if (w >= head_lo) w += p.tile_hi - head_lo;
    tile_part_forward1<GpuCtx, CS, MODE>(ctx, p, stream, w, pair);
}

// grid = (N / kCmacThreads, block groups, streams): one thread per bin of kCmacBlocks consecutive blocks
__global__ void __launch_bounds__(kCmacThreads) aw_part_cmac_kernel(TileParams p) {
    part_cmac_bin(p, (long long)blockIdx.z, (int)blockIdx.y * kCmacBlocks, (int)(blockIdx.x * kCmacThreads + threadIdx.x));
}

__global__ void __launch_bounds__(kThreads) aw_part_inverse_kernel(TileParams p, long long nwg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GpuCtx ctx{reinterpret_cast<cf *>(smem), nullptr};
    const long long id = xcd_remap((long long)blockIdx.x, nwg);
    tile_part_inverse<GpuCtx>(ctx, p, id / p.n_blocks, (int)(id % p.n_blocks));
}

// (real channels, batches): 2C pseudo-channels in batches of four
#define AW_FOR_EACH_VEC2(X) X(1, 1) X(2, 1) X(3, 2) X(5, 3) X(7, 4)      // 4, 6 and 8 channels live in ols2_even_kernels.hip (SLP on)

const char *fused_ols2_kernel_name(int C) {
    switch (C) {
        case 1: return "aw_fused_ols2_kernel<1, 1, true>";
        case 2: return "aw_fused_ols2_kernel<2, 1, true>";
        case 3: return "aw_fused_ols2_kernel<3, 2, true>";
        case 4: return "aw_fused_ols2_kernel<4, 2, true>";
        case 5: return "aw_fused_ols2_kernel<5, 3, true>";
        case 6: return "aw_fused_ols2_kernel<6, 3, true>";
        case 7: return "aw_fused_ols2_kernel<7, 4, true>";
        case 8: return "aw_fused_ols2_kernel<8, 4, true>";
        default: return "aw_fused_ols2_kernel<0, 0, false>";
    }
}
static bool has_vec2_variant(int C) { return C >= 1 && C <= 8; }
static bool ols2_slp_layout(int C) { return C == 4 || C == 6 || C == 8; }      // built in ols2_even_kernels.hip

// Kernel variants.  Vectorised interior kernels <CS, NP, true> exist for 2-16 channels (frames that are not whole float4s are
// loaded 16 B per lane at dword alignment, zero tables cancel the lanes that run into the next frame; 9-14 channels in one
// pass over two eight-channel groups, 15-16 in two passes); the boundary tiles of every layout, and mono, run the
// generic-addressing kernels <0, NP, false> (NP = compile-time pair count 1..4; NP = 0 loops over batches of two pairs at
// run time).  The (12, 0) (14, 0) (16, 0) entries are the round-1 run-time-loop kernels (AW_WIDE_TWO_PASS=0) and the wide
// layouts' forward kernels of the partitioned path.
#define AW_FOR_EACH_VEC(X) X(2, 1) X(3, 2) X(4, 2) X(5, 3) X(6, 3) X(7, 4) X(8, 4) X(12, 0) X(14, 0) X(16, 0)
#define AW_FOR_EACH_GEN(X) X(1) X(2) X(3) X(4) X(0)
// wide layouts (interior tiles): (channels, pairs of the first pass, pairs of the accumulating second pass)
#define AW_FOR_EACH_WIDE(X) X(10, 4, 1) X(12, 4, 2) X(14, 4, 3) X(15, 4, 4) X(16, 4, 4)
// 10-14 channels in one pass: (channels, pairs).  16 channels stay on two passes (one pass: 292 B of scratch per thread,
// 13.5 against 13.9 G frames/s)
#define AW_FOR_EACH_WIDE1(X) X(9, 5) X(10, 5) X(11, 6) X(12, 6) X(13, 7) X(14, 7)
// boundary tiles of the common layouts keep whole-frame vector loads (history / zero-page selects per frame)
#define AW_FOR_EACH_BVEC(X) X(2, 1) X(4, 2) X(8, 4)

hipError_t prepare_kernels(LaunchCfg *cfg) {
    hipError_t e = hipSuccess;
    if (cfg) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) {
            cfg->cus = cus;
            cfg->persistent_wgs = cus;            // one resident workgroup per CU (152 KB LDS each)
        }
        int &g_persistent_wgs = cfg->persistent_wgs;
        if (const char *e2 = getenv("AW_PERSISTENT_WGS")) g_persistent_wgs = atoi(e2) > 0 ? atoi(e2) : g_persistent_wgs;
        // the persistent kernels deal tiles to 8 XCD groups (blockIdx % 8): a grid below 8 workgroups with more tiles than
        // workgroups would leave groups without a workgroup and their tiles uncomputed
        if (g_persistent_wgs < 8) g_persistent_wgs = 8;
        if (const char *e3 = getenv("AW_WIDE_TWO_PASS")) cfg->wide_two_pass = atoi(e3) != 0;      // A/B: 1 = two passes, 0 = run-time loop
        cfg->debug_occupancy = getenv("AW_DEBUG_OCCUPANCY") != nullptr;
        if (const char *e5 = getenv("AW_STAMP_THREAD")) cfg->stamp_thread = atoi(e5);
        if (const char *e6 = getenv("AW_EQ_EAR_SPLIT")) cfg->eq_ear_split = atoi(e6);
        if (const char *e7 = getenv("AW_LW_ROWS_PB")) cfg->lw_rows_pb = atoi(e7) == 2 ? 2 : 1;
        if (const char *e8 = getenv("AW_HOP_ALIGN")) cfg->hop_align = atoi(e8);
        if (const char *e9 = getenv("AW_LW_ROWS_FORM")) cfg->lw_rows_form = atoi(e9) == 8 ? 8 : 16;
        if (const char *e10 = getenv("AW_LW_ROWS16_WGS")) cfg->lw_rows16_wgs = atoi(e10) >= 1 && atoi(e10) <= 4 ? atoi(e10) : cfg->lw_rows16_wgs;
        if (const char *e12 = getenv("AW_LW_TABLES")) cfg->lw_tables_on_gpu = std::strcmp(e12, "host") == 0 ? 0 : 1;
        if (const char *e13 = getenv("AW_OLA_MIN_BLOCKS")) cfg->ola_min_blocks_per_wg = atoi(e13) >= 0 ? atoi(e13) : cfg->ola_min_blocks_per_wg;
        if (const char *e14 = getenv("AW_HOST_OUT_ASYNC")) cfg->host_out_async = atoi(e14) != 0;
        if (const char *e11 = getenv("AW_HOST_CHUNK_MB")) cfg->host_chunk_mb = atoi(e11) >= 1 ? atoi(e11) : cfg->host_chunk_mb;
    }
#define AW_SET_VEC(CS, NP)                                                                           \
    if (e == hipSuccess)                                                                             \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_fused_ols_kernel<CS, NP, true>),  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
#define AW_SET_GEN(NP)                                                                               \
    if (e == hipSuccess)                                                                             \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_fused_ols_kernel<0, NP, false>),  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
#define AW_SET_BVEC(CS, NP)                                                                          \
    if (e == hipSuccess)                                                                             \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_fused_ols_kernel<CS, NP, false>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    AW_FOR_EACH_VEC(AW_SET_VEC)
#define AW_SET_WIDE(CS, NPA, NPB)                                                                              \
    if (e == hipSuccess)                                                                                       \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_fused_ols_kernel<CS, NPA, true, false>),    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);                        \
    if (e == hipSuccess)                                                                                       \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_fused_ols_kernel<CS, NPB, true, true>),     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    AW_FOR_EACH_WIDE(AW_SET_WIDE)
#undef AW_SET_WIDE
    AW_FOR_EACH_WIDE1(AW_SET_VEC)
#define AW_SET_GENACC(NP)                                                                                      \
    if (e == hipSuccess)                                                                                       \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_fused_ols_kernel<0, NP, false, true>),      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    AW_SET_GENACC(1) AW_SET_GENACC(2) AW_SET_GENACC(3) AW_SET_GENACC(4)
#undef AW_SET_GENACC
    AW_FOR_EACH_GEN(AW_SET_GEN)
    AW_FOR_EACH_BVEC(AW_SET_BVEC)
#define AW_SET_VEC2(CS, NB)                                                                          \
    if (e == hipSuccess)                                                                             \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_fused_ols2_kernel<CS, NB, true>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    AW_FOR_EACH_VEC2(AW_SET_VEC2)
#undef AW_SET_VEC2
#define AW_SET_BVEC2(CS, NB)                                                                          \
    if (e == hipSuccess)                                                                              \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_fused_ols2_kernel<CS, NB, false>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    AW_FOR_EACH_VEC2(AW_SET_BVEC2)
#undef AW_SET_BVEC2
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_fused_ols2_kernel<0, 0, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    if (e == hipSuccess) e = prepare_ols2_even();
#undef AW_SET_BVEC
#undef AW_SET_VEC
#undef AW_SET_GEN
#define AW_SET_FWD(CS, NP)                                                                              \
    if (e == hipSuccess)                                                                                \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_part_forward_kernel<CS, 1>),         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);                 \
    if (e == hipSuccess)                                                                                \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_part_forward_kernel<CS, 2>),         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    AW_FOR_EACH_VEC(AW_SET_FWD)
#undef AW_SET_FWD
#define AW_SET_FWD1(CS, NP)                                                                             \
    if (e == hipSuccess)                                                                                \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_part_forward1_kernel<CS, 1>),        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kInvLdsBytes);              \
    if (e == hipSuccess)                                                                                \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_part_forward1_kernel<CS, 2>),        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kInvLdsBytes);
    AW_FOR_EACH_VEC(AW_SET_FWD1)
#undef AW_SET_FWD1
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_part_forward1_kernel<0, 0>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kInvLdsBytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_part_forward_kernel<0, 0>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_part_inverse_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kInvLdsBytes);
    return e;
}

static bool has_vec_variant(int C) { return (C >= 2 && C <= 8) || C == 12 || C == 14 || C == 16; }
static bool has_fused_vec_variant(int C) { return has_vec_variant(C) || (C >= 9 && C <= 15); }       // 9, 10, 11, 13, 15 channels: the fused kernels only

const char *fused_ols_kernel_name(int C) {
    switch (C) {
        case 2: return "aw_fused_ols_kernel<2, 1, true>";
        case 3: return "aw_fused_ols_kernel<3, 2, true>";
        case 4: return "aw_fused_ols_kernel<4, 2, true>";
        case 5: return "aw_fused_ols_kernel<5, 3, true>";
        case 6: return "aw_fused_ols_kernel<6, 3, true>";
        case 7: return "aw_fused_ols_kernel<7, 4, true>";
        case 8: return "aw_fused_ols_kernel<8, 4, true>";
        case 9: return "aw_fused_ols_kernel<9, 5, true>";
        case 10: return "aw_fused_ols_kernel<10, 5, true>";
        case 11: return "aw_fused_ols_kernel<11, 6, true>";
        case 12: return "aw_fused_ols_kernel<12, 6, true>";
        case 13: return "aw_fused_ols_kernel<13, 7, true>";
        case 14: return "aw_fused_ols_kernel<14, 7, true>";
        case 15: return "aw_fused_ols_kernel<15, 4, true> + <15, 4, true, accumulate>";
        case 16: return "aw_fused_ols_kernel<16, 4, true> + <16, 4, true, accumulate>";
        default: return "aw_fused_ols_kernel<0, NP, false>";
    }
}

static dim3 persistent_grid(long long n_tiles, const TileParams &p) {
    const long long wgs = p.persistent_wgs >= 8 ? p.persistent_wgs : 256;
    return dim3((unsigned)(n_tiles < wgs ? n_tiles : wgs));
}

static void launch_vec(const TileParams &p, long long n_tiles, hipStream_t stream) {
    const dim3 grid = persistent_grid(n_tiles, p), block(kThreads);
    if (p.wide_two_pass == 2) {      // 10/12/14 channels in one pass over two eight-channel groups (the default)
        switch (p.n_channels) {
#define AW_CASE(CS, NP) case CS: hipLaunchKernelGGL((aw_fused_ols_kernel<CS, NP, true>), grid, block, kLdsBytes, stream, p, n_tiles); return;
            AW_FOR_EACH_WIDE1(AW_CASE)
#undef AW_CASE
            default: break;
        }
    }
    if (p.wide_two_pass || p.n_channels == 10) {
        // second pass: input and tables shifted by the first pass's 4 pairs (8 channels), result added to the output
        TileParams q = p;
        q.in = p.in + 8;
        q.tab = p.tab + 4 * (long long)kN;
        switch (p.n_channels) {
#define AW_CASE(CS, NPA, NPB) case CS:                                                                                               \
            hipLaunchKernelGGL((aw_fused_ols_kernel<CS, NPA, true, false>), grid, block, kLdsBytes, stream, p, n_tiles);           \
            hipLaunchKernelGGL((aw_fused_ols_kernel<CS, NPB, true, true>), grid, block, kLdsBytes, stream, q, n_tiles);            \
            return;
            AW_FOR_EACH_WIDE(AW_CASE)
#undef AW_CASE
            default: break;
        }
    }
    switch (p.n_channels) {
#define AW_CASE(CS, NP) case CS: hipLaunchKernelGGL((aw_fused_ols_kernel<CS, NP, true>), grid, block, kLdsBytes, stream, p, n_tiles); break;
        AW_FOR_EACH_VEC(AW_CASE)
#undef AW_CASE
        default: break;
    }
}

static void launch_gen(const TileParams &p, long long n_tiles, hipStream_t stream) {
    const dim3 grid = persistent_grid(n_tiles, p), block(kThreads);
    switch (p.n_channels) {
#define AW_CASE(CS, NP) case CS: hipLaunchKernelGGL((aw_fused_ols_kernel<CS, NP, false>), grid, block, kLdsBytes, stream, p, n_tiles); return;
        AW_FOR_EACH_BVEC(AW_CASE)
#undef AW_CASE
        default: break;
    }
    if (p.n_pairs > 4 && p.n_pairs <= 8 && p.wide_two_pass) {
        // 9-16 channels: compile-time 4-pair pass, then an accumulating compile-time pass over the remaining pairs
        // (input, history and tables shifted by 8 channels; ch_base keeps the padding-channel test right)
        TileParams q = p;
        q.in = p.in + 8; q.hist = p.hist + 8; q.tab = p.tab + 4 * (long long)kN; q.ch_base = 8;
        hipLaunchKernelGGL((aw_fused_ols_kernel<0, 4, false, false>), grid, block, kLdsBytes, stream, p, n_tiles);
        switch (p.n_pairs - 4) {
            case 1: hipLaunchKernelGGL((aw_fused_ols_kernel<0, 1, false, true>), grid, block, kLdsBytes, stream, q, n_tiles); break;
            case 2: hipLaunchKernelGGL((aw_fused_ols_kernel<0, 2, false, true>), grid, block, kLdsBytes, stream, q, n_tiles); break;
            case 3: hipLaunchKernelGGL((aw_fused_ols_kernel<0, 3, false, true>), grid, block, kLdsBytes, stream, q, n_tiles); break;
            default: hipLaunchKernelGGL((aw_fused_ols_kernel<0, 4, false, true>), grid, block, kLdsBytes, stream, q, n_tiles); break;
        }
        return;
    }
    const int np = p.n_pairs <= 4 ? p.n_pairs : 0;
    switch (np) {
#define AW_CASE(NP) case NP: hipLaunchKernelGGL((aw_fused_ols_kernel<0, NP, false>), grid, block, kLdsBytes, stream, p, n_tiles); break;
        AW_FOR_EACH_GEN(AW_CASE)
#undef AW_CASE
        default: break;
    }
}

// Interior tiles (window entirely inside the call's input) and boundary tiles are separate launches.
hipError_t launch_fused_ols(const TileParams &p_in, int n_streams, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1,
                            long long *dominant_tiles) {
    TileParams p = p_in;
    // tile i is interior iff  i*hop - hist_len >= 0  and  i*hop - hist_len + N <= frames
    long long lo = (p.hist_len + p.hop - 1) / p.hop;
    // layouts whose frames are not whole float4s read up to 3 floats past a frame: keep one frame of slack
    const long long usable = p.frames - ((p.n_channels % 4 != 0 && p.n_channels != 2) ? 1 : 0);
    long long hi = (usable - kN + p.hist_len) >= 0 ? (usable - kN + p.hist_len) / p.hop + 1 : 0;
    if (hi > p.tiles_per_stream) hi = p.tiles_per_stream;
    if (hi < lo) hi = lo;
    if (lo > p.tiles_per_stream) { lo = p.tiles_per_stream; hi = lo; }
    if (!has_fused_vec_variant(p.n_channels) || (p.wide_two_pass != 2 && p.n_channels > 8 && (p.n_channels & 1))) { lo = 0; hi = 0; }  // everything through the generic kernels
    p.tile_lo = (int)lo; p.tile_hi = (int)hi;
    const long long n_int = (long long)n_streams * (hi - lo);
    const long long n_bnd = (long long)n_streams * (p.tiles_per_stream - (hi - lo));
    if (n_int > 0x7fffffffLL || n_bnd > 0x7fffffffLL) return hipErrorInvalidValue;
    // the events bracket the DOMINANT launch only (interior tiles when the layout has a vector variant)
    const bool dom_int = n_int > 0;
    if (dominant_tiles) *dominant_tiles = dom_int ? n_int : n_bnd;
    if (ev0 && dom_int) (void)hipEventRecord(ev0, stream);
    if (n_int > 0) launch_vec(p, n_int, stream);
    if (ev1 && dom_int) (void)hipEventRecord(ev1, stream);
    if (n_bnd > 0) {
        TileParams pb = p;
        pb.dbg = nullptr;                 // diagnostic stamps describe the interior launch only
        if (ev0 && !dom_int) (void)hipEventRecord(ev0, stream);
        launch_gen(pb, n_bnd, stream);
        if (ev1 && !dom_int) (void)hipEventRecord(ev1, stream);
    }
    return hipGetLastError();
}

// 16384-frame windows: same interior / boundary split, in real frames.
hipError_t launch_fused_ols2(const TileParams &p_in, int n_streams, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1,
                             long long *dominant_tiles) {
    TileParams p = p_in;
    long long lo = (p.hist_len + p.hop - 1) / p.hop;
    // the last batch of a pseudo-frame reads up to 2 floats past it (2C not a multiple of 4): keep that much input after the window
    const long long usable = p.frames - (((2 * p.n_channels) % 4 != 0) ? (p.n_channels == 1 ? 2 : 1) : 0);
    long long hi = (usable - kN2 + p.hist_len) >= 0 ? (usable - kN2 + p.hist_len) / p.hop + 1 : 0;
    if (hi > p.tiles_per_stream) hi = p.tiles_per_stream;
    if (hi < lo) hi = lo;
    if (lo > p.tiles_per_stream) { lo = p.tiles_per_stream; hi = lo; }
    if (!has_vec2_variant(p.n_channels)) { lo = 0; hi = 0; }
    p.tile_lo = (int)lo; p.tile_hi = (int)hi;
    const long long n_int = (long long)n_streams * (hi - lo);
    const long long n_bnd = (long long)n_streams * (p.tiles_per_stream - (hi - lo));
    if (n_int > 0x7fffffffLL || n_bnd > 0x7fffffffLL) return hipErrorInvalidValue;
    const bool dom_int = n_int > 0;
    if (dominant_tiles) *dominant_tiles = dom_int ? n_int : n_bnd;
    if (n_int > 0) {
        const dim3 grid = persistent_grid(n_int, p), block(kThreads);
        if (ev0) (void)hipEventRecord(ev0, stream);
        if (ols2_slp_layout(p.n_channels)) launch_ols2_even(p, true, n_int, grid, stream);
        else switch (p.n_channels) {
#define AW_CASE(CS, NB) case CS: hipLaunchKernelGGL((aw_fused_ols2_kernel<CS, NB, true>), grid, block, kLdsBytes, stream, p, n_int); break;
            AW_FOR_EACH_VEC2(AW_CASE)
#undef AW_CASE
            default: break;
        }
        if (ev1) (void)hipEventRecord(ev1, stream);
    }
    if (n_bnd > 0) {
        p.dbg = nullptr;                 // diagnostic stamps describe the interior launch only
        if (ev0 && !dom_int) (void)hipEventRecord(ev0, stream);
        const dim3 grid = persistent_grid(n_bnd, p), block(kThreads);
        if (ols2_slp_layout(p.n_channels)) launch_ols2_even(p, false, n_bnd, grid, stream);
        else switch (p.n_channels) {      // compile-time channel and batch counts also for the boundary tiles (scalar loads)
#define AW_CASE(CS, NB) case CS: hipLaunchKernelGGL((aw_fused_ols2_kernel<CS, NB, false>), grid, block, kLdsBytes, stream, p, n_bnd); break;
            AW_FOR_EACH_VEC2(AW_CASE)
#undef AW_CASE
            default: hipLaunchKernelGGL((aw_fused_ols2_kernel<0, 0, false>), grid, block, kLdsBytes, stream, p, n_bnd); break;
        }
        if (ev1 && !dom_int) (void)hipEventRecord(ev1, stream);
    }
    return hipGetLastError();
}

hipError_t launch_part_forward(const TileParams &p_in, int n_streams, hipStream_t stream, StageTimer *tm) {
    TileParams p = p_in;
    const int n_windows = p.n_blocks + p.partitions - 1;
    if ((long long)n_streams * n_windows <= 0) return hipSuccess;
    // window w covers frames [(w - P) B, (w - P) B + N): interior iff it starts at >= 0 and ends inside the input
    // (one frame of slack for layouts whose frames are not whole float4s, see load_batch); head iff it starts in the
    // history (which reaches back P B frames: every window does) and still ends inside the input
    const long long usable = p.frames - ((p.n_channels % 4 != 0 && p.n_channels != 2) ? 1 : 0);
    long long lo = p.partitions;
    // windows [0, hi) end inside the input: (w - P) B + N <= usable  <=>  w <= floor((usable - N) / B) + P   (floor, not truncation:
    // a call shorter than one window leaves only windows that lie entirely in the history)
    const long long d = usable - kN;
    long long hi = (d >= 0 ? d / p.hop : -((-d + p.hop - 1) / p.hop)) + p.partitions + 1;
    long long head_lo = 0;
    if (hi > n_windows) hi = n_windows;
    if (lo > n_windows) lo = n_windows;
    if (hi < lo) { lo = hi; }                   // short call: even some head windows run past the end
    if (hi < 0) hi = 0;
    if (lo < 0) lo = 0;
    if (!has_vec_variant(p.n_channels)) { lo = 0; hi = 0; }
    p.tile_lo = (int)lo; p.tile_hi = (int)hi;
    const long long n_int = (long long)n_streams * (hi - lo), n_head = (long long)n_streams * (lo - head_lo);
    const long long n_bnd = (long long)n_streams * (n_windows - (hi - head_lo));
    if (n_int > 0x7fffffffLL || n_bnd > 0x7fffffffLL || n_head > 0x7fffffffLL) return hipErrorInvalidValue;
    if (p.fwd_one_pair) {             // one channel pair per workgroup, two workgroups per CU (the default)
        const long long np = p.n_pairs;
        if (n_int * np > 0x7fffffffLL || n_bnd * np > 0x7fffffffLL || n_head * np > 0x7fffffffLL) return hipErrorInvalidValue;
        if (n_int > 0) {
            if (tm) tm->begin();
            switch (p.n_channels) {
#define AW_CASE(CS, NP) case CS: hipLaunchKernelGGL((aw_part_forward1_kernel<CS, 1>), dim3((unsigned)(n_int * np)), dim3(kThreads), kInvLdsBytes, stream, p, n_int * np, (int)head_lo); break;
                AW_FOR_EACH_VEC(AW_CASE)
#undef AW_CASE
                default: break;
            }
            if (tm) tm->end("aw_part_forward1_kernel<CS, interior>");
        }
        if (n_head > 0) {
            if (tm) tm->begin();
            switch (p.n_channels) {
#define AW_CASE(CS, NP) case CS: hipLaunchKernelGGL((aw_part_forward1_kernel<CS, 2>), dim3((unsigned)(n_head * np)), dim3(kThreads), kInvLdsBytes, stream, p, n_head * np, (int)head_lo); break;
                AW_FOR_EACH_VEC(AW_CASE)
#undef AW_CASE
                default: break;
            }
            if (tm) tm->end("aw_part_forward1_kernel<CS, head>");
        }
        if (n_bnd > 0) {
            if (tm) tm->begin();
            hipLaunchKernelGGL((aw_part_forward1_kernel<0, 0>), dim3((unsigned)(n_bnd * np)), dim3(kThreads), kInvLdsBytes, stream, p, n_bnd * np, (int)head_lo);
            if (tm) tm->end("aw_part_forward1_kernel<generic>");
        }
        return hipGetLastError();
    }
    if (n_int > 0) {
        if (tm) tm->begin();
        switch (p.n_channels) {
#define AW_CASE(CS, NP) case CS: hipLaunchKernelGGL((aw_part_forward_kernel<CS, 1>), persistent_grid(n_int, p), dim3(kThreads), kLdsBytes, stream, p, n_int, (int)head_lo); break;
            AW_FOR_EACH_VEC(AW_CASE)
#undef AW_CASE
            default: break;
        }
        if (tm) tm->end("aw_part_forward_kernel<CS, interior>");
    }
    if (n_head > 0) {
        if (tm) tm->begin();
        switch (p.n_channels) {
#define AW_CASE(CS, NP) case CS: hipLaunchKernelGGL((aw_part_forward_kernel<CS, 2>), persistent_grid(n_head, p), dim3(kThreads), kLdsBytes, stream, p, n_head, (int)head_lo); break;
            AW_FOR_EACH_VEC(AW_CASE)
#undef AW_CASE
            default: break;
        }
        if (tm) tm->end("aw_part_forward_kernel<CS, head>");
    }
    if (n_bnd > 0) {
        if (tm) tm->begin();
        hipLaunchKernelGGL((aw_part_forward_kernel<0, 0>), persistent_grid(n_bnd, p), dim3(kThreads), kLdsBytes, stream, p, n_bnd, (int)head_lo);
        if (tm) tm->end("aw_part_forward_kernel<generic>");
    }
    return hipGetLastError();
}

hipError_t launch_part_cmac(const TileParams &p, int n_streams, hipStream_t stream, StageTimer *tm) {
    const int groups = (p.n_blocks + kCmacBlocks - 1) / kCmacBlocks;
    if (n_streams <= 0 || groups <= 0) return hipSuccess;
    if (groups > 65535 || n_streams > 65535) return hipErrorInvalidValue;
    if (tm) tm->begin();
    hipLaunchKernelGGL(aw_part_cmac_kernel, dim3(kN / kCmacThreads, (unsigned)groups, (unsigned)n_streams), dim3(kCmacThreads), 0,
                       stream, p);
    if (tm) tm->end("aw_part_cmac_kernel");
    return hipGetLastError();
}

hipError_t launch_part_inverse(const TileParams &p, int n_streams, hipStream_t stream, StageTimer *tm) {
    const long long nwg = (long long)n_streams * p.n_blocks;
    if (nwg <= 0) return hipSuccess;
    if (nwg > 0x7fffffffLL) return hipErrorInvalidValue;
    if (tm) tm->begin();
    hipLaunchKernelGGL(aw_part_inverse_kernel, dim3((unsigned)nwg), dim3(kThreads), kInvLdsBytes, stream, p, nwg);
    if (tm) tm->end("aw_part_inverse_kernel");
    return hipGetLastError();
}

// ---- history carry ---------------------------------------------------------------------------
__global__ void aw_hist_update_kernel(const float *__restrict__ in, const float *__restrict__ hist_old,
                                      float *__restrict__ hist_new, long long frames, int C, int hist_len) {
    const long long s = blockIdx.y;
    const long long per = (long long)hist_len * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long long)gridDim.x * blockDim.x) {
        const long long fi = i / C;
        const int ch = (int)(i - fi * C);
        const long long f = frames - hist_len + fi;
        const float v = f < 0 ? hist_old[s * per + (hist_len + f) * C + ch] : in[(s * frames + f) * C + ch];
        hist_new[s * per + i] = v;
    }
}

hipError_t launch_hist_update(const float *in, const float *hist_old, float *hist_new, long long frames,
                              int n_channels, int hist_len, int n_streams, hipStream_t stream) {
    if (hist_len <= 0 || n_streams <= 0) return hipSuccess;
    const long long per = (long long)hist_len * n_channels;
    unsigned gx = (unsigned)((per + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(aw_hist_update_kernel, dim3(gx, (unsigned)n_streams), dim3(256), 0, stream, in, hist_old,
                       hist_new, frames, n_channels, hist_len);
    return hipGetLastError();
}

// ---- synthetic input ------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void aw_synth_fill_kernel(float *__restrict__ dst, long long per_stream, unsigned long long seed,
                                     unsigned long long first_stream) {
    const unsigned long long s = blockIdx.y;
    const unsigned long long key0 = (seed + first_stream + s) * 0x9E3779B97F4A7C15ull;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per_stream; i += (long long)gridDim.x * blockDim.x) {
        const unsigned top = (unsigned)(splitmix64(key0 + (unsigned long long)i) >> 40);
        dst[s * per_stream + i] = (float)top * (1.0f / 16777216.0f) - 0.5f;
    }
}

hipError_t launch_synth_fill(float *dst, int n_streams, long long per_stream, unsigned long long seed,
                             unsigned long long first_stream, hipStream_t stream) {
    if (n_streams <= 0 || per_stream <= 0) return hipSuccess;
    long long gx = (per_stream + 255) / 256;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(aw_synth_fill_kernel, dim3((unsigned)gx, (unsigned)n_streams), dim3(256), 0, stream, dst,
                       per_stream, seed, first_stream);
    return hipGetLastError();
}

// ---- planar <-> interleaved stereo (plugin-shaped entry) ---------------------------------------
__global__ void aw_interleave2_kernel(const float *l, const float *r, float *dst, int frames) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < frames) { dst[2 * i] = l[i]; dst[2 * i + 1] = r[i]; }
}
__global__ void aw_deinterleave2_kernel(const float *src, float *l, float *r, int frames) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < frames) { l[i] = src[2 * i]; r[i] = src[2 * i + 1]; }
}
hipError_t launch_interleave2(const float *l, const float *r, float *dst, int frames, hipStream_t stream) {
    if (frames <= 0) return hipSuccess;
    hipLaunchKernelGGL(aw_interleave2_kernel, dim3((frames + 255) / 256), dim3(256), 0, stream, l, r, dst, frames);
    return hipGetLastError();
}
hipError_t launch_deinterleave2(const float *src, float *l, float *r, int frames, hipStream_t stream) {
    if (frames <= 0) return hipSuccess;
    hipLaunchKernelGGL(aw_deinterleave2_kernel, dim3((frames + 255) / 256), dim3(256), 0, stream, src, l, r, frames);
    return hipGetLastError();
}

}  // namespace awk
