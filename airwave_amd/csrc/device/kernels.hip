// kernels.hip — gfx950 kernels of the batch HRIR spatializer.
//
// The tile body lives in tile_ols.hpp (shared with the CPU emulation harness); this file adds the
// GPU execution context (LDS, barriers), the XCD-aware workgroup -> tile mapping and the small
// utility kernels (history carry, synthetic fill, planar<->interleaved).
#include "kernels.hpp"

#include <cstdlib>

namespace awk {

#ifndef AW_STAGGER_SLOTS
#define AW_STAGGER_SLOTS 0       // s_sleep argument (x64 cycles) for waves 4-7 after a barrier; 0 = off
#endif
#ifndef AW_LDS_NO_READ2
#define AW_LDS_NO_READ2 0
#endif
#ifndef AW_SCHED_FENCE
#define AW_SCHED_FENCE 0
#endif
#ifndef AW_STAMPS
#define AW_STAMPS 0
#endif

struct GpuCtx {
    cf *lds_;
    unsigned long long *dbg_;
    // Phase stamps (diagnostic build only: -DAW_STAMPS=1; never in the shipped kernel).  Every wave
    // reads the shader clock into SGPRs (uniform, no VGPRs, no branches in the timed code); thread 0
    // stores them once at the end into a buffer nothing else reads.
#if AW_STAMPS
    unsigned long long st_[kStamps];
#endif
    __device__ __forceinline__ void stamp(int i) {
#if AW_STAMPS
        unsigned long long tm;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm)::"memory");
        st_[i] = tm;
#else
        (void)i;
#endif
    }
    int stamp_thread_ = 0;
    __device__ __forceinline__ void flush_stamps() {
#if AW_STAMPS
        if ((int)threadIdx.x == stamp_thread_ && dbg_)
            for (int i = 0; i < kStamps; ++i) dbg_[i] = st_[i];
#endif
    }
    __device__ __forceinline__ int tid() const { return (int)threadIdx.x; }
    __device__ __forceinline__ int lane() const { return (int)(threadIdx.x & 63u); }
    // wave id as a provably wave-uniform (SGPR) value: row bases become scalar
    __device__ __forceinline__ int wave() const { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
    __device__ __forceinline__ cf *lds() const { return lds_; }
    __device__ __forceinline__ void barrier() const { __syncthreads(); }
    // Phase offset between the two waves of each SIMD (waves w and w+4): the younger half idles a
    // little after a barrier so that its LDS-exchange phases fall into the older half's butterfly
    // phases instead of colliding with them (MI355X_MICROARCH "Two waves per SIMD", item 9).
    __device__ __forceinline__ void stagger(int wave, int slots) const {
        (void)slots;
#if AW_STAGGER_SLOTS > 0
        // one opaque asm statement: a real branch here splits the block and wrecks register allocation
        asm volatile("s_cmp_lt_u32 %0, 4\n\ts_cbranch_scc1 1f\n\ts_sleep %1\n1:" ::"s"(wave), "n"(AW_STAGGER_SLOTS) : "scc");
#else
        (void)wave;
#endif
    }
    // LDS read of one complex value as a single ds_read_b64.  AW_LDS_NO_READ2: volatile 64-bit
    // access, which keeps hipcc from fusing neighbours into ds_read2_b64 / ds_read2st64_b64
    // (measured 8.3 cycles per wave-instruction against 2 x 2.6 for two ds_read_b64, tools/ubench/lds_rate.hip).
    __device__ __forceinline__ cf ld(const cf *p) const {
#if AW_LDS_NO_READ2
        const unsigned long long v = *reinterpret_cast<const volatile unsigned long long *>(p);
        cf r;
        r.x = __uint_as_float((unsigned)v);
        r.y = __uint_as_float((unsigned)(v >> 32));
        return r;
#else
        return *p;
#endif
    }
    // Scheduling fence (no instruction): keeps hipcc from interleaving the two rows' butterflies,
    // which doubles their temporaries at the register-pressure peak.
    // Cross-lane swap primitive of the register<->lane-field transposes (semantics checked by tools/ubench/xlane_swap.hip):
    //   lanes with bit b = 0: hi' = partner.lo ;  lanes with bit b = 1: lo' = partner.hi ;  partner = lane ^ (1 << b)
    __device__ __forceinline__ void xswap32(unsigned &lo, unsigned &hi, int bit) const {
        if (bit == 5) { auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false); lo = r[0]; hi = r[1]; }
        else if (bit == 4) { auto r = __builtin_amdgcn_permlane16_swap(lo, hi, false, false); lo = r[0]; hi = r[1]; }
        else {
            const unsigned t = hi;
            hi = __builtin_amdgcn_update_dpp(hi, lo, 0x128, 0xf, 0x3, false);   // lanes 0-7 of every row: hi <- lo of lane + 8
            lo = __builtin_amdgcn_update_dpp(lo, t, 0x128, 0xf, 0xc, false);    // lanes 8-15: lo <- old hi of lane - 8
        }
    }
    __device__ __forceinline__ void xswap(cf &lo, cf &hi, int bit) const {
        unsigned a = __float_as_uint(lo.x), b = __float_as_uint(hi.x);
        xswap32(a, b, bit);
        lo.x = __uint_as_float(a); hi.x = __uint_as_float(b);
        a = __float_as_uint(lo.y); b = __float_as_uint(hi.y);
        xswap32(a, b, bit);
        lo.y = __uint_as_float(a); hi.y = __uint_as_float(b);
    }
    // unconditional scheduling fence (bounds how far loads are hoisted)
    __device__ __forceinline__ void sched_fence_hard() const {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void sched_fence() const {
#if AW_SCHED_FENCE
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
    // Hides a value's provenance from the optimiser (no instruction emitted): stops LICM/CSE from
    // keeping re-computable values live across the whole tile.
    __device__ __forceinline__ int opaque_i(int v) const {
        asm volatile("" : "+v"(v));
        return v;
    }
    __device__ __forceinline__ cf opaque(cf v) const {
        asm volatile("" : "+v"(v.x), "+v"(v.y));
        return v;
    }
    // Exchanges inside one wave need no s_barrier: a wave's LDS instructions execute in issue
    // order.  The fences only stop the compiler from moving LDS accesses across the exchange.
    __device__ __forceinline__ void wave_sync() const {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};

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
template <int CS, int NP, bool INTERIOR>
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
    tiles_fused_ols<GpuCtx, CS, NP, INTERIOR>(ctx, p, lo + slot, per_xcd_wg, hi);
}

static int g_persistent_wgs = 256;      // one resident workgroup per CU (152 KB LDS each)

// Windows [tile_lo, tile_hi) of every stream lie inside the call's input (INTERIOR), the others touch the
// history or the zero page.  p.tile_lo/hi carry the window range here.
template <int CS, bool INTERIOR>
__global__ void __launch_bounds__(kThreads) aw_part_forward_kernel(TileParams p, long long nwg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GpuCtx ctx{reinterpret_cast<cf *>(smem), nullptr};
    const long long id = xcd_remap((long long)blockIdx.x, nwg);
    const int n_windows = p.n_blocks + p.partitions - 1;
    const int per = INTERIOR ? p.tile_hi - p.tile_lo : n_windows - (p.tile_hi - p.tile_lo);
    const long long stream = id / per;
    int w = (int)(id - stream * per);
    if (INTERIOR) w += p.tile_lo;
    else if (w >= p.tile_lo) w += p.tile_hi - p.tile_lo;
    tile_part_forward<GpuCtx, CS, INTERIOR>(ctx, p, stream, w);
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
// (real channels, batches): 2C pseudo-channels in batches of four
#define AW_FOR_EACH_VEC2(X) X(1, 1) X(2, 1) X(4, 2) X(6, 3) X(7, 4) X(8, 4)

const char *fused_ols2_kernel_name(int C) {
    switch (C) {
        case 1: return "aw_fused_ols2_kernel<1, 1, true>";
        case 2: return "aw_fused_ols2_kernel<2, 1, true>";
        case 4: return "aw_fused_ols2_kernel<4, 2, true>";
        case 6: return "aw_fused_ols2_kernel<6, 3, true>";
        case 7: return "aw_fused_ols2_kernel<7, 4, true>";
        case 8: return "aw_fused_ols2_kernel<8, 4, true>";
        default: return "aw_fused_ols2_kernel<0, 0, false>";
    }
}
static bool has_vec2_variant(int C) { return C == 1 || C == 2 || C == 4 || C == 6 || C == 7 || C == 8; }

// Kernel variants.  Vectorised interior kernels <CS, NP, true> exist for the channel counts whose
// frames are whole float4s/float2s (2, 4, 8, 12, 16 channels: stereo ... 7.1.4 + 4); every other
// case — the few boundary tiles of those, and all tiles of the other channel counts — runs the
// generic-addressing kernels <0, NP, false> (NP = compile-time pair count 1..4; NP = 0 loops over
// batches of two pairs at run time: more than 8 channels, where full unrolling only spills).
#define AW_FOR_EACH_VEC(X) X(2, 1) X(4, 2) X(6, 3) X(7, 4) X(8, 4) X(12, 0) X(14, 0) X(16, 0)
#define AW_FOR_EACH_GEN(X) X(1) X(2) X(3) X(4) X(0)
// boundary tiles of the common layouts keep whole-frame vector loads (history / zero-page selects per frame)
#define AW_FOR_EACH_BVEC(X) X(2, 1) X(4, 2) X(8, 4)

hipError_t prepare_kernels() {
    hipError_t e = hipSuccess;
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            g_persistent_wgs = cus;
        if (const char *e2 = getenv("AW_PERSISTENT_WGS")) g_persistent_wgs = atoi(e2) > 0 ? atoi(e2) : g_persistent_wgs;
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
#undef AW_SET_BVEC
#undef AW_SET_VEC
#undef AW_SET_GEN
#define AW_SET_FWD(CS, NP)                                                                              \
    if (e == hipSuccess)                                                                                \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_part_forward_kernel<CS, true>),      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    AW_FOR_EACH_VEC(AW_SET_FWD)
#undef AW_SET_FWD
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_part_forward_kernel<0, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_part_inverse_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kInvLdsBytes);
    return e;
}

static bool has_vec_variant(int C) { return C == 2 || C == 4 || C == 6 || C == 7 || C == 8 || C == 12 || C == 14 || C == 16; }

const char *fused_ols_kernel_name(int C) {
    switch (C) {
        case 2: return "aw_fused_ols_kernel<2, 1, true>";
        case 4: return "aw_fused_ols_kernel<4, 2, true>";
        case 6: return "aw_fused_ols_kernel<6, 3, true>";
        case 7: return "aw_fused_ols_kernel<7, 4, true>";
        case 8: return "aw_fused_ols_kernel<8, 4, true>";
        case 14: return "aw_fused_ols_kernel<14, 0, true>";
        case 12: return "aw_fused_ols_kernel<12, 0, true>";
        case 16: return "aw_fused_ols_kernel<16, 0, true>";
        default: return "aw_fused_ols_kernel<0, NP, false>";
    }
}

static dim3 persistent_grid(long long n_tiles) {
    return dim3((unsigned)(n_tiles < g_persistent_wgs ? n_tiles : g_persistent_wgs));
}

static void launch_vec(const TileParams &p, long long n_tiles, hipStream_t stream) {
    const dim3 grid = persistent_grid(n_tiles), block(kThreads);
    switch (p.n_channels) {
#define AW_CASE(CS, NP) case CS: hipLaunchKernelGGL((aw_fused_ols_kernel<CS, NP, true>), grid, block, kLdsBytes, stream, p, n_tiles); break;
        AW_FOR_EACH_VEC(AW_CASE)
#undef AW_CASE
        default: break;
    }
}

static void launch_gen(const TileParams &p, long long n_tiles, hipStream_t stream) {
    const dim3 grid = persistent_grid(n_tiles), block(kThreads);
    switch (p.n_channels) {
#define AW_CASE(CS, NP) case CS: hipLaunchKernelGGL((aw_fused_ols_kernel<CS, NP, false>), grid, block, kLdsBytes, stream, p, n_tiles); return;
        AW_FOR_EACH_BVEC(AW_CASE)
#undef AW_CASE
        default: break;
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
    if (!has_vec_variant(p.n_channels)) { lo = 0; hi = 0; }        // everything through the generic kernels
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
        const dim3 grid = persistent_grid(n_int), block(kThreads);
        if (ev0) (void)hipEventRecord(ev0, stream);
        switch (p.n_channels) {
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
        const dim3 grid = persistent_grid(n_bnd), block(kThreads);
        switch (p.n_channels) {      // compile-time channel and batch counts also for the boundary tiles (scalar loads)
#define AW_CASE(CS, NB) case CS: hipLaunchKernelGGL((aw_fused_ols2_kernel<CS, NB, false>), grid, block, kLdsBytes, stream, p, n_bnd); break;
            AW_FOR_EACH_VEC2(AW_CASE)
#undef AW_CASE
            default: hipLaunchKernelGGL((aw_fused_ols2_kernel<0, 0, false>), grid, block, kLdsBytes, stream, p, n_bnd); break;
        }
        if (ev1 && !dom_int) (void)hipEventRecord(ev1, stream);
    }
    return hipGetLastError();
}

hipError_t launch_part_forward(const TileParams &p_in, int n_streams, hipStream_t stream) {
    TileParams p = p_in;
    const int n_windows = p.n_blocks + p.partitions - 1;
    if ((long long)n_streams * n_windows <= 0) return hipSuccess;
    // window w covers frames [(w - P) B, (w - P) B + N): interior iff it starts at >= 0 and ends inside the input
    // (one frame of slack for layouts whose frames are not whole float4s, see load_batch)
    const long long usable = p.frames - ((p.n_channels % 4 != 0 && p.n_channels != 2) ? 1 : 0);
    long long lo = p.partitions;
    long long hi = usable >= kN ? (usable - kN) / p.hop + p.partitions + 1 : lo;
    if (hi > n_windows) hi = n_windows;
    if (lo > n_windows) lo = n_windows;
    if (hi < lo) hi = lo;
    if (!has_vec_variant(p.n_channels)) hi = lo;
    p.tile_lo = (int)lo; p.tile_hi = (int)hi;
    const long long n_int = (long long)n_streams * (hi - lo), n_bnd = (long long)n_streams * (n_windows - (hi - lo));
    if (n_int > 0x7fffffffLL || n_bnd > 0x7fffffffLL) return hipErrorInvalidValue;
    if (n_int > 0) {
        switch (p.n_channels) {
#define AW_CASE(CS, NP) case CS: hipLaunchKernelGGL((aw_part_forward_kernel<CS, true>), dim3((unsigned)n_int), dim3(kThreads), kLdsBytes, stream, p, n_int); break;
            AW_FOR_EACH_VEC(AW_CASE)
#undef AW_CASE
            default: break;
        }
    }
    if (n_bnd > 0)
        hipLaunchKernelGGL((aw_part_forward_kernel<0, false>), dim3((unsigned)n_bnd), dim3(kThreads), kLdsBytes, stream, p, n_bnd);
    return hipGetLastError();
}

hipError_t launch_part_cmac(const TileParams &p, int n_streams, hipStream_t stream) {
    const int groups = (p.n_blocks + kCmacBlocks - 1) / kCmacBlocks;
    if (n_streams <= 0 || groups <= 0) return hipSuccess;
    if (groups > 65535 || n_streams > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(aw_part_cmac_kernel, dim3(kN / kCmacThreads, (unsigned)groups, (unsigned)n_streams), dim3(kCmacThreads), 0,
                       stream, p);
    return hipGetLastError();
}

hipError_t launch_part_inverse(const TileParams &p, int n_streams, hipStream_t stream) {
    const long long nwg = (long long)n_streams * p.n_blocks;
    if (nwg <= 0) return hipSuccess;
    if (nwg > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(aw_part_inverse_kernel, dim3((unsigned)nwg), dim3(kThreads), kInvLdsBytes, stream, p, nwg);
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
