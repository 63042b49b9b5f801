// gpu_ctx.hpp — the GPU execution context of the tile code (LDS, barriers, fences); the CPU emulation harness has its
// own (tests/emu/emu_harness.cpp).  Included by kernels.hip only (and by one-kernel probe builds under tools/).
#pragma once
#include "tile_ols.hpp"

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
#if defined(AW_MARK_STAMPS)       // probe builds: an assembler comment per stamp, to find the phases in the ISA listing
        asm volatile("; AW_STAMP %0" ::"s"(i));
#elif AW_STAMPS
        unsigned long long tm;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm)::"memory");
        if (stamp_on_) st_[i] = tm;
#else
        (void)i;
#endif
    }
    int stamp_thread_ = 0;
    bool stamp_on_ = true;        // diagnostic builds: a persistent kernel records ONE of its tiles (a mid-kernel one: the last tile runs on a draining chip)
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
#ifdef AW_ABL_NOBARRIER       // timing ablation only (racy)
    __device__ __forceinline__ void barrier() const { __builtin_amdgcn_wave_barrier(); }
#else
    __device__ __forceinline__ void barrier() const { __syncthreads(); }
#endif
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
#elif AW_LDS_ATOMIC_READS
        return ld_single(p);
#else
        return *p;
#endif
    }
    // Eight LDS reads per row for two rows, element i at p + i * STRIDE (complex elements), as sixteen single ds_read_b64
    // in one asm block.  Left to itself hipcc fuses neighbouring reads into ds_read2_b64 / ds_read2st64_b64, which take 8 LDS
    // cycles per wave-instruction for 16 bytes per lane where two ds_read_b64 take 2 + 2 (MI355X_MICROARCH.md, LDS table);
    // a volatile access to stop the fusion turns into flat loads.  AW_ASM_LDS_READS=0: plain reads.
#ifndef AW_ASM_LDS_READS
#define AW_ASM_LDS_READS 0
#endif
#ifndef AW_LDS_ATOMIC_READS
#define AW_LDS_ATOMIC_READS 0
#endif
    template <int STRIDE>
    __device__ __forceinline__ void ld8x2(cf (&a)[8], const cf *p0, cf (&b)[8], const cf *p1) const {
#if AW_ASM_LDS_READS
        typedef float v2f __attribute__((ext_vector_type(2)));
        v2f r0, r1, r2, r3, r4, r5, r6, r7, q0, q1, q2, q3, q4, q5, q6, q7;
        const unsigned a0 = (unsigned)(unsigned long long)(p0), a1 = (unsigned)(unsigned long long)(p1);   // LDS addresses are 32-bit
        asm volatile(
            "ds_read_b64 %0, %16 offset:%18\n\tds_read_b64 %8, %17 offset:%18\n\t"
            "ds_read_b64 %1, %16 offset:%19\n\tds_read_b64 %9, %17 offset:%19\n\t"
            "ds_read_b64 %2, %16 offset:%20\n\tds_read_b64 %10, %17 offset:%20\n\t"
            "ds_read_b64 %3, %16 offset:%21\n\tds_read_b64 %11, %17 offset:%21\n\t"
            "ds_read_b64 %4, %16 offset:%22\n\tds_read_b64 %12, %17 offset:%22\n\t"
            "ds_read_b64 %5, %16 offset:%23\n\tds_read_b64 %13, %17 offset:%23\n\t"
            "ds_read_b64 %6, %16 offset:%24\n\tds_read_b64 %14, %17 offset:%24\n\t"
            "ds_read_b64 %7, %16 offset:%25\n\tds_read_b64 %15, %17 offset:%25\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7),
              "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6), "=&v"(q7)
            : "v"(a0), "v"(a1), "n"(0 * STRIDE * 8), "n"(1 * STRIDE * 8), "n"(2 * STRIDE * 8), "n"(3 * STRIDE * 8), "n"(4 * STRIDE * 8),
              "n"(5 * STRIDE * 8), "n"(6 * STRIDE * 8), "n"(7 * STRIDE * 8)
            : "memory");
        a[0] = mk(r0.x, r0.y); a[1] = mk(r1.x, r1.y); a[2] = mk(r2.x, r2.y); a[3] = mk(r3.x, r3.y);
        a[4] = mk(r4.x, r4.y); a[5] = mk(r5.x, r5.y); a[6] = mk(r6.x, r6.y); a[7] = mk(r7.x, r7.y);
        b[0] = mk(q0.x, q0.y); b[1] = mk(q1.x, q1.y); b[2] = mk(q2.x, q2.y); b[3] = mk(q3.x, q3.y);
        b[4] = mk(q4.x, q4.y); b[5] = mk(q5.x, q5.y); b[6] = mk(q6.x, q6.y); b[7] = mk(q7.x, q7.y);
#elif AW_LDS_ATOMIC_READS
        // relaxed wavefront-scope atomic loads: still ds_read_b64, but SILoadStoreOptimizer leaves atomics alone (no read2 fusion)
        // and the compiler keeps placing the waits itself
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = ld_single(p0 + i * STRIDE); }
#pragma unroll
        for (int i = 0; i < 8; ++i) { b[i] = ld_single(p1 + i * STRIDE); }
#else
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = p0[i * STRIDE]; b[i] = p1[i * STRIDE]; }
#endif
    }
    __device__ __forceinline__ cf ld_single(const cf *p) const {
        const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        cf r;
        r.x = __uint_as_float((unsigned)v);
        r.y = __uint_as_float((unsigned)(v >> 32));
        return r;
    }
    // Scheduling fence (no instruction): keeps hipcc from interleaving the two rows' butterflies,
    // which doubles their temporaries at the register-pressure peak.
    // Cross-lane swap primitive of the register<->lane-field transposes (semantics checked by tools/ubench/xlane_swap.hip):
    //   lanes with bit b = 0: hi' = partner.lo ;  lanes with bit b = 1: lo' = partner.hi ;  partner = lane ^ (1 << b)
    __device__ __forceinline__ void xswap32(unsigned &lo, unsigned &hi, int bit) const {
        if (bit == 5) { auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false); lo = r[0]; hi = r[1]; }
        else if (bit == 4) { auto r = __builtin_amdgcn_permlane16_swap(lo, hi, false, false); lo = r[0]; hi = r[1]; }
        else if (bit == 3) {
            const unsigned t = hi;
            hi = __builtin_amdgcn_update_dpp(hi, lo, 0x128, 0xf, 0x3, false);   // lanes 0-7 of every row: hi <- lo of lane + 8
            lo = __builtin_amdgcn_update_dpp(lo, t, 0x128, 0xf, 0xc, false);    // lanes 8-15: lo <- old hi of lane - 8
        } else {                                                                // bit == 2: partner = lane ^ 4, bank-masked row shifts by 4
            const unsigned t = hi;
            hi = __builtin_amdgcn_update_dpp(hi, lo, 0x104, 0xf, 0x5, false);   // row_shl:4, lanes 0-3 / 8-11 of every row: hi <- lo of lane + 4
            lo = __builtin_amdgcn_update_dpp(lo, t, 0x114, 0xf, 0xa, false);    // row_shr:4, lanes 4-7 / 12-15: lo <- old hi of lane - 4
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
    // Streaming (non-temporal) stores for data written once and read by a LATER kernel (spectra, W, the stereo output):
    // they do not occupy the XCD's L2, which then keeps the input lines that overlapping windows and the second channel
    // batch re-read (cfg 3 forward kernel: fabric-side reads 27.9 -> 15.7 GB per step, 10.7 -> 9.95 ms).  AW_NT_STORES=0: plain.
#ifndef AW_NT_STORES
#define AW_NT_STORES 1
#endif
    __device__ __forceinline__ void st_stream(cf *p, cf v) const {
#if AW_NT_STORES
        typedef float v2f __attribute__((ext_vector_type(2)));
        v2f x = {v.x, v.y};
        __builtin_nontemporal_store(x, reinterpret_cast<v2f *>(p));
#else
        *p = v;
#endif
    }
    // data read once (rows of the long-window path's scratch): a non-temporal load (tile_march.hpp: 7.1 against 6.4 TB/s)
    __device__ __forceinline__ cf ld_stream(const cf *p) const {
#ifdef AW_LD_STREAM_PLAIN        // A/B only
        return *p;
#endif
        typedef float v2f __attribute__((ext_vector_type(2)));
        const v2f v = __builtin_nontemporal_load(reinterpret_cast<const v2f *>(p));
        return mk(v.x, v.y);
    }
    __device__ __forceinline__ void st_stream4(float *p, float a, float b, float c, float d) const {   // dword-aligned 16 bytes
#if AW_NT_STORES
        typedef float v4fu __attribute__((ext_vector_type(4), aligned(4)));
        v4fu x = {a, b, c, d};
        __builtin_nontemporal_store(x, reinterpret_cast<v4fu *>(p));
#else
        f4u v; v.x = a; v.y = b; v.z = c; v.w = d;
        *reinterpret_cast<f4u *>(p) = v;
#endif
    }
    // value of lane ^ 1 (DPP quad_perm [1,0,3,2]): pairs neighbouring bins for 16-byte stores
    __device__ __forceinline__ cf xchg1(cf v) const {
        cf r;
        r.x = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.x), 0xB1, 0xf, 0xf, true));
        r.y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.y), 0xB1, 0xf, 0xf, true));
        return r;
    }
    // issue priority of this wave among the waves of its SIMD (s_setprio 0..3)
    template <int P> __device__ __forceinline__ void prio() const { __builtin_amdgcn_s_setprio(P); }
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
    // Buffer loads (tile_ola.hpp): a descriptor carries its byte size and the hardware returns zeros for every dword at or past it — the
    // ragged end of a stream, and the rows before a history window, cost no compare, no select and no zero page.  Raw buffer
    // (stride 0), 32-bit data format; `off` = per-lane byte offset (offen), `imm` = compile-time byte offset of the instruction.
    typedef __amdgpu_buffer_rsrc_t Buf;
    __device__ __forceinline__ Buf buf(const void *base, unsigned bytes) const {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
    }
    template <int N> __device__ __forceinline__ void buf_ld(const Buf &b, unsigned off, int imm, float *dst) const {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        if constexpr (N == 4) {
            const v4u v = __builtin_amdgcn_raw_buffer_load_b128(b, (int)(off + (unsigned)imm), 0, 0);
            dst[0] = __uint_as_float(v.x); dst[1] = __uint_as_float(v.y); dst[2] = __uint_as_float(v.z); dst[3] = __uint_as_float(v.w);
        } else if constexpr (N == 2) {
            const v2u v = __builtin_amdgcn_raw_buffer_load_b64(b, (int)(off + (unsigned)imm), 0, 0);
            dst[0] = __uint_as_float(v.x); dst[1] = __uint_as_float(v.y);
        } else {
            dst[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b, (int)(off + (unsigned)imm), 0, 0));
        }
    }
    // Exchanges inside one wave need no s_barrier: a wave's LDS instructions execute in issue
    // order.  The fences only stop the compiler from moving LDS accesses across the exchange.
    __device__ __forceinline__ void wave_sync() const {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};

}  // namespace awk
