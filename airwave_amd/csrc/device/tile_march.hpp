// tile_march.hpp — partitioned (long-HRIR) path, kernel 2: the frequency-domain delay line of
//   ConvolutionEngine.process            (Airwave/ConvolutionEngine.swift:256-350: FDL push, per-partition zvmul + zvadd)
// marched through time with BOTH the delay line and the partition tables in registers.
//
//   W_b[k] = sum over partitions q and channel pairs p of  Z_{b-q,p}[k] A_{q,p}[k] + conj(Z_{b-q,p}[N-k]) B_{q,p}[k]
//
// One thread owns one bin pair {k, N-k} of one channel pair for the whole timeline of a stream: it loads the two
// table entries of PQ partitions once (8 PQ floats), keeps the last PQ windows' two spectrum values (4 PQ floats) and
// per window loads 16 bytes and issues 16 PQ FMAs — every spectrum value is read from memory exactly once and no table
// byte is re-read (the block-group kernel this replaces read (8+P-1)/8 windows and P/8 table sets per block).  The
// channel pairs of a bin pair sit in adjacent lanes (a lane group of LG = 1, 2, 4 or 8) and are summed across lanes
// before the store.  Partition counts above PQ run as several passes (q0 = 0, PQ, ...) that accumulate into W.
//
// Shared with the CPU emulation harness (tests/emu/): march_thread() is the per-thread body; the cross-lane sum and
// the store are the caller's `emit`.
#pragma once
#include "tile_ols.hpp"

namespace awk {

constexpr int kMarchThreads = 256;
constexpr int kMarchSlots = kN / 2 + 1;     // bin-pair slots per (stream, channel pair): 4095 pairs {k, N-k} + the self-paired k = 0 and k = N/2

// Slot j -> storage indices (tile_ols.hpp: bin k = k1 + 16 k2 is stored at k1 * 512 + k2) of the slot's two bins.
// Partner of (k1, k2): row (16 - k1) & 15, column (512 - k2) & 511 for row 0, else 511 - k2.  Consecutive slots are
// consecutive addresses at i and descending addresses at pi: both loads of a wave touch whole 128-B lines.
struct MarchBins { int i, pi; };
AW_HD MarchBins march_bins(int j) {
    MarchBins r;
    if (j < 7 * kSub) {                          // rows 1..7 with rows 15..9
        const int row = 1 + (j >> 9), col = j & 511;
        r.i = row * kSub + col; r.pi = (16 - row) * kSub + (511 - col);
    } else if (j < 7 * kSub + 256) {             // row 0: column 0 is k = 0 (its own partner), columns 1..255 with 511..257
        const int col = j - 7 * kSub;
        r.i = col; r.pi = (512 - col) & 511;
    } else if (j < 8 * kSub) {                   // row 8: columns 0..255 with 511..256
        const int col = j - (7 * kSub + 256);
        r.i = 8 * kSub + col; r.pi = 8 * kSub + 511 - col;
    } else {                                     // k = N/2 (row 0, column 256), its own partner
        r.i = 256; r.pi = 256;
    }
    return r;
}

// Explicit fused multiply-adds: with the generic cfma/cfmac expressions hipcc's SLP vectoriser splits the chains into
// v_pk_mul + v_pk_add pairs (1790 VALU instructions per 8 steps instead of ~1100).
AW_HD cf march_fma(cf a, cf b, cf acc) {           // acc + a b
    acc.x = __builtin_fmaf(a.x, b.x, acc.x); acc.y = __builtin_fmaf(a.x, b.y, acc.y);
    acc.x = __builtin_fmaf(-a.y, b.y, acc.x); acc.y = __builtin_fmaf(a.y, b.x, acc.y);
    return acc;
}
AW_HD cf march_fmac(cf a, cf b, cf acc) {          // acc + conj(a) b
    acc.x = __builtin_fmaf(a.x, b.x, acc.x); acc.y = __builtin_fmaf(a.x, b.y, acc.y);
    acc.x = __builtin_fmaf(a.y, b.y, acc.x); acc.y = __builtin_fmaf(-a.y, b.x, acc.y);
    return acc;
}

// f(integral_constant<0>) ... f(integral_constant<N-1>): a loop whose index is a compile-time constant in the body
template <int I> struct MarchIdx { static constexpr int value = I; };
template <int N, int I = 0, class F>
AW_HD void march_unroll(F &&f) {
    if constexpr (I < N) {
        f(MarchIdx<I>{});
        march_unroll<N, I + 1>(f);
    }
}

template <int PQ> struct MarchState {
    cf2 ti[PQ], tp[PQ];     // {A, B} of partitions q0 .. q0 + PQ - 1 at bin i and at bin pi
    cf fz[PQ], fp[PQ];      // delay line: Z[i], Z[pi] of the last PQ windows; slot = step mod PQ
};

#ifndef AW_MARCH_CHAINS
#define AW_MARCH_CHAINS 4
#endif
#ifndef AW_MARCH_DEPTH
#define AW_MARCH_DEPTH 4     // windows in flight per thread ahead of the one being applied
#endif

// One spectrum value, read once: a non-temporal load (measured 7.1 against 6.4 TB/s for plain streaming loads).
#ifndef AW_MARCH_NT
#define AW_MARCH_NT 1
#endif
template <bool NT>
AW_HD cf march_ld(const cf *p) {
    if constexpr (!NT) return *p;
#if defined(__HIP_DEVICE_COMPILE__) && AW_MARCH_NT
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f v = __builtin_nontemporal_load(reinterpret_cast<const v2f *>(p));
    return mk(v.x, v.y);
#else
    return *p;
#endif
}

// emit(stream, block, bins, acc_i, acc_pi): acc_* are THIS lane's (channel pair's) partial sums of W_block at the two
// bins.  Lanes with pl >= n_pairs carry zero tables (they exist only to keep lane groups a power of two).
// The thread walks streams [stream0, stream1) one after the other with the same tables.
// p.herm_last: the last channel pair has a real input only (odd channel count), so its spectrum is Hermitian and the
// forward kernel stored rows 0..8 only: slots whose partner bin lies in rows 9..15 take conj(Z[i]) instead of a load.
// NT: non-temporal spectrum loads.  Right when a wave consumes whole 128-byte lines (lane groups of up to 4: 16 slots x 8 B
// per plane); with lane groups of 8 a wave takes HALF of each line and its sibling wave the other half — then the line must
// stay cached for the sibling (measured with NT on the 14-channel reading of cfg 3: 108 GB read for 55 GB of spectra).
template <int PQ, bool NT = true, class Emit>
AW_HD void march_thread(const TileParams &p, long long stream0, long long stream1, int j, int pl, int q0, Emit &&emit) {
    constexpr int D = AW_MARCH_DEPTH < PQ ? AW_MARCH_DEPTH : PQ;
    static_assert(PQ % D == 0, "the prefetch ring is indexed by step mod D inside a loop unrolled by PQ");
    const int P = p.partitions;
    const int nq = P - q0 < PQ ? P - q0 : PQ;
    const int n_windows = p.n_blocks + P - 1;
    const MarchBins mb = march_bins(j);
    const bool live = pl < p.n_pairs;
    const int pr = live ? pl : 0;
    const bool herm = p.herm_last && pr == p.n_pairs - 1 && j < 7 * kSub;      // partner row 9..15: not stored
    const cf zero = mk(0.f, 0.f);
    MarchState<PQ> s;
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
        const bool on = live && q < nq;
        const cf2 *t = p.tab + ((long long)(q0 + (q < nq ? q : 0)) * p.n_pairs + pr) * kN;
        const cf2 a = t[mb.i], b = t[mb.pi];
        s.ti[q].a = on ? a.a : zero; s.ti[q].b = on ? a.b : zero;
        s.tp[q].a = on ? b.a : zero; s.tp[q].b = on ? b.b : zero;
    }
    const long long wstride = (long long)p.n_pairs * kN;
    // Step t applies window u_first + t as partition q0 of block t - (PQ - 1).  Windows before 0 (only when nq < PQ)
    // and past the end (prefetch overrun) are clamped: whatever they hold meets a zero table or is never applied.
    const int u_first = P - 1 - q0 - (PQ - 1);
    // Hermitian lanes re-read Z[i] instead of the partner (same line: no extra bytes) and conjugate it in registers, so
    // the steady state has no branch at all — a conditional load makes hipcc drain vmcnt(0) at every step.
    const int ipart = herm ? mb.i : mb.pi;
    for (long long stream = stream0; stream < stream1; ++stream) {
        const cf *spec_s = p.spec + (stream * (long long)n_windows * p.n_pairs + pr) * (long long)kN;
        auto load2 = [&](int t, cf &z, cf &zp) {
            int u = u_first + t;
            u = u < 0 ? 0 : (u >= n_windows ? n_windows - 1 : u);
#ifdef AW_ABL_MARCH_NOLOAD      // timing ablation only (wrong results): every step re-reads window 0 (cache hits)
            u = 0;
#endif
            const cf *w = spec_s + (long long)u * wstride;
            z = march_ld<NT>(w + mb.i);
            zp = march_ld<NT>(w + ipart);
        };
        auto push = [&](int slot, cf z, cf zp) {
            s.fz[slot] = z;
            s.fp[slot] = herm ? conj(z) : zp;
        };
        cf pre_z[D], pre_p[D];
#pragma unroll
        for (int d = 0; d < D; ++d) load2(d, pre_z[d], pre_p[d]);
        // warm-up: windows of steps 0 .. PQ-2 enter the delay line, nothing is emitted
#pragma unroll
        for (int t = 0; t < PQ - 1; ++t) {
            push(t, pre_z[t % D], pre_p[t % D]);
            load2(t + D, pre_z[t % D], pre_p[t % D]);
        }
        // one step: block b = g * PQ + JJ; its newest window sits in slot (JJ + PQ - 1) % PQ, prefetch ring slot likewise mod D
        auto step = [&](int b, auto jj_c) {
            constexpr int JJ = decltype(jj_c)::value;
            constexpr int SLOT = (JJ + PQ - 1) % PQ, RING = (JJ + PQ - 1) % D;
            push(SLOT, pre_z[RING], pre_p[RING]);
            load2(b + PQ - 1 + D, pre_z[RING], pre_p[RING]);
            // two accumulator sets (even / odd partitions): eight independent FMA chains per thread instead of four
            cf ai = zero, ap = zero, ai2 = zero, ap2 = zero;
#pragma unroll
            for (int q = 0; q < PQ; ++q) {
                const int slot = (SLOT - q + PQ) % PQ;                // compile-time
                if (AW_MARCH_CHAINS == 8 && (q & 1)) {
                    ai2 = march_fma(s.fz[slot], s.ti[q].a, ai2);
                    ai2 = march_fmac(s.fp[slot], s.ti[q].b, ai2);
                    ap2 = march_fma(s.fp[slot], s.tp[q].a, ap2);
                    ap2 = march_fmac(s.fz[slot], s.tp[q].b, ap2);
                } else {
                    ai = march_fma(s.fz[slot], s.ti[q].a, ai);
                    ai = march_fmac(s.fp[slot], s.ti[q].b, ai);
                    ap = march_fma(s.fp[slot], s.tp[q].a, ap);
                    ap = march_fmac(s.fz[slot], s.tp[q].b, ap);
                }
            }
            if (AW_MARCH_CHAINS == 8) { ai = ai + ai2; ap = ap + ap2; }
            emit(stream, b, mb, ai, ap);
        };
        const int groups = p.n_blocks / PQ;
        for (int g = 0; g < groups; ++g) {                            // steady state: no branch inside
            march_unroll<PQ>([&](auto jj_c) { step(g * PQ + decltype(jj_c)::value, jj_c); });
        }
        {   // remainder (< PQ blocks)
            const int b0 = groups * PQ, rem = p.n_blocks - b0;
            march_unroll<PQ>([&](auto jj_c) { if (decltype(jj_c)::value < rem) step(b0 + decltype(jj_c)::value, jj_c); });
        }
    }
}

}  // namespace awk
