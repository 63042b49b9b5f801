// tile_lw.hpp — the LONG-WINDOW path for long calls (cfg 3's 32 768-tap HRIRs; past a measured HRIR length every layout,
// runtime.cpp lw_choose): single-partition overlap-save on windows of N = R x 4096 frames (R = 32, 64, 128: N = 131 072 ...
// 524 288), the transform run as a four-step FFT through HBM scratch.
//
// Replaces, for a whole window of one stream at once, what ConvolutionEngine.process does per 512-frame block with a
// frequency-domain delay line (Airwave/ConvolutionEngine.swift:232-367) and what RealtimeAudioProcessor.processPendingBlock
// sums over speakers (Airwave/RealtimeAudioProcessor.swift:141-172).  Against the partitioned path of tile_ols.hpp /
// tile_march.hpp (B = 4096, 2x overlap, spectra of every window written and read once per partition step) a window
// N >> taps overlaps by N / (N - taps) only (1.07 for 10 s streams on N = 524 288) and needs no delay line at all:
// 119 instead of 203 bytes of fabric traffic per output frame on cfg 3, measured.
//
// Odd-frequency transform.  All spectra are sampled at k + 1/2:  X[k] = sum_n x[n] w_N^{n (k + 1/2)}, w_N = exp(-2 pi i / N).
// The wrap-around of the product then carries a minus sign (negacyclic), which overlap-save discards anyway, and the
// Hermitian partner of bin k is N-1-k: with k = k1 + R k2 the partner of row k1 is row R-1-k1 — never the row itself —
// so ALL R/2 row pairs look alike (no DC / Nyquist special rows).
//
//   kernel 1  lw_split  (per stream, window, 64 consecutive t):   n = 4096 j + t
//       D[k1] = sum_j z[4096 j + t] w_R^{j (k1 + 1/2)}            odd DFT of size R over the stride-4096 frames, two levels:
//                                                                 j = j1 + 8 j2 in-thread over j2 (size RA = R/8), then
//                                                                 across the 8 waves (j1) through LDS
//       rows k1 <  R/2 :  S[k1][t] = D[k1] w_N^{t (k1 + 1/2)}
//       rows k1 >= R/2 :  S[k1][t] = conj(D[k1]) w_N^{t (R-1-k1 + 1/2)}        "conj-reversed": FFT_4096(S[k1])[k2] =
//                                                                 conj(X[N-1-k]) at k = (R-1-k1) + R k2 — the partner
//                                                                 bin of the lower row lands in the SAME lane
//       A real last channel (odd channel count) gives S[R-1-k1] == S[k1]: its upper rows are neither stored nor transformed.
//   kernel 2  lw_rows   (per stream, window, row pair {ra, rb = R-1-ra}; row pairs pinned to XCDs, the 512 KB table slice
//                        of a row pair stays in that XCD's L2 while every stream passes):
//       Z1 = FFT_4096(S[ra]),  V = FFT_4096(S[rb])  per channel pair        (8 x 512: radix-8 pass + per-wave 512-point sub-FFTs)
//       W1 += Z1 T0 + V T1 ;  W2 += V T2 + Z1 T3                          2x2 complex MAC per bin, all in-lane: no partner
//                                                                            exchange, no split of the packed real channels
//       s1 = IFFT_4096(W1), s2 = IFFT_4096(W2)  -> scratch
//   kernel 3  lw_merge  (per stream, window, 64 consecutive t): the mirror image of kernel 1 on (s1, s2), stores the
//       last `hop` frames of the window as interleaved stereo.
// Tables T0..T3 (host, float64, csrc/host/tables.cpp build_lw_tables): with A_p, B_p as in tile_ols.hpp but sampled at
// k + 1/2 and k' = N-1-k:  T0 = A[k], T1 = B[k], T2 = conj(A[k']), T3 = conj(B[k']).
// The algebra is checked in float64 by tools/lw/math_check.py; this file under CPU thread emulation by tests/test_emu_lw.py.
#pragma once
#include "tile_ols.hpp"

namespace awk {

constexpr int kLwM = 4096;                  // row length (frames between the R strided sub-sequences of a window)
constexpr int kLwInner = kLwM / kSub;       // 8 inner rows of 512 per row: radix of the row kernel's pass 1
constexpr int kLwTw = 64;                   // frames t per split / merge tile: one wave wide
constexpr int kLwChunks = kLwM / kLwTw;     // 64 tiles per (stream, window)

struct alignas(16) LwTab { cf t0, t1, t2, t3; };

struct LwParams {
    const float *in;        // [stream][frames][C] interleaved
    float *out;             // [stream][frames][2]
    const float *hist;      // [stream][hist_len][C]: the frames preceding in[...][0]
    float *hist_out;        // [stream][hist_len][C] or NULL: the split kernel also carries the convolution tail — the last hist_len
                            // frames of (history ++ input) — into the other history buffer (every such frame passes through it anyway)
    const float *zeros;     // >= 64 bytes of zeros
    long long frames;       // frames per stream in this call
    long long frame0;       // a call may run as several groups of windows, each with its own window length: this launch's windows start at
    long long frame_end;    // frame0 + win * hop (their history: the frames before, from `in` or, below 0, from `hist`) and store frames < frame_end
    int n_channels, n_pairs;
    const float *tail;      // wide split kernel (9-16 channels): a copy of the LAST frame of the last stream of `in`, followed by >= 4 zeros
    int n_streams;          // streams in this launch (the wide split kernel redirects the last frame of the last one to `tail`)
    int real_last;          // odd channel count: the last pair's second channel is absent (real input)
    int hist_len;           // N - hop >= taps - 1: window positions below it are discarded
    int hop;                // new output frames per window
    int n_windows;          // ceil(frames / hop)
    int R, N;               // R = 8 RA rows of kLwM frames
    cf *spec;               // [stream][window][pair][rows: R, or R/2 for a real last pair][4096]
    long long spec_per_sw;  // complex elements of spec per (stream, window)
    cf *wrows;              // [stream][window][row pair][2][4096]: s1, s2
    const LwTab *tab;       // [row pair][pair][8][512]: bin k2 = q1 + 8 q2 of the row pair at [q1][q2]
    // row twiddles w_N^{t (k1 + 1/2)}, t = 64 tc + lane, k1 = RA m + ka:  base(ka) = coarse[ka][tc] * fine[ka][lane], times step[m][t]
    const cf *tw_coarse;    // [RA][64]: w_{2N}^{64 tc (2 ka + 1)}
    const cf *tw_fine;      // [RA][64]: w_{2N}^{lane (2 ka + 1)}
    const cf *tw_step;      // [3][4096]: w_{2N}^{2 RA m t}, m = 1, 2, 3
    const cf *tw_r;         // [RA][8]: w_{2R}^{j1 (2 ka + 1)}
    const cf *tw1m;         // [512]: w_4096^{t}
    const cf *twa, *twb;    // sub-FFT twiddles of the context (tile_ols.hpp)
    int persistent_wgs;
    int rows_pairs_per_batch;   // rows kernel form: 2 = two pairs per batch, one workgroup per CU; 1 = one pair per batch, two workgroups per CU
    // rows kernel, 16-points-per-thread form (tile_lw16.hpp): 256-thread workgroups, one row at a time
    int rows_form;              // 16 = that form (tab16 / tw2 are set), otherwise the 8-point forms above (tab is set)
    const struct LwTab2 *tab16; // [row pair][pair][row ra | rb][m1 = 16][thread = 256]: {T0, T3} for row ra, {T1, T2} for row rb
    const cf *tw2;              // [m0 = 16][a = 16]: w_256^{a m0}
    int rows16_wgs;             // workgroups per CU of that form's persistent grid
};

// window lengths: R = 8 RA rows of 4096 frames, RA = 4 .. 16 without the primes 11 and 13 (the split / merge kernels run a DFT of that
// size in registers, lw_fft below; the direct 11- and 13-point products make them compute-bound: 33-36 against 40-44 G frames/s on cfg 3's layout)
constexpr bool lw_ra_ok(int ra) { return ra >= 4 && ra <= 16 && ra != 11 && ra != 13; }

template <int I> struct LwIdx { static constexpr int value = I; };
template <int NN, int I = 0, class F> AW_HD void lw_unroll(F &&f) {
    if constexpr (I < NN) { f(LwIdx<I>{}); lw_unroll<NN, I + 1>(f); }
}

// ---- compile-time unit roots for any order (the window lengths R = 8 RA with RA = 5, 6, 7, 10, 12, 14 need w_q, w_RA, w_2RA) ----
// cos / sin of 2 pi num / den in double by range reduction to [0, pi/4] and Taylor series (constexpr in C++17 on both compilers)
constexpr double lw_kPi = 3.14159265358979323846264338327950288;
constexpr double lw_taylor_sin(double x) {   // |x| <= pi/4
    const double x2 = x * x;
    return x * (1.0 + x2 * (-1.0 / 6 + x2 * (1.0 / 120 + x2 * (-1.0 / 5040 + x2 * (1.0 / 362880 + x2 * (-1.0 / 39916800 + x2 * (1.0 / 6227020800.0 + x2 * (-1.0 / 1307674368000.0))))))));
}
constexpr double lw_taylor_cos(double x) {
    const double x2 = x * x;
    return 1.0 + x2 * (-0.5 + x2 * (1.0 / 24 + x2 * (-1.0 / 720 + x2 * (1.0 / 40320 + x2 * (-1.0 / 3628800 + x2 * (1.0 / 479001600.0 + x2 * (-1.0 / 87178291200.0 + x2 * (1.0 / 20922789888000.0))))))));
}
struct LwUnit { double c, s; };
// exp(-2 pi i num / den) for integers num, den > 0: octant by exact integer arithmetic, then the series on the remainder
constexpr LwUnit lw_unit(long long num, long long den) {
    num %= den; if (num < 0) num += den;
    // angle = 2 pi num / den = (pi / 4) (8 num / den): octant o = floor(8 num / den), remainder r in [0, 1)
    const long long o = (8 * num) / den;
    const double r = (double)(8 * num - o * den) / (double)den;          // fraction of an octant
    const double a = r * (lw_kPi / 4);
    const double c0 = lw_taylor_cos(a), s0 = lw_taylor_sin(a);          // angle within the octant
    const double h = 0.70710678118654752440084436210484904;
    // rotate by o octants: (c, s) of o * pi/4 + a
    double c = 0, sn = 0;
    switch (o & 7) {
        case 0: c = c0; sn = s0; break;
        case 1: c = h * (c0 - s0); sn = h * (c0 + s0); break;
        case 2: c = -s0; sn = c0; break;
        case 3: c = -h * (c0 + s0); sn = h * (c0 - s0); break;
        case 4: c = -c0; sn = -s0; break;
        case 5: c = -h * (c0 - s0); sn = -h * (c0 + s0); break;
        case 6: c = s0; sn = -c0; break;
        default: c = h * (c0 + s0); sn = -h * (c0 - s0); break;
    }
    return LwUnit{c, -sn};                                                // forward kernel: exp(-i angle)
}
// a * exp(-2 pi i NUM / DEN) (forward) or times its conjugate (INV)
template <bool INV, int NUM, int DEN> AW_HD cf lw_mul_unit(cf a) {
    constexpr int n = ((NUM % DEN) + DEN) % DEN;
    if constexpr (n == 0) return a;
    else if constexpr (4 * n == DEN) return rot90<INV>(a);
    else if constexpr (2 * n == DEN) return mk(-a.x, -a.y);
    else if constexpr (4 * n == 3 * DEN) return rot90<!INV>(a);
    else {
        constexpr LwUnit u = lw_unit(n, DEN);
        constexpr float c = (float)u.c, s = (float)(INV ? -u.s : u.s);
        return mk(a.x * c - a.y * s, a.x * s + a.y * c);
    }
}

constexpr int lw_odd_part(int n) { return n % 2 == 0 ? lw_odd_part(n / 2) : n; }
constexpr int lw_small_factor(int n) { return n % 3 == 0 ? 3 : n % 5 == 0 ? 5 : n % 7 == 0 ? 7 : n; }     // smallest odd prime factor (n odd, < 49)

// DFT of size NN over the register index, natural order in and out (NN <= 16).  Powers of two are the butterflies of cplx.hpp; an odd
// prime is a direct product; every other size splits as NN = A B, j = B ja + jb, k = ka + A kb: DFT_A over ja for every jb, twiddle
// w_NN^{jb ka}, DFT_B over jb for every ka — A the odd part of an even size (the power of two last), the smallest prime of an odd one.
template <bool INV, int NN> AW_HD void lw_fft(cf (&v)[NN]) {
    static_assert(NN >= 1 && NN <= 16, "in-register DFT");
    constexpr int q = lw_odd_part(NN);
    if constexpr (NN == 1) {
    } else if constexpr (q == 1) {
        if constexpr (NN == 16) fft16<INV>(v);
        else if constexpr (NN == 8) fft8<INV>(v);
        else if constexpr (NN == 4) fft4<INV>(v[0], v[1], v[2], v[3]);
        else { const cf a = v[0] + v[1], b = v[0] - v[1]; v[0] = a; v[1] = b; }
    } else if constexpr (q == NN && lw_small_factor(NN) == NN) {          // odd prime: direct
        cf y[NN];
        lw_unroll<NN>([&](auto K) {
            constexpr int k = K.value;
            cf acc = v[0];
            lw_unroll<NN - 1>([&](auto J) { acc = acc + lw_mul_unit<INV, (J.value + 1) * k, NN>(v[J.value + 1]); });
            y[k] = acc;
        });
        lw_unroll<NN>([&](auto K) { v[K.value] = y[K.value]; });
    } else {
        constexpr int A = q == NN ? lw_small_factor(NN) : q, B = NN / A;
        cf y[NN];                                    // y[ka * B + jb]
        lw_unroll<B>([&](auto JB) {
            constexpr int jb = JB.value;
            cf z[A];
            lw_unroll<A>([&](auto JA) { z[JA.value] = v[JA.value * B + jb]; });
            lw_fft<INV, A>(z);
            lw_unroll<A>([&](auto KA) { y[KA.value * B + jb] = lw_mul_unit<INV, jb * KA.value, NN>(z[KA.value]); });
        });
        lw_unroll<A>([&](auto KA) {
            constexpr int ka = KA.value;
            cf z[B];
            lw_unroll<B>([&](auto JB) { z[JB.value] = y[ka * B + JB.value]; });
            lw_fft<INV, B>(z);
            lw_unroll<B>([&](auto KB) { v[ka + A * KB.value] = z[KB.value]; });
        });
    }
}

// odd DFT of size NN over the register index:  v[k] <- sum_j v[j] w_NN^{j (k + 1/2)}   (INV: conjugate kernel)
template <bool INV, int NN> AW_HD void lw_odd_dft(cf (&v)[NN]) {
    if constexpr (!INV) lw_unroll<NN>([&](auto J) { v[J.value] = lw_mul_unit<false, J.value, 2 * NN>(v[J.value]); });
    lw_fft<INV, NN>(v);
    if constexpr (INV) lw_unroll<NN>([&](auto J) { v[J.value] = lw_mul_unit<true, J.value, 2 * NN>(v[J.value]); });
}

struct __attribute__((packed, aligned(4))) f3u { float x, y, z; };

// One whole interleaved frame of CS channels -> registers (dword-aligned vector loads; exactly CS floats are read).
template <int CS, bool ALIGNED = (CS % 4 == 0)> AW_HD void lw_load_frame(const float *src, float (&d)[CS]) {
    constexpr int n4 = CS / 4, rem = CS % 4;
#pragma unroll
    for (int g = 0; g < n4; ++g) {
        if constexpr (ALIGNED) {
            const f4 v = *reinterpret_cast<const f4 *>(src + 4 * g);
            d[4 * g] = v.x; d[4 * g + 1] = v.y; d[4 * g + 2] = v.z; d[4 * g + 3] = v.w;
        } else {
            const f4u v = *reinterpret_cast<const f4u *>(src + 4 * g);
            d[4 * g] = v.x; d[4 * g + 1] = v.y; d[4 * g + 2] = v.z; d[4 * g + 3] = v.w;
        }
    }
    if constexpr (rem == 3) {
        const f3u v = *reinterpret_cast<const f3u *>(src + 4 * n4);
        d[4 * n4] = v.x; d[4 * n4 + 1] = v.y; d[4 * n4 + 2] = v.z;
    } else if constexpr (rem == 2) {
        if constexpr (CS % 2 == 0 && ALIGNED == (CS % 4 == 0)) { const f2 v = *reinterpret_cast<const f2 *>(src + 4 * n4); d[4 * n4] = v.x; d[4 * n4 + 1] = v.y; }
        else { const f2u v = *reinterpret_cast<const f2u *>(src + 4 * n4); d[4 * n4] = v.x; d[4 * n4 + 1] = v.y; }
    } else if constexpr (rem == 1) {
        d[4 * n4] = src[4 * n4];
    }
}

template <int CS, bool ALIGNED = (CS % 4 == 0)> AW_HD void lw_store_frame(float *dst, const float (&d)[CS]) {
    constexpr int n4 = CS / 4, rem = CS % 4;
#pragma unroll
    for (int g = 0; g < n4; ++g) {
        if constexpr (ALIGNED) { f4 v; v.x = d[4 * g]; v.y = d[4 * g + 1]; v.z = d[4 * g + 2]; v.w = d[4 * g + 3]; *reinterpret_cast<f4 *>(dst + 4 * g) = v; }
        else { f4u v; v.x = d[4 * g]; v.y = d[4 * g + 1]; v.z = d[4 * g + 2]; v.w = d[4 * g + 3]; *reinterpret_cast<f4u *>(dst + 4 * g) = v; }
    }
    if constexpr (rem == 3) { f3u v; v.x = d[4 * n4]; v.y = d[4 * n4 + 1]; v.z = d[4 * n4 + 2]; *reinterpret_cast<f3u *>(dst + 4 * n4) = v; }
    else if constexpr (rem == 2) {
        if constexpr (CS % 2 == 0 && ALIGNED == (CS % 4 == 0)) { f2 v; v.x = d[4 * n4]; v.y = d[4 * n4 + 1]; *reinterpret_cast<f2 *>(dst + 4 * n4) = v; }
        else { f2u v; v.x = d[4 * n4]; v.y = d[4 * n4 + 1]; *reinterpret_cast<f2u *>(dst + 4 * n4) = v; }
    } else if constexpr (rem == 1) dst[4 * n4] = d[4 * n4];
}

// The small twiddle tables of the split / merge kernels live in LDS behind the exchange buffer (global loads of them — two
// per output value — were most of the kernels' vector-memory instructions and their exposed latency: 174 of 330 per tile).
// FINE_LDS = false (merge kernel): the per-lane `fine` table stays in global memory (four coalesced loads per tile), so that the
// kernel's LDS footprint — 64 KB of exchange buffer — leaves room for two workgroups per CU.
template <int RA, bool FINE_LDS = true> constexpr int lw_small_elems() { return RA * 8 + RA * 64 + (FINE_LDS ? RA * 64 : 0); }
template <int RA, bool FINE_LDS = true, class Ctx>
AW_HD void lw_small_tables(Ctx &ctx, const LwParams &p, cf *sm) {            // [twr RA x 8][coarse RA x 64][fine RA x 64]; visible after the next barrier
    for (int i = ctx.tid(); i < RA * 8; i += kThreads) sm[i] = p.tw_r[i];
    for (int i = ctx.tid(); i < RA * 64; i += kThreads) {
        sm[RA * 8 + i] = p.tw_coarse[i];
        if constexpr (FINE_LDS) sm[RA * 8 + RA * 64 + i] = p.tw_fine[i];
    }
}
// the four row twiddles base(ka) * {1, S1, S2, S3} of one ka
template <int RA, bool FINE_LDS = true, class Ctx>
AW_HD void lw_row_twiddles(Ctx &ctx, const LwParams &p, const cf *sm, int ka, int tc, int lane, const cf (&S)[3], cf (&tau)[4]) {
    const cf fine = FINE_LDS ? ctx.ld(sm + RA * 8 + RA * 64 + ka * 64 + lane) : p.tw_fine[ka * 64 + lane];
    tau[0] = cmul(ctx.ld(sm + RA * 8 + ka * 64 + tc), fine);
#pragma unroll
    for (int m = 1; m < 4; ++m) tau[m] = cmul(tau[0], S[m - 1]);
}

// ---- kernel 1: split ---------------------------------------------------------------------------------------------
// Tile id = (stream, window) * 64 + tc.  512 threads: wave = j1, lane = t - 64 tc.  LDS: [2][RA][8][64] complex.
template <int RA> constexpr int lw_split_lds_elems() { return 2 * RA * 8 * 64 + lw_small_elems<RA>(); }

// Layouts of up to eight channels (CS == p.n_channels; 9-16 channels: lw_split_wide_tiles below).
template <class Ctx, int RA, int CS>
AW_HD void lw_split_tiles(Ctx &ctx, const LwParams &p, long long first, long long step, long long end) {
    static_assert(lw_ra_ok(RA), "R = 8 RA rows");
    static_assert(CS >= 1 && CS <= 8, "up to eight channels");
    constexpr bool AL = CS % 4 == 0;
    constexpr int CF = CS, c0 = 0, pair0 = 0;
    constexpr int NP = (CS + 1) / 2, NPASS = (NP + 1) / 2, NCOMBO = (2 * RA + 7) / 8;      // (pair of the pass, ka) combinations, one per wave and round
    if (first >= end) return;
    const int lane = ctx.lane(), wave = ctx.wave();
    cf *lds = ctx.lds();
    cf *sm = lds + 2 * RA * 8 * 64;
    lw_small_tables<RA>(ctx, p, sm);
    float raw[RA][CS];
    auto load_tile = [&](long long id) {
        const long long sw = (long long)((unsigned long long)id / (unsigned)kLwChunks);
        const int tc = (int)(id - sw * kLwChunks);
        const long long stream = sw / p.n_windows;
        const int win = (int)(sw - stream * p.n_windows);
        const float *in_s = p.in + stream * p.frames * CF + c0;
        const float *hist_s = p.hist + stream * (long long)p.hist_len * CF + c0;
        const long long fb = p.frame0 + (long long)win * p.hop - p.hist_len + (long long)kLwM * wave + tc * kLwTw + lane;
#pragma unroll
        for (int j2 = 0; j2 < RA; ++j2) {
            const long long f = fb + (long long)kLwM * 8 * j2;
            // frames before the call: the history buffer; past its end: a page of zeros (a pointer select, never a select on data)
            const float *src = f < 0 ? hist_s + ((long long)p.hist_len + f) * CF : (f >= p.frames ? p.zeros : in_s + f * CF);
#ifdef AW_LW_ABL_SPLIT_NOLOAD     // timing ablation only (wrong results)
#pragma unroll
            for (int c = 0; c < CS; ++c) raw[j2][c] = 0.001f * lane + (float)(src == nullptr);
            continue;
#endif
            lw_load_frame<CS, AL>(src, raw[j2]);
        }
    };
    load_tile(first);
    for (long long id = first; id < end; id += step) {
        const long long sw = (long long)((unsigned long long)id / (unsigned)kLwChunks);
        const int tc = (int)(id - sw * kLwChunks);
        const int t = tc * kLwTw + lane;
        cf *spec_sw = p.spec + sw * p.spec_per_sw;
        cf S[3];                      // issued here: they travel under step 1
#pragma unroll
        for (int m = 0; m < 3; ++m) S[m] = p.tw_step[m * kLwM + t];
        if (p.hist_out) {             // uniform.  The next call's history: frames [frames - hist_len, frames) of (history ++ input)
            const long long stream = sw / p.n_windows;
            const int win = (int)(sw - stream * p.n_windows);
            const long long fb = p.frame0 + (long long)win * p.hop - p.hist_len + (long long)kLwM * wave + t;
            float *ho = p.hist_out + stream * (long long)p.hist_len * CF + c0;
#pragma unroll
            for (int j2 = 0; j2 < RA; ++j2) {
                const long long i = fb + (long long)kLwM * 8 * j2 - (p.frames - p.hist_len);
                if (i >= 0 && i < p.hist_len) lw_store_frame<CS, AL>(ho + i * CF, raw[j2]);
            }
        }
        lw_unroll<NPASS>([&](auto PP) {
            constexpr int pp = PP.value;
            if (pp > 0 || id != first) ctx.barrier();                 // every wave is done reading the exchange buffer
            // step 1: odd DFT over j2 (this thread's RA frames) to LDS [q][ka][j1][lane]
            lw_unroll<2>([&](auto Q) {
                constexpr int q = Q.value, pair = 2 * pp + q;
                if constexpr (pair < NP) {
                    cf x[RA];
#pragma unroll
                    for (int j2 = 0; j2 < RA; ++j2) x[j2] = mk(raw[j2][2 * pair], 2 * pair + 1 < CS ? raw[j2][(2 * pair + 1) % CS] : 0.0f);
                    lw_odd_dft<false, RA>(x);
#pragma unroll
                    for (int ka = 0; ka < RA; ++ka)
                        lds[((q * RA + ka) * 8 + wave) * 64 + lane] = x[ka];
                }
            });
            if constexpr (pp == NPASS - 1) load_tile(id + step < end ? id + step : id);     // the frames are consumed: fetch the next tile's (the last one re-reads its own)
            ctx.barrier();
            // step 3: twiddle w_R^{j1 (ka + 1/2)}, DFT-8 over j1 -> kb, row k1 = RA kb + ka; row twiddle; rows to scratch (512
            // contiguous bytes per wave and row).  Row twiddles: lower rows (kb < 4) base(ka) S^kb, upper rows — conj-reversed,
            // the partner row R-1-k1 = RA (7 - kb) + (RA-1-ka) — base(RA-1-ka) S^(7-kb).
#pragma unroll
            for (int i = 0; i < NCOMBO; ++i) {
                const int combo = wave + 8 * i;                                 // uniform: (q, ka) = (combo / RA, combo % RA)
                if (combo >= 2 * RA) continue;
                const int q = combo / RA, ka = combo - q * RA;
                const int pair = 2 * pp + q;
                if (pair >= NP) continue;
                cf v[8];
#pragma unroll
                for (int j1 = 0; j1 < 8; ++j1) v[j1] = ctx.ld(lds + ((q * RA + ka) * 8 + j1) * 64 + lane);
#pragma unroll
                for (int j1 = 1; j1 < 8; ++j1) v[j1] = cmul(v[j1], ctx.ld(sm + ka * 8 + j1));
                fft8<false>(v);
                cf tlo[4], tup[4];
                lw_row_twiddles<RA>(ctx, p, sm, ka, tc, lane, S, tlo);
                lw_row_twiddles<RA>(ctx, p, sm, RA - 1 - ka, tc, lane, S, tup);
                const bool real_pair = (CS & 1) && pair == NP - 1;        // (an odd layout's last group carries its real last channel)
                cf *dst = spec_sw + (long long)(pair0 + pair) * p.N + t;
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) {
                    const int k1 = RA * kb + ka;
#ifdef AW_LW_ABL_SPLIT_NOSTORE    // timing ablation only (wrong results)
                    if (v[kb].x != 1.2345e-30f) continue;
#endif
                    if (kb < 4) {
                        ctx.st_stream(dst + (long long)k1 * kLwM, cmul(v[kb], tlo[kb & 3]));
                    } else if (!real_pair) {
                        ctx.st_stream(dst + (long long)k1 * kLwM, cmul(conj(v[kb]), tup[(7 - kb) & 3]));
                    }
                }
            }
        });
    }
}


// Layouts of 9-16 channels in ONE launch: a wave covers 32 consecutive frames t and both halves of their channels — lanes
// 0-31 channels 0-7 (pairs 0-3), lanes 32-63 channels 8..C-1 (pairs 4..) — so every 128-byte line of the input crosses the
// fabric once (one launch per group of eight channels read every line twice: cfg 3 with 14 channels 16.3 against 11.8 ms).
// Every lane issues the same two 16-byte loads per frame; the second half's may run up to 3 floats past its frame (into the
// next frame, the history buffer's slack or the zero page — values that are never used); the one place that would leave the
// caller's buffer, the last frame of the last stream, is read from a padded copy (p.tail).  Tile id = (stream, window) * 128 + tc.
constexpr int kLwTwW = 32;
constexpr int kLwChunksW = kLwM / kLwTwW;

template <class Ctx, int RA, int CS1>
AW_HD void lw_split_wide_tiles(Ctx &ctx, const LwParams &p, long long first, long long step, long long end) {
    static_assert(lw_ra_ok(RA), "R = 8 RA rows");
    static_assert(CS1 >= 1 && CS1 <= 8, "channels 8 .. 8 + CS1 - 1 in the second half");
    constexpr int C = 8 + CS1, NP1 = (CS1 + 1) / 2, NPASS = 2, NCOMBO = (2 * RA + 7) / 8;
    if (first >= end) return;
    const int lane = ctx.lane(), wave = ctx.wave();
    const int tl = lane & 31, half = lane >> 5;
    const int nph = half ? NP1 : 4;                      // pairs this lane's half holds
    cf *lds = ctx.lds();
    cf *sm = lds + 2 * RA * 8 * 64;
    lw_small_tables<RA>(ctx, p, sm);
    float raw[RA][8];
    auto load_tile = [&](long long id) {
        const long long sw = (long long)((unsigned long long)id / (unsigned)kLwChunksW);
        const int tc = (int)(id - sw * kLwChunksW);
        const long long stream = sw / p.n_windows;
        const int win = (int)(sw - stream * p.n_windows);
        const float *in_s = p.in + stream * p.frames * C + 8 * half;
        const float *hist_s = p.hist + stream * (long long)p.hist_len * C + 8 * half;
        const bool last_stream = stream == p.n_streams - 1;
        const long long fb = p.frame0 + (long long)win * p.hop - p.hist_len + (long long)kLwM * wave + tc * kLwTwW + tl;
#pragma unroll
        for (int j2 = 0; j2 < RA; ++j2) {
            const long long f = fb + (long long)kLwM * 8 * j2;
            const float *src = f < 0 ? hist_s + ((long long)p.hist_len + f) * C : (f >= p.frames ? p.zeros : in_s + f * C);
            if (last_stream && f == p.frames - 1) src = p.tail + 8 * half;
            // second half with at most four channels: its second load has nothing to fetch
            const float *src2 = (CS1 <= 4 && half) ? p.zeros : src + 4;
            const f4u a = *reinterpret_cast<const f4u *>(src);
            const f4u b = *reinterpret_cast<const f4u *>(src2);
            raw[j2][0] = a.x; raw[j2][1] = a.y; raw[j2][2] = a.z; raw[j2][3] = a.w;
            raw[j2][4] = b.x; raw[j2][5] = b.y; raw[j2][6] = b.z; raw[j2][7] = b.w;
        }
    };
    load_tile(first);
    for (long long id = first; id < end; id += step) {
        const long long sw = (long long)((unsigned long long)id / (unsigned)kLwChunksW);
        const int tc = (int)(id - sw * kLwChunksW);
        const int t = tc * kLwTwW + tl;
        const int tc64 = tc >> 1, lane64 = 32 * (tc & 1) + tl;      // the twiddle tables' (64-frame chunk, lane) coordinates of t
        cf *spec_sw = p.spec + sw * p.spec_per_sw;
        cf S[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) S[m] = p.tw_step[m * kLwM + t];
        if (p.hist_out) {             // uniform.  The next call's history: frames [frames - hist_len, frames) of (history ++ input)
            const long long stream = sw / p.n_windows;
            const int win = (int)(sw - stream * p.n_windows);
            const long long fb = p.frame0 + (long long)win * p.hop - p.hist_len + (long long)kLwM * wave + t;
            float *ho = p.hist_out + stream * (long long)p.hist_len * C + 8 * half;
#pragma unroll
            for (int j2 = 0; j2 < RA; ++j2) {
                const long long i = fb + (long long)kLwM * 8 * j2 - (p.frames - p.hist_len);
                if (i >= 0 && i < p.hist_len) {
                    if (half == 0) lw_store_frame<8, false>(ho + i * C, raw[j2]);
                    else {
                        float d[CS1];
#pragma unroll
                        for (int c = 0; c < CS1; ++c) d[c] = raw[j2][c];
                        lw_store_frame<CS1, false>(ho + i * C, d);
                    }
                }
            }
        }
        lw_unroll<NPASS>([&](auto PP) {
            constexpr int pp = PP.value;
            if (pp > 0 || id != first) ctx.barrier();
            // step 1: local pairs 2 pp, 2 pp + 1 of this lane's half (a half without that pair transforms zeros / an unused value)
            lw_unroll<2>([&](auto Q) {
                constexpr int q = Q.value, pair = 2 * pp + q;
                cf x[RA];
                // the second channel of a real last pair (odd channel count) must be a true zero
                const bool has_b = half == 0 || 2 * pair + 1 < CS1;
#pragma unroll
                for (int j2 = 0; j2 < RA; ++j2) x[j2] = mk(raw[j2][2 * pair], has_b ? raw[j2][2 * pair + 1] : 0.0f);
                lw_odd_dft<false, RA>(x);
#pragma unroll
                for (int ka = 0; ka < RA; ++ka) lds[((q * RA + ka) * 8 + wave) * 64 + lane] = x[ka];
            });
            if constexpr (pp == NPASS - 1) load_tile(id + step < end ? id + step : id);
            ctx.barrier();
#pragma unroll
            for (int i = 0; i < NCOMBO; ++i) {
                const int combo = wave + 8 * i;                                 // uniform
                if (combo >= 2 * RA) continue;
                const int q = combo / RA, ka = combo - q * RA;
                const int pair = 2 * pp + q;                                    // local to the half
                cf v[8];
#pragma unroll
                for (int j1 = 0; j1 < 8; ++j1) v[j1] = ctx.ld(lds + ((q * RA + ka) * 8 + j1) * 64 + lane);
#pragma unroll
                for (int j1 = 1; j1 < 8; ++j1) v[j1] = cmul(v[j1], ctx.ld(sm + ka * 8 + j1));
                fft8<false>(v);
                cf tlo[4], tup[4];
                lw_row_twiddles<RA>(ctx, p, sm, ka, tc64, lane64, S, tlo);
                lw_row_twiddles<RA>(ctx, p, sm, RA - 1 - ka, tc64, lane64, S, tup);
                const bool live = pair < nph;                                   // per lane half
                const bool real_pair = (CS1 & 1) && half == 1 && pair == NP1 - 1;
                cf *dst = spec_sw + (long long)(4 * half + pair) * p.N + t;
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) {
                    const int k1 = RA * kb + ka;
                    if (kb < 4) {
                        if (live) ctx.st_stream(dst + (long long)k1 * kLwM, cmul(v[kb], tlo[kb & 3]));
                    } else if (live && !real_pair) {
                        ctx.st_stream(dst + (long long)k1 * kLwM, cmul(conj(v[kb]), tup[(7 - kb) & 3]));
                    }
                }
            }
        });
    }
}

// ---- kernel 2: rows ------------------------------------------------------------------------------------------------
// One 512-point DFT over the lane dimension (sub_fft512x2 of tile_ols.hpp with one row).
template <bool INV, class Ctx>
AW_HD void lw_sub_fft512(Ctx &ctx, cf (&a)[8], cf *scr, const cf *twa, const cf *twb, int lane) {
    fft8<INV>(a);
#pragma unroll
    for (int ka = 1; ka < 8; ++ka) a[ka] = twmul<INV>(a[ka], ctx.ld(twa + ka * 64 + lane));
    const int l0 = lane & 7, kap = lane >> 3;
#pragma unroll
    for (int ka = 0; ka < 8; ++ka) scr[ka * 72 + lane] = a[ka];
    ctx.wave_sync();
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = ctx.ld(scr + kap * 72 + l0 + 8 * i);
    ctx.wave_sync();
    fft8<INV>(a);
#pragma unroll
    for (int kb = 1; kb < 8; ++kb) a[kb] = twmul<INV>(a[kb], ctx.ld(twb + kb * 8 + l0));
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) scr[(kb * 8 + kap) * 9 + l0] = a[kb];
    ctx.wave_sync();
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = ctx.ld(scr + lane * 9 + i);
    ctx.wave_sync();
    fft8<INV>(a);
}

// w^0 .. w^7 by a depth-3 product tree
AW_HD void lw_powers8(cf w, cf (&pw)[8]) {
    pw[0] = mk(1.f, 0.f);
    pw[1] = w;
    pw[2] = cmul(w, w);
    pw[3] = cmul(pw[2], w);
    pw[4] = cmul(pw[2], pw[2]);
    pw[5] = cmul(pw[4], w);
    pw[6] = cmul(pw[4], pw[2]);
    pw[7] = cmul(pw[4], pw[3]);
}

// Virtual tile id vid -> (row pair, stream-window): rp = rp0 + rp_step * (vid / n_sw), sw = vid % n_sw.  The GPU kernel
// gives every XCD the row pairs rp = xcd (mod 8), walked one after the other: the 32 workgroups of an XCD work on the
// same row pair at a time and its 512 KB table slice stays in their L2.
// Window lengths whose row-pair count is not a multiple of the group count (R = 40, 56, 72, 120: R/2 = 4 mod 8) leave `left` row
// pairs over after the whole rounds; those are dealt to ALL groups in equal shares of their (row pair, stream-window) tiles — group g
// takes the tiles [g L / groups, (g + 1) L / groups) of the L left-over tiles in row-pair-major order, so it still works on one table slice at a
// time (two at most) — instead of handing whole pairs to the first `left` groups (which then ran 8 row pairs against 7 at R = 120,
// and the launch as long as the fullest group: round-4 advisor finding).
struct LwRowTile { int rp; long long sw; };
struct LwRowMap { long long n_sw, full, left_lo, left_hi; int rp0, rp_step, rp_left; };
AW_HD LwRowMap lw_row_map(long long n_sw, int n_rp, int rp0, int rp_step) {
    LwRowMap m;
    const int rounds = n_rp / rp_step, left = n_rp % rp_step;
    m.n_sw = n_sw; m.rp0 = rp0; m.rp_step = rp_step;
    m.full = (long long)rounds * n_sw;
    m.rp_left = rounds * rp_step;
    const long long left_tiles = (long long)left * n_sw;
    m.left_lo = left_tiles * rp0 / rp_step;               // group g: tiles [g L / groups, (g + 1) L / groups) of the left-over pairs
    m.left_hi = left_tiles * (rp0 + 1) / rp_step;
    return m;
}
// virtual tiles of this group: its whole rounds, then its share of the left-over row pairs
AW_HD long long lw_row_count(const LwRowMap &m) { return m.full + (m.left_hi - m.left_lo); }
AW_HD LwRowTile lw_row_tile(const LwRowMap &m, long long vid) {
    LwRowTile r;
    if (vid < m.full) {
        const long long q = vid / m.n_sw;
        r.rp = m.rp0 + m.rp_step * (int)q;
        r.sw = vid - q * m.n_sw;
    } else {
        const long long l = m.left_lo + (vid - m.full);
        const long long q = l / m.n_sw;
        r.rp = m.rp_left + (int)q;
        r.sw = l - q * m.n_sw;
    }
    return r;
}

// PB = channel pairs per batch: 2 = two exchange buffers (152 KB of LDS, one workgroup per CU); 1 = one buffer (78 KB) and
// half the register batch, so that two workgroups share a CU and one's barriers and exposed latencies meet the other's work.
template <int PB> constexpr int lw_rows_lds_elems() { return PB * kBufElems + kTwaElems + kTwbElems; }

template <class Ctx, int NP, bool REAL_LAST, int PB = 2>
AW_HD void lw_rows_tiles(Ctx &ctx, const LwParams &p, long long first, long long step, long long n_sw, int rp0, int rp_step) {
    static_assert(NP >= 1 && NP <= 8, "channel pairs");
    static_assert(PB == 1 || PB == 2, "pairs per batch");
    constexpr int NB = (NP + PB - 1) / PB;
    const LwRowMap rmap = lw_row_map(n_sw, p.R / 2, rp0, rp_step);
    const long long end = lw_row_count(rmap);
    if (first >= end) return;
    const int t0 = ctx.tid();
    int t = t0, lane = ctx.lane();
    const int wave = ctx.wave();
    cf *buf0 = ctx.lds();
    cf *buf1 = buf0 + (PB - 1) * kBufElems;
    cf *twa = buf0 + PB * kBufElems;
    cf *twb = twa + kTwaElems;
    const cf w1 = p.tw1m[t];
    twa[t] = p.twa[t];
    if (t < kTwbElems) twb[t] = p.twb[t];
    const int R = p.R;

    cf raw[PB][2][8];         // [pair of the batch][row ra / rb][j2]: row samples t + 512 j2
    auto load_batch = [&](long long vid, int b) {
        const LwRowTile tl = lw_row_tile(rmap, vid);
        const cf *spec_sw = p.spec + tl.sw * p.spec_per_sw;
#pragma unroll
        for (int h = 0; h < PB; ++h) {
            const int pair = PB * b + h;
            if (pair >= NP) break;                                     // compile-time after unrolling
            const cf *base = spec_sw + (long long)pair * p.N + t;
#pragma unroll
#ifdef AW_LW_ABL_ROWS_NOLOAD      // timing ablation only (wrong results)
            for (int j2 = 0; j2 < 8; ++j2) { raw[h][0][j2] = mk(0.001f * t, (float)(base == nullptr)); raw[h][1][j2] = mk(0.002f * j2, 1.f); }
            continue;
#endif
            for (int j2 = 0; j2 < 8; ++j2) raw[h][0][j2] = ctx.ld_stream(base + (long long)tl.rp * kLwM + 512 * j2);
            if (!(REAL_LAST && pair == NP - 1)) {
#pragma unroll
                for (int j2 = 0; j2 < 8; ++j2) raw[h][1][j2] = ctx.ld_stream(base + (long long)(R - 1 - tl.rp) * kLwM + 512 * j2);
            }
        }
    };
    load_batch(first, 0);
    for (long long vid = first; vid < end; vid += step) {
        t = ctx.opaque_i(t0);                          // keeps lane-dependent addresses from living across the tile loop
        lane = t & 63;
        const LwRowTile tl = lw_row_tile(rmap, vid);
        cf wacc[2][8];                                 // W1, W2 of this wave's inner row: bins q2 = lane + 64 kc
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i) wacc[s][i] = mk(0.f, 0.f);

        lw_unroll<NB>([&](auto BB) {
            constexpr int b = BB.value;
            if (b > 0) ctx.barrier();                  // every wave is done reading buf0 / buf1
            {   // pass 1: radix-8 over j2 -> q1, twiddle w_4096^{t q1}; rows 0-7 of a buffer = row ra, rows 8-15 = row rb
                cf pw[8];
                lw_powers8(ctx.opaque(w1), pw);
#pragma unroll
                for (int h = 0; h < PB; ++h) {
                    constexpr int pair0 = PB * b;
                    if (pair0 + h >= NP) break;
                    cf *buf = h == 0 ? buf0 : buf1;
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        if (r == 1 && REAL_LAST && pair0 + h == NP - 1) break;
                        cf x[8];
#pragma unroll
                        for (int j2 = 0; j2 < 8; ++j2) x[j2] = raw[h][r][j2];
                        fft8<false>(x);
#pragma unroll
                        for (int q1 = 1; q1 < 8; ++q1) x[q1] = cmul(x[q1], pw[q1]);
#pragma unroll
                        for (int q1 = 0; q1 < 8; ++q1) buf[(8 * r + q1) * kRowStride + t] = x[q1];
                    }
                }
            }
            // the next batch's rows travel under this batch's sub-FFTs
            if constexpr (b + 1 < NB) load_batch(vid, b + 1);
            ctx.barrier();
#pragma unroll
            for (int h = 0; h < PB; ++h) {
                constexpr int pair0 = PB * b;
                const int pair = pair0 + h;
                if (pair >= NP) break;
                cf *buf = h == 0 ? buf0 : buf1;
                cf *row0 = buf + wave * kRowStride, *row1 = buf + (8 + wave) * kRowStride;
                const LwTab *tb = p.tab + (((long long)tl.rp * NP + pair) * kLwInner + wave) * kSub + lane;
                if (REAL_LAST && pair == NP - 1) {
                    // real channel: V == Z1; the host has folded T1 into T0 and T2 into T3
                    cf z[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) z[i] = ctx.ld(row0 + lane + 64 * i);
                    ctx.wave_sync();
                    lw_sub_fft512<false>(ctx, z, row0, twa, twb, lane);
#pragma unroll
                    for (int kc = 0; kc < 8; ++kc) {
#ifdef AW_LW_ABL_ROWS_NOTAB       // timing ablation only (wrong results)
                        const LwTab T{mk(1.f, 0.5f * lane), mk(0.f, 0.f), mk(0.f, 0.f), mk(0.25f * kc, 1.f * wave)};
#else
                        const LwTab T = tb[64 * kc];
#endif
                        wacc[0][kc] = cfma(z[kc], T.t0, wacc[0][kc]);
                        wacc[1][kc] = cfma(z[kc], T.t3, wacc[1][kc]);
                    }
                } else {
                    cf z[2][8];
                    ctx.template ld8x2<64>(z[0], row0 + lane, z[1], row1 + lane);
                    ctx.wave_sync();
                    sub_fft512x2<false>(ctx, z, row0, row1, twa, twb, lane);
#pragma unroll
                    for (int kc = 0; kc < 8; ++kc) {
#ifdef AW_LW_ABL_ROWS_NOTAB       // timing ablation only (wrong results)
                        const LwTab T{mk(1.f, 0.5f * lane), mk(0.3f, 0.1f), mk(0.2f * pair, 0.7f), mk(0.25f * kc, 1.f * wave)};
#else
                        const LwTab T = tb[64 * kc];
#endif
                        wacc[0][kc] = cfma(z[0][kc], T.t0, wacc[0][kc]);
                        wacc[0][kc] = cfma(z[1][kc], T.t1, wacc[0][kc]);
                        wacc[1][kc] = cfma(z[1][kc], T.t2, wacc[1][kc]);
                        wacc[1][kc] = cfma(z[0][kc], T.t3, wacc[1][kc]);
                    }
                }
            }
        });
        // inverse: per-wave 512-point inverse sub-FFTs of (W1, W2), rows published in buf0 (rows w and 8 + w are this wave's
        // own until the barrier), then radix-8 across the inner rows
        {
            cf *row0 = buf0 + wave * kRowStride, *row1 = buf0 + (8 + wave) * kRowStride;
            ctx.wave_sync();                                  // this wave's forward reads of its rows are complete
            sub_fft512x2<true>(ctx, wacc, row0, row1, twa, twb, lane);
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) { row0[lane + 64 * kc] = wacc[0][kc]; row1[lane + 64 * kc] = wacc[1][kc]; }
        }
        // the next tile's first batch travels under the final pass only: fetched a batch earlier — after the last pass 1 or before the
        // inverse sub-FFTs — it costs 104-540 B of scratch per thread and measures slower (5.6 -> 6.7 ms on cfg 3)
        load_batch(vid + step < end ? vid + step : vid, 0);
        ctx.barrier();
        {
            cf pw[8];
            lw_powers8(ctx.opaque(w1), pw);
            cf *dst = p.wrows + (tl.sw * (R / 2) + tl.rp) * (long long)(2 * kLwM) + t;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                cf y[8];
#pragma unroll
                for (int q1 = 0; q1 < 8; ++q1) y[q1] = ctx.ld(buf0 + (8 * r + q1) * kRowStride + t);
#pragma unroll
                for (int q1 = 1; q1 < 8; ++q1) y[q1] = cmulc(y[q1], pw[q1]);
                fft8<true>(y);
#pragma unroll
                for (int j2 = 0; j2 < 8; ++j2) {
#ifdef AW_LW_ABL_ROWS_NOSTORE     // timing ablation only (wrong results)
                    if (y[j2].x != 1.2345e-30f) continue;
#endif
                    ctx.st_stream(dst + r * kLwM + 512 * j2, y[j2]);
                }
            }
        }
        ctx.barrier();                                        // the final exchange has been read before buf0 is rewritten
    }
}

// ---- kernel 3: merge -------------------------------------------------------------------------------------------------
// Tile id = (stream, window) * 64 + tc.  512 threads.  LDS: [RA][8][64] complex.
template <int RA> constexpr int lw_merge_lds_elems() { return RA * 8 * 64 + lw_small_elems<RA, false>(); }

template <class Ctx, int RA>
AW_HD void lw_merge_tiles(Ctx &ctx, const LwParams &p, long long first, long long step, long long end) {
    static_assert(lw_ra_ok(RA), "R = 8 RA rows");
    constexpr int R = 8 * RA, NKA = (RA + 7) / 8;
    const int lane = ctx.lane(), wave = ctx.wave();
    cf *lds = ctx.lds();
    cf *sm = lds + RA * 8 * 64;
    lw_small_tables<RA, false>(ctx, p, sm);
    if (first < end) ctx.barrier();
    for (long long id = first; id < end; id += step) {
        const long long sw = (long long)((unsigned long long)id / (unsigned)kLwChunks);
        const int tc = (int)(id - sw * kLwChunks);
        const int t = tc * kLwTw + lane;
        const long long stream = sw / p.n_windows;
        const int win = (int)(sw - stream * p.n_windows);
        const cf *wr = p.wrows + sw * (long long)p.N + t;
        if (id != first) ctx.barrier();
        // step 1: rows k1 = RA kb + ka, kb = 0..7: untwiddle (upper rows: conj-reversed form), inverse DFT-8 over kb -> j1
#pragma unroll
        for (int i = 0; i < NKA; ++i) {
            const int ka = wave + 8 * i;
            if (ka >= RA) break;                                   // RA = 4: waves 4-7 only take part in step 2
            cf g[8];
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                const int k1 = RA * kb + ka;
                const int rp = kb < 4 ? k1 : R - 1 - k1;
                g[kb] = ctx.ld_stream(wr + ((long long)rp * 2 + (kb < 4 ? 0 : 1)) * kLwM);
            }
            cf S[3], tlo[4], tup[4];
#pragma unroll
            for (int m = 0; m < 3; ++m) S[m] = p.tw_step[m * kLwM + t];
            lw_row_twiddles<RA, false>(ctx, p, sm, ka, tc, lane, S, tlo);
            lw_row_twiddles<RA, false>(ctx, p, sm, RA - 1 - ka, tc, lane, S, tup);
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {          // rows of the lower half: s1 of row pair k1; upper half: s2 of row pair R-1-k1, conj-reversed
                const cf v = cmulc(g[kb], kb < 4 ? tlo[kb & 3] : tup[(7 - kb) & 3]);
                g[kb] = kb < 4 ? v : conj(v);
            }
            fft8<true>(g);
            lds[(ka * 8) * 64 + lane] = g[0];
#pragma unroll
            for (int j1 = 1; j1 < 8; ++j1) lds[(ka * 8 + j1) * 64 + lane] = cmulc(g[j1], ctx.ld(sm + ka * 8 + j1));
        }
        ctx.barrier();
        // step 2: wave = j1; inverse odd DFT over ka -> j2; frame 4096 (j1 + 8 j2) + t of the window
        cf F[RA];
#pragma unroll
        for (int ka = 0; ka < RA; ++ka) F[ka] = ctx.ld(lds + (ka * 8 + wave) * 64 + lane);
        lw_odd_dft<true, RA>(F);
        const long long f0 = p.frame0 + (long long)win * p.hop - p.hist_len;
#pragma unroll
        for (int j2 = 0; j2 < RA; ++j2) {
            const int n = kLwM * (wave + 8 * j2) + t;
            const long long f = f0 + n;
            if (n >= p.hist_len && f < p.frame_end) ctx.st_stream(reinterpret_cast<cf *>(p.out + (stream * p.frames + f) * 2), F[j2]);
        }
    }
}

}  // namespace awk
