// tile_olsh.hpp — EXPERIMENTAL (opt-in: AW_KERNEL_H=2): the 8192-frame overlap-save tile of tile_ols.hpp, re-cut so
// that TWO workgroups fit on a CU (four waves per SIMD instead of two).
//
// Same mathematics as tiles_fused_ols (ConvolutionEngine.swift:232-367 + the RealtimeAudioProcessor.swift:152-163
// downmix, see tile_ols.hpp), same tables, same twiddles, same results to rounding.  What changes is how the
// FFT_8192 is cut: the outermost stage is a radix-2 decimation in frequency,
//
//     Z[2m + q] = FFT_4096( (z[n] + (-1)^q z[n + 4096]) * W_8192^{n q} )[m] ,      q = 0 (even bins), 1 (odd bins)
//
// A half is a 4096-point problem: 8 rows of 512 (row r of half q is row 2r + q of tile_ols.hpp's [16][512] layout,
// so pair tables and partner-bin rules carry over), eight points per thread instead of sixteen, a 36 KB exchange
// buffer instead of 72 KB.  With two pairs in flight (one per buffer) a workgroup needs 78 KB of LDS and — without
// the register-resident prefetch of the one-workgroup-per-CU kernel — 128 VGPRs.
//
//   per batch of two pairs:
//     load 16 frames x 4 channels per thread ; fold n / n + 4096 (q = 1: times W_8192^n) ;
//     barrier ; pass 1 = radix-8 over j -> r, twiddle W_4096^{t r}, scatter to rows (pair 0 -> buf0, pair 1 -> buf1) ;
//     barrier ; wave w: 512-point sub-FFTs of row w of both pairs (sub_fft512x2) ; publish Z rows ; barrier ;
//     W_q += Z A + conj(Z[N-k]) B   (partner row: (8 - r) & 7 for q = 0, 7 - r for q = 1)
//   inverse: sub-FFT of W_q's row w ; barrier ; radix-8 across rows ;
//     y[n] = E[n] + conj(W_8192^n) O[n],  y[n + 4096] = E[n] - conj(W_8192^n) O[n]   (E, O: the halves' inverses)
//
// Measured (cfg 2): parity-green but 1.82 ms per launch against 1.45 ms for tile_ols.hpp; it wins
// ~10 % only for 12- and 14-channel layouts with long HRIRs (tools/kernel_sweep.py).  Kept behind AW_KERNEL_H=2.
#pragma once
#include "tile_ols.hpp"

namespace awk {

constexpr int kHRows = 8;                                   // rows of one half-spectrum
constexpr int kHBufElems = kHRows * kRowStride;             // one exchange buffer (36,864 B)
constexpr int kHLdsElems = 2 * kHBufElems + kTwaElems + kTwbElems;
constexpr int kHLdsBytes = kHLdsElems * 8;                  // 78,336 B: two workgroups per CU

// w^0 .. w^7 by a product tree
AW_HD void tw_powers8(cf w, cf (&pw)[8]) {
    pw[0] = mk(1.f, 0.f);
    pw[1] = w;
    pw[2] = cmul(w, w);
    pw[3] = cmul(pw[2], w);
    pw[4] = cmul(pw[2], pw[2]);
    pw[5] = cmul(pw[4], w);
    pw[6] = cmul(pw[4], pw[2]);
    pw[7] = cmul(pw[4], pw[3]);
}

// a * W_16^J (forward) or a * conj(W_16^J) (INV), J = 0..7
template <bool INV, int J> AW_HD cf mul_w16j(cf a) {
    if constexpr (J == 0) return a;
    else if constexpr (J == 4) return rot90<INV>(a);
    else if constexpr (J == 2) return mul_w8_1<INV>(a);
    else if constexpr (J == 6) return mul_w8_3<INV>(a);
    else {
        constexpr float wr = (J == 1) ? kC8 : (J == 3) ? kS8 : (J == 5) ? -kS8 : -kC8;
        constexpr float wf = (J == 1) ? -kS8 : (J == 3) ? -kC8 : (J == 5) ? -kC8 : -kS8;
        const float wi = INV ? -wf : wf;
        return mk(a.x * wr - a.y * wi, a.x * wi + a.y * wr);
    }
}

template <bool INV, int Q> AW_HD void half_twist(cf (&x)[8], cf w1) {
    // x[j] *= W_8192^{(t + 512 j) Q}  (conjugated for INV): W_16^j by constants, W_8192^t from the table
    if constexpr (Q == 1) {
        x[0] = twmul<INV>(x[0], w1);
        x[1] = twmul<INV>(mul_w16j<INV, 1>(x[1]), w1);
        x[2] = twmul<INV>(mul_w16j<INV, 2>(x[2]), w1);
        x[3] = twmul<INV>(mul_w16j<INV, 3>(x[3]), w1);
        x[4] = twmul<INV>(mul_w16j<INV, 4>(x[4]), w1);
        x[5] = twmul<INV>(mul_w16j<INV, 5>(x[5]), w1);
        x[6] = twmul<INV>(mul_w16j<INV, 6>(x[6]), w1);
        x[7] = twmul<INV>(mul_w16j<INV, 7>(x[7]), w1);
    }
}

// Frames t + 512 j and t + 512 (j + 8) for j = J0 .. J0+3 (four channels from c0): the folding partners of the
// radix-2 stage, eight loads in flight at a time (32 VGPRs) instead of load_batch's sixteen.
template <int CS, bool INTERIOR, int J0>
AW_HD void load_fold4(const TileParams &p, const float *in_s, const float *hist_s, long long f0, int t, int c0,
                      float (&lo)[4][kBatchCh], float (&hi)[4][kBatchCh]) {
    if constexpr (INTERIOR && CS > 0) {
        const float *lane_base = in_s + f0 * CS + c0;          // uniform
        const int lane_off = t * CS;                            // per lane, 32-bit
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int j = J0 + (jj & 3) + 8 * (jj >> 2);
            float(&dst)[kBatchCh] = (jj < 4) ? lo[jj & 3] : hi[jj & 3];
            const float *src = lane_base + (long long)j * 512 * CS + lane_off;
            if constexpr (CS % 4 == 0) {
                const f4 v = *reinterpret_cast<const f4 *>(src);
                dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
            } else if constexpr (CS == 2) {
                const f2 v = *reinterpret_cast<const f2 *>(src);
                dst[0] = v.x; dst[1] = v.y; dst[2] = 0.f; dst[3] = 0.f;
            } else {
                const f4u v = *reinterpret_cast<const f4u *>(src);     // see load_batch: stray lanes meet zero tables
                dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
            }
        }
    } else {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            load_frame<CS>(p, in_s, hist_s, f0 + t + 512 * (J0 + jj), lo[jj], c0);
            load_frame<CS>(p, in_s, hist_s, f0 + t + 512 * (J0 + jj + 8), hi[jj], c0);
        }
    }
}


// ---- sibling form: the two half-spectra of a tile go to TWO workgroups ------------------------------------------
// A first form ran both halves one after the other in one workgroup (removed): each of its four passes over a tile's
// frames (2 batches x 2 halves) touches every 128-byte line of the 256 KB window, 64 tiles are in flight per XCD, and
// the 4 MB L2 cannot hold them — HBM-side reads grew 2.8x (rocprofv3 PMC, cfg 2), 1.97 ms per launch.  Here workgroup
// 2s of an XCD takes the even bins (q = 0) and workgroup 2s + 1 the odd bins (q = 1) of the SAME tile at the same
// time: 32 tiles in flight per XCD, two passes each, the sibling's loads hit the lines its partner just fetched.
// The halves meet only
// in the output, y = E +- conj(W^n) O: the even workgroup stores E and raises the tile's flag, the odd one waits
// for the flag and adds its term (it is dispatched after its sibling — higher workgroup id — so the wait cannot
// deadlock).  CAVEAT: the cheap flag protocol (gpu_ctx.hpp, AW_SIB_SYNC=1) relies on both siblings sharing one L2,
// i.e. on workgroup id % 8 selecting the XCD; agent-scope fences (AW_SIB_SYNC=2) are portable but write back and
// invalidate the whole L2 on gfx950 (7.5 ms per launch).  Guards: the runtime enables these kernels only after a probe
// launch of the same shape has shown blocks b and b + 8 on one XCD (kernels.hip: probe_sibling_placement, HW_REG_XCC_ID),
// and every flag wait is bounded — a timeout sets the launch's error word, which the next call turns into an error.
template <bool INV, class Ctx>
AW_HD void sub_fft512x1(Ctx &ctx, cf (&a)[8], cf *scr, const cf *twa, const cf *twb, int lane) {
    fft8<INV>(a);
#pragma unroll
    for (int ka = 1; ka < 8; ++ka) a[ka] = twmul<INV>(a[ka], ctx.ld(twa + ka * 64 + lane));
    const int l0 = lane & 7, kap = lane >> 3;
#pragma unroll
    for (int ka = 0; ka < 8; ++ka) scr[ka * 72 + lane] = a[ka];
    ctx.wave_sync();
#pragma unroll
    for (int l1 = 0; l1 < 8; ++l1) a[l1] = ctx.ld(scr + kap * 72 + l0 + 8 * l1);
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

template <int J0>
AW_HD void fold4s(const float (&lo)[4][kBatchCh], const float (&hi)[4][kBatchCh], float sgn, cf (&u0)[8], cf (&u1)[8]) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        u0[J0 + jj] = mk(lo[jj][0] + sgn * hi[jj][0], lo[jj][1] + sgn * hi[jj][1]);     // sgn = +-1: exact
        u1[J0 + jj] = mk(lo[jj][2] + sgn * hi[jj][2], lo[jj][3] + sgn * hi[jj][3]);
    }
}

AW_HD void sib_pass1(cf (&x)[8], const cf (&pw)[8], cf *buf, int t) {
    fft8<false>(x);
#pragma unroll
    for (int r = 1; r < 8; ++r) x[r] = cmul(x[r], pw[r]);
#pragma unroll
    for (int r = 0; r < 8; ++r) buf[r * kRowStride + t] = x[r];
}

template <int q, class Ctx>
AW_HD void sib_cmac(Ctx &ctx, const cf (&z)[8], const cf2 (&tab)[8], const cf *buf, int lane, int wave, cf (&wq)[8]) {
    const int prow = q ? 7 - wave : (8 - wave) & 7;
    const cf *pr = buf + prow * kRowStride;
    const int bidx = 511 - lane + ((q == 0 && wave == 0) ? 1 : 0);
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) {
        int idx = bidx - 64 * kc;
        if (kc == 0) idx &= 511;
        const cf zp = ctx.ld(pr + idx);
        wq[kc] = cfma(z[kc], tab[kc].a, wq[kc]);
        wq[kc] = cfmac(zp, tab[kc].b, wq[kc]);
    }
}

template <int q> AW_HD void sib_load_tab(const TileParams &p, int pair, int wave, int lane, cf2 (&tab)[8]) {
#ifdef AW_ABL_NOTAB      // timing ablation only (wrong results): no table traffic
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) { tab[kc].a = mk(1.0f + pair, 0.5f * lane); tab[kc].b = mk(0.25f * kc, 1.0f * wave); }
    return;
#endif
    const cf2 *row = p.tab + ((long long)pair * kN + (2 * wave + q) * kSub + lane);
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) tab[kc] = row[64 * kc];
}

template <class Ctx, int CS, bool INTERIOR, int q>
AW_HD void sib_step(Ctx &ctx, const TileParams &p, const float *in_s, const float *hist_s, long long f0, int t, int lane,
                    int wave, int pair0, bool two, cf *buf0, cf *buf1, const cf *twa, const cf *twb, cf (&wq)[8]) {
    {
        cf u0[8], u1[8];
        cf w1;
        {
            ctx.sched_fence_hard();
            const int tl = ctx.opaque_i(t);
            w1 = p.tw1[tl];
            const float sgn = q ? -1.0f : 1.0f;
            float lo[4][kBatchCh], hi[4][kBatchCh];
            load_fold4<CS, INTERIOR, 0>(p, in_s, hist_s, f0, tl, 2 * pair0, lo, hi);
            fold4s<0>(lo, hi, sgn, u0, u1);
            ctx.sched_fence_hard();
            load_fold4<CS, INTERIOR, 4>(p, in_s, hist_s, f0, tl, 2 * pair0, lo, hi);
            fold4s<4>(lo, hi, sgn, u0, u1);
        }
        if constexpr (q != 0) {                            // the odd half's W_8192^n
            half_twist<false, 1>(u0, w1);
            if (two) half_twist<false, 1>(u1, w1);
        }
        ctx.barrier();
        cf pw[8];
        tw_powers8(cmul(w1, w1), pw);
        sib_pass1(u0, pw, buf0, t);
        if (two) sib_pass1(u1, pw, buf1, t);
    }
    ctx.barrier();
    cf *row0 = buf0 + wave * kRowStride;
    cf *row1 = buf1 + wave * kRowStride;
    cf z[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { z[0][j] = ctx.ld(row0 + lane + 64 * j); z[1][j] = ctx.ld(row1 + lane + 64 * j); }
    ctx.wave_sync();
    sub_fft512x2<false>(ctx, z, row0, row1, twa, twb, lane);
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) { row0[lane + 64 * kc] = z[0][kc]; row1[lane + 64 * kc] = z[1][kc]; }
    cf2 tab[8];
    sib_load_tab<q>(p, pair0, wave, lane, tab);
    ctx.barrier();
    sib_cmac<q>(ctx, z[0], tab, buf0, lane, wave, wq);
    if (two) {
        sib_load_tab<q>(p, pair0 + 1, wave, lane, tab);
        sib_cmac<q>(ctx, z[1], tab, buf1, lane, wave, wq);
    }
}

// first/step/end walk tile ids like tiles_fused_ols; q = 0: even bins (stores E, raises the flag), q = 1: odd bins.
template <class Ctx, int CS, int NP, bool INTERIOR, int q>
AW_HD void tiles_fused_olsq_half(Ctx &ctx, const TileParams &p, long long first, long long step, long long end) {
    const int t0 = ctx.tid();
    const int wave = ctx.wave();
    cf *buf0 = ctx.lds();
    cf *buf1 = buf0 + kHBufElems;
    cf *twa = buf1 + kHBufElems;
    cf *twb = twa + kTwaElems;
    const int Cn = CS > 0 ? CS : p.n_channels;
    if (first >= end) return;
    twa[t0] = p.twa[t0];
    if (t0 < kTwbElems) twb[t0] = p.twb[t0];

    for (long long id = first; id < end; id += step) {
        const int t = ctx.opaque_i(t0);
        const int lane = t & 63;
        const TileId cur = tile_of<INTERIOR>(p, id);
        const long long stream = cur.stream;
        const float *in_s = p.in + stream * p.frames * Cn;
        const float *hist_s = p.hist + stream * (long long)p.hist_len * Cn;
        const long long f0 = (long long)cur.tile * p.hop - p.hist_len;

        cf wq[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) wq[i] = mk(0.f, 0.f);
        if constexpr (NP > 0) {
#pragma unroll
            for (int b = 0; b < (NP + 1) / 2; ++b)
                sib_step<Ctx, CS, INTERIOR, q>(ctx, p, in_s, hist_s, f0, t, lane, wave, 2 * b, 2 * b + 1 < NP, buf0, buf1, twa, twb, wq);
        } else {
            for (int pair0 = 0; pair0 < p.n_pairs; pair0 += 2)
                sib_step<Ctx, CS, INTERIOR, q>(ctx, p, in_s, hist_s, f0, t, lane, wave, pair0, true, buf0, buf1, twa, twb, wq);
        }

        ctx.barrier();                                     // partner reads of the last step are done
        {
            cf *row0 = buf0 + wave * kRowStride;
            sub_fft512x1<true>(ctx, wq, row0, twa, twb, lane);
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) row0[lane + 64 * kc] = wq[kc];
        }
        const int ts = ctx.opaque_i(t);
        const cf w1 = p.tw1[ts];
        ctx.barrier();
        cf v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = ctx.ld(buf0 + r * kRowStride + ts);
        {
            cf pw[8];
            tw_powers8(cmul(w1, w1), pw);
#pragma unroll
            for (int r = 1; r < 8; ++r) v[r] = cmulc(v[r], pw[r]);
            fft8<true>(v);
            if constexpr (q != 0) half_twist<true, 1>(v, w1);
        }
        const int first_valid = p.hist_len;
        float *out_s = p.out + (long long)stream * p.frames * 2;
        if constexpr (q == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int m = ts + 512 * j;
                const long long f = f0 + m;
                if (m >= first_valid && f < p.frames) *reinterpret_cast<cf *>(out_s + f * 2) = v[j];
                if (m + 4096 >= first_valid && f + 4096 < p.frames) *reinterpret_cast<cf *>(out_s + (f + 4096) * 2) = v[j];
            }
            ctx.flag_release(p.flags + id, p.epoch);
        } else {
            ctx.flag_acquire(p.flags + id, p.epoch, p.flags - 1);      // p.flags[-1]: the launch error word
            cf a[8], b[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int m = ts + 512 * j;
                const long long f = f0 + m;
                a[j] = mk(0.f, 0.f); b[j] = mk(0.f, 0.f);
                if (m >= first_valid && f < p.frames) a[j] = ctx.ld_out(out_s + f * 2);
                if (m + 4096 >= first_valid && f + 4096 < p.frames) b[j] = ctx.ld_out(out_s + (f + 4096) * 2);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int m = ts + 512 * j;
                const long long f = f0 + m;
                if (m >= first_valid && f < p.frames) *reinterpret_cast<cf *>(out_s + f * 2) = a[j] + v[j];
                if (m + 4096 >= first_valid && f + 4096 < p.frames) *reinterpret_cast<cf *>(out_s + (f + 4096) * 2) = b[j] - v[j];
            }
        }
    }
}

// q is a compile-time constant of the tile code (a run-time q, although uniform, cost 136 spilled VGPRs): the
// workgroup picks its copy once.
template <class Ctx, int CS, int NP, bool INTERIOR>
AW_HD void tiles_fused_olsq(Ctx &ctx, const TileParams &p, long long first, long long step, long long end, int q) {
    if (q == 0) tiles_fused_olsq_half<Ctx, CS, NP, INTERIOR, 0>(ctx, p, first, step, end);
    else tiles_fused_olsq_half<Ctx, CS, NP, INTERIOR, 1>(ctx, p, first, step, end);
}

}  // namespace awk
