// tile_ols2.hpp — overlap-save on 16384-frame windows through the 8192-point machinery of tile_ols.hpp.
//
// A window of N2 = 16384 frames makes one tile emit hop = N2 - taps new frames instead of N - (taps-1)
// of 8192 (12064 against 3873 for the bundled 4320-tap HRIRs: 74 % of every transform is new output
// instead of 47 %), but an N2-point exchange buffer plus accumulators does not fit a CU.  Polyphase form:
// two consecutive frames are ONE frame of a signal at half the rate with twice the channels,
//     [frames][C]  ==  [frames/2][2C]           (same bytes; pseudo-channel c' = parity * C + c)
// and the stereo output at the full rate is four output channels at half the rate,
//     u_e[m] = y_L[2m] + i y_R[2m],   u_o[m] = y_L[2m+1] + i y_R[2m+1],
// each of which is a sum over pseudo-channels of half-rate convolutions with the polyphase components of
// the HRIRs (host/tables.cpp: build_poly_tables).  So a tile is: the 2C pseudo-channels in pairs through
// the SAME forward transform (pass 1 + per-wave 512-point sub-FFTs), every pair accumulated into TWO
// spectra W_e, W_o with two table sets, two inverse transforms, and the results interleaved back into
// full-rate frames (16 bytes per thread and j: frames 2m and 2m+1).  The window is 8192 half-rate frames.
#pragma once
#include "tile_ols.hpp"

#ifndef AW_OLS2_TABPRE
#define AW_OLS2_TABPRE 2     // table sets of a pair's first row issued before its sub-FFTs (0, 1 or 2)
#endif
#ifndef AW_OLS2_H
#define AW_OLS2_H AW_OLS_H   // mono and stereo: the row transforms on the 16-point core, a half-wave per row (tile_ols.hpp, sub_fft512h_*); 0: the 8 x 8 x 8 form everywhere
#endif
#ifndef AW_OLS2_SPLIT
#define AW_OLS2_SPLIT 4      // measured (cfg 2 on 16384 windows / cfg 4 kernel ms): 16 -> 2.17 / 9.1, 8 -> 2.17 / 9.1, 4 -> 1.85 / 7.5, 0 -> 1.83 / 7.9
#endif

namespace awk {

constexpr int kN2 = 2 * kN;                      // full-rate frames per window

struct alignas(16) cf4 {                         // one table entry of the two-output path: {A,B} for u_e, {A,B} for u_o
    cf2 e, o;
};

// Pseudo-frame batch: thread t, j -> half-rate frame m = t + 512 j = full-rate frames f0 + 2m, f0 + 2m + 1;
// four consecutive pseudo-channels starting at c0 (multiple of 4) of that 2C-float pseudo-frame.
// J0..J1: which of the thread's 16 pseudo-frames (a batch can be fetched in two halves to bound live registers).
template <int CS, bool INTERIOR, int J0 = 0, int J1 = 16>
AW_HD void load_batch2(const TileParams &p, const float *in_s, const float *hist_s, long long f0, int t, int c0,
                       float (&raw)[16][kBatchCh]) {
    const int C = CS > 0 ? CS : p.n_channels;
    if constexpr (INTERIOR && CS > 0) {
        const float *lane_base = in_s + f0 * CS + c0;          // uniform
        const int lane_off = t * 2 * CS;                        // per lane, 32-bit
#pragma unroll
        for (int j = J0; j < J1; ++j) {
            const float *src = lane_base + (long long)j * 512 * 2 * CS + lane_off;
            // dword-aligned 16-byte load (frames of an odd stream offset are only 8-byte aligned); lanes past the
            // 2C floats of the pseudo-frame belong to phantom pseudo-channels whose tables are zero
            const f4u v = *reinterpret_cast<const f4u *>(src);
            raw[j][0] = v.x; raw[j][1] = v.y; raw[j][2] = v.z; raw[j][3] = v.w;
        }
    } else {
#pragma unroll
        for (int j = J0; j < J1; ++j) {
            const long long fe = f0 + 2 * (long long)(t + 512 * j);
#pragma unroll
            for (int c = 0; c < kBatchCh; ++c) {
                const int cp = c0 + c;                           // pseudo-channel
                const int par = cp >= C ? 1 : 0;
                const int ch = cp - par * C;
                const long long f = fe + par;
                const float *src = f < 0 ? hist_s + ((long long)p.hist_len + f) * C + ch : in_s + f * C + ch;
                if (cp >= 2 * C || f >= p.frames) src = p.zeros;
                raw[j][c] = *src;
            }
        }
    }
}

// pass 1 with the twiddle powers formed on the fly from w, w^2, w^4, w^8 (8 VGPRs instead of the 30 of the
// product tree held across the butterfly; depth <= 4 multiplications per power as in the tree).
AW_HD void pair_pass1_lean(cf (&x)[16], cf w, cf *buf, int t) {
    fft16<false>(x);
    const cf w2 = cmul(w, w), w4 = cmul(w2, w2), w8 = cmul(w4, w4);
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) {
        cf pw = (k1 & 8) ? w8 : mk(1.f, 0.f);
        bool one = !(k1 & 8);
        if (k1 & 4) { pw = one ? w4 : cmul(pw, w4); one = false; }
        if (k1 & 2) { pw = one ? w2 : cmul(pw, w2); one = false; }
        if (k1 & 1) { pw = one ? w : cmul(pw, w); one = false; }
        x[k1] = cmul(x[k1], pw);
    }
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) buf[k1 * kRowStride + t] = x[k1];
}

// One output set's table entries of one row: o = 0 (u_e) or 1 (u_o).
AW_HD void load_tab2(const TileParams &p, int pair, int wave, int lane, int s, int o, cf2 (&tab)[8]) {
#ifdef AW_ABL_NOTAB      // timing ablation only (wrong results): no table traffic
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) { tab[kc].a = mk(1.0f + pair, 0.5f * lane + o); tab[kc].b = mk(0.25f * kc + s, 1.0f * wave); }
    return;
#endif
    const cf2 *row = p.tab + (((long long)pair * kN + wave_row(wave, s) * kSub + lane) * 2 + o);
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) tab[kc] = row[64 * kc * 2];
}

// Per wave: the two 512-point sub-FFTs of its rows, then W_e, W_o += Z A + conj(Z[N-k]) B with their own tables.
// One (row, output) at a time: 8 table entries (32 VGPRs) in flight instead of 32 entries — the accumulators of
// this path already take 64 VGPRs.
template <class Ctx>
AW_HD void pair_subfft_cmac2(Ctx &ctx, const TileParams &p, int pair, cf *buf, const cf *twa, const cf *twb, int lane, int wave,
                             cf (&we)[2][8], cf (&wo)[2][8]) {
    cf *row0 = buf + wave_row(wave, 0) * kRowStride;
    cf *row1 = buf + wave_row(wave, 1) * kRowStride;
    cf z[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { z[0][j] = ctx.ld(row0 + lane + 64 * j); z[1][j] = ctx.ld(row1 + lane + 64 * j); }
    ctx.wave_sync();
    // Row 0's table entries for both outputs are issued BEFORE the sub-FFTs (their L2 latency, ~2 k cycles each
    // when exposed — phase stamps, tools/archive/stamps2.py — hides under the three radix-8 passes); row 1's are issued
    // into the same registers as soon as row 0's are consumed.  64 table VGPRs in flight at most.
#if defined(AW_ABL2) && (AW_ABL2 & 1)      // timing ablation only (wrong results): forward transform kept, CMAC phase dropped
    sub_fft512x2<false>(ctx, z, row0, row1, twa, twb, lane);
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) { we[0][kc] = we[0][kc] + z[0][kc]; wo[1][kc] = wo[1][kc] + z[1][kc]; }
    ctx.wave_sync();
    return;
#endif
    cf2 ta[8], tb[8];
    if (AW_OLS2_TABPRE >= 1) load_tab2(p, pair, wave, lane, 0, 0, ta);
    if (AW_OLS2_TABPRE >= 2) load_tab2(p, pair, wave, lane, 0, 1, tb);
    sub_fft512x2<false>(ctx, z, row0, row1, twa, twb, lane);
    if (AW_OLS2_TABPRE < 1) load_tab2(p, pair, wave, lane, 0, 0, ta);
    if (AW_OLS2_TABPRE < 2) load_tab2(p, pair, wave, lane, 0, 1, tb);
    ctx.stamp(22);
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) { row0[lane + 64 * kc] = z[0][kc]; row1[lane + 64 * kc] = z[1][kc]; }
    ctx.wave_sync();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const cf *prow = (wave == 0) ? (s == 0 ? row0 : row1) : (s == 0 ? row1 : row0);
        const int bidx = 511 - lane + ((wave == 0 && s == 0) ? 1 : 0);
        cf zp[8];
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            int idx = bidx - 64 * kc;
            if (kc == 0) idx &= 511;
            zp[kc] = ctx.ld(prow + idx);
        }
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            we[s][kc] = cfma(z[s][kc], ta[kc].a, we[s][kc]);
            we[s][kc] = cfmac(zp[kc], ta[kc].b, we[s][kc]);
        }
        if (s == 0) load_tab2(p, pair, wave, lane, 1, 0, ta);
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            wo[s][kc] = cfma(z[s][kc], tb[kc].a, wo[s][kc]);
            wo[s][kc] = cfmac(zp[kc], tb[kc].b, wo[s][kc]);
        }
        if (s == 0) load_tab2(p, pair, wave, lane, 1, 1, tb);
    }
    ctx.wave_sync();    // partner reads done before this wave reuses its rows as scratch
    ctx.stamp(23);
}

// The same on the half-wave row transforms (AW_OLS2_H): a lane holds 16 bins of ONE row, Z[col + 32 kb].  Tables in parts of
// G bins per output (G = 8: 64 table VGPRs in flight at most, as above; G = 4: 32, for the layouts whose accumulators and frame
// batch leave less room).
template <int G>
AW_HD void load_tab2_h(const TileParams &p, int pair, int wave, int lane, int o, int part, cf2 (&tab)[G]) {
#ifdef AW_ABL_NOTAB      // timing ablation only (wrong results): no table traffic
#pragma unroll
    for (int i = 0; i < G; ++i) { tab[i].a = mk(1.0f + pair, 0.5f * lane + o); tab[i].b = mk(0.25f * i + part, 1.0f * wave); }
    return;
#endif
    const cf2 *row = p.tab + (((long long)pair * kN + wave_row(wave, lane >> 5) * kSub + hl_col(lane) + 32 * G * part) * 2 + o);
#pragma unroll
    for (int i = 0; i < G; ++i) tab[i] = row[32 * i * 2];
}
template <int G, class Ctx>
AW_HD void pair_subfft_cmac2_h(Ctx &ctx, const TileParams &p, int pair, cf *buf, const cf *twh, int lane, int wave, cf (&we)[16], cf (&wo)[16]) {
    constexpr int NPART = 16 / G;
    const HLane L = hl_make(ctx, buf, twh, lane, wave);
    cf z[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) z[j] = ctx.ld(L.row + L.h + 32 * j);
    cf2 ta[G], tb[G];
    if (AW_OLS2_TABPRE >= 1) load_tab2_h<G>(p, pair, wave, lane, 0, 0, ta);
    if (AW_OLS2_TABPRE >= 2) load_tab2_h<G>(p, pair, wave, lane, 1, 0, tb);
    sub_fft512h_fwd(ctx, z, L);
    if (AW_OLS2_TABPRE < 1) load_tab2_h<G>(p, pair, wave, lane, 0, 0, ta);
    if (AW_OLS2_TABPRE < 2) load_tab2_h<G>(p, pair, wave, lane, 1, 0, tb);
    ctx.stamp(22);
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) L.row[L.col + 32 * kb] = z[kb];
    ctx.wave_sync();
#pragma unroll
    for (int part = 0; part < NPART; ++part) {
        cf zp[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            int idx = L.pidx - 32 * (G * part + i);
            if (part == 0 && i == 0) idx &= 511;               // only (row 0, column 0) wraps: 512 -> 0
            zp[i] = ctx.ld(L.prow + idx);
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            we[G * part + i] = cfma(z[G * part + i], ta[i].a, we[G * part + i]);
            we[G * part + i] = cfmac(zp[i], ta[i].b, we[G * part + i]);
        }
        if (part + 1 < NPART) load_tab2_h<G>(p, pair, wave, lane, 0, part + 1, ta);
#pragma unroll
        for (int i = 0; i < G; ++i) {
            wo[G * part + i] = cfma(z[G * part + i], tb[i].a, wo[G * part + i]);
            wo[G * part + i] = cfmac(zp[i], tb[i].b, wo[G * part + i]);
        }
        if (part + 1 < NPART) load_tab2_h<G>(p, pair, wave, lane, 1, part + 1, tb);
    }
    ctx.wave_sync();    // partner reads done before this wave reuses its rows as scratch
    ctx.stamp(23);
}

// Final pass of both inverse transforms: radix-16 across the rows of buf0 (u_e) and buf1 (u_o), then the
// half-rate results interleaved into full-rate frames: window position 2m (+1) = frame f0 + 2m (+1).
template <class Ctx, bool INTERIOR>
AW_HD void tile_inverse_final2(Ctx &ctx, const TileParams &p, cf *buf0, cf *buf1, cf w1, int t, long long stream, long long f0,
                               int first_valid) {
    ctx.barrier();
    cf ye[16], yo[16];
    {
        cf pw[16];
        tw_powers(ctx.opaque(w1), pw);
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) { ye[k1] = ctx.ld(buf0 + k1 * kRowStride + t); yo[k1] = ctx.ld(buf1 + k1 * kRowStride + t); }
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) { ye[k1] = cmulc(ye[k1], pw[k1]); yo[k1] = cmulc(yo[k1], pw[k1]); }
    }
    fft16<true>(ye);
    fft16<true>(yo);
    float *out_s = p.out + stream * p.frames * 2;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int n = 2 * (t + 512 * j);                 // window position of the even frame
        const long long f = f0 + n;
        if (n < first_valid) continue;                   // first_valid is even: both frames of the pair are new or neither
        if (INTERIOR || f + 1 < p.frames) {
            ctx.st_stream4(out_s + f * 2, ye[j].x, ye[j].y, yo[j].x, yo[j].y);
        } else if (f < p.frames) {
            ctx.st_stream(reinterpret_cast<cf *>(out_s + f * 2), ye[j]);
        }
    }
}

// Persistent tile loop, same shape as tiles_fused_ols.  CS: real channel count at compile time (0 = runtime),
// NB: compile-time number of pseudo-channel batches (4 pseudo-channels = 2 pseudo-pairs each; 0 = runtime loop).
// p.hop / p.hist_len / p.frames / f0 are in REAL frames; p.n_pairs = pseudo-pairs.
template <class Ctx, int CS, int NB, bool INTERIOR>
AW_HD void tiles_fused_ols2(Ctx &ctx, const TileParams &p, long long first, long long step, long long end) {
    const int t0 = ctx.tid();
    int t = t0, lane = ctx.lane();
    const int wave = ctx.wave();
    cf *buf0 = ctx.lds();
    cf *buf1 = buf0 + kBufElems;
    cf *twa = buf1 + kBufElems;
    cf *twb = twa + kTwaElems;
    const int Cn = CS > 0 ? CS : p.n_channels;
    if (first >= end) return;
    const cf w1 = p.tw1[t];
    // Row transform: the half-wave form of tile_ols.hpp.  Measured per layout on 128 streams x 10 s, 4320 taps (tools/ols2_ab.py, G frames/s,
    // 8 x 8 x 8 -> half-wave): mono 205 -> 230, stereo 135 -> 146 with the tables in parts of eight bins; with eight-bin parts 3 / 5 / 7 channels
    // spill 19-39 VGPRs (sixteen live values per transform next to two accumulators and the frame batch) and lose 2-7 %, in parts of FOUR
    // bins nothing spills and every layout gains: 3 / 4 / 5 / 6 / 7 / 8 channels 99 -> 104, 60.7 -> 68, 60.5 -> 64, 41.7 -> 47, 40.8 -> 43.4, 31.9 -> 33.9.
    constexpr bool kH = AW_OLS2_H != 0;
    constexpr int kHG = (CS == 1 || CS == 2) ? 8 : 4;     // table entries per part
    if constexpr (kH) {
        twa[t] = hl_twiddle(p.twa, t);                   // [16][32] row twiddles of the half-wave form
        (void)twb;
    } else {
        twa[t] = p.twa[t];
        if (t < kTwbElems) twb[t] = p.twb[t];
    }

    float raw[16][kBatchCh];
    {
        const TileId id0 = tile_of<INTERIOR>(p, first);
        load_batch2<CS, INTERIOR>(p, p.in + id0.stream * p.frames * Cn, p.hist + id0.stream * (long long)p.hist_len * Cn,
                                  (long long)id0.tile * p.hop - p.hist_len, t, 0, raw);
    }
    for (long long id = first; id < end; id += step) {
        t = ctx.opaque_i(t0);
        lane = t & 63;
        const TileId cur = tile_of<INTERIOR>(p, id);
        const long long stream = cur.stream;
        const float *in_s = p.in + stream * p.frames * Cn;
        const float *hist_s = p.hist + stream * (long long)p.hist_len * Cn;
        const long long f0 = (long long)cur.tile * p.hop - p.hist_len;     // real frame of window position 0

        cf we[16], wo[16];            // half-wave form: 16 bins of the lane's row; 8 x 8 x 8 form: [row][8 bins]
#pragma unroll
        for (int i = 0; i < 16; ++i) { we[i] = mk(0.f, 0.f); wo[i] = mk(0.f, 0.f); }
        auto subfft_cmac_ = [&](int pr, cf *bf) {
            if constexpr (kH) pair_subfft_cmac2_h<kHG>(ctx, p, pr, bf, twa, lane, wave, we, wo);
            else pair_subfft_cmac2(ctx, p, pr, bf, twa, twb, lane, wave, reinterpret_cast<cf (&)[2][8]>(we), reinterpret_cast<cf (&)[2][8]>(wo));
        };

        const int n_batches = NB > 0 ? NB : (p.n_pairs + 1) / 2;
        auto batch = [&](int b, bool more) {
            if (b > 0) ctx.barrier();                    // every wave is done reading buf0/buf1
            ctx.stamp(4 * (b & 3));                      // diagnostic builds: batch top / pass 1 done / barrier / first pair done
            t = ctx.opaque_i(t);                         // per-batch addresses are recomputed, not held across batches
            lane = t & 63;
            // an odd pseudo-pair count (odd channel counts: 7 pseudo-pairs for 7 channels, 1 for mono) leaves the last
            // batch with one pair; with a compile-time layout its phantom partner is skipped instead of transformed
            const bool two = !(CS > 0 && NB > 0 && CS % 2 == 1 && b == NB - 1);
            {
                cf x[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) x[j] = mk(raw[j][0], raw[j][1]);
                pair_pass1_lean(x, ctx.opaque(w1), buf0, t);
                if (two) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) x[j] = mk(raw[j][2], raw[j][3]);
                    pair_pass1_lean(x, ctx.opaque(w1), buf1, t);
                }
            }
            ctx.stamp(4 * (b & 3) + 1);
            ctx.barrier();
            ctx.stamp(4 * (b & 3) + 2);
            subfft_cmac_(2 * b, buf0);
            ctx.stamp(4 * (b & 3) + 3);
            // the next batch's frames travel under the second pair's sub-FFTs, in two halves: all 16 pseudo-frames
            // at once (64 VGPRs) on top of the 64 accumulator registers is what hipcc spills
            if (more) load_batch2<CS, INTERIOR, 0, AW_OLS2_SPLIT>(p, in_s, hist_s, f0, t, 4 * (b + 1), raw);
            if (two) subfft_cmac_(2 * b + 1, buf1);   // run-time layouts: a phantom pair hits the zero pair
            if (more) load_batch2<CS, INTERIOR, AW_OLS2_SPLIT, 16>(p, in_s, hist_s, f0, t, 4 * (b + 1), raw);
        };
        if constexpr (NB > 0) {
#pragma unroll
            for (int b = 0; b < NB; ++b) batch(b, b + 1 < NB);
        } else {
            for (int b = 0; b < n_batches; ++b) batch(b, b + 1 < n_batches);
        }

        ctx.stamp(30);
        if constexpr (kH) {
            tile_inverse_rows_h(ctx, we, buf0, twa);
            tile_inverse_rows_h(ctx, wo, buf1, twa);
        } else {
            tile_inverse_rows(ctx, reinterpret_cast<cf (&)[2][8]>(we), buf0, twa, twb);
            tile_inverse_rows(ctx, reinterpret_cast<cf (&)[2][8]>(wo), buf1, twa, twb);
        }
        {   // next tile's first batch (unconditional, see tiles_fused_ols)
            const TileId nx = tile_of<INTERIOR>(p, id + step < end ? id + step : id);
            load_batch2<CS, INTERIOR>(p, p.in + nx.stream * p.frames * Cn, p.hist + nx.stream * (long long)p.hist_len * Cn,
                                      (long long)nx.tile * p.hop - p.hist_len, t, 0, raw);
        }
        tile_inverse_final2<Ctx, INTERIOR>(ctx, p, buf0, buf1, w1, t, stream, f0, p.hist_len);
        ctx.stamp(31);
        ctx.flush_stamps();
        ctx.barrier();
    }
}

}  // namespace awk
