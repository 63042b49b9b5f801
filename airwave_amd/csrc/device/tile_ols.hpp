// tile_ols.hpp — one overlap-save tile of the batch HRIR spatializer, written once for
// the HIP kernel (kernels.hip) and for the CPU thread-emulation harness (tests/emu/).
//
// Replaces, for a whole window of N frames of one stream at once, the per-512-frame loop
//   RealtimeAudioProcessor.processPendingBlock      (Airwave/RealtimeAudioProcessor.swift:141-172)
//     -> ConvolutionEngine.process per (speaker, ear) (Airwave/ConvolutionEngine.swift:232-367)
//     -> vDSP_vadd downmix into blockLeft/blockRight  (RealtimeAudioProcessor.swift:152-163)
// with one workgroup-level computation:
//
//   window  : N = 8192 frames, the last `hop` of which are new output (overlap-save, like
//             ConvolutionEngine.swift:237-243 + :366 but with hop = N - (taps-1) instead of N/2)
//   forward : input channels are packed in PAIRS  z_p[n] = x_{2p}[n] + i x_{2p+1}[n]  (an
//             interleaved frame is already an array of such complex numbers), one complex
//             FFT_N per pair                                   (replaces vDSP_ctoz + fft_zrip :247-252)
//   CMAC    : W[k] += Z_p[k] * A_p[k] + conj(Z_p[N-k]) * B_p[k]          (replaces :304-350)
//             where, for ear E in {L,R} and the pair's channels a,b with spectra H_aE, H_bE,
//               A_p = G1_L + i G1_R,  G1_E = (H_aE - i H_bE) / (2N)
//               B_p = G2_L + i G2_R,  G2_E = (H_aE + i H_bE) / (2N)
//             so that W = (Y_L + i Y_R)/N summed over ALL input channels: the channel split,
//             both ears, the cross-speaker downmix and the 1/N scale cost no extra pass.
//   inverse : one complex IFFT_N gives y_L + i y_R, i.e. the interleaved stereo frame
//                                                        (replaces fft_zrip inverse + vsmul + ztoc :353-363)
//
// FFT_N = 16 x 512:  thread t (512 per workgroup) holds window samples t + 512 j, j < 16.
//   pass 1 (all threads): radix-16 over j -> k1, twiddle W_N^{t k1}, exchange through LDS
//   sub-FFT (per wave)  : 512-point DFT over t for a fixed k1 as three radix-8 passes with
//                         wave-private LDS exchanges.  Wave w owns rows k1 in {w, 16-w}
//                         ({0, 8} for w = 0), which is closed under k -> N-k: the CMAC and
//                         the inverse sub-FFT never leave the wave.
//   inverse             : mirror image; last pass is the radix-16 across rows, so thread t
//                         ends with output samples t + 512 j: coalesced stores.
// One workgroup barrier per pair FFT (+1 for the inverse); LDS = 2 x [16][576] complex.
#pragma once
#include "cplx.hpp"

namespace awk {

constexpr int kN = 8192;          // window / FFT length (frames)
constexpr int kThreads = 512;     // threads per workgroup (8 waves)
constexpr int kRows = 16;         // radix of pass 1
constexpr int kSub = 512;         // sub-FFT length = kN / kRows = kThreads
constexpr int kRowStride = 576;   // complex elements per LDS row (512 + room for padded exchanges)
constexpr int kBufElems = kRows * kRowStride;          // one exchange buffer
constexpr int kLdsBytes = 2 * kBufElems * 8;           // 147456 B (<= 160 KiB)
constexpr int kBatchCh = 4;       // input channels held in registers at once (two pairs)

struct alignas(16) cf2 {          // one table entry: A[k], B[k]
    cf a, b;
};

struct TileParams {
    const float *in;        // [stream][frames][C] interleaved, device
    float *out;             // [stream][frames][2]
    const float *hist;      // [stream][hist_len][C]: the hist_len frames preceding in[...][0]
    const cf2 *tab;         // [pair][16][512] {A,B}, index k = k1 + 16 k2 stored at [k1][k2]
    const cf *tw1;          // [512][16]: W_N^{t k1}      (pass-1 twiddles, one 128-B row per thread)
    const cf *twa;          // [64][8]  : W_512^{lane ka} (sub-FFT pass A)
    const cf *twb;          // [8][8]   : W_64^{l0 kb}    (sub-FFT pass B)
    long long frames;       // frames per stream in this call
    int n_channels;         // C
    int n_pairs;            // ceil(C / 2)
    int hop;                // new output frames per tile, hop <= N - (taps - 1)
    int hist_len;           // = N - hop
    int tiles_per_stream;   // ceil(frames / hop)
};

// ---- per-wave ownership of rows -------------------------------------------------------------
AW_HD int wave_row(int wave, int slot) {
    // slot 0/1 -> k1 ; {0,8} for wave 0, {w, 16-w} otherwise
    return wave == 0 ? (slot == 0 ? 0 : 8) : (slot == 0 ? wave : 16 - wave);
}

// 512-point DFT over the lane dimension, one row held as a[j] = row[lane + 64 j].
// On return a[kc] = X[lane + 64 kc].  `scr` points at this row's private 576-element scratch.
template <bool INV, class Ctx>
AW_HD void sub_fft512(Ctx &ctx, cf (&a)[8], cf *scr, const TileParams &p, int lane) {
    // pass A: radix-8 over j -> ka, twiddle W_512^{lane ka}
    fft8<INV>(a);
    {
        const cf *w = p.twa + lane * 8;
#pragma unroll
        for (int ka = 1; ka < 8; ++ka) a[ka] = twmul<INV>(a[ka], w[ka]);
    }
    // exchange A: write [ka][lane] (row stride 72), read [ka'][l0' + 8 l1] with lane = l0' + 8 ka'
#pragma unroll
    for (int ka = 0; ka < 8; ++ka) scr[ka * 72 + lane] = a[ka];
    ctx.wave_sync();
    const int l0 = lane & 7, kap = lane >> 3;
#pragma unroll
    for (int l1 = 0; l1 < 8; ++l1) a[l1] = scr[kap * 72 + l0 + 8 * l1];
    ctx.wave_sync();
    // pass B: radix-8 over l1 -> kb, twiddle W_64^{l0 kb}
    fft8<INV>(a);
    {
        const cf *w = p.twb + l0 * 8;
#pragma unroll
        for (int kb = 1; kb < 8; ++kb) a[kb] = twmul<INV>(a[kb], w[kb]);
    }
    // exchange B: chunk (kb, ka') of 8 (+1 pad) elements indexed by l0; read by lane = ka'' + 8 kb''
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) scr[(kb * 8 + kap) * 9 + l0] = a[kb];
    ctx.wave_sync();
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = scr[lane * 9 + i];      // chunk index kb''*8 + ka'' == lane
    ctx.wave_sync();
    // pass C: radix-8 over l0 -> kc.  Now a[kc] = X[ka'' + 8 kb'' + 64 kc] = X[lane + 64 kc]
    fft8<INV>(a);
}

// ---- the tile ----------------------------------------------------------------------------------
struct alignas(16) f4 { float x, y, z, w; };
struct alignas(8) f2 { float x, y; };

// One interleaved frame -> registers (channels c0 .. c0+3, zero padded).  Frames before the call
// come from the history buffer (previous calls' tail; zeros after create/reset), frames past the
// end read as zero.  CS = compile-time channel count for vector loads (0 = generic scalar path).
template <int CS>
AW_HD void load_frame(const TileParams &p, const float *in_s, const float *hist_s, long long f,
                      float (&dst)[kBatchCh], int c0) {
    // in_s / hist_s: this stream's first frame in the call input / history buffer.
    // Branch-free: always load from a clamped, valid address, then zero what lies past the end.
    const int C = CS > 0 ? CS : p.n_channels;
    const bool before = f < 0;                         // f >= -hist_len by construction
    const bool past = f >= p.frames;
    const float *base = before ? hist_s : in_s;
    long long idx = before ? (long long)p.hist_len + f : (past ? p.frames - 1 : f);
    const float *src = base + idx * C;
    if constexpr (CS > 0 && CS % 4 == 0) {
        const f4 v = *reinterpret_cast<const f4 *>(src + c0);
        dst[0] = past ? 0.f : v.x; dst[1] = past ? 0.f : v.y; dst[2] = past ? 0.f : v.z; dst[3] = past ? 0.f : v.w;
    } else if constexpr (CS == 2) {
        const f2 v = *reinterpret_cast<const f2 *>(src);
        dst[0] = past ? 0.f : v.x; dst[1] = past ? 0.f : v.y; dst[2] = 0.0f; dst[3] = 0.0f;
    } else {
#pragma unroll
        for (int c = 0; c < kBatchCh; ++c) {
            const int ch = c0 + c;
            const float v = src[ch < C ? ch : C - 1];
            dst[c] = (!past && ch < C) ? v : 0.0f;
        }
    }
}

// pass 1 of one pair: radix-16 over the thread's 16 window samples, twiddle, scatter to rows.
template <class Ctx>
AW_HD void pair_pass1(const TileParams &p, cf (&x)[16], cf *buf, int t) {
    fft16<false>(x);
    {
        const cf *w = p.tw1 + t * 16;
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) x[k1] = cmul(x[k1], w[k1]);
    }
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) buf[k1 * kRowStride + t] = x[k1];
}

// Per wave: the two 512-point sub-FFTs of its rows, then W += Z A + conj(Z[N-k]) B.
template <class Ctx>
AW_HD void pair_subfft_cmac(Ctx &ctx, const TileParams &p, cf *buf, int pair, int lane, int wave, cf (&wacc)[2][8]) {
    cf z[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        cf *row = buf + wave_row(wave, s) * kRowStride;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[s][j] = row[lane + 64 * j];
    }
    ctx.wave_sync();
#pragma unroll
    for (int s = 0; s < 2; ++s) sub_fft512<false>(ctx, z[s], buf + wave_row(wave, s) * kRowStride, p, lane);
    // publish Z rows inside the wave, then CMAC against the partner bins
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        cf *row = buf + wave_row(wave, s) * kRowStride;
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) row[lane + 64 * kc] = z[s][kc];
    }
    ctx.wave_sync();
    const cf2 *tab = p.tab + (long long)pair * kN;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int k1 = wave_row(wave, s);
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            const int k2 = lane + 64 * kc;
            const int k = k1 + 16 * k2;
            const int kk = (kN - k) & (kN - 1);
            const cf zp = buf[(kk & 15) * kRowStride + (kk >> 4)];
            const cf2 ab = tab[k1 * kSub + k2];
            wacc[s][kc] = cfma(z[s][kc], ab.a, wacc[s][kc]);
            wacc[s][kc] = cfmac(zp, ab.b, wacc[s][kc]);
        }
    }
    ctx.wave_sync();    // partner reads done before this wave reuses its rows as scratch
}

// Register batch: 16 frames x 4 channels (two pairs) per thread.
template <int CS>
AW_HD void load_batch(const TileParams &p, const float *in_s, const float *hist_s, long long f0, int t, int c0,
                      float (&raw)[16][kBatchCh]) {
#pragma unroll
    for (int j = 0; j < 16; ++j) load_frame<CS>(p, in_s, hist_s, f0 + t + 512 * j, raw[j], c0);
}

template <class Ctx, int CS>
AW_HD void tile_fused_ols(Ctx &ctx, const TileParams &p, long long stream, int tile) {
    const int t = ctx.tid();
    const int lane = ctx.lane(), wave = ctx.wave();
    cf *buf0 = ctx.lds();
    const int Cn = CS > 0 ? CS : p.n_channels;
    const float *in_s = p.in + stream * p.frames * Cn;
    const float *hist_s = p.hist + stream * (long long)p.hist_len * Cn;
    cf *buf1 = buf0 + kBufElems;
    const long long f0 = (long long)tile * p.hop - p.hist_len;     // frame of window position 0

    cf wacc[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) wacc[s][i] = mk(0.f, 0.f);

    // Schedule per batch of two pairs (4 barriers per tile at C = 8):
    //   pass 1 of both pairs -> buf0, buf1 ; issue the NEXT batch's global loads (their latency
    //   hides under the sub-FFTs; the lines were touched by this batch a moment ago: L2 hits) ;
    //   barrier ; per-wave sub-FFTs + CMAC on buf0 then buf1.
    float raw[16][kBatchCh];
    load_batch<CS>(p, in_s, hist_s, f0, t, 0, raw);
    for (int pair0 = 0; pair0 < p.n_pairs; pair0 += 2) {
        const bool two = pair0 + 1 < p.n_pairs;          // uniform across the workgroup
        if (pair0 > 0) ctx.barrier();                    // every wave is done reading buf0/buf1
        {
            cf x[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) x[j] = mk(raw[j][0], raw[j][1]);
            pair_pass1<Ctx>(p, x, buf0, t);
        }
        if (two) {
            cf x[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) x[j] = mk(raw[j][2], raw[j][3]);
            pair_pass1<Ctx>(p, x, buf1, t);
        }
        if (pair0 + 2 < p.n_pairs) load_batch<CS>(p, in_s, hist_s, f0, t, 2 * (pair0 + 2), raw);
        ctx.barrier();
        pair_subfft_cmac(ctx, p, buf0, pair0, lane, wave, wacc);
        if (two) pair_subfft_cmac(ctx, p, buf1, pair0 + 1, lane, wave, wacc);
    }

    // ---- inverse: per-wave 512-point inverse sub-FFTs of W (scratch = own rows of buf0, which
    // only this wave touches until the barrier), exchange, radix-16 across rows ----
#pragma unroll
    for (int s = 0; s < 2; ++s) sub_fft512<true>(ctx, wacc[s], buf0 + wave_row(wave, s) * kRowStride, p, lane);
    // other waves may still be reading their partner rows of buf1/buf0 for the CMAC: they only
    // read rows they own, and this wave only writes rows it owns.
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        cf *row = buf0 + wave_row(wave, s) * kRowStride;
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) row[lane + 64 * kc] = wacc[s][kc];
    }
    ctx.barrier();
    cf y[16];
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) y[k1] = buf0[k1 * kRowStride + t];
    {
        const cf *w = p.tw1 + t * 16;
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) y[k1] = cmulc(y[k1], w[k1]);
    }
    fft16<true>(y);
    // ---- store the valid part of the window: positions m >= N - hop, frames < p.frames ----
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int m = t + 512 * j;
        const long long f = f0 + m;
        if (m >= p.hist_len && f < p.frames)
            *reinterpret_cast<cf *>(p.out + ((long long)stream * p.frames + f) * 2) = y[j];
    }
}

}  // namespace awk
