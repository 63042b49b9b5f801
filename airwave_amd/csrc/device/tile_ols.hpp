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
// Two workgroup barriers per batch of two pairs (+1 for the inverse); LDS = 2 x [16][576]
// complex exchange buffers + 4.5 KiB of sub-FFT twiddles.
#pragma once
#include "cplx.hpp"

namespace awk {

constexpr int kN = 8192;          // window / FFT length (frames)
constexpr int kThreads = 512;     // threads per workgroup (8 waves)
constexpr int kRows = 16;         // radix of pass 1
constexpr int kSub = 512;         // sub-FFT length = kN / kRows = kThreads
constexpr int kRowStride = 576;   // complex elements per LDS row (512 + room for padded exchanges)
constexpr int kBufElems = kRows * kRowStride;          // one exchange buffer
constexpr int kTwaElems = 8 * 64;                      // [ka][lane]  W_512^{lane ka}
constexpr int kTwbElems = 8 * 8;                       // [kb][l0]    W_64^{l0 kb}
constexpr int kLdsElems = 2 * kBufElems + kTwaElems + kTwbElems;
constexpr int kLdsBytes = kLdsElems * 8;               // 152064 B (<= 160 KiB)
#ifndef AW_XA_REG
#define AW_XA_REG 0      // 1: first sub-FFT exchange through cross-lane swaps instead of LDS (correct; measured 1.54 -> 1.68 ms: +735 VALU instructions per tile and wave for -120 LDS ones)
#endif
#ifndef AW_PREFETCH_RAW_EARLY
#define AW_PREFETCH_RAW_EARLY 1
#endif
constexpr bool kPrefetchRawEarly = AW_PREFETCH_RAW_EARLY != 0;
#ifndef AW_TAB_EARLY
#define AW_TAB_EARLY 0
#endif
constexpr bool kTabEarly0 = (AW_TAB_EARLY & 1) != 0;   // first pair of a batch: table loads issued before the barrier
constexpr bool kTabEarly1 = (AW_TAB_EARLY & 2) != 0;   // second pair: issued right after the first pair's CMAC   // issue the next batch's frame loads before pair 1's sub-FFTs
#ifndef AW_SKIP_PHANTOM
#define AW_SKIP_PHANTOM 1
#endif
constexpr bool kSkipPhantom = AW_SKIP_PHANTOM != 0;   // run-time batch loop: an odd pair count's last batch transforms one pair, not a phantom second one
#ifndef AW_EVEN_F2_LOADS
#define AW_EVEN_F2_LOADS 0
#endif
constexpr int kBatchCh = 4;       // input channels held in registers at once (two pairs)

struct alignas(16) cf2 {          // one table entry: A[k], B[k]
    cf a, b;
};

struct TileParams {
    const float *in;        // [stream][frames][C] interleaved, device
    float *out;             // [stream][frames][2]
    const float *hist;      // [stream][hist_len][C]: the hist_len frames preceding in[...][0]
    const cf2 *tab;         // [pair][16][512] {A,B}, index k = k1 + 16 k2 stored at [k1][k2]; fused path: + one all-zero pair at the end
    const cf *tw1;          // [512]  : W_N^t (pass-1 twiddle base; powers are formed in registers)
    const cf *twa;          // [8][64]: W_512^{lane ka}, ka-major (sub-FFT pass A)
    const cf *twb;          // [8][8] : W_64^{l0 kb}, kb-major   (sub-FFT pass B)
    const float *zeros;     // >= 64 floats of zeros: source of frames past the end of the call
    long long frames;       // frames per stream in this call
    int n_channels;         // C
    int n_pairs;            // ceil(C / 2)
    int hop;                // new output frames per tile, hop <= N - (taps - 1)
    int hist_len;           // = N - hop
    int tiles_per_stream;   // ceil(frames / hop)
    int tile_lo, tile_hi;   // tiles [tile_lo, tile_hi) of every stream are INTERIOR (window inside the input)
    // partitioned (long-HRIR) path only:
    cf *spec;               // [stream][window][pair][16][512] input-window spectra (scratch)
    cf *wspec;              // [stream][block][16][512] accumulated output spectra W (scratch)
    int partitions;         // P = ceil(taps / hop); tables are [partition][pair][N]
    int n_blocks;           // output blocks of `hop` frames per stream in this call
    int first_valid;        // first window position that is stored (N - hop)
    int persistent_wgs;     // grid of the persistent kernels (LaunchCfg, from the context); 0 = 256
    int wide_two_pass;      // LaunchCfg::wide_two_pass
    int debug_occupancy;    // LaunchCfg::debug_occupancy
    int ch_base;            // second pass of a wide layout: `in`, `hist` and `tab` are shifted by this many channels (8); 0 otherwise
    int fwd_one_pair;       // forward kernel form: 1 = one channel pair per workgroup (two workgroups per CU), 0 = all pairs in one workgroup
    int herm_last;          // odd channel count: the last pair's input is real, its spectrum Hermitian — rows 9..15 are neither stored nor read
    int stagger;            // tuning: waves 4-7 idle this many 64-cycle slots after each barrier (phase offset)
    unsigned long long *dbg; // diagnostic builds only (AW_STAMPS): [workgroup][16] s_memtime stamps of wave 0
};
constexpr int kStamps = 32;

// ---- per-wave ownership of rows -------------------------------------------------------------
AW_HD int wave_row(int wave, int slot) {
    // slot 0/1 -> k1 ; {0,8} for wave 0, {w, 16-w} otherwise
    return wave == 0 ? (slot == 0 ? 0 : 8) : (slot == 0 ? wave : 16 - wave);
}

// w^1 .. w^15 from w by a depth-4 product tree (no memory, 14 complex multiplies)
AW_HD void tw_powers(cf w, cf (&pw)[16]) {
    pw[0] = mk(1.f, 0.f);
    pw[1] = w;
    pw[2] = cmul(w, w);
    pw[3] = cmul(pw[2], w);
    pw[4] = cmul(pw[2], pw[2]);
    pw[5] = cmul(pw[4], w);
    pw[6] = cmul(pw[4], pw[2]);
    pw[7] = cmul(pw[4], pw[3]);
    pw[8] = cmul(pw[4], pw[4]);
    pw[9] = cmul(pw[8], w);
    pw[10] = cmul(pw[8], pw[2]);
    pw[11] = cmul(pw[8], pw[3]);
    pw[12] = cmul(pw[8], pw[4]);
    pw[13] = cmul(pw[8], pw[5]);
    pw[14] = cmul(pw[8], pw[6]);
    pw[15] = cmul(pw[8], pw[7]);
}

// Two 512-point DFTs over the lane dimension at once (the wave's two rows; independent work
// interleaved for ILP).  Row s is held as a[s][j] = row_s[lane + 64 j]; on return
// a[s][kc] = X_s[lane + 64 kc].  scr0/scr1: the rows' private 576-element scratch.
template <bool INV, class Ctx>
AW_HD void sub_fft512x2(Ctx &ctx, cf (&a)[2][8], cf *scr0, cf *scr1, const cf *twa, const cf *twb, int lane) {
    constexpr int SB = INV ? 24 : 16;
    ctx.stamp(SB + 0);
    // pass A: radix-8 over j -> ka, twiddle W_512^{lane ka}
    fft8<INV>(a[0]);
    ctx.sched_fence();
    fft8<INV>(a[1]);
    ctx.sched_fence();
#pragma unroll
    for (int ka = 1; ka < 8; ++ka) {
        const cf w = ctx.ld(twa + ka * 64 + lane);
        a[0][ka] = twmul<INV>(a[0][ka], w);
        a[1][ka] = twmul<INV>(a[1][ka], w);
    }
    ctx.stamp(SB + 1);
    // exchange A: element (ka, lane = l0 + 8 l1) -> (register l1, lane = l0 + 8 ka): an 8x8 transpose between the
    // register index and lane bits 3-5.
    const int l0 = lane & 7, kap = lane >> 3;
#if AW_XA_REG
    // In registers: three swap stages, one per (register bit, lane bit) pair — v_permlane32_swap, v_permlane16_swap
    // and a bank-masked DPP row_ror:8.  No LDS traffic: the LDS store path (6 cycles per ds_write_b64 for the whole CU)
    // is this kernel's busiest resource, VALU issue is not.
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int r = 0; r < 4; ++r) ctx.xswap(a[s][r], a[s][r + 4], 5);
#pragma unroll
        for (int r = 0; r < 4; ++r) ctx.xswap(a[s][(r & 1) + 4 * (r >> 1)], a[s][(r & 1) + 4 * (r >> 1) + 2], 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) ctx.xswap(a[s][2 * r], a[s][2 * r + 1], 3);
    }
#else
    // through LDS: write [ka][lane] (row stride 72), read [ka'][l0' + 8 l1] with lane = l0' + 8 ka'
#pragma unroll
    for (int ka = 0; ka < 8; ++ka) { scr0[ka * 72 + lane] = a[0][ka]; scr1[ka * 72 + lane] = a[1][ka]; }
    ctx.wave_sync();
    ctx.template ld8x2<8>(a[0], scr0 + kap * 72 + l0, a[1], scr1 + kap * 72 + l0);
    ctx.wave_sync();
#endif
    ctx.stamp(SB + 2);
    // pass B: radix-8 over l1 -> kb, twiddle W_64^{l0 kb}
    fft8<INV>(a[0]);
    ctx.sched_fence();
    fft8<INV>(a[1]);
    ctx.sched_fence();
#pragma unroll
    for (int kb = 1; kb < 8; ++kb) {
        const cf w = ctx.ld(twb + kb * 8 + l0);
        a[0][kb] = twmul<INV>(a[0][kb], w);
        a[1][kb] = twmul<INV>(a[1][kb], w);
    }
    ctx.stamp(SB + 3);
    // exchange B: chunk (kb, ka') of 8 (+1 pad) elements indexed by l0; read by lane = ka'' + 8 kb''
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) { scr0[(kb * 8 + kap) * 9 + l0] = a[0][kb]; scr1[(kb * 8 + kap) * 9 + l0] = a[1][kb]; }
    ctx.wave_sync();
    ctx.template ld8x2<1>(a[0], scr0 + lane * 9, a[1], scr1 + lane * 9);   // chunk kb''*8 + ka'' == lane
    ctx.wave_sync();
    ctx.stamp(SB + 4);
    // pass C: radix-8 over l0 -> kc.  Now a[s][kc] = X_s[ka'' + 8 kb'' + 64 kc] = X_s[lane + 64 kc]
    fft8<INV>(a[0]);
    ctx.sched_fence();
    fft8<INV>(a[1]);
    ctx.sched_fence();
    ctx.stamp(SB + 5);
}

// ---- the same row transforms on the 16-point core (round 4): a HALF-wave per row ------------------
// Lanes 0-31 of a wave own its first row, lanes 32-63 its second; a lane holds SIXTEEN values of its row.  512 = 16 x 2 x 16:
//   radix 16 over j (x[h + 32 j], h = lane & 31) -> ka, twiddle w_512^{h ka} (a [16][32] table in LDS),
//   radix 2 over lane bit 4 as a register<->lane swap (v_permlane16_swap: registers ka, ka + 8) + an in-lane butterfly,
//   twiddle w_32^{h0} on the odd half (h0 = lane & 15), ONE wave-private 16 x 16 LDS transpose (17-element pitch, 272 per group of
//   16 lanes: conflict-free both ways), radix 16 over h0 -> kb.
// One LDS exchange per row instead of two (63 LDS instructions per lane and row pair against 94):
// tools/ubench/fft_core.hip 3.03 -> 2.56 us per pair transform per CU.
// Bin order: on return a[kb] = X[hl_col(lane) + 32 kb].
#ifndef AW_OLS_H
#define AW_OLS_H 1        // 0: the fused 8192-frame tile on the 8 x 8 x 8 row transforms of rounds 1-3
#endif

AW_HD int hl_col(int lane) { return (lane & 7) + 8 * ((lane >> 4) & 1) + 16 * ((lane >> 3) & 1); }   // lane & 31 with bits 3 and 4 exchanged
struct HLane {
    cf *row;        // the lane's row of the exchange buffer (also its half-wave's transpose scratch: 2 x 272 <= 576 elements)
    cf *prow;       // the row holding the conjugate partners of this row's bins: (16 - k1) & 15
    const cf *twh;  // w_512^{h ka} at twh[32 ka]
    int h, e2w, e2r, col, pidx;
    cf w32;
};
// The row twiddles of this form, [ka][h] = w_512^{h ka} (16 x 32), take the place of the 8 x 8 x 8 form's [8][64] table in LDS; one
// entry per thread, formed once per launch from two entries of that table: w_512^m = w_512^{m & 63} w_8^{m >> 6}.
AW_HD cf hl_twiddle(const cf *twa_g, int t) {
    const int m = (t >> 5) * (t & 31), j = m >> 6;
    const cf lo = twa_g[64 + (m & 63)];                 // twa_g[ka][lane] = w_512^{lane ka}
    cf hi = twa_g[2 * (j & 3) * 64 + 32];               // w_512^{64 (j & 3)}
    if (j & 4) hi = mk(-hi.x, -hi.y);
    return cmul(lo, hi);
}
template <class Ctx> AW_HD HLane hl_make(Ctx &ctx, cf *buf, const cf *twh, int lane, int wave) {
    HLane L;
    const int s = lane >> 5, g = (lane >> 4) & 1, h0 = lane & 15;
    L.h = lane & 31;
    L.row = buf + wave_row(wave, s) * kRowStride;
    L.prow = buf + wave_row(wave, wave == 0 ? s : 1 - s) * kRowStride;
    L.twh = twh + L.h;
    L.e2w = 272 * g + h0;
    L.e2r = 272 * g + 17 * h0;
    L.col = hl_col(lane);
    L.pidx = 511 - L.col + ((wave == 0 && s == 0) ? 1 : 0);          // column (512 - k2) & 511 on row 0, 511 - k2 elsewhere
    L.w32 = ctx.ld(twh + 8 * 32 + 2 * h0);                           // w_32^{h0} = w_512^{8 (2 h0)}
    return L;
}
// a[ka] *= w_512^{h ka} (INV: conjugate); the entries are requested eight at a time before their multiplies
template <bool INV, class Ctx> AW_HD void hl_tw_apply(Ctx &ctx, cf (&a)[16], const HLane &L) {
#ifdef AW_ABL_NOTWLDS          // timing ablation only (wrong results): row twiddles without their LDS table reads
#pragma unroll
    for (int m = 1; m < 16; ++m) a[m] = twmul<INV>(a[m], mk(L.w32.x + 0.001f * m, L.w32.y));
    return;
#endif
#pragma unroll
    for (int m = 1; m < 16; m += 8) {
        cf w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) if (m + i < 16) w[i] = ctx.ld(L.twh + 32 * (m + i));
        ctx.sched_fence();
#pragma unroll
        for (int i = 0; i < 8; ++i) if (m + i < 16) a[m + i] = twmul<INV>(a[m + i], w[i]);
    }
}
// forward: a[j] = x[h + 32 j]  ->  a[kb] = X[col + 32 kb]
template <class Ctx> AW_HD void sub_fft512h_fwd(Ctx &ctx, cf (&a)[16], const HLane &L) {
    ctx.stamp(16);
    fft16<false>(a);
    ctx.stamp(17);
    hl_tw_apply<false>(ctx, a, L);
    ctx.stamp(18);
    const cf w32 = ctx.opaque(L.w32);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        ctx.xswap(a[q], a[q + 8], 4);        // lanes with bit 4 clear now hold ka = q of both halves h1 = 0, 1; the others ka = q + 8
        const cf e = a[q] + a[q + 8], o = a[q] - a[q + 8];
        a[q] = e;
        a[q + 8] = cmul(o, w32);
    }
    ctx.stamp(19);
#ifndef AW_ABL_NOXCHG          // (timing ablation, wrong results: the 16 x 16 LDS transpose of the row transform removed)
    ctx.wave_sync();                         // the half-wave's loads of its row have returned
#pragma unroll
    for (int r = 0; r < 16; ++r) L.row[L.e2w + 17 * r] = a[r];
    ctx.wave_sync();
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = ctx.ld(L.row + L.e2r + r);
    ctx.wave_sync();
#endif
    ctx.stamp(20);
    fft16<false>(a);
    ctx.stamp(21);
}
// inverse (unnormalised), the mirror image: a[kb] = X[col + 32 kb]  ->  a[j] = x[h + 32 j]
template <class Ctx> AW_HD void sub_fft512h_inv(Ctx &ctx, cf (&a)[16], const HLane &L) {
    fft16<true>(a);
#ifndef AW_ABL_NOXCHG
    ctx.wave_sync();
#pragma unroll
    for (int r = 0; r < 16; ++r) L.row[L.e2r + r] = a[r];
    ctx.wave_sync();
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = ctx.ld(L.row + L.e2w + 17 * r);
    ctx.wave_sync();
#endif
    const cf w32 = ctx.opaque(L.w32);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const cf o = cmulc(a[q + 8], w32);
        const cf e = a[q];
        a[q] = e + o;
        a[q + 8] = e - o;
        ctx.xswap(a[q], a[q + 8], 4);
    }
    hl_tw_apply<true>(ctx, a, L);
    fft16<true>(a);
}

// ---- the tile ----------------------------------------------------------------------------------
struct alignas(16) f4 { float x, y, z, w; };
struct alignas(8) f2 { float x, y; };
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };   // 16 bytes at dword alignment
struct __attribute__((packed, aligned(4))) f2u { float x, y; };    // 8 bytes at dword alignment

// One interleaved frame -> registers (channels c0 .. c0+3, zero padded).  Frames before the call
// come from the history buffer (previous calls' tail; zeros after create/reset), frames past the
// end read as zero.  CS = compile-time channel count for vector loads (0 = generic scalar path).
template <int CS>
AW_HD void load_frame(const TileParams &p, const float *in_s, const float *hist_s, long long f,
                      float (&dst)[kBatchCh], int c0) {
    // in_s / hist_s: this stream's first frame in the call input / history buffer.
    // Branch-free and with NO arithmetic on the loaded data: frames before the call come from the
    // history buffer, frames past the end from a page of zeros, so the 16 loads of a batch have no
    // consumers until pass 1 and all stay in flight together (a select on the loaded value made
    // hipcc reuse one destination register and serialise the loads: 16 exposed round trips).
    const int C = CS > 0 ? CS : p.n_channels;
    const bool before = f < 0;                         // f >= -hist_len by construction
    const bool past = f >= p.frames;
    const float *src = before ? hist_s + ((long long)p.hist_len + f) * C : (past ? p.zeros : in_s + f * C);
    if constexpr (CS > 0 && CS % 4 == 0) {
        const f4 v = *reinterpret_cast<const f4 *>(src + c0);
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    } else if constexpr (CS == 2) {
        const f2 v = *reinterpret_cast<const f2 *>(src);
        dst[0] = v.x; dst[1] = v.y; dst[2] = 0.0f; dst[3] = 0.0f;
    } else {
#pragma unroll
        for (int c = 0; c < kBatchCh; ++c) {
            const int ch = c0 + c;
            const float *q = ch + p.ch_base < C ? src + ch : p.zeros;      // padding channels of the last batch read zeros
            dst[c] = *q;
        }
    }
}

// Register batch: 16 frames x 4 channels (two pairs) per thread.
// INTERIOR tiles (window entirely inside the call's input: every tile but the first one or two
// and the last of a stream) use wave-uniform frame bases + one per-lane offset, which hipcc turns
// into scalar-base global loads: no per-frame address registers, no clamping arithmetic.
template <int CS, bool INTERIOR>
AW_HD void load_batch(const TileParams &p, const float *in_s, const float *hist_s, long long f0, int t, int c0,
                      float (&raw)[16][kBatchCh]) {
#ifdef AW_ABL_NOLOAD           // timing ablation only: no global loads of frames
    {
#pragma unroll
        for (int j = 0; j < 16; ++j) { raw[j][0] = 0.001f * t; raw[j][1] = 0.002f * j; raw[j][2] = 0.003f * c0; raw[j][3] = 1.0f; }
        return;
    }
#endif
    if constexpr (INTERIOR && CS > 0) {
        const float *lane_base = in_s + f0 * CS + c0;          // uniform
        const int lane_off = t * CS;                            // per lane, 32-bit
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float *src = lane_base + (long long)j * 512 * CS + lane_off;
            if constexpr (CS % 4 == 0) {
                const f4 v = *reinterpret_cast<const f4 *>(src);
                raw[j][0] = v.x; raw[j][1] = v.y; raw[j][2] = v.z; raw[j][3] = v.w;
            } else if constexpr (CS == 2) {
                const f2 v = *reinterpret_cast<const f2 *>(src);
                raw[j][0] = v.x; raw[j][1] = v.y; raw[j][2] = 0.f; raw[j][3] = 0.f;
            } else if constexpr (CS % 2 == 0 && AW_EVEN_F2_LOADS) {
                // 6, 10, 14 channels: frames are multiples of 8 bytes — two naturally aligned 8-byte loads instead of one
                // dword-aligned 16-byte load (which the memory pipeline splits)
                const f2 v0 = *reinterpret_cast<const f2 *>(src);
                const f2 v1 = *reinterpret_cast<const f2 *>(src + 2);
                raw[j][0] = v0.x; raw[j][1] = v0.y; raw[j][2] = v1.x; raw[j][3] = v1.y;
            } else {
                // frames that are not whole float4s (6, 7, 14 channels): one dword-aligned 16-B load all the
                // same.  The last batch of a frame runs up to 3 floats into the next frame; those lanes belong
                // to channels >= CS, whose filters are zero tables (build_pair_tables / the zero pair), so the
                // finite stray samples contribute nothing.  The launch keeps one frame of slack at the end.
                const f4u v = *reinterpret_cast<const f4u *>(src);
                raw[j][0] = v.x; raw[j][1] = v.y; raw[j][2] = v.z; raw[j][3] = v.w;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) load_frame<CS>(p, in_s, hist_s, f0 + t + 512 * j, raw[j], c0);
    }
}

// Both four-channel batches of the thread's 16 frames in one go (interior tiles, 5-8 channels): the two 16-byte loads of a
// frame are issued back to back, so the second one hits the line the first has just brought into L1 — the batches no longer
// pull every line through the L2 -> L1 path twice, half a tile apart (tools/ubench/tile_bench.hip: frame loads are 20 % of a
// cfg-2 tile, a throughput cost of that path).  Same registers as issuing batch 1 during batch 0's processing.
#ifndef AW_WHOLE_FRAMES
#define AW_WHOLE_FRAMES 1
#endif
#ifndef AW_WIDE_LATE_B
#define AW_WIDE_LATE_B 0     // 1: the second batch of a wide layout's second channel group is fetched one pair later (measured: 14 channels 17.1 -> 15.9 G frames/s)
#endif
// SECOND: floats of the second batch that are fetched — 4 (all), 2 (a seventh pair is the group's last: 13-14 channels) or
// 0 (9-12 channels: the last eight-channel group has one batch).
template <int CS, int SECOND = 4, bool FIRST = true>
AW_HD void load_batch2(const float *in_s, long long f0, int t, float (&ra)[16][kBatchCh], float (&rb)[16][kBatchCh]) {
    static_assert(CS >= 5, "two batches of four channels (the first eight channels from the base pointer; wide layouts shift the base)");
    const float *lane_base = in_s + f0 * CS;            // uniform
    const int lane_off = t * CS;                         // per lane, 32-bit
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float *src = lane_base + (long long)j * 512 * CS + lane_off;
        if constexpr (FIRST) {
            if constexpr (CS % 4 == 0) {
                const f4 a = *reinterpret_cast<const f4 *>(src);
                ra[j][0] = a.x; ra[j][1] = a.y; ra[j][2] = a.z; ra[j][3] = a.w;
            } else {
                const f4u a = *reinterpret_cast<const f4u *>(src);
                ra[j][0] = a.x; ra[j][1] = a.y; ra[j][2] = a.z; ra[j][3] = a.w;
            }
        }
        if constexpr (SECOND == 4) {
            if constexpr (CS % 4 == 0) {
                const f4 b = *reinterpret_cast<const f4 *>(src + 4);
                rb[j][0] = b.x; rb[j][1] = b.y; rb[j][2] = b.z; rb[j][3] = b.w;
            } else {
                const f4u b = *reinterpret_cast<const f4u *>(src + 4);      // runs up to 3 floats into the next frame: zero tables (see load_batch)
                rb[j][0] = b.x; rb[j][1] = b.y; rb[j][2] = b.z; rb[j][3] = b.w;
            }
        } else if constexpr (SECOND == 2) {
            if constexpr (CS % 2 == 0) {
                const f2 b = *reinterpret_cast<const f2 *>(src + 4);
                rb[j][0] = b.x; rb[j][1] = b.y;
            } else {
                const f2u b = *reinterpret_cast<const f2u *>(src + 4);      // .y is the next frame's first sample: zero tables
                rb[j][0] = b.x; rb[j][1] = b.y;
            }
            rb[j][2] = 0.f; rb[j][3] = 0.f;
        }
    }
}

// HEAD windows of the partitioned path (the first P windows of a call: part history, part input, nothing past the
// end): per frame a pointer select between the history buffer and the input, then the same whole-frame vector load as
// the interior windows.  Layouts whose frames are not whole float4s read up to 3 floats past a frame: the history
// allocation carries that slack (runtime.cpp), the launch keeps one input frame of slack.
template <int CS>
AW_HD void load_batch_head(const TileParams &p, const float *in_s, const float *hist_s, long long f0, int t, int c0,
                           float (&raw)[16][kBatchCh]) {
    static_assert(CS > 0, "vector layouts only");
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const long long f = f0 + t + 512 * j;
        const float *src = (f < 0 ? hist_s + ((long long)p.hist_len + f) * CS : in_s + f * CS) + c0;
        if constexpr (CS % 4 == 0) {
            const f4 v = *reinterpret_cast<const f4 *>(src);
            raw[j][0] = v.x; raw[j][1] = v.y; raw[j][2] = v.z; raw[j][3] = v.w;
        } else if constexpr (CS == 2) {
            const f2 v = *reinterpret_cast<const f2 *>(src);
            raw[j][0] = v.x; raw[j][1] = v.y; raw[j][2] = 0.f; raw[j][3] = 0.f;
        } else {
            const f4u v = *reinterpret_cast<const f4u *>(src);
            raw[j][0] = v.x; raw[j][1] = v.y; raw[j][2] = v.z; raw[j][3] = v.w;
        }
    }
}

// Both four-channel batches of a head window's frames back to back (5-8 channels), as load_batch2 does for interior windows.
template <int CS>
AW_HD void load_batch2_head(const TileParams &p, const float *in_s, const float *hist_s, long long f0, int t,
                            float (&ra)[16][kBatchCh], float (&rb)[16][kBatchCh]) {
    static_assert(CS >= 5 && CS <= 8, "two batches of four channels");
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const long long f = f0 + t + 512 * j;
        const float *src = f < 0 ? hist_s + ((long long)p.hist_len + f) * CS : in_s + f * CS;
        if constexpr (CS % 4 == 0) {
            const f4 a = *reinterpret_cast<const f4 *>(src);
            const f4 b = *reinterpret_cast<const f4 *>(src + 4);
            ra[j][0] = a.x; ra[j][1] = a.y; ra[j][2] = a.z; ra[j][3] = a.w;
            rb[j][0] = b.x; rb[j][1] = b.y; rb[j][2] = b.z; rb[j][3] = b.w;
        } else {
            const f4u a = *reinterpret_cast<const f4u *>(src);
            const f4u b = *reinterpret_cast<const f4u *>(src + 4);      // up to 3 floats into the next frame (history: its allocation's slack)
            ra[j][0] = a.x; ra[j][1] = a.y; ra[j][2] = a.z; ra[j][3] = a.w;
            rb[j][0] = b.x; rb[j][1] = b.y; rb[j][2] = b.z; rb[j][3] = b.w;
        }
    }
}

// pass 1 of one pair: radix-16 over the thread's 16 window samples, twiddle, scatter to rows.
AW_HD void pair_pass1(cf (&x)[16], const cf (&pw)[16], cf *buf, int t) {
    fft16<false>(x);
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) x[k1] = cmul(x[k1], pw[k1]);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) buf[k1 * kRowStride + t] = x[k1];
}

// This lane's 16 table entries of one pair (issued early; consumed after the sub-FFTs).
AW_HD void load_tab(const TileParams &p, int pair, int wave, int lane, cf2 (&tab)[2][8]) {
#ifdef AW_ABL_NOTAB      // timing ablation only (wrong results): no table traffic
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) { tab[s][kc].a = mk(1.0f + pair, 0.5f * lane); tab[s][kc].b = mk(0.25f * kc, 1.0f * wave); }
    return;
#endif
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const cf2 *row = p.tab + ((long long)pair * kN + wave_row(wave, s) * kSub + lane);
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) tab[s][kc] = row[64 * kc];
    }
}

// Per wave: the two 512-point sub-FFTs of its rows, then W += Z A + conj(Z[N-k]) B.
template <class Ctx>
AW_HD void pair_subfft_cmac(Ctx &ctx, const TileParams &p, int pair, cf *buf, const cf *twa, const cf *twb,
                            cf2 (&tab)[2][8], int lane, int wave, cf (&wacc)[2][8], bool tab_loaded) {
    cf *row0 = buf + wave_row(wave, 0) * kRowStride;
    cf *row1 = buf + wave_row(wave, 1) * kRowStride;
    cf z[2][8];
    ctx.template ld8x2<64>(z[0], row0 + lane, z[1], row1 + lane);
    ctx.wave_sync();
    sub_fft512x2<false>(ctx, z, row0, row1, twa, twb, lane);
#ifdef AW_ABL_NOCMAC          // timing ablation only (wrong results): no tables, no partner exchange, no multiply-accumulate
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) wacc[s][kc] = wacc[s][kc] + z[s][kc];
    return;
#endif
    if (!tab_loaded) load_tab(p, pair, wave, lane, tab);
    ctx.stamp(22);
#ifdef AW_ABL_NOPARTNER       // timing ablation only (wrong results): no partner exchange through LDS
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            wacc[s][kc] = cfma(z[s][kc], tab[s][kc].a, wacc[s][kc]);
            wacc[s][kc] = cfmac(z[1 - s][7 - kc], tab[s][kc].b, wacc[s][kc]);
        }
    return;
#endif
    // publish Z rows inside the wave, then CMAC against the partner bins
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) { row0[lane + 64 * kc] = z[0][kc]; row1[lane + 64 * kc] = z[1][kc]; }
    ctx.wave_sync();
    // partner of bin (k1, k2): row (16 - k1) & 15 (the wave's other row; itself for k1 in {0, 8}),
    // column (512 - k2) & 511 for k1 == 0, else 511 - k2.
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const cf *prow = (wave == 0) ? (s == 0 ? row0 : row1) : (s == 0 ? row1 : row0);
        const int bidx = 511 - lane + ((wave == 0 && s == 0) ? 1 : 0);
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            int idx = bidx - 64 * kc;
            if (kc == 0) idx &= 511;                       // only (row 0, lane 0, kc 0) wraps: 512 -> 0
            const cf zp = ctx.ld(prow + idx);
            wacc[s][kc] = cfma(z[s][kc], tab[s][kc].a, wacc[s][kc]);
            wacc[s][kc] = cfmac(zp, tab[s][kc].b, wacc[s][kc]);
        }
    }
    ctx.wave_sync();    // partner reads done before this wave reuses its rows as scratch
    ctx.stamp(23);
}

// Inverse half of a tile, part 1: per-wave 512-point inverse sub-FFTs of W (scratch = the wave's
// own rows of buf0, which only it touches until the barrier), results published row-wise.
template <class Ctx>
AW_HD void tile_inverse_rows(Ctx &ctx, cf (&wacc)[2][8], cf *buf0, const cf *twa, const cf *twb) {
    const int lane = ctx.lane(), wave = ctx.wave();
    cf *row0 = buf0 + wave_row(wave, 0) * kRowStride;
    cf *row1 = buf0 + wave_row(wave, 1) * kRowStride;
    sub_fft512x2<true>(ctx, wacc, row0, row1, twa, twb, lane);
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) { row0[lane + 64 * kc] = wacc[0][kc]; row1[lane + 64 * kc] = wacc[1][kc]; }
}

// The same three steps on the half-wave row transforms (AW_OLS_H): a lane holds 16 bins of ONE row.
AW_HD void load_tab_h(const TileParams &p, int pair, int wave, int lane, cf2 (&tab)[16]) {
#ifdef AW_ABL_NOTAB      // timing ablation only (wrong results): no table traffic
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) { tab[kb].a = mk(1.0f + pair, 0.5f * lane); tab[kb].b = mk(0.25f * kb, 1.0f * wave); }
    return;
#endif
    const cf2 *row = p.tab + ((long long)pair * kN + wave_row(wave, lane >> 5) * kSub + hl_col(lane));
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) tab[kb] = row[32 * kb];          // per half-wave 512 contiguous bytes (whole 128-byte lines per 8 lanes)
}
// G: table entries per part when the tables are fetched after the transform (G = 16: all at once, 64 VGPRs; G = 4: two parts of
// four in flight, 32 VGPRs — the next part travels while the current one is consumed).
template <int G>
AW_HD void load_tab_part_h(const TileParams &p, int pair, int wave, int lane, int part, cf2 *tab) {
#ifdef AW_ABL_NOTAB      // timing ablation only (wrong results): no table traffic
#pragma unroll
    for (int i = 0; i < G; ++i) { tab[i].a = mk(1.0f + pair, 0.5f * lane); tab[i].b = mk(0.25f * i + part, 1.0f * wave); }
    return;
#endif
    const cf2 *row = p.tab + ((long long)pair * kN + wave_row(wave, lane >> 5) * kSub + hl_col(lane) + 32 * G * part);
#pragma unroll
    for (int i = 0; i < G; ++i) tab[i] = row[32 * i];
}
template <int G = 16, class Ctx>
AW_HD void pair_subfft_cmac_h(Ctx &ctx, const TileParams &p, int pair, cf *buf, const cf *twa, cf2 (&tab)[16], int lane, int wave,
                              cf (&wacc)[16], bool tab_loaded) {
    const HLane L = hl_make(ctx, buf, twa, lane, wave);
    cf z[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) z[j] = ctx.ld(L.row + L.h + 32 * j);
    sub_fft512h_fwd(ctx, z, L);
    if constexpr (G == 16) { if (!tab_loaded) load_tab_h(p, pair, wave, lane, tab); }
    else load_tab_part_h<G>(p, pair, wave, lane, 0, tab);
    ctx.stamp(22);
    // publish Z in natural column order inside the wave, then multiply-accumulate against the partner bins
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) L.row[L.col + 32 * kb] = z[kb];
    ctx.wave_sync();
    if constexpr (G == 16) {
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            int idx = L.pidx - 32 * kb;
            if (kb == 0) idx &= 511;                               // only (row 0, column 0) wraps: 512 -> 0
            const cf zp = ctx.ld(L.prow + idx);
            wacc[kb] = cfma(z[kb], tab[kb].a, wacc[kb]);
            wacc[kb] = cfmac(zp, tab[kb].b, wacc[kb]);
        }
    } else {
        static_assert(2 * G <= 16, "two parts in flight inside tab[16]");
#pragma unroll
        for (int part = 0; part < 16 / G; ++part) {
            cf2 *cur = tab + G * (part & 1);
            if (part + 1 < 16 / G) load_tab_part_h<G>(p, pair, wave, lane, part + 1, tab + G * ((part + 1) & 1));
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int kb = G * part + i;
                int idx = L.pidx - 32 * kb;
                if (kb == 0) idx &= 511;
                const cf zp = ctx.ld(L.prow + idx);
                wacc[kb] = cfma(z[kb], cur[i].a, wacc[kb]);
                wacc[kb] = cfmac(zp, cur[i].b, wacc[kb]);
            }
        }
    }
    ctx.wave_sync();    // partner reads done before this wave reuses its rows as scratch
    ctx.stamp(23);
}
template <class Ctx>
AW_HD void tile_inverse_rows_h(Ctx &ctx, cf (&wacc)[16], cf *buf0, const cf *twa) {
    const int lane = ctx.lane(), wave = ctx.wave();
    const HLane L = hl_make(ctx, buf0, twa, lane, wave);
    sub_fft512h_inv(ctx, wacc, L);
    ctx.wave_sync();
#pragma unroll
    for (int j = 0; j < 16; ++j) L.row[L.h + 32 * j] = wacc[j];
}

// Part 2: barrier, radix-16 across rows, then store window positions m >= first_valid whose
// frame f0 + m lies inside the call.
template <class Ctx>
AW_HD void tile_inverse_final(Ctx &ctx, const TileParams &p, cf *buf0, cf w1, int t, long long stream, long long f0,
                              int first_valid, bool accumulate = false) {
    ctx.stamp(11);
    cf old[16];
    if (accumulate) {          // uniform.  The first pass's output of this tile, issued before the barrier: it lands under the final pass
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int m = t + 512 * j;
            const long long f = f0 + m;
            old[j] = mk(0.f, 0.f);
            if (m >= first_valid && f < p.frames) old[j] = *reinterpret_cast<const cf *>(p.out + ((long long)stream * p.frames + f) * 2);
        }
    }
    ctx.barrier();
    ctx.stamp(12);
    cf y[16];
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) y[k1] = ctx.ld(buf0 + k1 * kRowStride + t);
    {
        cf pw[16];
        tw_powers(ctx.opaque(w1), pw);
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) y[k1] = cmulc(y[k1], pw[k1]);
    }
    fft16<true>(y);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int m = t + 512 * j;
        const long long f = f0 + m;
        if (m >= first_valid && f < p.frames) {
            cf *o = reinterpret_cast<cf *>(p.out + ((long long)stream * p.frames + f) * 2);
            if (accumulate) y[j] = y[j] + old[j];      // second pass over a wide layout's remaining channels (uniform)
#ifdef AW_ABL_NOSTORE         // timing ablation only
            if (y[j].x == 1.2345e-30f)
#endif
            ctx.st_stream(o, y[j]);
        }
    }
    ctx.stamp(13);
    ctx.flush_stamps();
}

// NP: compile-time pair count (straight-line schedule, no phis around the prefetches); NP = 0 is
// the generic variant: a runtime loop over batches that always processes two pairs (a phantom
// pair has all-zero input, so whatever table it multiplies contributes nothing).
// Persistent form: the workgroup walks tiles first, first + step, ... < end (ids in the launch's
// tile order, see tile_of()).  The next tile's first frame batch is issued before the inverse of
// the current one, so its HBM latency (measured ~10 k cycles when exposed at tile start, 18 % of a
// tile) hides under the inverse transform, and no workgroup launch gap separates tiles.
struct TileId { long long stream; int tile; };

template <bool INTERIOR>
AW_HD TileId tile_of(const TileParams &p, long long id) {
    const unsigned per = (unsigned)(INTERIOR ? p.tile_hi - p.tile_lo : p.tiles_per_stream - (p.tile_hi - p.tile_lo));
    TileId r;
    const unsigned q = (unsigned)id / per;               // tile ids fit 31 bits (checked at launch): 32-bit divide
    r.stream = q;
    int tile = (int)((unsigned)id - q * per);
    if (INTERIOR) tile += p.tile_lo;
    else if (tile >= p.tile_lo) tile += p.tile_hi - p.tile_lo;
    r.tile = tile;
    return r;
}

// ACC: the tile ADDS its result to the output (second pass of a wide layout, see launch_fused_ols: layouts of 9-16
// channels run the compile-time 4-pair kernel on channels 0-7 and a second compile-time pass on the rest, because one
// kernel over 5-8 pairs — unrolled or as a run-time batch loop — spills 56-87 VGPRs at the 256-register limit).
template <class Ctx, int CS, int NP, bool INTERIOR, bool ACC = false>
AW_HD void tiles_fused_ols(Ctx &ctx, const TileParams &p, long long first, long long step, long long end) {
    const int t0 = ctx.tid();
    int t = t0, lane = ctx.lane();
    const int wave = ctx.wave();
    cf *buf0 = ctx.lds();
    cf *buf1 = buf0 + kBufElems;
    cf *twa = buf1 + kBufElems;        // LDS copies of the sub-FFT twiddles
    cf *twb = twa + kTwaElems;
    const int Cn = CS > 0 ? CS : p.n_channels;
    if (first >= end) return;
    const cf w1 = p.tw1[t];
#if AW_OLS_H
    twa[t] = hl_twiddle(p.twa, t);                       // [16][32] row twiddles of the half-wave form, one per thread
    (void)twb;
#else
    twa[t] = p.twa[t];                                   // 512 entries, one per thread
    if (t < kTwbElems) twb[t] = p.twb[t];                // visible after the first barrier below
#endif

    // whole-frame mode: both batches of a tile are loaded together (load_batch2), batch 1 waits in raw_b
    // kWide (9-16 channels in ONE pass, 5-8 compile-time pairs): two groups of eight channels; the second group is fetched
    // while the first one's last pair is transformed — raw is free since batch 0's pass 1, raw_b since batch 1's — and takes
    // the first group's place.  One accumulator, one inverse transform and one output store for all pairs (the two-pass form
    // pays a second inverse and an output read-modify-write: 9 transforms instead of 8 for 14 channels).
    constexpr bool kWide = AW_WHOLE_FRAMES != 0 && INTERIOR && !ACC && CS >= 9 && NP >= 5 && NP <= 8;
    constexpr bool kWhole = kWide || (AW_WHOLE_FRAMES != 0 && INTERIOR && CS >= 5 && (NP == 3 || NP == 4));    // also both passes of the two-pass wide form
    float raw[16][kBatchCh];
    float raw_b[kWhole ? 16 : 1][kBatchCh];
    {
        const TileId id0 = tile_of<INTERIOR>(p, first);
        if constexpr (kWhole) load_batch2<CS>(p.in + id0.stream * p.frames * Cn, (long long)id0.tile * p.hop - p.hist_len, t, raw, raw_b);
        else load_batch<CS, INTERIOR>(p, p.in + id0.stream * p.frames * Cn, p.hist + id0.stream * (long long)p.hist_len * Cn,
                                      (long long)id0.tile * p.hop - p.hist_len, t, 0, raw);
    }
    for (long long id = first; id < end; id += step) {
    // Per-iteration opaque thread index: keeps lane-dependent addresses from being hoisted out of
    // the tile loop and held live across it.
    t = ctx.opaque_i(t0);
    lane = t & 63;
    const TileId cur = tile_of<INTERIOR>(p, id);
    const long long stream = cur.stream;
    const float *in_s = p.in + stream * p.frames * Cn;
    const float *hist_s = p.hist + stream * (long long)p.hist_len * Cn;
    const long long f0 = (long long)cur.tile * p.hop - p.hist_len;     // frame of window position 0
    ctx.stamp(0);

#if AW_OLS_H
    cf wacc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) wacc[i] = mk(0.f, 0.f);
#else
    cf wacc[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) wacc[s][i] = mk(0.f, 0.f);
#endif

    // Schedule per batch of two pairs:
    //   pass 1 of both pairs -> buf0, buf1 ; issue pair 0's table loads ; barrier ;
    //   sub-FFTs + CMAC of pair 0 ; issue pair 1's table loads and the NEXT batch's frame loads
    //   (their latency hides under pair 1's sub-FFTs) ; sub-FFTs + CMAC of pair 1.
    const int n_pairs = NP > 0 ? NP : p.n_pairs;
    auto batch = [&](int pair0, bool two, bool more) {
        if (pair0 > 0) ctx.barrier();                    // every wave is done reading buf0/buf1
        ctx.stamp(pair0 > 0 ? 6 : 1);
        {
            cf pw[16];
            tw_powers(ctx.opaque(w1), pw);       // opaque: keep the 15 powers out of long-lived registers
            cf x[16];
            if constexpr (kWhole) {
                if ((pair0 & 2) != 0) {          // compile-time after unrolling: the second batch of a group waits in raw_b
#pragma unroll
                    for (int j = 0; j < 16; ++j) x[j] = mk(raw_b[j][0], raw_b[j][1]);
                    pair_pass1(x, pw, buf0, t);
                    if (two) {
#pragma unroll
                        for (int j = 0; j < 16; ++j) x[j] = mk(raw_b[j][2], raw_b[j][3]);
                        pair_pass1(x, pw, buf1, t);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) x[j] = mk(raw[j][0], raw[j][1]);
                    pair_pass1(x, pw, buf0, t);
#pragma unroll
                    for (int j = 0; j < 16; ++j) x[j] = mk(raw[j][2], raw[j][3]);
                    pair_pass1(x, pw, buf1, t);
                }
            } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) x[j] = mk(raw[j][0], raw[j][1]);
            pair_pass1(x, pw, buf0, t);
            if (two) {
#pragma unroll
                for (int j = 0; j < 16; ++j) x[j] = mk(raw[j][2], raw[j][3]);
                pair_pass1(x, pw, buf1, t);
            }
            }
        }
#if AW_OLS_H
        cf2 tab[16];
        auto load_tab_ = [&](int pr) { load_tab_h(p, pr, wave, lane, tab); };
        // tables after the transform in one go (64 VGPRs) or in two parts of eight bins.  Measured per layout (tools/ols2_ab.py WINDOW=8192,
        // tools/ubench/tile_bench.hip): 13 / 14 channels 22.5 / 21.9 -> 23.9 / 23.2 G frames/s (19 spilled VGPRs either way, fewer live through the
        // multiply-accumulate), 9 / 10 +-0, 11 / 12 channels 28.5 / 29.2 -> 24.0 / 23.0 (!), 8 channels +-0 (parts of four: -5 %) — seven pairs only.
#ifndef AW_TAB_G7
#define AW_TAB_G7 8
#endif
        constexpr int kTabG = (kTabEarly0 || kTabEarly1) ? 16 : NP == 7 ? AW_TAB_G7 : 16;
        auto subfft_cmac_ = [&](int pr, cf *bf, bool loaded) { pair_subfft_cmac_h<kTabG>(ctx, p, pr, bf, twa, tab, lane, wave, wacc, loaded); };
#else
        cf2 tab[2][8];
        auto load_tab_ = [&](int pr) { load_tab(p, pr, wave, lane, tab); };
        auto subfft_cmac_ = [&](int pr, cf *bf, bool loaded) { pair_subfft_cmac(ctx, p, pr, bf, twa, twb, tab, lane, wave, wacc, loaded); };
#endif
        if (kTabEarly0) load_tab_(pair0);
        ctx.stamp(pair0 > 0 ? 7 : 2);
        ctx.barrier();
        ctx.stagger(wave, p.stagger);
        ctx.stamp(pair0 > 0 ? 8 : 3);
        subfft_cmac_(pair0, buf0, kTabEarly0);
        ctx.stamp(pair0 > 0 ? 9 : 4);
        const int pair1 = pair0 + 1;     // a phantom second pair (odd pair count, runtime loop) lands on the zero pair after the last one
        if (two && kTabEarly1) load_tab_(pair1);
        if constexpr (kWide) {
            if (pair0 == 2) load_batch2<CS, (NP == 8 && !AW_WIDE_LATE_B ? 4 : NP == 7 && !AW_WIDE_LATE_B ? 2 : 0)>(in_s + 8, f0, t, raw, raw_b);      // channels 8-15 (a 9-12 channel layout: 8-11 only)
        }
        if (!kWhole && kPrefetchRawEarly && more) load_batch<CS, INTERIOR>(p, in_s, hist_s, f0, t, 2 * (pair0 + 2), raw);
        if (two) subfft_cmac_(pair1, buf1, kTabEarly1);
        if constexpr (kWide && AW_WIDE_LATE_B != 0 && NP >= 7) {      // the group's second batch one pair later (an L2 hit): fewer values live through the CMAC
            if (pair0 == 2) load_batch2<CS, (NP == 8 ? 4 : 2), false>(in_s + 8, f0, t, raw, raw_b);
        }
        if (!kWhole && !kPrefetchRawEarly && more) load_batch<CS, INTERIOR>(p, in_s, hist_s, f0, t, 2 * (pair0 + 2), raw);
        ctx.stamp(pair0 > 0 ? 10 : 5);
    };
    if constexpr (NP > 0) {
#pragma unroll
        for (int b = 0; b < (NP + 1) / 2; ++b) batch(2 * b, 2 * b + 1 < NP, 2 * b + 2 < NP);
    } else {
        // an odd pair count's last batch transforms one pair, not a phantom second one (9, 10, 13, 14 channels: +5-10 %);
        // layouts known at compile time to have an even pair count keep the branch-free body
        constexpr bool kEvenPairs = CS > 0 && ((CS + 1) / 2) % 2 == 0;
        for (int pair0 = 0; pair0 < n_pairs; pair0 += 2)
            batch(pair0, (kSkipPhantom && !kEvenPairs) ? pair0 + 1 < n_pairs : true, pair0 + 2 < n_pairs);
    }

#if AW_OLS_H
    tile_inverse_rows_h(ctx, wacc, buf0, twa);
#else
    tile_inverse_rows(ctx, wacc, buf0, twa, twb);
#endif
    {   // prefetch the next tile's first batch here, where few registers are live (unconditional:
        // the last iteration re-reads its own batch; a branch would put phis on 64 registers)
        const TileId nx = tile_of<INTERIOR>(p, id + step < end ? id + step : id);
        if constexpr (kWhole) load_batch2<CS>(p.in + nx.stream * p.frames * Cn, (long long)nx.tile * p.hop - p.hist_len, t, raw, raw_b);
        else load_batch<CS, INTERIOR>(p, p.in + nx.stream * p.frames * Cn, p.hist + nx.stream * (long long)p.hist_len * Cn,
                                      (long long)nx.tile * p.hop - p.hist_len, t, 0, raw);
    }
    tile_inverse_final(ctx, p, buf0, w1, t, stream, f0, p.hist_len, ACC);
    ctx.barrier();                                       // the final exchange has been read before buf0 is rewritten
    }
}

// ---- partitioned path (taps too long for one window: Y[b] = sum_q X[b-q] . H_q, the same
// frequency-domain delay line as ConvolutionEngine.swift:256-350 with B = N/2 = 4096) -----------
// Kernel 1: spectra of one input window for every pair -> global scratch.
// MODE 1 (interior): the window lies inside the call's input (whole-frame vector loads off a uniform base);
// MODE 2 (head): history + input, nothing past the end (vector loads behind a per-frame pointer select);
// MODE 0: anything (scalar loads; history, input or the zero page per frame).
// One 512-bin row of a spectrum from registers (lane holds bins lane + 64 kc) to memory as 16-byte stores: neighbouring
// lanes trade one value per pair of chunks, so that even lanes store bins (lane, lane + 1) of chunk 2m and odd lanes bins
// (lane - 1, lane) of chunk 2m + 1 — half the store instructions of 8-byte stores, every instruction still two whole
// 512-byte runs (the spectrum stores are the forward kernel's issue-bound tail: 10.9 ms with them, 7.9 ms without).
#ifndef AW_FWD_STORE16
#define AW_FWD_STORE16 0     // measured cfg 3: 10.95 ms with 16-byte stores against 10.66 ms with 8-byte ones — the stores are not issue-bound
#endif
template <class Ctx>
AW_HD void store_row16(Ctx &ctx, cf *row, const cf (&z)[8], int lane) {
#if AW_FWD_STORE16
    const bool odd = (lane & 1) != 0;
    cf *base = row + (lane & ~1);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const cf na = ctx.xchg1(z[2 * m]), nb = ctx.xchg1(z[2 * m + 1]);
        cf2 v;
        v.a = odd ? nb : z[2 * m];
        v.b = odd ? z[2 * m + 1] : na;
        *reinterpret_cast<cf2 *>(base + 64 * (2 * m + (odd ? 1 : 0))) = v;
    }
#else
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) ctx.st_stream(row + lane + 64 * kc, z[kc]);
#endif
}

// Persistent form: the workgroup walks window ids first, first + step, ... < end of its launch; id -> (stream, window)
// through `per` windows per stream starting at window `w0` (MODE 0: the windows outside [w0, w0 + skip) — see the launcher).
// The next window's first frame batch is issued right after the current window's last pass 1, so its latency hides under
// the last sub-FFTs, and the spectrum stores of one window drain while the next one computes (a workgroup that ends
// after its stores cannot overlap them with anything: measured 11.2 ms with, 7.9 ms without the stores).
struct PartWin { long long stream; int w; };
template <int MODE>
AW_HD PartWin part_window_of(long long id, int per, int w0, int skip) {
    PartWin r;
    const unsigned q = (unsigned)id / (unsigned)per;          // ids fit 31 bits (checked at launch)
    r.stream = q;
    int w = (int)((unsigned)id - q * (unsigned)per);
    if (MODE == 0) { if (w >= w0) w += skip; }
    else w += w0;
    r.w = w;
    return r;
}

template <class Ctx, int CS, int MODE>
AW_HD void tiles_part_forward(Ctx &ctx, const TileParams &p, long long first, long long step, long long end, int per, int w0, int skip) {
    auto load = [&](const float *in_s, const float *hist_s, long long f0, int t, int c0, float (&raw)[16][kBatchCh]) {
        if constexpr (MODE == 2) load_batch_head<CS>(p, in_s, hist_s, f0, t, c0, raw);
        else load_batch<CS, MODE == 1>(p, in_s, hist_s, f0, t, c0, raw);
        // Frames that are not whole float4s: the whole-frame vector loads put the next frame's first samples into the padding lanes
        // of the last batch.  Zero tables would cancel finite values, but (a) the Hermitian shortcut (herm_last) needs an odd layout's
        // last pair REAL and (b) the frame after a stream's last history frame belongs to the NEXT stream: a NaN there must not
        // reach this stream (0 x NaN) — streams are independent in the reference.  So the padding lanes are zeroed.
        if constexpr (MODE != 0 && CS > 2 && CS % 4 != 0) {
            if (c0 + kBatchCh > CS) {                              // uniform: the last batch
#pragma unroll
                for (int j = 0; j < 16; ++j)
#pragma unroll
                    for (int c = 0; c < kBatchCh; ++c)
                        if (c0 + c >= CS) raw[j][c] = 0.f;
            }
        }
    };
    if (first >= end) return;
    const int t0 = ctx.tid();
    int t = t0, lane = ctx.lane();
    const int wave = ctx.wave();
    cf *buf0 = ctx.lds();
    cf *buf1 = buf0 + kBufElems;
    cf *twa = buf1 + kBufElems;
    cf *twb = twa + kTwaElems;
    const int Cn = CS > 0 ? CS : p.n_channels;
    const int n_windows = p.n_blocks + p.partitions - 1;
    const cf w1 = p.tw1[t];
    twa[t] = p.twa[t];
    if (t < kTwbElems) twb[t] = p.twb[t];

    // whole-frame mode (interior windows of 5-8 channel layouts): both four-channel batches of a window are loaded together
    // (load_batch2: every line crosses the L2 -> L1 path once), the second batch waits in raw_b
    constexpr bool kWhole = AW_WHOLE_FRAMES != 0 && (MODE == 1 || MODE == 2) && CS >= 5 && CS <= 8;
    float raw[16][kBatchCh];
    float raw_b[kWhole ? 16 : 1][kBatchCh];
    auto load_window = [&](const PartWin &w) {
        const float *in_w = p.in + w.stream * p.frames * Cn;
        const long long fw = ((long long)w.w - p.partitions) * p.hop;
        if constexpr (kWhole) {
            if constexpr (MODE == 2) load_batch2_head<CS>(p, in_w, p.hist + w.stream * (long long)p.hist_len * Cn, fw, t, raw, raw_b);
            else load_batch2<CS>(in_w, fw, t, raw, raw_b);
            if constexpr (CS % 4 != 0) {                    // the padding lanes of the last batch must be real zeros (herm_last; no NaN from the next stream's history)
#pragma unroll
                for (int j = 0; j < 16; ++j)
#pragma unroll
                    for (int c = 0; c < kBatchCh; ++c)
                        if (4 + c >= CS) raw_b[j][c] = 0.f;
            }
        } else {
            load(in_w, p.hist + w.stream * (long long)p.hist_len * Cn, fw, t, 0, raw);
        }
    };
    load_window(part_window_of<MODE>(first, per, w0, skip));
    for (long long id = first; id < end; id += step) {
        t = ctx.opaque_i(t0);                       // keeps lane-dependent addresses from living across the window loop
        lane = t & 63;
        const PartWin cur = part_window_of<MODE>(id, per, w0, skip);
        const float *in_s = p.in + cur.stream * p.frames * Cn;
        const float *hist_s = p.hist + cur.stream * (long long)p.hist_len * Cn;
        const long long f0 = ((long long)cur.w - p.partitions) * p.hop;      // window w covers blocks (w-P, w-P+1)
        cf *spec_w = p.spec + ((cur.stream * n_windows + cur.w) * p.n_pairs) * (long long)kN;
        for (int pair0 = 0; pair0 < p.n_pairs; pair0 += 2) {
            const bool more = pair0 + 2 < p.n_pairs;
            if (pair0 > 0 || id != first) ctx.barrier();          // every wave is done reading buf0/buf1
            {
                cf pw[16];
                tw_powers(ctx.opaque(w1), pw);
                cf x[16];
                if (kWhole && pair0 > 0) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) x[j] = mk(raw_b[kWhole ? j : 0][0], raw_b[kWhole ? j : 0][1]);
                    pair_pass1(x, pw, buf0, t);
#pragma unroll
                    for (int j = 0; j < 16; ++j) x[j] = mk(raw_b[kWhole ? j : 0][2], raw_b[kWhole ? j : 0][3]);
                    pair_pass1(x, pw, buf1, t);
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) x[j] = mk(raw[j][0], raw[j][1]);
                    pair_pass1(x, pw, buf0, t);
#pragma unroll
                    for (int j = 0; j < 16; ++j) x[j] = mk(raw[j][2], raw[j][3]);
                    pair_pass1(x, pw, buf1, t);
                }
            }
            // the next batch's frames — of this window, or the first batch of the workgroup's next window — travel while
            // this batch's sub-FFTs run (no accumulators or tables live here)
            if (more) { if (!kWhole) load(in_s, hist_s, f0, t, 2 * (pair0 + 2), raw); }
            else load_window(part_window_of<MODE>(id + step < end ? id + step : id, per, w0, skip));     // the last one re-reads its own frames
            ctx.barrier();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (pair0 + h >= p.n_pairs) break;            // uniform
                cf *buf = h == 0 ? buf0 : buf1;
                cf *row0 = buf + wave_row(wave, 0) * kRowStride;
                cf *row1 = buf + wave_row(wave, 1) * kRowStride;
                cf z[2][8];
                ctx.template ld8x2<64>(z[0], row0 + lane, z[1], row1 + lane);
                ctx.wave_sync();
                sub_fft512x2<false>(ctx, z, row0, row1, twa, twb, lane);
                cf *dst = spec_w + (long long)(pair0 + h) * kN;
                // a real-only last pair (odd channel count): Z[N-k] = conj(Z[k]); rows 9..15 (slot 1 of waves 1..7) are not stored
                const bool skip1 = p.herm_last && pair0 + h == p.n_pairs - 1 && wave != 0;      // uniform
#ifdef AW_ABL_FWD_NOSTORE      // timing ablation only (wrong results): one store per wave instead of 16
                if (z[0][0].x == 1.2345e-30f) dst[lane] = z[1][7];
                continue;
#endif
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    if (s == 1 && skip1) break;
                    store_row16(ctx, dst + wave_row(wave, s) * kSub, z[s], lane);
                }
            }
        }
    }
}

// one window (the CPU emulation harness and non-persistent launches)
template <class Ctx, int CS, int MODE>
AW_HD void tile_part_forward(Ctx &ctx, const TileParams &p, long long stream, int widx) {
    const int n_windows = p.n_blocks + p.partitions - 1;
    tiles_part_forward<Ctx, CS, MODE>(ctx, p, stream * n_windows + widx, 1, stream * n_windows + widx + 1, n_windows, MODE == 0 ? n_windows : 0, 0);
}

// Kernel 1, one-pair form: ONE channel pair of one input window per workgroup — one exchange buffer (78 KB of LDS) and
// no register batch of four channels, so two workgroups share a CU like the inverse kernel's (which moves the same bytes
// per transform and runs 4.4 us per transform against 6.2 here with one 152-KB workgroup per CU).  The four pair
// workgroups of a window are neighbours in the launch order of one XCD: the lines they all read meet in its L2.

template <class Ctx, int CS, int MODE>
AW_HD void tile_part_forward1(Ctx &ctx, const TileParams &p, long long stream, int widx, int pair) {
    const int t = ctx.tid();
    const int lane = ctx.lane(), wave = ctx.wave();
    cf *buf0 = ctx.lds();
    cf *twa = buf0 + kBufElems;
    cf *twb = twa + kTwaElems;
    const int Cn = CS > 0 ? CS : p.n_channels;
    const float *in_s = p.in + stream * p.frames * Cn;
    const float *hist_s = p.hist + stream * (long long)p.hist_len * Cn;
    const long long f0 = ((long long)widx - p.partitions) * p.hop;      // window widx covers blocks (widx-P, widx-P+1)
    const int n_windows = p.n_blocks + p.partitions - 1;
    cf *dst = p.spec + ((stream * n_windows + widx) * p.n_pairs + pair) * (long long)kN;
    const int c0 = 2 * pair;
    const bool has_b = c0 + 1 < Cn;                 // uniform: an odd channel count's last pair has no second channel
    cf x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const long long f = f0 + t + 512 * j;
        if constexpr (MODE != 0 && CS > 0) {
            const float *src = (MODE == 2 && f < 0 ? hist_s + ((long long)p.hist_len + f) * CS : in_s + f * CS) + c0;
            if constexpr (CS % 2 == 0) {
                const f2 v = *reinterpret_cast<const f2 *>(src);
                x[j] = mk(v.x, v.y);
            } else {
                const f2u v = *reinterpret_cast<const f2u *>(src);      // the last pair's .y is the next frame's first sample: dropped below
                x[j] = mk(v.x, v.y);
            }
        } else {
            const bool before = f < 0, past = f >= p.frames;
            const float *src = before ? hist_s + ((long long)p.hist_len + f) * Cn : (past ? p.zeros : in_s + f * Cn);
            const float *qa = src + c0, *qb = has_b ? src + c0 + 1 : p.zeros;
            x[j] = mk(*qa, *qb);
        }
    }
    if constexpr (MODE != 0 && CS > 0 && (CS & 1)) {
        if (!has_b) {
#pragma unroll
            for (int j = 0; j < 16; ++j) x[j].y = 0.f;
        }
    }
    const cf w1 = p.tw1[t];
    twa[t] = p.twa[t];
    if (t < kTwbElems) twb[t] = p.twb[t];
    {
        cf pw[16];
        tw_powers(ctx.opaque(w1), pw);
        pair_pass1(x, pw, buf0, t);
    }
    ctx.barrier();
    cf *row0 = buf0 + wave_row(wave, 0) * kRowStride;
    cf *row1 = buf0 + wave_row(wave, 1) * kRowStride;
    cf z[2][8];
    ctx.template ld8x2<64>(z[0], row0 + lane, z[1], row1 + lane);
    ctx.wave_sync();
    sub_fft512x2<false>(ctx, z, row0, row1, twa, twb, lane);
    const bool skip1 = p.herm_last && pair == p.n_pairs - 1 && wave != 0;      // uniform (see tile_part_forward)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (s == 1 && skip1) break;
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) ctx.st_stream(dst + wave_row(wave, s) * kSub + lane + 64 * kc, z[s][kc]);
    }
}

// Kernel 2 (no LDS, one thread per frequency bin): W_b[k] = sum over partitions q and pairs of
//   Z_{b-q}[k] A_q[k] + conj(Z_{b-q}[N-k]) B_q[k]      for kCmacBlocks consecutive output blocks b.
// A thread keeps its bin's table entries of kCmacQ partitions in registers and applies them to every
// block of the group, and the blocks share all but kCmacBlocks-1 of their input windows: per block
// the kernel reads (R+P-1)/R windows and P/R table sets instead of P and P (R = kCmacBlocks).  All
// loads of a (pair, partition chunk) step are independent, and without LDS the kernel runs at full
// occupancy — the fused CMAC+inverse kernel it replaces was load-latency bound at 8 waves per CU.
#ifndef AW_CMAC_BLOCKS
#define AW_CMAC_BLOCKS 8      // measured cfg 4 / cfg 3 Gframes/s: 2 -> 11.8 / 10.1, 4 -> 13.4 / 12.4, 8 -> 13.8 / 13.4, 12 -> 13.8 / 13.8, 16 -> 13.5 / 13.6
#endif
constexpr int kCmacBlocks = AW_CMAC_BLOCKS;
constexpr int kCmacQ = 8;
constexpr int kCmacThreads = 256;

AW_HD void part_cmac_bin(const TileParams &p, long long stream, int block0, int i) {
    const int k1 = i >> 9, k2 = i & 511;
    // partner bin N-k in the [k1][k2] storage order: row (16-k1)&15, column (512-k2)&511 for row 0, else 511-k2
    const int pi = (((16 - k1) & 15) << 9) + (k1 == 0 ? ((512 - k2) & 511) : (511 - k2));
    const int P = p.partitions;
    const int n_windows = p.n_blocks + P - 1;
    const int nb = p.n_blocks - block0 < kCmacBlocks ? p.n_blocks - block0 : kCmacBlocks;
    cf w[kCmacBlocks];
#pragma unroll
    for (int r = 0; r < kCmacBlocks; ++r) w[r] = mk(0.f, 0.f);
    const cf *spec_s = p.spec + stream * (long long)n_windows * p.n_pairs * kN;
    for (int pair = 0; pair < p.n_pairs; ++pair) {
        for (int q0 = 0; q0 < P; q0 += kCmacQ) {
            const int nq = P - q0 < kCmacQ ? P - q0 : kCmacQ;
            cf2 tb[kCmacQ];
#pragma unroll
            for (int j = 0; j < kCmacQ; ++j) {
                const int q = q0 + (j < nq ? j : 0);                        // phantom entries re-read a valid one, never used
                tb[j] = p.tab[((long long)q * p.n_pairs + pair) * kN + i];
            }
            // block r (of the group) and partition q0+j read window  block0 + r + (P-1) - (q0+j):
            // walk the diagonals d = r - j so that both the block and the table index are compile-time
#pragma unroll
            for (int d = -(kCmacQ - 1); d < kCmacBlocks; ++d) {
                const int u = d + P - 1 - q0;                                // window relative to block0
                const bool used = (d >= -(nq - 1)) && (d <= nb - 1);        // some (r, j) on this diagonal is real (uniform)
                if (!used) continue;
                const cf *zs = spec_s + ((long long)(block0 + u) * p.n_pairs + pair) * kN;
                const cf z = zs[i], zp = zs[pi];
#pragma unroll
                for (int r = 0; r < kCmacBlocks; ++r) {
                    const int j = r - d;
                    if (j < 0 || j >= kCmacQ) continue;                      // compile-time
                    if (r < nb && j < nq) {
                        w[r] = cfma(z, tb[j].a, w[r]);
                        w[r] = cfmac(zp, tb[j].b, w[r]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < kCmacBlocks; ++r)
        if (r < nb) p.wspec[((stream * p.n_blocks) + block0 + r) * (long long)kN + i] = w[r];
}

// Kernel 3: inverse transform of one output block's W, store the last hop frames.  One exchange
// buffer only (78 KB of LDS: two workgroups per CU).
constexpr int kInvLdsElems = kBufElems + kTwaElems + kTwbElems;
constexpr int kInvLdsBytes = kInvLdsElems * 8;

template <class Ctx>
AW_HD void tile_part_inverse(Ctx &ctx, const TileParams &p, long long stream, int block) {
    const int t = ctx.tid();
    const int lane = ctx.lane(), wave = ctx.wave();
    cf *buf0 = ctx.lds();
    cf *twa = buf0 + kBufElems;
    cf *twb = twa + kTwaElems;
    const cf w1 = p.tw1[t];
    twa[t] = p.twa[t];
    if (t < kTwbElems) twb[t] = p.twb[t];
    const cf *ws = p.wspec + ((stream * p.n_blocks) + block) * (long long)kN;
    cf wacc[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) wacc[s][kc] = ws[wave_row(wave, s) * kSub + lane + 64 * kc];
    ctx.barrier();                                        // twiddles visible
    const long long f0 = ((long long)block - 1) * p.hop;  // window of block b: frames [(b-1)B, (b+1)B)
    tile_inverse_rows(ctx, wacc, buf0, twa, twb);
    tile_inverse_final(ctx, p, buf0, w1, t, stream, f0, p.first_valid);
}

}  // namespace awk
