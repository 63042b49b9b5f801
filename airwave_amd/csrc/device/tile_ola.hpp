// tile_ola.hpp — the fused 8192-frame tile as OVERLAP-ADD with the carry in registers (round 6).
//
// Same transform machinery, same tables and same arithmetic per window as tile_ols.hpp (FFT_8192 = 16 x 512, half-wave row
// transforms, W += Z A + conj(Z[N-k]) B), but the window is a BLOCK of hop = 512 H NEW frames followed by zeros instead of
// `taps - 1` old frames followed by `hop` new ones — the formulation of the reference itself (ConvolutionEngine.swift:232-367 is
// an overlap-add FIR; the overlap-save tile re-reads N - hop frames of every window).  What that buys on this machine:
//
//   * a thread holds window positions t + 512 j, and only j < H of them are frames: H x C floats instead of 16 x C.  For 14-channel
//     frames and H = 7 that is 98 VGPRs — the WHOLE frame of every position at once, where the overlap-save tile could hold eight
//     channels (128 VGPRs) and pulled every 128-byte line of a 459-KB window through L2 -> L1 twice per tile (DESIGN 4.1, 4.5c:
//     the same arithmetic 22 % slower on 56-byte frames).  Here every input line is read exactly once per call: 200 KB per tile.
//   * the inverse transform leaves thread t with block positions t + 512 j, j < 16: j < H are finished output frames (after adding
//     the carry), j >= H is the tail that belongs to the next blocks.  hop is a multiple of 512, so position t + 512 j of this block
//     is position t + 512 (j - H) of the next one: THE SAME THREAD.  The overlap-add is a register rename (`carry`, 16 - H complex
//     values per thread) — no output read-modify-write, no atomics, no LDS.
//   * a workgroup walks a contiguous run of blocks of one stream.  A run that starts in the middle of a stream (or at its start, with
//     the history of earlier calls before it) first transforms the ceil((taps-1)/hop) blocks before its first one with the stores
//     suppressed: that rebuilds the carry from the input alone, so runs are independent and the result does not depend on how the
//     launch cuts the streams (bit-reproducible).  Cost: <= 2 blocks per run, ~3 % at cfg 2's 67 blocks per workgroup.
//   * frames are fetched with BUFFER loads: the descriptor of a block's source (the stream's input, or its history rows for blocks
//     before frame 0) carries the byte size, and the hardware returns zeros past it.  The ragged last block of a stream, the blocks
//     that start in the history and the part of the history window before its first row therefore run the very same code as interior
//     blocks: ONE launch, no boundary kernel, no per-frame pointer selects, no zero page.
//
// State between calls stays what every other path of the runtime keeps: the last `hist_len` INPUT frames (aw_hist_update_kernel), so
// a spatializer may take this tile for one call and any other kernel family for the next.
//
// Shared with the CPU thread-emulation harness (tests/emu/): the Ctx supplies buf() / buf_ld<N>().
#pragma once
#include <utility>
#include "tile_ols.hpp"

namespace awk {

constexpr unsigned kOlaOutOfRange = 0x7fffff00u;     // byte offset past every descriptor this tile makes (the launch checks sizes <= kOlaMaxBytes)
constexpr long long kOlaMaxBytes = 0x7ffffe00LL;

// When the next block's frames are requested: late = between the inverse row transforms and the final pass (few registers live), early =
// before the last batch's row transforms (H x C more registers live through its CMAC).  tools/ubench/ola_bench (128 streams x 10 s, 4320
// taps, G frames/s late / early): 8 channels 52.4 / 49.9, 12 channels 36.0 / 33.9, 14 channels 25.7 / 29.7 — wide layouts have a long tail of pairs to hide the
// burst under, narrow ones only lose registers.  -DAW_OLA_PREFETCH_EARLY=0|1 forces one form for every layout (A/B).
#ifdef AW_OLA_PREFETCH_EARLY
template <int NP> struct OlaEarly { static constexpr bool value = AW_OLA_PREFETCH_EARLY != 0; };
#else
template <int NP> struct OlaEarly { static constexpr bool value = NP >= 7; };
#endif
#ifndef AW_OLA_TABG
#define AW_OLA_TABG 0                  // table entries per part (pair_subfft_cmac_h): 0 = per layout (8 from seven pairs, else 16)
#endif
#ifndef AW_OLA_TAB_EARLY
#define AW_OLA_TAB_EARLY 0             // bit 0: layouts of up to four pairs request a pair's tables before its row transform
#endif
#ifndef AW_OLA_PAIR2
#define AW_OLA_PAIR2 0                 // 1: layouts of up to four pairs run the two pairs of a batch through the row transforms together (ola_subfft_cmac2)
#endif
#ifndef AW_OLA_PAIR2_G
#define AW_OLA_PAIR2_G 4
#endif
#ifndef AW_OLA_PAIR2_MAXNP
#define AW_OLA_PAIR2_MAXNP 4
#endif
#ifndef AW_OLA_PREFETCH_MID
#define AW_OLA_PREFETCH_MID 0          // 1: the next block's frames are requested right behind the block's last table request (ola_subfft_cmac)
#endif

// One block of one stream: frames idx0 + t + 512 j (j < H) of the source behind `src`, every channel, into raw[j][*].
// idx is relative to the descriptor's first row and may be negative (history blocks reach before the first kept row): those
// lanes ask for an offset no descriptor covers and get zeros like the rows past the end.
template <int CS, int H, class Ctx>
AW_HD void ola_load_block(Ctx &ctx, const typename Ctx::Buf &src, int idx0, int t, float (&raw)[H][2 * ((CS + 1) / 2)]) {
    constexpr int CP = 2 * ((CS + 1) / 2);
#ifdef AW_ABL_OLA_NOLOAD       // timing ablation only (wrong results): no frame loads
#pragma unroll
    for (int j = 0; j < H; ++j)
#pragma unroll
        for (int c = 0; c < CP; ++c) raw[j][c] = 0.001f * (float)(t + c) + (float)idx0 + 0.5f * j;
    return;
#endif
#pragma unroll
    for (int j = 0; j < H; ++j) {
        const int idx = idx0 + t + 512 * j;
        const unsigned off = idx < 0 ? kOlaOutOfRange : (unsigned)idx * (unsigned)(CS * 4);
        int c = 0;
#pragma unroll
        for (; c + 4 <= CS; c += 4) ctx.template buf_ld<4>(src, off, 4 * c, &raw[j][c]);
        if constexpr (CS % 4 >= 2) { ctx.template buf_ld<2>(src, off, 4 * (CS & ~3), &raw[j][CS & ~3]); }
        if constexpr (CS % 2 == 1) { ctx.template buf_ld<1>(src, off, 4 * (CS - 1), &raw[j][CS - 1]); raw[j][CP - 1] = 0.0f; }
    }
}

// pass 1 of one pair on a block: x[j] = 0 for j >= H.  The first layer of the 4 x 4 radix-16 form sees (a, b, 0, 0) or (a, 0, 0, 0):
// written out, because x + 0 does not fold under IEEE rules (-0 + 0 = +0).
template <int H>
AW_HD void ola_pass1(const cf (&xin)[H], const cf (&pw)[16], cf *buf, int t) {
    static_assert(H >= 1 && H <= 8, "blocks of at most half a window");
    cf v[16];
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) {
        if (n0 + 4 < H) {                       // (a, b, 0, 0)
            const cf a = xin[n0], b = xin[n0 + 4], r = rot90<false>(b);
            v[n0] = a + b; v[n0 + 4] = a + r; v[n0 + 8] = a - b; v[n0 + 12] = a - r;
        } else if (n0 < H) {                    // (a, 0, 0, 0)
            v[n0] = xin[n0]; v[n0 + 4] = xin[n0]; v[n0 + 8] = xin[n0]; v[n0 + 12] = xin[n0];
        } else {
            v[n0] = mk(0.f, 0.f); v[n0 + 4] = mk(0.f, 0.f); v[n0 + 8] = mk(0.f, 0.f); v[n0 + 12] = mk(0.f, 0.f);
        }
    }
    // second layer and transpose: as fft16<false> (cplx.hpp)
    fft4<false>(v[0], v[1], v[2], v[3]);
    fft4_w16<false, 1>(v[4], v[5], v[6], v[7]);
    fft4_w8<false>(v[8], v[9], v[10], v[11]);
    fft4_w16<false, 3>(v[12], v[13], v[14], v[15]);
    cf s;
    s = v[1]; v[1] = v[4]; v[4] = s;
    s = v[2]; v[2] = v[8]; v[8] = s;
    s = v[3]; v[3] = v[12]; v[12] = s;
    s = v[6]; v[6] = v[9]; v[9] = s;
    s = v[7]; v[7] = v[13]; v[13] = s;
    s = v[11]; v[11] = v[14]; v[14] = s;
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) v[k1] = cmul(v[k1], pw[k1]);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) buf[k1 * kRowStride + t] = v[k1];
}

// Row transforms + multiply-accumulate of one pair (pair_subfft_cmac_h of tile_ols.hpp) with two hooks for this tile:
//   TAB_EARLY  the pair's 16 table entries were requested by the caller BEFORE the row transform (64 VGPRs that the overlap-save tile,
//              with 128 registers of frames, never had): they travel under the transform instead of being waited for after it;
//   PREFETCH   right behind the pair's LAST table request, the next block's frame loads are queued into `raw` — vector-memory
//              results return in issue order, so frames requested before a table load would hold that load's consumer up for an HBM
//              round trip, frames requested after the last one delay nobody.
template <int N, class F, int... I> AW_HD void ola_static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> AW_HD void ola_static_for(F &&f) { ola_static_for_impl<N>(static_cast<F &&>(f), std::make_integer_sequence<int, N>{}); }

template <class Buf> struct OlaNext { Buf src; int idx0; };     // the next block's source and first row (wave-uniform)
template <int G, bool TAB_EARLY, bool PREFETCH, int CS, int H, class Ctx>
AW_HD void ola_subfft_cmac(Ctx &ctx, const TileParams &p, int pair, cf *buf, const cf *twa, cf2 (&tab)[16], int lane, int wave, cf (&wacc)[16],
                           const OlaNext<typename Ctx::Buf> &next, int t, float (&raw)[H][2 * ((CS + 1) / 2)]) {
    static_assert(!TAB_EARLY || G == 16, "early tables come whole");
    const HLane L = hl_make(ctx, buf, twa, lane, wave);
    auto after = [&] { if constexpr (PREFETCH) ola_load_block<CS, H>(ctx, next.src, next.idx0, t, raw); };
    if constexpr (TAB_EARLY) after();
    cf z[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) z[j] = ctx.ld(L.row + L.h + 32 * j);
    sub_fft512h_fwd(ctx, z, L);
    if constexpr (G == 16) { if constexpr (!TAB_EARLY) { load_tab_h(p, pair, wave, lane, tab); after(); } }
    else load_tab_part_h<G>(p, pair, wave, lane, 0, tab);
#ifndef AW_ABL_OLA_NOPARTNER    // (timing ablation, wrong results: no publish of Z / no partner reads through LDS)
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) L.row[L.col + 32 * kb] = z[kb];
    ctx.wave_sync();
#endif
    if constexpr (G == 16) {
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            int idx = L.pidx - 32 * kb;
            if (kb == 0) idx &= 511;                               // only (row 0, column 0) wraps: 512 -> 0
#ifdef AW_ABL_OLA_NOPARTNER
            const cf zp = z[15 - kb];
#else
            const cf zp = ctx.ld(L.prow + idx);
#endif
            wacc[kb] = cfma(z[kb], tab[kb].a, wacc[kb]);
            wacc[kb] = cfmac(zp, tab[kb].b, wacc[kb]);
        }
    } else {
        static_assert(2 * G <= 16, "two parts in flight inside tab[16]");
#pragma unroll
        for (int part = 0; part < 16 / G; ++part) {
            cf2 *cur = tab + G * (part & 1);
            if (part + 1 < 16 / G) {
                load_tab_part_h<G>(p, pair, wave, lane, part + 1, tab + G * ((part + 1) & 1));
                if (part + 2 == 16 / G) after();
            }
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
}

// Two pairs of a batch through the row transforms and the multiply-accumulate TOGETHER (AW_OLA_PAIR2): the same instruction sequence as
// ola_subfft_cmac, with every step issued for both pairs before the wave waits — pair p in buf0, pair p + 1 in buf1, so the exchanges do
// not collide — which halves the number of LDS round trips a wave sits through per pair (the tile is bound by that chain, not by its
// butterflies: tools/ubench/ola_bench with AW_ABL_NOFFT runs no faster) and shares the row twiddles between the two transforms.
// 32 more VGPRs for the second spectrum and table parts of four entries for both pairs: layouts of up to four pairs only.
template <int G, class Ctx>
AW_HD void ola_subfft_cmac2(Ctx &ctx, const TileParams &p, int pair, cf *buf_a, cf *buf_b, const cf *twa, int lane, int wave, cf (&wacc)[16]) {
    static_assert(G == 4 || G == 8, "table parts in flight for two pairs");
    const HLane La = hl_make(ctx, buf_a, twa, lane, wave), Lb = hl_make(ctx, buf_b, twa, lane, wave);
    cf za[16], zb[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { za[j] = ctx.ld(La.row + La.h + 32 * j); zb[j] = ctx.ld(Lb.row + Lb.h + 32 * j); }
    // sub_fft512h_fwd for both (tile_ols.hpp), step by step
    fft16<false>(za);
    fft16<false>(zb);
#pragma unroll
    for (int m = 1; m < 16; m += 8) {                 // row twiddles w_512^{h ka}: one LDS read serves both transforms
        cf w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) if (m + i < 16) w[i] = ctx.ld(La.twh + 32 * (m + i));
#pragma unroll
        for (int i = 0; i < 8; ++i) if (m + i < 16) { za[m + i] = twmul<false>(za[m + i], w[i]); zb[m + i] = twmul<false>(zb[m + i], w[i]); }
    }
    const cf w32 = ctx.opaque(La.w32);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        ctx.xswap(za[q], za[q + 8], 4);
        ctx.xswap(zb[q], zb[q + 8], 4);
        const cf ea = za[q] + za[q + 8], oa = za[q] - za[q + 8];
        const cf eb = zb[q] + zb[q + 8], ob = zb[q] - zb[q + 8];
        za[q] = ea; za[q + 8] = cmul(oa, w32);
        zb[q] = eb; zb[q + 8] = cmul(ob, w32);
    }
    ctx.wave_sync();
#pragma unroll
    for (int r = 0; r < 16; ++r) { La.row[La.e2w + 17 * r] = za[r]; Lb.row[Lb.e2w + 17 * r] = zb[r]; }
    ctx.wave_sync();
#pragma unroll
    for (int r = 0; r < 16; ++r) { za[r] = ctx.ld(La.row + La.e2r + r); zb[r] = ctx.ld(Lb.row + Lb.e2r + r); }
    ctx.wave_sync();
    fft16<false>(za);
    fft16<false>(zb);
    // tables: parts of G entries per pair, two parts per pair in flight
    cf2 ta[2 * G], tb[2 * G];
    load_tab_part_h<G>(p, pair, wave, lane, 0, ta);
    load_tab_part_h<G>(p, pair + 1, wave, lane, 0, tb);
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) { La.row[La.col + 32 * kb] = za[kb]; Lb.row[Lb.col + 32 * kb] = zb[kb]; }
    ctx.wave_sync();
#pragma unroll
    for (int part = 0; part < 16 / G; ++part) {
        cf2 *ca = ta + G * (part & 1), *cb = tb + G * (part & 1);
        if (part + 1 < 16 / G) {
            load_tab_part_h<G>(p, pair, wave, lane, part + 1, ta + G * ((part + 1) & 1));
            load_tab_part_h<G>(p, pair + 1, wave, lane, part + 1, tb + G * ((part + 1) & 1));
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int kb = G * part + i;
            int idx = La.pidx - 32 * kb;
            if (kb == 0) idx &= 511;                               // only (row 0, column 0) wraps: 512 -> 0
            const cf pa = ctx.ld(La.prow + idx), pb = ctx.ld(Lb.prow + idx);
            wacc[kb] = cfma(za[kb], ca[i].a, wacc[kb]);
            wacc[kb] = cfmac(pa, ca[i].b, wacc[kb]);
            wacc[kb] = cfma(zb[kb], cb[i].a, wacc[kb]);
            wacc[kb] = cfmac(pb, cb[i].b, wacc[kb]);
        }
    }
    ctx.wave_sync();    // partner reads done before this wave reuses its rows as scratch
}

// where a run's next block comes from (all wave-uniform)
struct OlaCursor {
    long long id;          // first tile id of the launch order that no segment has claimed yet
    long long stream;
    int k, k_store, k_end; // block being transformed; blocks [k_store, k_end) are stored, the ones before only rebuild the carry
};

// Persistent form: the workgroup owns tile ids [first, end) of the launch order (stream-major, blocks of a stream consecutive).
//   CS: channels per frame, NP = ceil(CS / 2) pairs, H: hop = 512 H frames per block (H <= 8).
template <class Ctx, int CS, int NP, int H>
AW_HD void tiles_fused_ola(Ctx &ctx, const TileParams &p, long long first, long long end) {
    static_assert(NP == (CS + 1) / 2, "every pair of the layout in one pass");
    constexpr int CP = 2 * NP;
    constexpr int NC = 16 - H;               // carry values per thread
    constexpr int hop = 512 * H;
    if (first >= end) return;
    const int t0 = ctx.tid();
    int t = t0, lane = ctx.lane();
    const int wave = ctx.wave();
    cf *buf0 = ctx.lds();
    cf *buf1 = buf0 + kBufElems;
    cf *twa = buf1 + kBufElems;
    const cf w1 = p.tw1[t];
    twa[t] = hl_twiddle(p.twa, t);           // [16][32] row twiddles of the half-wave form; visible after the first barrier
    const int K = p.tiles_per_stream;
    const int warm = (p.hist_len + hop - 1) / hop;        // blocks whose tail reaches into a given block: ceil((taps - 1) / hop)

    OlaCursor cur;
    cur.id = first;
    auto begin_segment = [&](OlaCursor &c) {
        c.stream = c.id / K;
        const int tile0 = (int)(c.id - c.stream * K);
        const long long left = end - c.id;
        const int n = left < (long long)(K - tile0) ? (int)left : K - tile0;
        c.k_store = tile0;
        c.k = tile0 - warm;
        c.k_end = tile0 + n;
        c.id += n;
    };
    // a block's frames: input rows k hop + ..., or (k < 0) history rows hist_len + k hop + ...
    auto source = [&](const OlaCursor &c) {
        const bool from_hist = c.k < 0;                                                        // uniform: blocks never straddle frame 0
        const float *base = from_hist ? p.hist + c.stream * (long long)p.hist_len * CS : p.in + c.stream * p.frames * CS;
        const unsigned bytes = (unsigned)((from_hist ? (long long)p.hist_len : p.frames) * (CS * 4));
        return OlaNext<typename Ctx::Buf>{ctx.buf(base, bytes), c.k * hop + (from_hist ? p.hist_len : 0)};
    };
    auto load = [&](const OlaCursor &c, float (&raw)[H][CP]) {
        const OlaNext<typename Ctx::Buf> s = source(c);
        ola_load_block<CS, H>(ctx, s.src, s.idx0, t, raw);
    };
    begin_segment(cur);
    float raw[H][CP];
    load(cur, raw);
    cf carry[NC];
    bool fresh = true;                       // the block about to be transformed is the first of its segment: the carry starts from zero

    for (;;) {
        t = ctx.opaque_i(t0);                // keeps lane-dependent addresses from being hoisted out of the block loop
        lane = t & 63;
        if (fresh) {                         // uniform
#pragma unroll
            for (int i = 0; i < NC; ++i) carry[i] = mk(0.f, 0.f);
        }
        cf wacc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) wacc[i] = mk(0.f, 0.f);

        // the cursor of the block after this one (for the prefetch); `last`: nothing follows
        OlaCursor nx = cur;
        bool last = false, nx_fresh = false;
        nx.k = cur.k + 1;
        if (nx.k == cur.k_end) {
            if (cur.id >= end) { last = true; nx = cur; }
            else { begin_segment(nx); nx_fresh = true; }
        }

        // batches of two pairs through buf0 / buf1, as tiles_fused_ols: pass 1 of both, barrier, row transforms + CMAC of each.
        // (a compile-time loop: which batch is the last one, and whether it has a second pair, select template instantiations)
        const OlaNext<typename Ctx::Buf> nsrc = source(nx);
        ola_static_for<(NP + 1) / 2>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            constexpr int pair0 = 2 * b;
            constexpr bool two = pair0 + 1 < NP;
            constexpr bool last_batch = b == (NP + 1) / 2 - 1;
            constexpr bool kMid = AW_OLA_PREFETCH_MID != 0;
            if (b > 0) ctx.barrier();                    // every wave is done reading buf0 / buf1
            {
                cf pw[16];
                tw_powers(ctx.opaque(w1), pw);
                cf x[H];
#pragma unroll
                for (int j = 0; j < H; ++j) x[j] = mk(raw[j][2 * pair0], raw[j][2 * pair0 + 1]);
                ola_pass1<H>(x, pw, buf0, t);
                if constexpr (two) {
#pragma unroll
                    for (int j = 0; j < H; ++j) x[j] = mk(raw[j][2 * pair0 + 2], raw[j][2 * pair0 + 3]);
                    ola_pass1<H>(x, pw, buf1, t);
                }
            }
            ctx.barrier();
            if constexpr (!kMid && OlaEarly<NP>::value && last_batch) load(nx, raw);        // every channel of this block has been through pass 1
            constexpr int kTabG = AW_OLA_TABG ? AW_OLA_TABG : NP >= 7 ? 8 : 16;
            constexpr bool kTabEarly = (AW_OLA_TAB_EARLY & 1) != 0 && NP <= 4 && kTabG == 16;
            if constexpr (AW_OLA_PAIR2 != 0 && two && NP <= AW_OLA_PAIR2_MAXNP) {
                ola_subfft_cmac2<AW_OLA_PAIR2_G>(ctx, p, pair0, buf0, buf1, twa, lane, wave, wacc);
            } else {
                cf2 tab[16];
                if constexpr (kTabEarly) load_tab_h(p, pair0, wave, lane, tab);
                ola_subfft_cmac<kTabG, kTabEarly, kMid && last_batch && !two, CS, H>(ctx, p, pair0, buf0, twa, tab, lane, wave, wacc, nsrc, t, raw);
                if constexpr (two) {
                    if constexpr (kTabEarly) load_tab_h(p, pair0 + 1, wave, lane, tab);
                    ola_subfft_cmac<kTabG, kTabEarly, kMid && last_batch, CS, H>(ctx, p, pair0 + 1, buf1, twa, tab, lane, wave, wacc, nsrc, t, raw);
                }
            }
        });

        tile_inverse_rows_h(ctx, wacc, buf0, twa);
        if (!AW_OLA_PREFETCH_MID && !OlaEarly<NP>::value) load(nx, raw);       // few registers are live here; the last block re-reads its own frames (no branch around 100 registers)

        // radix-16 across rows: y[j] = block position t + 512 j; add the carry; j < H are frames, the rest is the new carry
        ctx.barrier();
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
        const bool store = cur.k >= cur.k_store;                                      // uniform
        float *out_s = p.out + cur.stream * p.frames * 2;
        const long long f0 = (long long)cur.k * hop;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j < NC) y[j] = y[j] + carry[j];
            if (j < H) {
                const long long f = f0 + t + 512 * j;
#ifdef AW_ABL_OLA_NOSTORE      // timing ablation only (wrong results): no output stores
                if (store && f < p.frames && y[j].x == 1.2345e-30f)
#else
                if (store && f < p.frames)
#endif
                    ctx.st_stream(reinterpret_cast<cf *>(out_s + f * 2), y[j]);
            }
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) carry[i] = y[i + H];
        ctx.barrier();                       // the final exchange has been read before buf0 is rewritten
        if (last) break;
        cur = nx;
        fresh = nx_fresh;
    }
}

}  // namespace awk
