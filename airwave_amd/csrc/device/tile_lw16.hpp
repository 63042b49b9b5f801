// tile_lw16.hpp — rows kernel of the long-window path (tile_lw.hpp, kernel 2) on a different transform core:
// 256-thread workgroups, 16 points of ONE row per thread, 4096 = 16 x 16 x 16.
//
// Same arithmetic as lw_rows_tiles (per row pair {ra, rb = R-1-ra}: Z1 = FFT_4096(S[ra]), V = FFT_4096(S[rb]) per channel pair,
// W1 += Z1 T0 + V T1, W2 += V T2 + Z1 T3, s1 = IFFT_4096(W1), s2 = IFFT_4096(W2); the reference semantics are still
// ConvolutionEngine.process, Airwave/ConvolutionEngine.swift:232-367, summed over speakers, RealtimeAudioProcessor.swift:141-172),
// different decomposition.  The 8-point core of tile_ols.hpp runs a 4096-point row as 8 x (8 x 8 x 8): one workgroup-level LDS
// exchange plus two wave-level ones per transform, 12 LDS operations per value, ten workgroup barriers of eight waves per tile, and
// the two rows of a pair side by side in every thread.  Here a row is three radix-16 passes:
//
//   row sample n = a + 16 b + 256 j          thread (a, b): a = lane & 15, b = (lane >> 4) + 4 wave, i.e. n = thread + 256 j: every global
//                                            access of a wave is a 512-byte run in lane order; register j
//   pass 1  radix-16 over j -> k2,           twiddle w_4096^{(a + 16 b) k2}    (powers of one per-thread base, formed in registers)
//   E1      LDS: register k2 <-> thread field b                                (16 stores + 16 loads per thread, two barriers of four waves)
//   pass 2  radix-16 over b -> m0,           twiddle w_256^{a m0}              (256-entry table in LDS)
//   E2      LDS, wave-private: register m0 <-> lane field a                    (16 + 16, no barrier: after E1 a wave reads only its own
//                                                                               quarter of the buffer, which it then reuses for E2)
//   pass 3  radix-16 over a -> m1:           bin k = kappa + 16 alpha + 256 m1 in thread (alpha = lane & 15, kappa = (lane >> 4) + 4 wave), register m1
//
// 4 LDS operations per value instead of 12, all bank-conflict free, and the same number of vector instructions as the 8-point core
// (E2 in registers — permlane swaps and bank-masked DPP moves, built and measured first — forces the field a onto lane bits 5..2,
// which scatters every global access over four lines per 16 lanes: 5.5 against 4.6 ms, and costs 150 of 890 vector instructions per
// transform in a core that is bound by vector issue).  One row at a time: a thread carries 16 values of the row in flight, 16 + 16
// accumulators (W1, W2) and the 16 prefetched values of the next row; the row-ra pass multiplies by {T0, T3}, the row-rb pass by
// {T1, T2} (tables stored that way, in thread order: a wave's table loads are 1 KB runs).  Three independent 4-wave workgroups per CU
// instead of two 8-wave ones.  The inverse is the mirror image (conjugate twiddles, passes in reverse), so the s1 / s2 rows leave
// in natural order as 512-byte runs per wave, exactly where lw_merge_tiles reads them.
#pragma once
#include "tile_lw.hpp"

namespace awk {

#ifndef AW_R16_PRIO
#define AW_R16_PRIO 3           // s_setprio: 1 = raised while the next row's loads are issued, 2 = during the table / multiply-accumulate phase (4.31 -> 4.16 ms), 4 / 8 = during the E1 / E2 exchanges
#endif
constexpr int kR16Threads = 256;
// E1 through LDS as [register of the contiguous side][thread], row stride 272: the contiguous side (forward stores, inverse loads) is a
// 512-byte run per wave; on the strided side lane (a, x = lane >> 4) touches 272 (x + 4 wave) + a + 16 bb — 16 consecutive slots per
// 16-lane group, and the two groups of a 32-lane load group 272 = 16 (mod 32) slots apart: no bank conflicts either way.
// E2 inside the wave's own four rows (1088 slots): element (a, m0) of lane group x at 272 x + 17 m0 + a, stored a-contiguous, loaded
// as 17 alpha + rho: 17 alpha (mod 32) takes 16 values whose complement is the set shifted by 16 — again conflict free.
constexpr int kR16Stride = 272;
constexpr int kR16BufElems = 16 * kR16Stride;
constexpr int kR16Tw2Elems = 256;
constexpr int kR16LdsElems = kR16BufElems + kR16Tw2Elems;
constexpr int kR16LdsBytes = kR16LdsElems * 8;          // 36 864 B: three workgroups per CU
// (Variants measured on tools/ubench/rows_bench in round 4 and removed from the source in round 5 — DESIGN.md §4.5b has the numbers: the
// pass-1 powers w^1 .. w^8 in LDS, uniform-base addressing, the next row's requests spread over the transform, table batches requested
// before the last pass, no row prefetch with four workgroups per CU, all sixteen table entries requested before the transform.)

struct alignas(16) LwTab2 { cf u, w; };

// FFT_4096 output bin held by (thread, register m1) of the 16-point core
AW_HD int r16_bin(int thread, int m1) { return (thread >> 4) + 16 * (thread & 15) + 256 * m1; }

// Per-thread constants of the core
struct R16Thread {
    cf w1;            // w_4096^{a + 16 b} = w_4096^{thread}
    cf *lin;          // E1, contiguous side: buf + thread                      (+ 272 k2)
    cf *str;          // E1, strided side:    buf + 272 b + a                   (+ 16 bb)
    cf *e2w;          // E2 store base: quarter + 272 (lane >> 4) + a           (+ 17 m0)
    cf *e2r;          // E2 load base:  quarter + 272 (lane >> 4) + 17 alpha    (+ rho)
    const cf *tw2;    // w_256^{a m0} at tw2[16 m0]
};

// E2: 16 x 16 transpose between the register index and the lane field a, through the wave's own quarter of the E1 buffer
template <class Ctx> AW_HD void r16_lane_transpose(Ctx &ctx, cf (&v)[16], const R16Thread &th) {
    ctx.wave_sync();                                 // this wave's E1 loads have returned (other waves never read this quarter)
#pragma unroll
    for (int m0 = 0; m0 < 16; ++m0) th.e2w[17 * m0] = v[m0];
    ctx.wave_sync();
#pragma unroll
    for (int rho = 0; rho < 16; ++rho) v[rho] = ctx.ld(th.e2r + rho);
    ctx.wave_sync();
}

// v[m0] *= w_256^{a m0} (INV: conjugate), the table in LDS at tw2[16 m0].  The entries are requested AW_R16_T2_BATCH at a time before
// their multiplies (left in one loop hipcc awaits them one by one: 1200 cycles for 15 reads and 60 multiplies).
#ifndef AW_R16_T2_BATCH
#define AW_R16_T2_BATCH 8
#endif
template <bool INV, class Ctx> AW_HD void r16_tw2_apply(Ctx &ctx, cf (&v)[16], const cf *tw2) {
    constexpr int B = AW_R16_T2_BATCH;
#pragma unroll
    for (int m = 1; m < 16; m += B) {
        cf w[B];
#pragma unroll
        for (int i = 0; i < B; ++i) if (m + i < 16) w[i] = ctx.ld(tw2 + 16 * (m + i));
        ctx.sched_fence();
#pragma unroll
        for (int i = 0; i < B; ++i) if (m + i < 16) v[m + i] = twmul<INV>(v[m + i], w[i]);
    }
}

// forward: v[j] = row[a + 16 b + 256 j]  ->  v[m1] = X[kappa + 16 alpha + 256 m1].  SB >= 0: phase stamps SB.. (diagnostic builds)
template <int SB = -1, class Ctx> AW_HD void r16_forward(Ctx &ctx, cf (&v)[16], const R16Thread &th) {
    auto stamp = [&](int i) {              // (the values pass through an opaque asm first: arithmetic does not float across the stamp)
        if constexpr (SB >= 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = ctx.opaque(v[k]);
            ctx.stamp(SB + i);
        } else (void)i;
    };
    fft16<false>(v);
    stamp(0);
    r16_pow_apply(v, ctx.opaque(th.w1));
    stamp(1);
    if constexpr (AW_R16_PRIO & 4) ctx.template prio<1>();
    ctx.barrier();                                   // every wave has read the previous transform's exchange
    stamp(2);
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) th.lin[kR16Stride * k2] = v[k2];
    stamp(3);
    ctx.barrier();
    stamp(4);
#pragma unroll
    for (int bb = 0; bb < 16; ++bb) v[bb] = ctx.ld(th.str + 16 * bb);
    if constexpr (AW_R16_PRIO & 4) ctx.template prio<0>();
    fft16<false>(v);
    stamp(5);
    r16_tw2_apply<false>(ctx, v, th.tw2);
    stamp(6);
    if constexpr (AW_R16_PRIO & 8) ctx.template prio<1>();
    r16_lane_transpose(ctx, v, th);
    if constexpr (AW_R16_PRIO & 8) ctx.template prio<0>();
    stamp(7);
    fft16<false>(v);
    stamp(8);
}

// inverse (unnormalised): v[m1] = W[kappa + 16 alpha + 256 m1]  ->  v[j] = w[a + 16 b + 256 j].  AFTER_INVERSE: the previous transform was
// an inverse too, whose E1 loads (the contiguous side) read every wave's quarter: a barrier before this wave reuses its own for E2.
template <bool AFTER_INVERSE, class Ctx> AW_HD void r16_inverse(Ctx &ctx, cf (&v)[16], const R16Thread &th) {
    fft16<true>(v);
    if constexpr (AFTER_INVERSE) ctx.barrier();
    r16_lane_transpose(ctx, v, th);
    r16_tw2_apply<true>(ctx, v, th.tw2);
    fft16<true>(v);
    ctx.barrier();
#pragma unroll
    for (int bb = 0; bb < 16; ++bb) th.str[16 * bb] = v[bb];
    ctx.barrier();
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) v[k2] = ctx.ld(th.lin + kR16Stride * k2);
    r16_pow_apply(v, conj(ctx.opaque(th.w1)));
    fft16<true>(v);
}

#ifndef AW_R16_TAB_K
#define AW_R16_TAB_K 2          // table entries per batch of the multiply-accumulate phase
#endif
#ifndef AW_R16_TAB_DEPTH
#define AW_R16_TAB_DEPTH 2      // batches in flight (1: issue, await, use)
#endif
#ifndef AW_R16_STAMP_TILE
#define AW_R16_STAMP_TILE 40    // diagnostic builds (-DAW_STAMPS=1): the tile of every workgroup whose phases are recorded
#endif

// Tiles as in lw_rows_tiles: virtual id -> (row pair, stream-window), row pairs pinned to XCDs by the launcher.
template <class Ctx, int NP, bool REAL_LAST>
AW_HD void lw_rows16_tiles(Ctx &ctx, const LwParams &p, long long first, long long step, long long n_sw, int rp0, int rp_step) {
    static_assert(NP >= 1 && NP <= 8, "channel pairs");
    constexpr int NROWS = 2 * NP - (REAL_LAST ? 1 : 0);           // forward row transforms per tile
    const LwRowMap rmap = lw_row_map(n_sw, p.R / 2, rp0, rp_step);
    const long long end = lw_row_count(rmap);
    if (first >= end) return;
    const int tid = ctx.tid(), lane = ctx.lane(), wave = ctx.wave();
    const int a = lane & 15, b = (lane >> 4) + 4 * wave, nth = tid;        // nth = a + 16 b
    cf *buf = ctx.lds();
    cf *tw2 = buf + kR16BufElems;
    tw2[tid] = p.tw2[tid];                                       // visible after the first transform's first barrier
    R16Thread th;
    th.w1 = p.tw1m[nth];
    th.lin = buf + tid;
    th.str = buf + kR16Stride * b + a;
    th.e2w = buf + kR16Stride * (4 * wave + (lane >> 4)) + a;
    th.e2r = buf + kR16Stride * (4 * wave + (lane >> 4)) + 17 * a;
    th.tw2 = tw2 + a;
    const int R = p.R;

    auto row_src = [&](const LwRowTile &tl, int idx) -> const cf * {     // idx = 2 pair + (0: row ra, 1: row rb)
        const int pair = idx >> 1;
        const int row = (idx & 1) ? R - 1 - tl.rp : tl.rp;
        return p.spec + tl.sw * p.spec_per_sw + (long long)pair * p.N + (long long)row * kLwM + nth;
    };
    cf raw[16];
    auto load_row = [&](const cf *src, cf (&d)[16]) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
#ifdef AW_LW_ABL_ROWS_NOLOAD      // timing ablation only (wrong results)
            d[j] = mk(0.001f * tid, (float)(src == nullptr) + 0.002f * j);
#else
            d[j] = ctx.ld_stream(src + 256 * j);
#endif
        }
    };
    load_row(row_src(lw_row_tile(rmap, first), 0), raw);
    for (long long vid = first; vid < end; vid += step) {
        const LwRowTile tl = lw_row_tile(rmap, vid);
#if defined(AW_STAMPS) && AW_STAMPS
        ctx.stamp_on_ = (vid - first) / step == AW_R16_STAMP_TILE;       // diagnostic builds: one mid-kernel tile
#endif
        cf w1acc[16], w2acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { w1acc[i] = mk(0.f, 0.f); w2acc[i] = mk(0.f, 0.f); }
        lw_unroll<NROWS>([&](auto I) {
            constexpr int idx = I.value, pair = idx >> 1, r = idx & 1;
            cf v[16];
            constexpr int SB = idx == (NROWS > 2 ? 2 : 0) ? 1 : -1;       // diagnostic builds stamp one row of the tile
            if constexpr (SB >= 0) ctx.stamp(0);
            const LwTab2 *tb = p.tab16 + ((((long long)tl.rp * NP + pair) * 2 + r) * 16) * kR16Threads + tid;
            auto tab_entry = [&](int m1) -> LwTab2 {
#ifdef AW_LW_ABL_ROWS_NOTAB       // timing ablation only (wrong results)
                return LwTab2{mk(1.f, 0.5f * lane), mk(0.25f * m1, 1.f * wave + (float)(tb == nullptr))};
#else
                return tb[m1 * kR16Threads];
#endif
            };
            // the next row (of this tile, or the first one of the next tile; the last tile re-reads its own): requested at the start of this
            // row's transform.  Vector-memory results return in issue order — a wait for the table entries also waits for every load issued
            // before them — so the tables are requested after the transform.
            const cf *next_src = idx + 1 < NROWS ? row_src(tl, idx + 1) : row_src(lw_row_tile(rmap, vid + step < end ? vid + step : vid), 0);
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = raw[j];
            ctx.sched_fence_hard();               // keeps hipcc from hoisting several rows' loads to the top of the unrolled tile
            if constexpr (AW_R16_PRIO & 1) ctx.template prio<3>();
            load_row(next_src, raw);
            if constexpr (AW_R16_PRIO & 1) ctx.template prio<0>();
            ctx.sched_fence_hard();
            // Table entries in batches of K, DEPTH batches in flight, fenced: left in one loop hipcc emits load, s_waitcnt vmcnt(0),
            // eight multiply-accumulates, sixteen times over — sixteen exposed L2 round trips per row (8.9 k of its 15.6 k cycles).
            constexpr int K = AW_R16_TAB_K, NBATCH = 16 / K, DEPTH = AW_R16_TAB_DEPTH;
            LwTab2 tq[DEPTH][K];
            auto issue = [&](int bi) {
#pragma unroll
                for (int i = 0; i < K; ++i) tq[bi % DEPTH][i] = tab_entry(bi * K + i);
            };
            r16_forward<SB>(ctx, v, th);
            ctx.sched_fence_hard();               // (table loads hoisted above the transform end up in scratch)
            if constexpr (SB >= 0) ctx.stamp(10);
            if constexpr (AW_R16_PRIO & 2) ctx.template prio<2>();
#pragma unroll
            for (int bi = 0; bi < DEPTH - 1 && bi < NBATCH; ++bi) issue(bi);
#pragma unroll
            for (int bi = 0; bi < NBATCH; ++bi) {
                if (bi + DEPTH - 1 < NBATCH) issue(bi + DEPTH - 1);
                ctx.sched_fence_hard();
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const int m1 = bi * K + i;
                    w1acc[m1] = cfma(v[m1], tq[bi % DEPTH][i].u, w1acc[m1]);
                    w2acc[m1] = cfma(v[m1], tq[bi % DEPTH][i].w, w2acc[m1]);
                }
                ctx.sched_fence_hard();
            }
            if constexpr (AW_R16_PRIO & 2) ctx.template prio<0>();
            // pins the multiply-accumulates here: left free, hipcc sinks every row's to the end of the tile and keeps the rows' spectra
            // and table values in scratch until then (2 KB per thread)
#pragma unroll
            for (int m1 = 0; m1 < 16; ++m1) { w1acc[m1] = ctx.opaque(w1acc[m1]); w2acc[m1] = ctx.opaque(w2acc[m1]); }
            if constexpr (SB >= 0) ctx.stamp(11);
        });
        ctx.stamp(12);
        cf *dst = p.wrows + (tl.sw * (R / 2) + tl.rp) * (long long)(2 * kLwM) + nth;
        r16_inverse<false>(ctx, w1acc, th);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
#ifdef AW_LW_ABL_ROWS_NOSTORE     // timing ablation only (wrong results)
            if (w1acc[j].x != 1.2345e-30f) continue;
#endif
            ctx.st_stream(dst + 256 * j, w1acc[j]);
        }
        r16_inverse<true>(ctx, w2acc, th);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
#ifdef AW_LW_ABL_ROWS_NOSTORE
            if (w2acc[j].x != 1.2345e-30f) continue;
#endif
            ctx.st_stream(dst + kLwM + 256 * j, w2acc[j]);
        }
        ctx.stamp(13);
    }
}

}  // namespace awk
