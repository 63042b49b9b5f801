// emu_harness.cpp — TEST-ONLY thread emulation of the HIP tile kernel.
//
// Compiles airwave_amd/csrc/device/tile_ols.hpp (the exact code the GPU runs) with g++ and
// executes one workgroup as 512 std::threads: workgroup barriers are std::barrier(512),
// wave-level syncs are std::barrier(64).  It exists so index math, twiddles and LDS hazards can
// be checked on the CPU-only build container before spending GPU minutes.  It is NOT part of the
// product library and nothing under airwave_amd/ links it.
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "../../airwave_amd/csrc/device/tile_ols.hpp"
#include "../../airwave_amd/csrc/device/tile_march.hpp"
#include "../../airwave_amd/csrc/device/tile_lw.hpp"
#include "../../airwave_amd/csrc/device/tile_lw16.hpp"
#include "../../airwave_amd/csrc/device/eq_cascade.hpp"
#include "../../airwave_amd/csrc/host/eq.hpp"
#include "../../airwave_amd/csrc/host/tables.hpp"

namespace {

alignas(16) float g_zeros[1024] = {0};

struct EmuShared {
    std::barrier<> wg;
    std::vector<std::unique_ptr<std::barrier<>>> wave;
    std::vector<awk::cf> lds;
    std::vector<awk::cf> xs;      // cross-lane swap mailbox: [thread][2]
    explicit EmuShared(int threads = awk::kThreads, size_t lds_elems = (size_t)awk::kLdsElems) : wg(threads), lds(lds_elems), xs((size_t)threads * 2) {
        for (int w = 0; w < threads / 64; ++w) wave.emplace_back(new std::barrier<>(64));
    }
};

struct EmuCtx {
    int tid_;
    EmuShared *sh;
    int tid() const { return tid_; }
    int lane() const { return tid_ & 63; }
    int wave() const { return tid_ >> 6; }
    awk::cf *lds() const { return sh->lds.data(); }
    awk::cf opaque(awk::cf v) const { return v; }
    int opaque_i(int v) const { return v; }
    void stamp(int) const {}
    void sched_fence() const {}
    void sched_fence_hard() const {}
    void flush_stamps() const {}
    awk::cf ld(const awk::cf *p) const { return *p; }
    void stagger(int, int) const {}
    void barrier() const { sh->wg.arrive_and_wait(); }
    // lanes with bit b = 0: hi' = partner.lo ; lanes with bit b = 1: lo' = partner.hi ; partner = lane ^ (1 << b)
    void xswap(awk::cf &lo, awk::cf &hi, int bit) const {
        sh->xs[(size_t)tid_ * 2] = lo; sh->xs[(size_t)tid_ * 2 + 1] = hi;
        sh->wave[tid_ >> 6]->arrive_and_wait();
        const int p = tid_ ^ (1 << bit);
        if ((tid_ >> bit) & 1) lo = sh->xs[(size_t)p * 2 + 1]; else hi = sh->xs[(size_t)p * 2];
        sh->wave[tid_ >> 6]->arrive_and_wait();
    }
    void wave_sync() const { sh->wave[tid_ >> 6]->arrive_and_wait(); }
    template <int STRIDE> void ld8x2(awk::cf (&a)[8], const awk::cf *p0, awk::cf (&b)[8], const awk::cf *p1) const {
        for (int i = 0; i < 8; ++i) { a[i] = p0[i * STRIDE]; b[i] = p1[i * STRIDE]; }
    }
    void st_stream(awk::cf *q, awk::cf v) const { *q = v; }
    awk::cf ld_stream(const awk::cf *q) const { return *q; }
    void st_stream4(float *q, float a, float b, float c, float d) const { q[0] = a; q[1] = b; q[2] = c; q[3] = d; }
    awk::cf xchg1(awk::cf v) const {          // value of lane ^ 1
        sh->xs[(size_t)tid_ * 2] = v;
        sh->wave[tid_ >> 6]->arrive_and_wait();
        const awk::cf r = sh->xs[(size_t)(tid_ ^ 1) * 2];
        sh->wave[tid_ >> 6]->arrive_and_wait();
        return r;
    }
    // in-register scan moves of eq_cascade.hpp (DPP on the GPU): value of another lane of the wave, or zero / fill
    double lane_value(double v, int src_lane, double otherwise) const {     // src_lane < 0: no source
        double *box = reinterpret_cast<double *>(&sh->xs[(size_t)tid_ * 2]);
        *box = v;
        sh->wave[tid_ >> 6]->arrive_and_wait();
        const double r = src_lane < 0 ? otherwise : *reinterpret_cast<const double *>(&sh->xs[(size_t)((tid_ & ~63) + src_lane) * 2]);
        sh->wave[tid_ >> 6]->arrive_and_wait();
        return r;
    }
    void fma_in_place(double &x, double a, double c, double, double) const { x = __builtin_fma(a, x, c); }
    template <int D> double row_shr(double v) const { const int l = tid_ & 63; return lane_value(v, (l & 15) >= D ? l - D : -1, 0.0); }
    double row_bcast15(double v) const { const int l = tid_ & 63; return lane_value(v, ((l >> 4) & 1) ? (l & ~15) - 1 : -1, 0.0); }
    double row_bcast31(double v) const { const int l = tid_ & 63; return lane_value(v, l >= 32 ? 31 : -1, 0.0); }
    double wave_shr1(double v, double fill) const { const int l = tid_ & 63; return lane_value(v, l > 0 ? l - 1 : -1, fill); }
};

}  // namespace

extern "C" {

// Builds tables + twiddles on the host and runs every tile of every stream through the emulated
// workgroup.  Layouts as TileParams.  hist may be NULL (zeros).
// The 16384-frame window path (tile_ols2.hpp): hop and history follow from the taps; hist is [stream][hist][C].
static int emu_fused_ols2(const float *in, float *out, const float *hist, const float *tracks, int n_tracks, int taps,
                          int n_channels, const int32_t *left_track, const int32_t *right_track, long long frames,
                          int n_streams) {
    using namespace awk;
    awh::Twiddles tw;
    awh::build_twiddles(tw);
    std::vector<cf4> tab;
    awh::build_poly_tables(tracks, n_tracks, taps, n_channels, left_track, right_track, tab);
    tab.resize(tab.size() + kN, cf4{{mk(0, 0), mk(0, 0)}, {mk(0, 0), mk(0, 0)}});      // the zero pair
    TileParams p{};
    p.in = in; p.out = out; p.tab = reinterpret_cast<const cf2 *>(tab.data());
    p.tw1 = tw.tw1.data(); p.twa = tw.twa.data(); p.twb = tw.twb.data(); p.zeros = g_zeros;
    p.frames = frames; p.n_channels = n_channels; p.n_pairs = (2 * n_channels + 1) / 2;
    p.hist_len = awh::poly_history_frames(taps); p.hop = kN2 - p.hist_len;
    p.tiles_per_stream = (int)((frames + p.hop - 1) / p.hop);
    std::vector<float> zero_hist;
    if (!hist) { zero_hist.assign((size_t)n_streams * p.hist_len * n_channels + 4, 0.f); hist = zero_hist.data(); }
    p.hist = hist;
    const long long usable = frames - (((2 * n_channels) % 4 != 0) ? (n_channels == 1 ? 2 : 1) : 0);
    long long lo = (p.hist_len + p.hop - 1) / p.hop;
    long long hi = (usable - kN2 + p.hist_len) >= 0 ? (usable - kN2 + p.hist_len) / p.hop + 1 : 0;
    if (hi > p.tiles_per_stream) hi = p.tiles_per_stream;
    if (hi < lo) hi = lo;
    if (lo > p.tiles_per_stream) { lo = p.tiles_per_stream; hi = lo; }
    const bool vec = n_channels == 1 || n_channels == 2 || n_channels == 3 || n_channels == 5 || n_channels == 7 || n_channels == 8;
    if (!vec) { lo = 0; hi = 0; }
    p.tile_lo = (int)lo; p.tile_hi = (int)hi;
    EmuShared sh;
    auto run = [&](bool interior, long long n_tiles) {
        const long long G = n_tiles < 3 ? n_tiles : 3;
        for (long long g = 0; g < G; ++g) {
            std::vector<std::thread> th;
            th.reserve(kThreads);
            for (int t = 0; t < kThreads; ++t)
                th.emplace_back([&, t]() {
                    EmuCtx ctx{t, &sh};
                    if (interior) {
                        switch (n_channels) {
                            case 1: tiles_fused_ols2<EmuCtx, 1, 1, true>(ctx, p, g, G, n_tiles); break;
                            case 2: tiles_fused_ols2<EmuCtx, 2, 1, true>(ctx, p, g, G, n_tiles); break;
                            case 3: tiles_fused_ols2<EmuCtx, 3, 2, true>(ctx, p, g, G, n_tiles); break;
                            case 5: tiles_fused_ols2<EmuCtx, 5, 3, true>(ctx, p, g, G, n_tiles); break;
                            case 7: tiles_fused_ols2<EmuCtx, 7, 4, true>(ctx, p, g, G, n_tiles); break;
                            default: tiles_fused_ols2<EmuCtx, 8, 4, true>(ctx, p, g, G, n_tiles); break;
                        }
                    } else {
                        tiles_fused_ols2<EmuCtx, 0, 0, false>(ctx, p, g, G, n_tiles);
                    }
                });
            for (auto &x : th) x.join();
        }
    };
    run(true, (long long)n_streams * (hi - lo));
    run(false, (long long)n_streams * (p.tiles_per_stream - (hi - lo)));
    return 0;
}

int emu_fused_ols(const float *in, float *out, const float *hist, const float *tracks, int n_tracks,
                  int taps, int n_channels, const int32_t *left_track, const int32_t *right_track,
                  long long frames, int n_streams, int hop, int variant) {
    using namespace awk;
    if (variant == 2) return emu_fused_ols2(in, out, hist, tracks, n_tracks, taps, n_channels, left_track, right_track, frames, n_streams);
    if (hop <= 0 || hop > kN - (taps - 1)) return -1;
    awh::Twiddles tw;
    std::vector<cf2> tab;
    awh::build_twiddles(tw);
    awh::build_pair_tables(tracks, n_tracks, taps, n_channels, left_track, right_track, 0, taps, tab);
    tab.resize(tab.size() + kN, cf2{mk(0, 0), mk(0, 0)});            // the zero pair (as runtime.cpp appends it)
    std::vector<float> zero_hist;
    TileParams p{};
    p.in = in; p.out = out; p.tab = tab.data(); p.tw1 = tw.tw1.data(); p.twa = tw.twa.data(); p.twb = tw.twb.data(); p.zeros = g_zeros;
    p.frames = frames; p.n_channels = n_channels; p.n_pairs = (n_channels + 1) / 2;
    p.hop = hop; p.hist_len = kN - hop;
    p.tiles_per_stream = (int)((frames + hop - 1) / hop);
    if (!hist) {
        zero_hist.assign((size_t)n_streams * p.hist_len * n_channels, 0.f);
        hist = zero_hist.data();
    }
    p.hist = hist;
    // same interior/boundary split as awk::launch_fused_ols
    long long lo = (p.hist_len + p.hop - 1) / p.hop;
    const long long usable = p.frames - ((n_channels % 4 != 0 && n_channels != 2) ? 1 : 0);
    long long hi = (usable - kN + p.hist_len) >= 0 ? (usable - kN + p.hist_len) / p.hop + 1 : 0;
    if (hi > p.tiles_per_stream) hi = p.tiles_per_stream;
    if (hi < lo) hi = lo;
    if (lo > p.tiles_per_stream) { lo = p.tiles_per_stream; hi = lo; }
    const bool vec = (n_channels >= 2 && n_channels <= 8) ||
                     (n_channels >= 9 && n_channels <= 15 && (variant != 5 || !(n_channels & 1))) || n_channels == 16;
    if (!vec || (variant != 1 && variant != 5)) { lo = 0; hi = 0; }
    p.tile_lo = (int)lo; p.tile_hi = (int)hi;
    EmuShared sh;
    // emulate a persistent launch with a few workgroups, each walking several tiles
    auto run = [&](bool interior, long long n_tiles) {
        const long long G = n_tiles < 3 ? n_tiles : 3;
        for (long long g = 0; g < G; ++g) {
            std::vector<std::thread> th;
            th.reserve(kThreads);
            for (int t = 0; t < kThreads; ++t)
                th.emplace_back([&, t]() {
                    EmuCtx ctx{t, &sh};
                    // mirror of awk::launch_fused_ols' variant choice
                    if (interior) {
                        // wide layouts: first pass over 4 pairs, then an accumulating pass over the rest (input and tables shifted)
                        TileParams q = p;
                        q.in = p.in + 8;
                        q.tab = p.tab + 4 * (long long)kN;
                        switch (n_channels) {
                            case 2: tiles_fused_ols<EmuCtx, 2, 1, true>(ctx, p, g, G, n_tiles); break;
                            case 3: tiles_fused_ols<EmuCtx, 3, 2, true>(ctx, p, g, G, n_tiles); break;
                            case 4: tiles_fused_ols<EmuCtx, 4, 2, true>(ctx, p, g, G, n_tiles); break;
                            case 5: tiles_fused_ols<EmuCtx, 5, 3, true>(ctx, p, g, G, n_tiles); break;
                            case 6: tiles_fused_ols<EmuCtx, 6, 3, true>(ctx, p, g, G, n_tiles); break;
                            case 7: tiles_fused_ols<EmuCtx, 7, 4, true>(ctx, p, g, G, n_tiles); break;
                            case 8: tiles_fused_ols<EmuCtx, 8, 4, true>(ctx, p, g, G, n_tiles); break;
                            // variant 1 (the default): one pass over two eight-channel groups; variant 5: the two-pass form (AW_WIDE_TWO_PASS=1)
                            case 15: tiles_fused_ols<EmuCtx, 15, 4, true>(ctx, p, g, G, n_tiles); ctx.barrier(); tiles_fused_ols<EmuCtx, 15, 4, true, true>(ctx, q, g, G, n_tiles); break;
                            case 9: tiles_fused_ols<EmuCtx, 9, 5, true>(ctx, p, g, G, n_tiles); break;
                            case 11: tiles_fused_ols<EmuCtx, 11, 6, true>(ctx, p, g, G, n_tiles); break;
                            case 13: tiles_fused_ols<EmuCtx, 13, 7, true>(ctx, p, g, G, n_tiles); break;
                            case 10: if (variant != 5) { tiles_fused_ols<EmuCtx, 10, 5, true>(ctx, p, g, G, n_tiles); break; }
                                     tiles_fused_ols<EmuCtx, 10, 4, true>(ctx, p, g, G, n_tiles); ctx.barrier(); tiles_fused_ols<EmuCtx, 10, 1, true, true>(ctx, q, g, G, n_tiles); break;
                            case 14: if (variant != 5) { tiles_fused_ols<EmuCtx, 14, 7, true>(ctx, p, g, G, n_tiles); break; }
                                     tiles_fused_ols<EmuCtx, 14, 4, true>(ctx, p, g, G, n_tiles); ctx.barrier(); tiles_fused_ols<EmuCtx, 14, 3, true, true>(ctx, q, g, G, n_tiles); break;
                            case 12: if (variant != 5) { tiles_fused_ols<EmuCtx, 12, 6, true>(ctx, p, g, G, n_tiles); break; }
                                     tiles_fused_ols<EmuCtx, 12, 4, true>(ctx, p, g, G, n_tiles); ctx.barrier(); tiles_fused_ols<EmuCtx, 12, 2, true, true>(ctx, q, g, G, n_tiles); break;
                            default: tiles_fused_ols<EmuCtx, 16, 4, true>(ctx, p, g, G, n_tiles); ctx.barrier(); tiles_fused_ols<EmuCtx, 16, 4, true, true>(ctx, q, g, G, n_tiles); break;
                        }
                    } else if (p.n_pairs > 4 && p.n_pairs <= 8 && (variant == 1 || variant == 5)) {      // launch_gen's two passes for 9-16 channels
                        TileParams q = p;
                        q.in = p.in + 8; q.hist = p.hist + 8; q.tab = p.tab + 4 * (long long)kN; q.ch_base = 8;
                        tiles_fused_ols<EmuCtx, 0, 4, false>(ctx, p, g, G, n_tiles);
                        ctx.barrier();
                        switch (p.n_pairs - 4) {
                            case 1: tiles_fused_ols<EmuCtx, 0, 1, false, true>(ctx, q, g, G, n_tiles); break;
                            case 2: tiles_fused_ols<EmuCtx, 0, 2, false, true>(ctx, q, g, G, n_tiles); break;
                            case 3: tiles_fused_ols<EmuCtx, 0, 3, false, true>(ctx, q, g, G, n_tiles); break;
                            default: tiles_fused_ols<EmuCtx, 0, 4, false, true>(ctx, q, g, G, n_tiles); break;
                        }
                    } else {
                        switch (p.n_pairs <= 4 && (variant == 1 || variant == 5) ? p.n_pairs : 0) {
                            case 1: tiles_fused_ols<EmuCtx, 0, 1, false>(ctx, p, g, G, n_tiles); break;
                            case 2: tiles_fused_ols<EmuCtx, 0, 2, false>(ctx, p, g, G, n_tiles); break;
                            case 3: tiles_fused_ols<EmuCtx, 0, 3, false>(ctx, p, g, G, n_tiles); break;
                            case 4: tiles_fused_ols<EmuCtx, 0, 4, false>(ctx, p, g, G, n_tiles); break;
                            default: tiles_fused_ols<EmuCtx, 0, 0, false>(ctx, p, g, G, n_tiles); break;
                        }
                    }
                });
            for (auto &x : th) x.join();
        }
    };
    run(true, (long long)n_streams * (hi - lo));
    run(false, (long long)n_streams * (p.tiles_per_stream - (hi - lo)));
    return 0;
}

// Partitioned (long-HRIR) path: kernel 1 (window spectra) then kernel 2 (CMAC over partitions + inverse).
int emu_partitioned(const float *in, float *out, const float *hist, const float *tracks, int n_tracks, int taps,
                    int n_channels, const int32_t *left_track, const int32_t *right_track, long long frames,
                    int n_streams, int cmac_variant) {
    using namespace awk;
    const int B = kN / 2, P = (taps + B - 1) / B;
    awh::Twiddles tw;
    awh::build_twiddles(tw);
    std::vector<cf2> tab, all;
    for (int q = 0; q < P; ++q) {
        awh::build_pair_tables(tracks, n_tracks, taps, n_channels, left_track, right_track, q * B, B, tab);
        all.insert(all.end(), tab.begin(), tab.end());
    }
    TileParams p{};
    p.in = in; p.out = out; p.tab = all.data(); p.tw1 = tw.tw1.data(); p.twa = tw.twa.data(); p.twb = tw.twb.data(); p.zeros = g_zeros;
    p.frames = frames; p.n_channels = n_channels; p.n_pairs = (n_channels + 1) / 2;
    p.hop = B; p.hist_len = P * B; p.partitions = P; p.n_blocks = (int)((frames + B - 1) / B);
    p.tiles_per_stream = p.n_blocks; p.first_valid = kN - B;
    p.herm_last = (cmac_variant != 1 && (n_channels & 1)) ? 1 : 0;      // as runtime.cpp sets it
    const int n_windows = p.n_blocks + P - 1;
    // history with the 4 floats of slack the runtime allocates (whole-frame vector loads of 7-float frames)
    std::vector<float> hist_pad((size_t)n_streams * p.hist_len * n_channels + 4, 0.f);
    if (hist) std::memcpy(hist_pad.data(), hist, ((size_t)n_streams * p.hist_len * n_channels) * sizeof(float));
    p.hist = hist_pad.data();
    std::vector<cf> spec((size_t)n_streams * n_windows * p.n_pairs * kN, mk(NAN, NAN));      // a read of a bin that was never stored poisons the output
    p.spec = spec.data();
    EmuShared sh;
    auto run = [&](auto fn, int count) {
        for (int s = 0; s < n_streams; ++s)
            for (int i = 0; i < count; ++i) {
                std::vector<std::thread> th;
                th.reserve(kThreads);
                for (int t = 0; t < kThreads; ++t) th.emplace_back([&, t]() { EmuCtx ctx{t, &sh}; fn(ctx, s, i); });
                for (auto &x : th) x.join();
            }
    };
    {   // interior / head / generic split of awk::launch_part_forward (vector variants are emulated for 5, 7 and 8 channels here)
        const long long usable = frames - ((n_channels % 4 != 0 && n_channels != 2) ? 1 : 0);
        const long long d = usable - kN;
        long long lo = P, hi = (d >= 0 ? d / B : -((-d + B - 1) / B)) + P + 1;       // floor((usable - N) / B) + P + 1, as awk::launch_part_forward
        if (hi > n_windows) hi = n_windows;
        if (lo > n_windows) lo = n_windows;
        if (hi < lo) lo = hi;
        if (hi < 0) hi = 0;
        if (lo < 0) lo = 0;
        if (n_channels != 7 && n_channels != 8 && n_channels != 5) { lo = 0; hi = 0; }
        if (cmac_variant == 1)             // with the block-group CMAC: the all-pairs-per-workgroup forward kernel
        run([&](EmuCtx &ctx, int s, int w) {
            if (w >= lo && w < hi) {
                if (n_channels == 7) tile_part_forward<EmuCtx, 7, 1>(ctx, p, s, w);
                else if (n_channels == 5) tile_part_forward<EmuCtx, 5, 1>(ctx, p, s, w);
                else tile_part_forward<EmuCtx, 8, 1>(ctx, p, s, w);
            } else if (w < lo) {
                if (n_channels == 7) tile_part_forward<EmuCtx, 7, 2>(ctx, p, s, w);
                else if (n_channels == 5) tile_part_forward<EmuCtx, 5, 2>(ctx, p, s, w);
                else tile_part_forward<EmuCtx, 8, 2>(ctx, p, s, w);
            } else tile_part_forward<EmuCtx, 0, 0>(ctx, p, s, w);
        }, n_windows);
        else                               // default: one channel pair per workgroup
        for (int pair = 0; pair < p.n_pairs; ++pair)
        run([&](EmuCtx &ctx, int s, int w) {
            if (w >= lo && w < hi) {
                if (n_channels == 7) tile_part_forward1<EmuCtx, 7, 1>(ctx, p, s, w, pair);
                else if (n_channels == 5) tile_part_forward1<EmuCtx, 5, 1>(ctx, p, s, w, pair);
                else tile_part_forward1<EmuCtx, 8, 1>(ctx, p, s, w, pair);
            } else if (w < lo) {
                if (n_channels == 7) tile_part_forward1<EmuCtx, 7, 2>(ctx, p, s, w, pair);
                else if (n_channels == 5) tile_part_forward1<EmuCtx, 5, 2>(ctx, p, s, w, pair);
                else tile_part_forward1<EmuCtx, 8, 2>(ctx, p, s, w, pair);
            } else tile_part_forward1<EmuCtx, 0, 0>(ctx, p, s, w, pair);
        }, n_windows);
    }
    std::vector<cf> wspec((size_t)n_streams * p.n_blocks * kN, mk(0.f, 0.f));
    p.wspec = wspec.data();
    if (cmac_variant == 1) {
        for (int s = 0; s < n_streams; ++s)                   // block-group kernel: one "thread" per bin and block group
            for (int b0 = 0; b0 < p.n_blocks; b0 += kCmacBlocks)
                for (int i = 0; i < kN; ++i) part_cmac_bin(p, s, b0, i);
    } else {
        // marched kernel: one "thread" per (bin-pair slot, channel pair); the lane-group sum is the += below.
        // Passes of at most 8 partitions, PQ as awk::launch_part_march picks it.
        for (int s = 0; s < n_streams; ++s)
            for (int q0 = 0; q0 < P; q0 += 8)
                for (int j = 0; j < kMarchSlots; ++j)
                    for (int pl = 0; pl < p.n_pairs; ++pl) {
                        auto emit = [&](long long st, int b, const MarchBins &mb, cf ai, cf ap) {
                            cf *w = p.wspec + (st * p.n_blocks + b) * kN;
                            w[mb.i] = w[mb.i] + ai;
                            if (mb.pi != mb.i) w[mb.pi] = w[mb.pi] + ap;
                        };
                        if (P - q0 <= 4) march_thread<4, true>(p, s, s + 1, j, pl, q0, emit);
                        else march_thread<8, true>(p, s, s + 1, j, pl, q0, emit);
                    }
    }
    run([&](EmuCtx &ctx, int s, int b) { tile_part_inverse<EmuCtx>(ctx, p, s, b); }, p.n_blocks);
    return 0;
}

// The long-window path (tile_lw.hpp): split -> rows -> merge on windows of N = R x 4096 frames, R in {32, 64, 128}.
// hist: [stream][hist_len][C] or NULL; hist_len = N - hop must be >= taps - 1 (hop given by the caller).
int emu_longwin(const float *in, float *out, const float *hist, const float *tracks, int n_tracks, int taps, int n_channels,
                const int32_t *left_track, const int32_t *right_track, long long frames, int n_streams, int R, int hop, int rows_pb, float *hist_out) {
    using namespace awk;
    if (R != 32 && R != 64 && R != 128) return -1;
    const long long N = (long long)R * kLwM;
    if (hop <= 0 || N - hop < taps - 1 || n_channels < 1 || n_channels > 16) return -2;
    awh::Twiddles tw;
    awh::build_twiddles(tw);
    awh::LwTables lt;
    const bool form16 = rows_pb == 16;                   // the 16-points-per-thread rows kernel (tile_lw16.hpp)
    awh::build_lw_tables(tracks, n_tracks, taps, n_channels, left_track, right_track, R, lt, form16 ? 16 : 8);
    LwParams p{};
    p.rows_form = form16 ? 16 : 8; p.tab16 = lt.tab16.data(); p.tw2 = lt.tw2.data();
    p.in = in; p.out = out; p.zeros = g_zeros; p.frames = frames;
    p.n_channels = n_channels; p.n_pairs = (n_channels + 1) / 2; p.real_last = n_channels & 1;
    p.hop = hop; p.hist_len = (int)(N - hop); p.n_windows = (int)((frames + hop - 1) / hop);
    p.R = R; p.N = (int)N;
    std::vector<float> hist_pad((size_t)n_streams * p.hist_len * n_channels + 4, 0.f);
    if (hist) std::memcpy(hist_pad.data(), hist, ((size_t)n_streams * p.hist_len * n_channels) * sizeof(float));
    p.hist = hist_pad.data();
    p.hist_out = hist_out;
    p.spec_per_sw = (long long)(p.n_pairs - p.real_last) * N + (p.real_last ? N / 2 : 0);
    const long long n_sw = (long long)n_streams * p.n_windows;
    std::vector<cf> spec((size_t)(n_sw * p.spec_per_sw), mk(NAN, NAN)), wrows((size_t)(n_sw * N), mk(NAN, NAN));
    p.spec = spec.data(); p.wrows = wrows.data();
    p.tab = lt.tab.data(); p.tw_coarse = lt.coarse.data(); p.tw_fine = lt.fine.data(); p.tw_step = lt.step.data(); p.tw_r = lt.tw_r.data(); p.tw1m = lt.tw1m.data();
    p.twa = tw.twa.data(); p.twb = tw.twb.data();
    EmuShared sh;
    auto run = [&](auto fn) {            // one emulated persistent workgroup walks every tile
        std::vector<std::thread> th;
        th.reserve(kThreads);
        for (int t = 0; t < kThreads; ++t) th.emplace_back([&, t]() { EmuCtx ctx{t, &sh}; fn(ctx); });
        for (auto &x : th) x.join();
    };
    const long long n_st = n_sw * kLwChunks;
    std::vector<float> tail(32, 0.f);
    std::memcpy(tail.data(), in + ((size_t)n_streams * frames - 1) * n_channels, n_channels * sizeof(float));
    p.tail = tail.data(); p.n_streams = n_streams;
    auto split = [&](auto RA) {
        constexpr int ra = decltype(RA)::value;
        if (n_channels > 8) {              // one launch: both channel halves of a frame in one wave
            const long long n_stw = n_sw * kLwChunksW;
            run([&](EmuCtx &ctx) {
                switch (n_channels - 8) {
                    case 1: lw_split_wide_tiles<EmuCtx, ra, 1>(ctx, p, 0, 1, n_stw); break; case 2: lw_split_wide_tiles<EmuCtx, ra, 2>(ctx, p, 0, 1, n_stw); break;
                    case 3: lw_split_wide_tiles<EmuCtx, ra, 3>(ctx, p, 0, 1, n_stw); break; case 4: lw_split_wide_tiles<EmuCtx, ra, 4>(ctx, p, 0, 1, n_stw); break;
                    case 5: lw_split_wide_tiles<EmuCtx, ra, 5>(ctx, p, 0, 1, n_stw); break; case 6: lw_split_wide_tiles<EmuCtx, ra, 6>(ctx, p, 0, 1, n_stw); break;
                    case 7: lw_split_wide_tiles<EmuCtx, ra, 7>(ctx, p, 0, 1, n_stw); break; default: lw_split_wide_tiles<EmuCtx, ra, 8>(ctx, p, 0, 1, n_stw); break;
                }
            });
            return;
        }
        run([&](EmuCtx &ctx) {
            switch (n_channels) {
                case 1: lw_split_tiles<EmuCtx, ra, 1>(ctx, p, 0, 1, n_st); break; case 2: lw_split_tiles<EmuCtx, ra, 2>(ctx, p, 0, 1, n_st); break;
                case 3: lw_split_tiles<EmuCtx, ra, 3>(ctx, p, 0, 1, n_st); break; case 4: lw_split_tiles<EmuCtx, ra, 4>(ctx, p, 0, 1, n_st); break;
                case 5: lw_split_tiles<EmuCtx, ra, 5>(ctx, p, 0, 1, n_st); break; case 6: lw_split_tiles<EmuCtx, ra, 6>(ctx, p, 0, 1, n_st); break;
                case 7: lw_split_tiles<EmuCtx, ra, 7>(ctx, p, 0, 1, n_st); break; default: lw_split_tiles<EmuCtx, ra, 8>(ctx, p, 0, 1, n_st); break;
            }
        });
    };
    if (R == 32) split(LwIdx<4>{}); else if (R == 64) split(LwIdx<8>{}); else split(LwIdx<16>{});
    const long long n_rt = n_sw * (R / 2);
    auto rows = [&](auto PBB) {
        constexpr int pb = decltype(PBB)::value;
        run([&](EmuCtx &ctx) {
            auto go = [&](auto NPP, auto REAL) {
                lw_rows_tiles<EmuCtx, decltype(NPP)::value, (decltype(REAL)::value != 0), (decltype(NPP)::value > 4 ? 1 : pb)>(ctx, p, 0, 1, n_rt, n_sw, 0, 1);
            };
            const int np = p.n_pairs;
            auto go_np = [&](auto REAL) {
                switch (np) {
                    case 1: go(LwIdx<1>{}, REAL); break; case 2: go(LwIdx<2>{}, REAL); break; case 3: go(LwIdx<3>{}, REAL); break; case 4: go(LwIdx<4>{}, REAL); break;
                    case 5: go(LwIdx<5>{}, REAL); break; case 6: go(LwIdx<6>{}, REAL); break; case 7: go(LwIdx<7>{}, REAL); break; default: go(LwIdx<8>{}, REAL); break;
                }
            };
            if (p.real_last) go_np(LwIdx<1>{}); else go_np(LwIdx<0>{});
        });
    };
    if (form16) {
        EmuShared sh16(kR16Threads, (size_t)kR16LdsElems);
        std::vector<std::thread> th;
        th.reserve(kR16Threads);
        for (int t = 0; t < kR16Threads; ++t)
            th.emplace_back([&, t]() {
                EmuCtx ctx{t, &sh16};
                auto go = [&](auto NPP, auto REAL) {
                    lw_rows16_tiles<EmuCtx, decltype(NPP)::value, (decltype(REAL)::value != 0)>(ctx, p, 0, 1, n_rt, n_sw, 0, 1);
                };
                auto go_np = [&](auto REAL) {
                    switch (p.n_pairs) {
                        case 1: go(LwIdx<1>{}, REAL); break; case 2: go(LwIdx<2>{}, REAL); break; case 3: go(LwIdx<3>{}, REAL); break; case 4: go(LwIdx<4>{}, REAL); break;
                        case 5: go(LwIdx<5>{}, REAL); break; case 6: go(LwIdx<6>{}, REAL); break; case 7: go(LwIdx<7>{}, REAL); break; default: go(LwIdx<8>{}, REAL); break;
                    }
                };
                if (p.real_last) go_np(LwIdx<1>{}); else go_np(LwIdx<0>{});
            });
        for (auto &x : th) x.join();
    } else if (rows_pb == 1) rows(LwIdx<1>{}); else rows(LwIdx<2>{});
    run([&](EmuCtx &ctx) {
        if (R == 32) lw_merge_tiles<EmuCtx, 4>(ctx, p, 0, 1, n_st);
        else if (R == 64) lw_merge_tiles<EmuCtx, 8>(ctx, p, 0, 1, n_st);
        else lw_merge_tiles<EmuCtx, 16>(ctx, p, 0, 1, n_st);
    });
    return 0;
}

// Butterfly unit checks: n in {4, 8, 16}; data is [n] complex interleaved, in place.
int emu_fft_small(float *data, int n, int inverse) {
    using namespace awk;
    if (n == 4) {
        cf a = mk(data[0], data[1]), b = mk(data[2], data[3]), c = mk(data[4], data[5]), d = mk(data[6], data[7]);
        if (inverse) fft4<true>(a, b, c, d); else fft4<false>(a, b, c, d);
        cf r[4] = {a, b, c, d};
        std::memcpy(data, r, sizeof(r));
    } else if (n == 8) {
        cf v[8]; std::memcpy(v, data, sizeof(v));
        if (inverse) fft8<true>(v); else fft8<false>(v);
        std::memcpy(data, v, sizeof(v));
    } else if (n == 16) {
        cf v[16]; std::memcpy(v, data, sizeof(v));
        if (inverse) fft16<true>(v); else fft16<false>(v);
        std::memcpy(data, v, sizeof(v));
    } else return -1;
    return 0;
}

// The host-built EQ tables of a filter set (awh::eq_prepare), for the table-precision test: tab [K][kEqTabDoubles],
// plane [K][64][4].  Returns K (or a negative prepare error); *tab_doubles = kEqTabDoubles, *chunk = kEqChunk.
int emu_eq_tables(double sample_rate, double preamp_db, const double *filters, int n_filters, double *tab, double *plane,
                  int *tab_doubles, int *chunk) {
    awh::EqDefinition def;
    def.preamp_db = preamp_db;
    for (int i = 0; i < n_filters; ++i) {
        awh::EqFilter f;
        f.type = (int)filters[i * 4]; f.frequency_hz = filters[i * 4 + 1]; f.gain_db = filters[i * 4 + 2]; f.q = filters[i * 4 + 3];
        def.filters.push_back(f);
    }
    awh::EqPrepared prep;
    int bi = 0, bk = 0;
    const int rc = awh::eq_prepare(&def, sample_rate, prep, &bi, &bk);
    if (rc) return -rc;
    *tab_doubles = awk::kEqTabDoubles; *chunk = awk::kEqChunk;
    if (tab) std::memcpy(tab, prep.tab.data(), prep.tab.size() * sizeof(double));
    if (plane) std::memcpy(plane, prep.plane.data(), prep.plane.size() * sizeof(double));
    return prep.n_filters;
}


// Parametric EQ cascade: one emulated workgroup (kEqThreads) per stream for the chunk-aligned part,
// eq_sequential for the tail — the same split runtime.cpp makes.  filters: [n][4] = type, fc, gain, q.
// z: [stream][K][4] state, carried in and out.  Returns K or a negative prepare error.
int emu_eq_process(const float *in, float *out, double *z, int n_streams, long long frames, double sample_rate,
                   double preamp_db, const double *filters, int n_filters, int ear_split) {
    using namespace awk;
    awh::EqDefinition def;
    def.preamp_db = preamp_db;
    for (int i = 0; i < n_filters; ++i) {
        awh::EqFilter f;
        f.type = (int)filters[i * 4]; f.frequency_hz = filters[i * 4 + 1]; f.gain_db = filters[i * 4 + 2]; f.q = filters[i * 4 + 3];
        def.filters.push_back(f);
    }
    awh::EqPrepared prep;
    int bi = 0, bk = 0;
    const int rc = awh::eq_prepare(&def, sample_rate, prep, &bi, &bk);
    if (rc) return -rc;
    EqParams p{};
    p.in = in; p.out = out; p.z = z;
    p.t.tab = prep.tab.data(); p.t.plane = prep.plane.data();
    p.t.preamp = prep.preamp; p.t.n_filters = prep.n_filters;
    p.stride_frames = frames;
    const long long body = frames - frames % kEqChunk;
    if (body > 0) {
        p.frames = body;
        EmuShared sh(kEqThreads, (size_t)kEqLdsBytes / sizeof(cf));
        for (int s = 0; s < n_streams; ++s) {
            std::vector<std::thread> th;
            th.reserve(kEqThreads);
            for (int ear = 0; ear < (ear_split ? 2 : 1); ++ear) {        // E = 1: one emulated workgroup per (stream, ear)
                th.clear();
                for (int t = 0; t < kEqThreads; ++t)
                    th.emplace_back([&, t]() {
                        EmuCtx ctx{t, &sh};
                        if (ear_split) eq_cascade_stream<EmuCtx, 1>(ctx, p, s, ear);
                        else eq_cascade_stream<EmuCtx, 2>(ctx, p, s, 0);
                    });
                for (auto &x : th) x.join();
            }
        }
    }
    if (frames > body) {
        p.in = in + body * 2; p.out = out + body * 2; p.frames = frames - body;
        for (int s = 0; s < n_streams; ++s)
            for (int ear = 0; ear < 2; ++ear) eq_sequential(p, s, ear);
    }
    return prep.n_filters;
}

}  // extern "C"
