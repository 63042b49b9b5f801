// emu_harness.cpp — TEST-ONLY thread emulation of the HIP tile kernels (fused, partitioned, EQ); the long-window kernels: emu_lw.cpp.
#include "emu_ctx.hpp"

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

// The overlap-add tile (tile_ola.hpp): blocks of hop = 512 H frames, the carry in registers, one launch for every block of every
// stream.  `n_workgroups` contiguous runs over the stream-major block list (runs that start mid-stream rebuild their carry).
int emu_fused_ola(const float *in, float *out, const float *hist, const float *tracks, int n_tracks, int taps, int n_channels,
                  const int32_t *left_track, const int32_t *right_track, long long frames, int n_streams, int hist_len, int H, int n_workgroups) {
    using namespace awk;
    const int hop = 512 * H;
    if (H < 1 || H > 8 || taps - 1 > kN - hop || hist_len < taps - 1) return -1;
    awh::Twiddles tw;
    std::vector<cf2> tab;
    awh::build_twiddles(tw);
    awh::build_pair_tables(tracks, n_tracks, taps, n_channels, left_track, right_track, 0, taps, tab);
    std::vector<float> zero_hist;
    TileParams p{};
    p.in = in; p.out = out; p.tab = tab.data(); p.tw1 = tw.tw1.data(); p.twa = tw.twa.data(); p.twb = tw.twb.data(); p.zeros = g_zeros;
    p.frames = frames; p.n_channels = n_channels; p.n_pairs = (n_channels + 1) / 2;
    p.hop = hop; p.hist_len = hist_len;
    p.tiles_per_stream = (int)((frames + hop - 1) / hop);
    if (!hist) { zero_hist.assign((size_t)n_streams * hist_len * n_channels, 0.f); hist = zero_hist.data(); }
    p.hist = hist;
    const long long n_tiles = (long long)n_streams * p.tiles_per_stream;
    const long long G = n_workgroups < 1 ? 1 : (n_workgroups > n_tiles ? n_tiles : n_workgroups);
    EmuShared sh;
    int rc = 0;
    for (long long g = 0; g < G; ++g) {
        const long long first = n_tiles * g / G, end = n_tiles * (g + 1) / G;
        std::vector<std::thread> th;
        th.reserve(kThreads);
        for (int t = 0; t < kThreads; ++t)
            th.emplace_back([&, t]() {
                EmuCtx ctx{t, &sh};
#define AW_OLA_CASE(CS, HH) if (n_channels == CS && H == HH) { tiles_fused_ola<EmuCtx, CS, (CS + 1) / 2, HH>(ctx, p, first, end); return; }
                AW_OLA_CASE(2, 7) AW_OLA_CASE(2, 8) AW_OLA_CASE(7, 7) AW_OLA_CASE(7, 8) AW_OLA_CASE(8, 7) AW_OLA_CASE(8, 8) AW_OLA_CASE(8, 5)
                AW_OLA_CASE(14, 7) AW_OLA_CASE(14, 8) AW_OLA_CASE(14, 6) AW_OLA_CASE(16, 7) AW_OLA_CASE(5, 4) AW_OLA_CASE(12, 7) AW_OLA_CASE(6, 7) AW_OLA_CASE(13, 7) AW_OLA_CASE(9, 8)
#undef AW_OLA_CASE
                if (t == 0) rc = -2;
            });
        for (auto &x : th) x.join();
    }
    return rc;
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

// One wave's pair of 512-point row transforms in the half-wave form (tile_ols.hpp, sub_fft512h_fwd / _inv): rows = [2][512] complex
// (interleaved floats); fwd = the spectra in NATURAL bin order [2][512]; back = inverse of those spectra (unnormalised: 512 x rows).
int emu_sub_fft512h(const float *rows, float *fwd, float *back) {
    using namespace awk;
    awh::Twiddles tw;
    awh::build_twiddles(tw);
    EmuShared sh;
    cf *buf = sh.lds.data();
    cf *twh = buf + 2 * kBufElems;
    for (int t = 0; t < kThreads; ++t) twh[t] = hl_twiddle(tw.twa.data(), t);
    const int wave = 3;                                    // rows 3 and 13 of the exchange buffer
    for (int s = 0; s < 2; ++s)
        for (int k = 0; k < kSub; ++k) buf[wave_row(wave, s) * kRowStride + k] = mk(rows[(s * kSub + k) * 2], rows[(s * kSub + k) * 2 + 1]);
    std::vector<std::thread> th;
    for (int lane = 0; lane < 64; ++lane)
        th.emplace_back([&, lane] {
            EmuCtx ctx{wave * 64 + lane, &sh};
            const HLane L = hl_make(ctx, buf, twh, lane, wave);
            cf z[16];
            for (int j = 0; j < 16; ++j) z[j] = ctx.ld(L.row + L.h + 32 * j);
            sub_fft512h_fwd(ctx, z, L);
            const int s = lane >> 5;
            for (int kb = 0; kb < 16; ++kb) {
                const int k = L.col + 32 * kb;
                fwd[(s * kSub + k) * 2] = z[kb].x; fwd[(s * kSub + k) * 2 + 1] = z[kb].y;
            }
            sub_fft512h_inv(ctx, z, L);
            for (int j = 0; j < 16; ++j) {
                const int n = L.h + 32 * j;
                back[(s * kSub + n) * 2] = z[j].x; back[(s * kSub + n) * 2 + 1] = z[j].y;
            }
        });
    for (auto &t : th) t.join();
    return 0;
}

}  // extern "C"
