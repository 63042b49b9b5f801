// emu_lw.cpp — TEST-ONLY thread emulation of the long-window kernels (tile_lw.hpp, tile_lw16.hpp); compiled as four
// translation units (-DEMU_LW_PART=0..3) side by side with emu_harness.cpp (tests/emu/emu.py).
#include "emu_ctx.hpp"

#ifndef EMU_LW_PART
#define EMU_LW_PART 0
#endif

// The split and merge kernels of one window length R = 8 RA, every channel count: instantiated per RA in separate compilations of this
// file (-DEMU_LW_PART=1..3; part 0 holds the driver), so that the 176 instantiations build side by side.
template <int RA> void emu_lw_split_or_merge(const awk::LwParams &p, long long n_sw, bool merge);

#define EMU_LW_RAS_1(X) X(4) X(5) X(6) X(7)
#define EMU_LW_RAS_2(X) X(8) X(9) X(10) X(12)
#define EMU_LW_RAS_3(X) X(14) X(15) X(16)

#if EMU_LW_PART != 0
template <int RA> void emu_lw_split_or_merge(const awk::LwParams &p, long long n_sw, bool merge) {
    using namespace awk;
    EmuShared sh;
    auto run = [&](auto fn) {            // one emulated persistent workgroup walks every tile
        std::vector<std::thread> th;
        th.reserve(kThreads);
        for (int t = 0; t < kThreads; ++t) th.emplace_back([&, t]() { EmuCtx ctx{t, &sh}; fn(ctx); });
        for (auto &x : th) x.join();
    };
    const long long n_st = n_sw * kLwChunks;
    const int n_channels = p.n_channels;
    constexpr int ra = RA;
    if (merge) { run([&](EmuCtx &ctx) { lw_merge_tiles<EmuCtx, ra>(ctx, p, 0, 1, n_st); }); return; }
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
}
#define EMU_LW_INST(RA) template void emu_lw_split_or_merge<RA>(const awk::LwParams &, long long, bool);
#if EMU_LW_PART == 1
EMU_LW_RAS_1(EMU_LW_INST)
#elif EMU_LW_PART == 2
EMU_LW_RAS_2(EMU_LW_INST)
#else
EMU_LW_RAS_3(EMU_LW_INST)
#endif
#else
#define EMU_LW_EXT(RA) extern template void emu_lw_split_or_merge<RA>(const awk::LwParams &, long long, bool);
EMU_LW_RAS_1(EMU_LW_EXT) EMU_LW_RAS_2(EMU_LW_EXT) EMU_LW_RAS_3(EMU_LW_EXT)

extern "C" {

// The long-window path (tile_lw.hpp): split -> rows -> merge on windows of N = R x 4096 frames, R in {32, 64, 128}.
// hist: [stream][hist_len][C] or NULL; hist_len = N - hop must be >= taps - 1 (hop given by the caller).
int emu_longwin(const float *in, float *out, const float *hist, const float *tracks, int n_tracks, int taps, int n_channels,
                const int32_t *left_track, const int32_t *right_track, long long frames, int n_streams, int R, int hop, int rows_pb, float *hist_out) {
    using namespace awk;
    if (R % 8 != 0 || !lw_ra_ok(R / 8)) return -1;
    const long long N = (long long)R * kLwM;
    if (hop <= 0 || N - hop < taps - 1 || n_channels < 1 || n_channels > 16) return -2;
    awh::Twiddles tw;
    awh::build_twiddles(tw);
    awh::LwTables lt;
    const bool form16 = rows_pb == 16;                   // the 16-points-per-thread rows kernel (tile_lw16.hpp)
    awh::build_lw_tables(tracks, n_tracks, taps, n_channels, left_track, right_track, R, lt, form16 ? 16 : 8);
    LwParams p{};
    p.rows_form = form16 ? 16 : 8; p.tab16 = lt.tab16.data(); p.tw2 = lt.tw2.data();
    p.in = in; p.out = out; p.zeros = g_zeros; p.frames = frames; p.frame0 = 0; p.frame_end = frames;
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
    auto with_ra = [&](auto &&fn) {
        switch (R / 8) {
            case 4: fn(LwIdx<4>{}); break; case 5: fn(LwIdx<5>{}); break; case 6: fn(LwIdx<6>{}); break; case 7: fn(LwIdx<7>{}); break; case 8: fn(LwIdx<8>{}); break;
            case 9: fn(LwIdx<9>{}); break; case 10: fn(LwIdx<10>{}); break; case 12: fn(LwIdx<12>{}); break;
            case 14: fn(LwIdx<14>{}); break; case 15: fn(LwIdx<15>{}); break; default: fn(LwIdx<16>{}); break;
        }
    };
    with_ra([&](auto RA) { emu_lw_split_or_merge<decltype(RA)::value>(p, n_sw, false); });
    auto rows = [&](auto PBB) {
        constexpr int pb = decltype(PBB)::value;
        run([&](EmuCtx &ctx) {
            auto go = [&](auto NPP, auto REAL) {
                lw_rows_tiles<EmuCtx, decltype(NPP)::value, (decltype(REAL)::value != 0), (decltype(NPP)::value > 4 ? 1 : pb)>(ctx, p, 0, 1, n_sw, 0, 1);
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
                    lw_rows16_tiles<EmuCtx, decltype(NPP)::value, (decltype(REAL)::value != 0)>(ctx, p, 0, 1, n_sw, 0, 1);
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
    with_ra([&](auto RA) { emu_lw_split_or_merge<decltype(RA)::value>(p, n_sw, true); });
    (void)n_st;
    return 0;
}

// In-register DFTs of the long-window split / merge kernels (tile_lw.hpp lw_fft, lw_odd_dft): n = 2^a q; odd != 0: the odd-frequency form.
int emu_lw_dft(float *data, int n, int inverse, int odd) {
    using namespace awk;
    auto go = [&](auto NN) {
        constexpr int nn = decltype(NN)::value;
        cf v[nn]; std::memcpy(v, data, sizeof(v));
        if (odd) { if (inverse) lw_odd_dft<true, nn>(v); else lw_odd_dft<false, nn>(v); }
        else { if (inverse) lw_fft<true, nn>(v); else lw_fft<false, nn>(v); }
        std::memcpy(data, v, sizeof(v));
    };
    switch (n) {
        case 2: go(LwIdx<2>{}); break; case 3: go(LwIdx<3>{}); break; case 4: go(LwIdx<4>{}); break; case 5: go(LwIdx<5>{}); break; case 6: go(LwIdx<6>{}); break;
        case 7: go(LwIdx<7>{}); break; case 8: go(LwIdx<8>{}); break; case 9: go(LwIdx<9>{}); break; case 10: go(LwIdx<10>{}); break; case 11: go(LwIdx<11>{}); break;
        case 12: go(LwIdx<12>{}); break; case 13: go(LwIdx<13>{}); break; case 14: go(LwIdx<14>{}); break; case 15: go(LwIdx<15>{}); break; case 16: go(LwIdx<16>{}); break;
        default: return -1;
    }
    return 0;
}


// The rows kernels' tile map over `groups` XCD groups (lw_row_map / lw_row_tile / lw_row_count, tile_lw.hpp): writes how often every
// (row pair, stream-window) is visited to hits[n_rp][n_sw] and every group's tile count to per_group[groups]; returns the number of tiles.
long long emu_lw_row_map(int n_rp, long long n_sw, int groups, int32_t *hits, long long *per_group) {
    using namespace awk;
    long long total = 0;
    for (int g = 0; g < groups; ++g) {
        const LwRowMap m = lw_row_map(n_sw, n_rp, g, groups);
        const long long n = lw_row_count(m);
        per_group[g] = n;
        int last_rp = -1, slices = 0;
        for (long long v = 0; v < n; ++v) {
            const LwRowTile t = lw_row_tile(m, v);
            if (t.rp < 0 || t.rp >= n_rp || t.sw < 0 || t.sw >= n_sw) return -1;
            if (t.rp != last_rp) { ++slices; last_rp = t.rp; }
            hits[(long long)t.rp * n_sw + t.sw] += 1;
        }
        if (slices > n_rp / groups + 2) return -2;          // a group walks one table slice at a time: its whole rounds + at most two left-over pairs
        total += n;
    }
    return total;
}

}  // extern "C"
#endif      // EMU_LW_PART == 0
