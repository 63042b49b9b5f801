// tables.hpp — host-side (double precision) preparation of the frequency-domain filter tables
// and twiddles consumed by the tile kernels.  This is the MI355X analogue of the per-engine
// HRIR partition FFTs in ConvolutionEngine.init (Airwave/ConvolutionEngine.swift:141-175) and of
// FFTSetupManager's twiddle cache (Airwave/FFTSetupManager.swift:41-60): computed once per
// spatializer, shared by every stream.
#pragma once
#include <cstdint>
#include <functional>
#include <vector>

#include "../device/tile_ols2.hpp"
#include "../device/tile_lw16.hpp"

namespace awh {

// Twiddle tables (see TileParams): tw1[512] = W_N^t, twa[8][64] (ka-major), twb[8][8] (kb-major).
struct Twiddles {
    std::vector<awk::cf> tw1, twa, twb;
};
void build_twiddles(Twiddles &tw);

// Tables for taps [tap_offset, tap_offset + tap_count) of every (pair, ear) filter, laid out
// [pair][k1][k2] with k = k1 + 16 k2 (see tile_ols.hpp).  Channels whose track index is < 0
// contribute a zero filter (skipped speakers, HRIRManager.swift:370-372).  `scale` is folded
// into the tables (1/N for the inverse transform).
// pair_threads: one host thread per channel pair (false: the caller already runs several calls side by side).  false = out of host memory.
bool build_pair_tables(const float *tracks, int n_tracks, int taps, int n_channels,
                       const int32_t *left_track, const int32_t *right_track, int tap_offset,
                       int tap_count, std::vector<awk::cf2> &out, bool pair_threads = false);
// fn(0) .. fn(n - 1) on up to n host threads; never throws; false = some fn threw (bad_alloc)
bool parallel_for(int n, const std::function<void(int)> &fn);

// Two-output (polyphase) tables of the 16384-frame window path (device/tile_ols2.hpp): the 2C pseudo-channels
// (parity * C + channel) against the half-rate polyphase components of every HRIR,
//   even outputs: even-frame channel -> h[2j],   odd-frame channel -> h[2j-1] (j >= 1)
//   odd  outputs: even-frame channel -> h[2j+1], odd-frame channel -> h[2j]
// laid out [pseudo-pair][k1][k2] of {A_e, B_e, A_o, B_o}.  Half-rate filter length = taps / 2 + 1.
bool build_poly_tables(const float *tracks, int n_tracks, int taps, int n_channels, const int32_t *left_track,
                       const int32_t *right_track, std::vector<awk::cf4> &out);
inline int poly_history_frames(int taps) { return 2 * (taps / 2); }          // real frames kept between calls (= N2 - hop)

// Long-window path (device/tile_lw.hpp): tables and twiddles of one window length N = R x 4096, R in {32, 64, 128}.
//   tab    [R/2][pairs][8][512] {T0, T1, T2, T3}: the odd-frequency spectra (k + 1/2) of the pair filters at bin
//          k = ra + R k2 (k2 = q1 + 8 q2 stored at [q1][q2]) and at its partner k' = N-1-k, scale 1/(2N) folded in; a real last
//          channel (odd channel count) has T1 folded into T0 and T2 into T3
//   row twiddles w_2N^{t (2 k1 + 1)}, t = 64 tc + lane, k1 = RA m + ka (RA = R/8) as coarse[ka][tc] fine[ka][lane] step[m][t]:
//   coarse [RA][64]: w_2N^{64 tc (2 ka + 1)};  fine [RA][64]: w_2N^{lane (2 ka + 1)};  step [3][4096]: w_2N^{2 RA m t}, m = 1..3;
//   tw_r [RA][8]: w_2R^{j1 (2 ka + 1)};  tw1m [512]: w_4096^t
//   rows_form 16 (device/tile_lw16.hpp) instead of `tab`:
//   tab16  [R/2][pairs][2][16][256] {u, w}: the same four values per bin, split by the row they multiply — [0] = {T0, T3} (row ra),
//          [1] = {T1, T2} (row rb; all zero for a folded real last channel) — at [m1][thread] for the bin k2 = r16_bin(thread, m1)
//   tw2    [16][16]: w_256^{a m0} at [m0][a]
struct LwTables {
    std::vector<awk::LwTab> tab;
    std::vector<awk::LwTab2> tab16;
    std::vector<awk::cf> coarse, fine, step, tw_r, tw1m, tw2;
};
// filters = false: only the small twiddle tables (the filter tables then come from the device: device/prep_kernels.hip)
bool build_lw_tables(const float *tracks, int n_tracks, int taps, int n_channels, const int32_t *left_track,
                     const int32_t *right_track, int R, LwTables &out, int rows_form = 8, bool filters = true);

}  // namespace awh
