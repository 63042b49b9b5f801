// tables.hpp — host-side (double precision) preparation of the frequency-domain filter tables
// and twiddles consumed by the tile kernels.  This is the MI355X analogue of the per-engine
// HRIR partition FFTs in ConvolutionEngine.init (Airwave/ConvolutionEngine.swift:141-175) and of
// FFTSetupManager's twiddle cache (Airwave/FFTSetupManager.swift:41-60): computed once per
// spatializer, shared by every stream.
#pragma once
#include <cstdint>
#include <vector>

#include "../device/tile_ols2.hpp"

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
void build_pair_tables(const float *tracks, int n_tracks, int taps, int n_channels,
                       const int32_t *left_track, const int32_t *right_track, int tap_offset,
                       int tap_count, std::vector<awk::cf2> &out);

// Two-output (polyphase) tables of the 16384-frame window path (device/tile_ols2.hpp): the 2C pseudo-channels
// (parity * C + channel) against the half-rate polyphase components of every HRIR,
//   even outputs: even-frame channel -> h[2j],   odd-frame channel -> h[2j-1] (j >= 1)
//   odd  outputs: even-frame channel -> h[2j+1], odd-frame channel -> h[2j]
// laid out [pseudo-pair][k1][k2] of {A_e, B_e, A_o, B_o}.  Half-rate filter length = taps / 2 + 1.
void build_poly_tables(const float *tracks, int n_tracks, int taps, int n_channels, const int32_t *left_track,
                       const int32_t *right_track, std::vector<awk::cf4> &out);
inline int poly_history_frames(int taps) { return 2 * (taps / 2); }          // real frames kept between calls (= N2 - hop)

}  // namespace awh
