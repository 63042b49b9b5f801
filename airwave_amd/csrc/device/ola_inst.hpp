// ola_inst.hpp — which overlap-add tile kernels (tile_ola.hpp) the library carries, and the translation units that build them.
// (channels, rows H of 512 frames per block): hop = 512 H, HRIRs of up to 8193 - 512 H taps.  H = 8 serves 3585 .. 4097 taps (the bundled
// HRIRs at 44.1 kHz: 3969) and everything shorter that the policy sends here, H = 7 4098 .. 4609 (the bundled HRIRs at 48 kHz: 4320),
// Layouts: 4 and 6 - 16 channels (mono, stereo, 3 and 5 channels run faster on their 16384-frame overlap-save tiles: 5 channels measured 0.80 - 0.99,
// profiles/round6_v5/ola_sweep_odd.txt).  H = 6 up to 5121 taps — beyond that the long-window kernels (or the 16384-frame tile) measure faster for every layout: blocks of 5 rows
// were built and measured (profiles/round6_v1/ola_sweep.txt: 0.88 - 1.02 of the best alternative) and are not carried.
#pragma once
#include "kernels.hpp"

#define AW_OLA_FOR_H(X, CS) X(CS, 6) X(CS, 7) X(CS, 8)
#define AW_OLA_LAYOUTS_A(X) AW_OLA_FOR_H(X, 4) AW_OLA_FOR_H(X, 6)
#define AW_OLA_LAYOUTS_B(X) AW_OLA_FOR_H(X, 7) AW_OLA_FOR_H(X, 8)
#define AW_OLA_LAYOUTS_C(X) AW_OLA_FOR_H(X, 10) AW_OLA_FOR_H(X, 12)
#define AW_OLA_LAYOUTS_D(X) AW_OLA_FOR_H(X, 14) AW_OLA_FOR_H(X, 13)
#define AW_OLA_LAYOUTS_E(X) AW_OLA_FOR_H(X, 16) AW_OLA_FOR_H(X, 15)
#define AW_OLA_LAYOUTS_F(X) AW_OLA_FOR_H(X, 9) AW_OLA_FOR_H(X, 11)

namespace awk {
// each returns false when the (channels, H) pair is not one of its unit's
bool launch_ola_a(const TileParams &p, int H, dim3 grid, long long n_tiles, hipStream_t stream);
bool launch_ola_b(const TileParams &p, int H, dim3 grid, long long n_tiles, hipStream_t stream);
bool launch_ola_c(const TileParams &p, int H, dim3 grid, long long n_tiles, hipStream_t stream);
bool launch_ola_d(const TileParams &p, int H, dim3 grid, long long n_tiles, hipStream_t stream);
bool launch_ola_e(const TileParams &p, int H, dim3 grid, long long n_tiles, hipStream_t stream);
bool launch_ola_f(const TileParams &p, int H, dim3 grid, long long n_tiles, hipStream_t stream);
hipError_t prepare_ola_a();
hipError_t prepare_ola_b();
hipError_t prepare_ola_c();
hipError_t prepare_ola_d();
hipError_t prepare_ola_e();
hipError_t prepare_ola_f();
}  // namespace awk
