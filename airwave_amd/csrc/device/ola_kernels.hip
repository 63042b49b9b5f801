// ola_kernels.hip — launcher of the overlap-add tile kernels (tile_ola.hpp; instantiations in ola_kernels_a .. e.hip).
#include "ola_inst.hpp"
#include "tile_ola.hpp"

#include <cstdio>

namespace awk {

// rows H of the block a layout's HRIR length maps to (0: no kernel): the largest H <= 8 whose window still holds the tail
int fused_ola_rows(int n_channels, int taps) {
    if (taps < 2) return 0;
    int H = (kN - (taps - 1)) / 512;
    if (H > 8) H = 8;
    if (H < 6) return 0;
    switch (n_channels) {
        case 4: case 6: case 7: case 8: case 9: case 10: case 11: case 12: case 13: case 14: case 15: case 16: return H;
        default: return 0;
    }
}

const char *fused_ola_kernel_name(int C, int H) {
    static thread_local char name[64];
    std::snprintf(name, sizeof name, "aw_fused_ola_kernel<%d, %d, %d>", C, (C + 1) / 2, H);
    return name;
}

hipError_t prepare_ola_kernels() {
    hipError_t e = prepare_ola_a();
    if (e == hipSuccess) e = prepare_ola_b();
    if (e == hipSuccess) e = prepare_ola_c();
    if (e == hipSuccess) e = prepare_ola_d();
    if (e == hipSuccess) e = prepare_ola_e();
    if (e == hipSuccess) e = prepare_ola_f();
    return e;
}

// One launch for every block of every stream (history blocks, interior blocks, the ragged last block).  p.hop = 512 H,
// p.tiles_per_stream = ceil(frames / hop), p.hist_len = rows of the history buffer (>= taps - 1).
hipError_t launch_fused_ola(const TileParams &p, int H, int n_streams, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    const long long n_tiles = (long long)n_streams * p.tiles_per_stream;
    if (n_tiles <= 0) return hipSuccess;
    if (p.hop != 512 * H || p.hist_len > kN - p.hop) return hipErrorInvalidValue;
    // buffer descriptors address a stream's input (and its history rows) with 32-bit byte offsets
    if (p.frames * p.n_channels * 4LL > kOlaMaxBytes || (long long)p.hist_len * p.n_channels * 4LL > kOlaMaxBytes) return hipErrorInvalidValue;
    const long long wgs = p.persistent_wgs >= 1 ? p.persistent_wgs : 256;
    const dim3 grid((unsigned)(n_tiles < wgs ? n_tiles : wgs));
    if (ev0) (void)hipEventRecord(ev0, stream);
    const bool ok = launch_ola_a(p, H, grid, n_tiles, stream) || launch_ola_b(p, H, grid, n_tiles, stream) || launch_ola_c(p, H, grid, n_tiles, stream) ||
                    launch_ola_d(p, H, grid, n_tiles, stream) || launch_ola_e(p, H, grid, n_tiles, stream) || launch_ola_f(p, H, grid, n_tiles, stream);
    if (ev1) (void)hipEventRecord(ev1, stream);
    return ok ? hipGetLastError() : hipErrorInvalidValue;
}

}  // namespace awk
