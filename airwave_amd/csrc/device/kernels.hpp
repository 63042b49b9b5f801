// kernels.hpp — host-callable launchers for the HIP kernels (defined in kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "tile_ols2.hpp"
#include "tile_ola.hpp"
#include "tile_march.hpp"
#include "tile_lw.hpp"
#include "tile_lw16.hpp"

namespace awk {

// Optional per-launch timing (profiling runs only): begin() right before a kernel launch, end(name) right after it.
struct StageTimer {
    virtual void begin() = 0;
    virtual void end(const char *name) = 0;
    virtual ~StageTimer() = default;
};

// Fused overlap-save spatializer: one workgroup per (stream, tile).  Returns hipSuccess or the
// launch error.  `n_streams * p.tiles_per_stream` workgroups of kThreads.
// ev0/ev1 (optional) are recorded around the dominant launch only: the interior tiles when the
// channel count has a vectorised variant, else the single generic launch; *dominant_tiles = its tile count.
hipError_t launch_fused_ols(const TileParams &p, int n_streams, hipStream_t stream, hipEvent_t ev0 = nullptr,
                            hipEvent_t ev1 = nullptr, long long *dominant_tiles = nullptr);
const char *fused_ols_kernel_name(int n_channels);
// 16384-frame windows (tile_ols2.hpp); p.hop / p.hist_len in real frames, p.tab = cf4 tables, p.n_pairs = pseudo-pairs.
hipError_t launch_fused_ols2(const TileParams &p, int n_streams, hipStream_t stream, hipEvent_t ev0 = nullptr,
                             hipEvent_t ev1 = nullptr, long long *dominant_tiles = nullptr);
const char *fused_ols2_kernel_name(int n_channels);

// Overlap-add form of the fused tile (tile_ola.hpp, ola_kernels*.hip): blocks of 512 H frames, the carry in registers, ONE launch for the whole
// call.  fused_ola_rows(): H for a layout and HRIR length, 0 when the library carries no such kernel.
int fused_ola_rows(int n_channels, int taps);
hipError_t prepare_ola_kernels();
hipError_t launch_fused_ola(const TileParams &p, int H, int n_streams, hipStream_t stream, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
const char *fused_ola_kernel_name(int n_channels, int H);

// Partitioned (long-HRIR) path: window spectra -> scratch; per-bin CMAC over partitions for groups of
// consecutive blocks -> W scratch; inverse transform of every block's W.
hipError_t launch_part_forward(const TileParams &p, int n_streams, hipStream_t stream, StageTimer *tm = nullptr);
hipError_t launch_part_cmac(const TileParams &p, int n_streams, hipStream_t stream, StageTimer *tm = nullptr);      // block-group kernel (A/B: AW_PART_CMAC=group)
hipError_t launch_part_march(const TileParams &p, int n_streams, hipStream_t stream, StageTimer *tm = nullptr);     // marched kernel (tile_march.hpp), the default
hipError_t launch_part_inverse(const TileParams &p, int n_streams, hipStream_t stream, StageTimer *tm = nullptr);

// Long-window path (tile_lw.hpp, lw_kernels.hip): windows of R x 4096 frames through split -> rows -> merge.
hipError_t prepare_lw_kernels();
hipError_t launch_lw_split(const LwParams &p, int n_streams, hipStream_t stream, StageTimer *tm = nullptr);
hipError_t launch_lw_rows(const LwParams &p, int n_streams, hipStream_t stream, StageTimer *tm = nullptr);
hipError_t launch_lw_merge(const LwParams &p, int n_streams, hipStream_t stream, StageTimer *tm = nullptr);

// hist_new[s][i][c] <- frame (frames - hist_len + i) of (hist_old ++ in), for every stream.
hipError_t launch_hist_update(const float *in, const float *hist_old, float *hist_new, long long frames,
                              int n_channels, int hist_len, int n_streams, hipStream_t stream);

// dst[s][i] = U(-0.5,0.5) counter RNG (oracle/airwave_oracle.h: orc_synth_value)
hipError_t launch_synth_fill(float *dst, int n_streams, long long per_stream, unsigned long long seed,
                             unsigned long long first_stream, hipStream_t stream);

// planar L/R (host layout staged on device) <-> interleaved helpers for the plugin-shaped entry
hipError_t launch_interleave2(const float *left, const float *right, float *dst, int frames, hipStream_t stream);
hipError_t launch_deinterleave2(const float *src, float *left, float *right, int frames, hipStream_t stream);

// HRIR prep on the device (prep_kernels.hip): the long-window filter tables of one window length R x 4096 in the 16-point rows kernel's
// layout, float64 on the GPU.  d_scratch: lw_prep_scratch_bytes() of temporary device memory; everything is queued on `stream`.
size_t lw_prep_scratch_bytes(int n_channels, int R);
hipError_t prepare_prep_kernels();
hipError_t launch_lw_prep(const float *d_tracks, int n_tracks, int taps, int n_channels, const int *d_left, const int *d_right, int R,
                          void *d_scratch, LwTab2 *d_tab16, hipStream_t stream);

// Measured-ceiling probe (probe_kernels.hip; aw_context_bandwidth_probe): what 0 read / 1 write / 2 copy over `bytes` of src / dst
hipError_t launch_bw_probe(int what, bool nt, const void *src, void *dst, size_t bytes, int cus, float *sink, hipStream_t stream, size_t *bytes_moved);

// Launch configuration of one context: device properties and tuning knobs, read ONCE at aw_context_create (never on a
// process path, never process-global: contexts on different devices keep their own).
struct LaunchCfg {
    int cus = 256;                // compute units of the context's device
    int persistent_wgs = 256;     // grid of the persistent tile kernels (AW_PERSISTENT_WGS; >= 8: the kernels deal tiles to 8 XCD groups)
    int wide_two_pass = 2;        // 10/12/14 channels: 2 = one pass over two eight-channel groups (16: two passes), 1 = two compile-time passes (AW_WIDE_TWO_PASS=1), 0 = the run-time-loop kernels (AW_WIDE_TWO_PASS=0)
    int debug_occupancy = 0;      // AW_DEBUG_OCCUPANCY
    int stamp_thread = 0;         // AW_STAMP_THREAD (diagnostic builds)
    int eq_ear_split = -1;        // AW_EQ_EAR_SPLIT: -1 automatic, 0 / 1 forced
    int lw_rows_pb = 1;           // long-window rows kernel, 8-point forms: channel pairs per batch (AW_LW_ROWS_PB; 1 = one exchange buffer, two workgroups per CU)
    int lw_rows_form = 16;        // long-window rows kernel: 16 = 256-thread workgroups, 16 points of one row per thread (tile_lw16.hpp); 8 = the 8-point forms (AW_LW_ROWS_FORM)
    int lw_rows16_wgs = 3;        // workgroups per CU of the 16-point rows kernel's persistent grid (AW_LW_ROWS16_WGS)
    int hop_align = 64;           // fused windows start on multiples of this many frames (AW_HOP_ALIGN; 1 = off)
    int lw_tables_on_gpu = 1;     // long-window filter tables built by the prep kernels (prep_kernels.hip); 0 = the float64 host builder (AW_LW_TABLES=host)
    int ola_min_blocks_per_wg = -1; // overlap-add tile: calls with fewer blocks per persistent workgroup run the overlap-save tile (AW_OLA_MIN_BLOCKS; -1 = per layout, runtime.cpp ola_min_blocks(); 0 = always the overlap-add tile)
    int host_out_async = 1;       // host entry on pageable buffers: the copy OUT of a chunk runs on output copy threads beside the next copy IN (AW_HOST_OUT_ASYNC=0: on the driver thread, round 5's form)
    int host_chunk_mb = 96;       // host entry of a multi-stream batch: input megabytes per staged chunk of streams (AW_HOST_CHUNK_MB; a few ms of PCIe Gen5 = the pipeline's fill / drain)
};
hipError_t prepare_kernels(LaunchCfg *cfg);   // fills cfg from the current device + environment; sets the dynamic-LDS attribute on every tile kernel

}  // namespace awk
