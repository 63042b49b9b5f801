// runtime.hpp — internal C++ objects behind the opaque C handles of include/airwave_hip.h.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/airwave_hip.h"
#include "device/kernels.hpp"

namespace awr {

void set_error(const std::string &msg);
aw_status fail(aw_status code, const std::string &msg);
aw_status fail_eq_filter(int enabled_index, int kind, int source_line, const std::string &msg);      // AW_ERR_EQ_INVALID_FILTER + aw_last_eq_filter_error
aw_status hip_fail(hipError_t e, const char *what);
bool context_literal_resampler(const aw_context *ctx);
// The C ABI promises "no exceptions": every entry point that can allocate host memory (containers, strings, threads) is a
// function-try-block whose handler returns caught() — std::bad_alloc / std::length_error become AW_ERR_OUT_OF_MEMORY, anything else
// AW_ERR_INVALID_ARGUMENT with the exception's text in aw_last_error_message().  Handles under construction are held by Owner<> until
// the entry hands them out, so an exception on the way leaks nothing.
aw_status caught() noexcept;
#define AW_NOEXCEPT_TAIL catch (...) { return ::awr::caught(); }

#define AW_HIP_TRY(expr)                                          \
    do {                                                          \
        hipError_t _e = (expr);                                   \
        if (_e != hipSuccess) return ::awr::hip_fail(_e, #expr);  \
    } while (0)

template <class T> using Owner = std::unique_ptr<T, void (*)(T *)>;

}  // namespace awr

struct aw_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    awk::cf *d_tw1 = nullptr, *d_twa = nullptr, *d_twb = nullptr;   // twiddle rows (FFTSetupManager analogue)
    float *d_zeros = nullptr;                                       // page of zeros (frames past the end of a call)
    bool literal_resampler = false;                                 // aw_context_set_resampler: aw_preset_activate resamples HRIRs with the literal vgenp call
    awk::LaunchCfg cfg;                                             // device properties + tuning knobs, read once at creation
    // Scratch pool (round 5): the grow-only HBM scratch of the partitioned and long-window kernels belongs to the CONTEXT and is
    // shared by every spatializer created on it — their calls are ordered on the context's stream, so one buffer of the largest
    // need serves them all, and a second spatializer (a preset change: HRIRManager.activatePreset builds a new renderer network
    // while the old one still plays, HRIRManager.swift:347-446) pays no second multi-GB hipMalloc.  launch_mu keeps one call's
    // launch sequence contiguous on the stream when two owners drive two handles of one context from two threads.
    awk::cf *d_pool = nullptr;
    size_t pool_capacity = 0;                                       // complex elements
    std::mutex launch_mu;
    std::atomic<long long> device_allocs{0};                        // hipMalloc / hipHostMalloc calls made on behalf of this context's handles (tests: a reserved process path makes none)
    std::atomic<long long> sync_copies{0};                          // blocking hipMemcpy calls likewise (table uploads)
    // host-entry pipeline (aw_spatializer_process_host on a multi-stream batch): H2D of chunk k+1 || kernels of chunk k || D2H of chunk k-1
    struct CopyPool;                                                // a few host threads that copy slices of one buffer at a time (runtime.cpp)
    CopyPool *copy_pool_out = nullptr;                              // a second, smaller pool for the output direction (copy OUT of chunk k-1 beside copy IN of chunk k+1)
    CopyPool *copy_pool = nullptr;                                  // made with the pipeline objects; pageable caller buffers are bounced through page-locked chunks by it
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_run[2] = {nullptr, nullptr}, ev_d2h[2] = {nullptr, nullptr};
};

struct aw_hrir {
    aw_context *ctx = nullptr;
    int n_tracks = 0, taps = 0;
    double sample_rate = 0.0;
    std::vector<float> tracks;   // [n_tracks][taps], host
};

struct aw_spatializer {
    aw_context *ctx = nullptr;
    int n_channels = 0, n_pairs = 0, n_streams = 0, taps = 0;
    int path = 0;             // 0 fused single-partition overlap-save, 1 partitioned
    bool fused2 = false;      // path 0 on 16384-frame windows (polyphase, two output spectra): device/tile_ols2.hpp
    int ola_h = 0;            // path 0 on 8192-frame windows: calls with enough blocks run the overlap-add tile with blocks of 512 ola_h frames (device/tile_ola.hpp); 0: never
    int last_ola_h = 0;       // ola_h of the last call if it ran that tile, else 0
    int hop = 0, hist_len = 0, partitions = 1;
    awk::cf2 *d_tab = nullptr;          // [partitions][pairs][N]
    float *d_hist[2] = {nullptr, nullptr};
    int hist_cur = 0;
    // scratch of the partitioned / long-window kernels: the context's pool (aw_context::d_pool), grow-only; reserved_pool = what
    // aw_spatializer_reserve() asked of it for this spatializer (elements)
    size_t reserved_pool = 0;
    size_t scratch_budget = 0;          // bytes per stream chunk (AW_SPEC_SCRATCH_MB at create; 0 = from free memory at first use)
    bool cmac_group = false;            // partitioned path: block-group CMAC kernel instead of the marched one (> 8 pairs; AW_PART_CMAC=group)
    bool fwd_one_pair = false;          // partitioned path: forward kernel with one channel pair per workgroup (default for more than 4 pairs; AW_PART_FWD=1|2 forces either form)
    bool herm_ok = true;                // partitioned path, odd channel count: store/read only the non-redundant half of the last pair's spectrum
    // long-window path (device/tile_lw.hpp): chosen per call (lw_choose) for long calls of a path-1 spatializer and, past a measured HRIR length,
    // of a path-0 one; tables per window length, built on first use (aw_spatializer_reserve builds those of the plan its max_frames implies)
    struct LwPlan { int R = 0; awk::LwTab *d_tab = nullptr; awk::LwTab2 *d_tab16 = nullptr; awk::cf *d_tw2 = nullptr; awk::cf *d_coarse = nullptr, *d_fine = nullptr, *d_step = nullptr, *d_tw_r = nullptr, *d_tw1m = nullptr; };
    std::vector<LwPlan> lw_plans;
    int lw_mode = -1;                   // AW_LW at create: -1 automatic (cost model), 0 never, 32/64/128 force that R where it fits
    std::vector<float> lw_tracks;       // the HRIR and channel map, kept for the lazily built tables
    std::vector<int32_t> lw_left, lw_right;
    int lw_n_tracks = 0;
    float *d_lw_tracks = nullptr;       // device copies of the same for the prep kernels (device/prep_kernels.hip), made with the first table set
    int32_t *d_lw_left = nullptr, *d_lw_right = nullptr;
    float *d_tail = nullptr;            // 32 floats: the last frame of the last stream of a call + zeros (wide split kernel, tile_lw.hpp)
    int last_lw_R = 0;                  // R of the last call's (first group of) windows (0: the partitioned kernels ran)
    int last_lw_R2 = 0;                 // R of its remainder window when the call ran as two groups
    int64_t reserved_frames = 0;        // aw_spatializer_reserve(): buffers are sized for calls up to this many frames
    // host-entry staging (grow-only; a multi-stream batch is staged in two chunks of streams each way: aw_spatializer_process_host)
    float *d_stage_in = nullptr, *d_stage_out = nullptr;
    size_t stage_in_cap = 0, stage_out_cap = 0;   // floats
    // one-stream, callback-sized calls (aw_spatializer_process_planar, aw_engine_*, aw_realtime_*): page-locked staging that the kernels read
    // and write DIRECTLY over PCIe — no copy engine, no (de)interleave kernels: the caller's samples are (de)interleaved by the CPU on the way
    float *h_pin_in = nullptr, *h_pin_out = nullptr;
    size_t pin_in_cap = 0, pin_out_cap = 0;       // floats
    // multi-stream host entry on PAGEABLE caller buffers: page-locked bounce chunks (two each way), filled / drained by the context's copy threads
    float *h_bounce_in = nullptr, *h_bounce_out = nullptr;
    size_t bounce_in_cap = 0, bounce_out_cap = 0; // floats (both slots)
    int64_t host_chunk_streams = 0;               // streams per staged chunk of the last host call (0: the whole batch in one piece, serial)
    int64_t host_chunk_reserved = 0, host_reserved_frames = 0;   // aw_spatializer_reserve_host: the chunking its buffers were sized for, and up to which call length
    // what the last aw_spatializer_reserve spent where (microseconds): float64 table build on host threads, table upload (hipMalloc +
    // hipMemcpy), scratch pool growth (hipMalloc) — bench.py's config.activation
    int64_t reserve_tables_us = 0, reserve_upload_us = 0, reserve_scratch_us = 0;
    unsigned long long *d_dbg = nullptr;   // AW_STAMPS diagnostic builds only
    size_t dbg_cap = 0;                     // words
    long long dbg_nwg = 0;
    // profiling of the dominant kernel
    bool profiling = false;
    hipEvent_t k0 = nullptr, k1 = nullptr;
    double kernel_ms_sum = 0.0;
    int kernel_launches = 0;
    long long dominant_frames = 0;         // output frames produced by the timed (dominant) launch of the last call
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;   // recorded, not yet read
    // per-launch stage timing (profiling only): every kernel of the step, by name
    struct StageRec { const char *name; hipEvent_t e0, e1; };
    struct StageStat { const char *name; double ms_sum; int launches; };
    std::vector<StageRec> stage_pending;
    std::vector<StageStat> stage_stats;
    std::vector<hipEvent_t> event_pool;
};

struct aw_engine {
    aw_context *ctx = nullptr;
    aw_hrir *hrir = nullptr;
    aw_spatializer *sp = nullptr;
    int block_size = 0;
    std::vector<float> tmp_out;   // [block][2]
};

struct aw_realtime {
    aw_context *ctx = nullptr;
    aw_spatializer *sp = nullptr;     // 1 stream, 2 input channels (L, R) -> first min(n,2) renderers
    int block_size = 0, max_frames = 0, fifo_capacity = 0, n_renderers = 0;
    std::vector<float> pending;       // [block][2] interleaved L,R
    std::vector<float> ready_in;      // completed blocks of this callback, interleaved
    std::vector<float> ready_out;
    std::vector<float> fifo_left, fifo_right;
    int pending_count = 0, fifo_read_index = 0, fifo_count = 0;
};
