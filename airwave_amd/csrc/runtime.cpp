// runtime.cpp — device-side objects of the C ABI: context, HRIR set, batch spatializer, the mono
// engine trio and the callback-size adapter.  Compiled by hipcc as host C++.
#include "runtime.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <new>
#include <stdexcept>
#include <thread>

#include "device/eq_kernels.hpp"
#include "host/tables.hpp"

// Default fused window: 8192 frames; 16384 (tile_ols2.hpp) where measurements favour it.
#ifndef AW_DEFAULT_WINDOW
#define AW_DEFAULT_WINDOW 8192
#endif

// table sets a spatializer can hold: one per long-window length (kLwRowChoices below); lw_plans reserves this many at create
static constexpr size_t kLwPlanSlots = 16;

// Host threads that copy slices of ONE buffer at a time: the multi-stream host entry bounces pageable caller memory through page-locked
// chunks with them (one thread copies ~10 GB/s, PCIe Gen5 moves 57 GB/s each way).  copy() blocks; the caller copies a slice too.
struct aw_context::CopyPool {
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    char *dst = nullptr; const char *src = nullptr;
    size_t bytes = 0, slice = 0;
    int n_slices = 0, next = 0, done = 0;
    unsigned long long gen = 0;
    bool stop = false;
    explicit CopyPool(int n) {
        for (int i = 0; i < n; ++i) {
            try { workers.emplace_back([this] { run(); }); } catch (...) { break; }          // fewer threads: the caller's own slice loop still finishes the job
        }
    }
    ~CopyPool() {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv_work.notify_all();
        for (auto &w : workers) if (w.joinable()) w.join();
    }
    bool take(int &i) { if (next >= n_slices) return false; i = next++; return true; }       // under m
    void slice_copy(int i) {
        const size_t off = (size_t)i * slice, n = std::min(slice, bytes - off);
        std::memcpy(dst + off, src + off, n);
    }
    void run() {
        unsigned long long seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv_work.wait(lk, [&] { return stop || (gen != seen && next < n_slices); });
            if (stop) return;
            seen = gen;
            int i;
            while (take(i)) {
                lk.unlock();
                slice_copy(i);
                lk.lock();
                if (++done == n_slices) cv_done.notify_all();
            }
        }
    }
    // start() hands a buffer to the workers and returns; wait() blocks until it is copied (a pool with no worker copies in wait()).  One
    // job at a time per pool: the host entry uses a second, smaller pool for the output direction so that the copy OUT of chunk k-1 runs
    // beside the copy IN of chunk k+1 instead of after it on the driver thread (round 6).
    void start(void *d, const void *s_, size_t n) {
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return done == n_slices; });
        if (n == 0) return;
        dst = static_cast<char *>(d); src = static_cast<const char *>(s_); bytes = n;
        const int parts = (int)std::min<size_t>(std::max<size_t>(workers.size(), 1), std::max<size_t>(1, n >> 20));
        slice = ((n + parts - 1) / parts + 4095) & ~(size_t)4095;
        n_slices = (int)((n + slice - 1) / slice); next = 0; done = 0; ++gen;
        cv_work.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        int i;
        while (workers.empty() && take(i)) {          // no thread could be started: the waiter does the work
            lk.unlock();
            slice_copy(i);
            lk.lock();
            ++done;
        }
        cv_done.wait(lk, [&] { return done == n_slices; });
    }
    void copy(void *d, const void *s_, size_t n) {
        if (n == 0) return;
        std::unique_lock<std::mutex> lk(m);
        dst = static_cast<char *>(d); src = static_cast<const char *>(s_); bytes = n;
        const int parts = (int)std::min<size_t>(workers.size() + 1, std::max<size_t>(1, n >> 20));      // slices of >= 1 MiB
        slice = ((n + parts - 1) / parts + 4095) & ~(size_t)4095;
        n_slices = (int)((n + slice - 1) / slice); next = 0; done = 0; ++gen;
        cv_work.notify_all();
        int i;
        while (take(i)) {
            lk.unlock();
            slice_copy(i);
            lk.lock();
            ++done;
        }
        cv_done.wait(lk, [&] { return done == n_slices; });
    }
};

namespace awr {

static thread_local std::string g_last_error;
// the structured part of AW_ERR_EQ_INVALID_FILTER (aw_last_eq_filter_error): valid until the thread's next failing call
static thread_local struct { bool valid; int index, kind, line; } g_eq_filter_error = {false, 0, 0, 0};

void set_error(const std::string &msg) { g_last_error = msg; }
aw_status fail(aw_status code, const std::string &msg) {
    g_last_error = msg;
    g_eq_filter_error.valid = false;
    return code;
}
aw_status fail_eq_filter(int enabled_index, int kind, int source_line, const std::string &msg) {
    g_last_error = msg;
    g_eq_filter_error = {true, enabled_index, kind, source_line};
    return AW_ERR_EQ_INVALID_FILTER;
}
aw_status hip_fail(hipError_t e, const char *what) {
    g_eq_filter_error.valid = false;
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return e == hipErrorOutOfMemory ? AW_ERR_OUT_OF_MEMORY : AW_ERR_HIP;
}

bool context_literal_resampler(const aw_context *ctx) { return ctx && ctx->literal_resampler; }

aw_status caught() noexcept {
    aw_status code = AW_ERR_INVALID_ARGUMENT;
    try {
        try { throw; }
        catch (const std::bad_alloc &) { code = AW_ERR_OUT_OF_MEMORY; g_last_error = "host memory exhausted"; }
        catch (const std::length_error &) { code = AW_ERR_OUT_OF_MEMORY; g_last_error = "host container size limit"; }
        catch (const std::exception &e) { g_last_error = std::string("internal error: ") + e.what(); }
        catch (...) { g_last_error = "internal error"; }
    } catch (...) {          // (the message itself could not be stored)
    }
    return code;
}

}  // namespace awr

using awr::fail;

extern "C" {

const char *aw_version(void) { return "airwave-hip 0.1 (gfx950)"; }

const char *aw_last_error_message(void) { return awr::g_last_error.c_str(); }

int32_t aw_last_eq_filter_error(int32_t *enabled_index, int32_t *error_kind, int32_t *source_line) {
    if (!awr::g_eq_filter_error.valid) return 0;
    if (enabled_index) *enabled_index = awr::g_eq_filter_error.index;
    if (error_kind) *error_kind = awr::g_eq_filter_error.kind;
    if (source_line) *source_line = awr::g_eq_filter_error.line;
    return 1;
}

const char *aw_status_string(aw_status s) {
    switch (s) {
        case AW_OK: return "ok";
        case AW_ERR_INVALID_ARGUMENT: return "invalid argument";
        case AW_ERR_OUT_OF_MEMORY: return "out of memory";
        case AW_ERR_HIP: return "HIP runtime error";
        case AW_ERR_NO_DEVICE: return "no HIP device";
        case AW_ERR_INVALID_CHANNEL_MAPPING: return "Invalid channel mapping";            // HRIRManager.swift:754
        case AW_ERR_CONVOLUTION_SETUP_FAILED: return "Failed to set up convolution";      // :752
        case AW_ERR_INVALID_CHANNEL_COUNT: return "Invalid channel count";                // :748, WAVLoader.swift:137
        case AW_ERR_WAV_FILE_READ: return "WAV file read error";                          // WAVLoader.swift:135
        case AW_ERR_WAV_EMPTY_FILE: return "WAV file is empty (0 frames)";                // :139
        case AW_ERR_WAV_UNSUPPORTED_FORMAT: return "Unsupported WAV format";              // :143
        case AW_ERR_BLOCK_SIZE_MISMATCH: return "frame count differs from the engine block size";
        case AW_ERR_EQ_PARSE: return "Could not read the equalizer preset";                // EqualizerAPOParser.swift:19
        case AW_ERR_EQ_INVALID_SAMPLE_RATE: return "Sample rate must be finite and positive.";   // ParametricEqualizerProcessor.swift:106-107
        case AW_ERR_EQ_NON_FINITE_PREAMP: return "Preamp must produce a finite linear gain.";   // :108-109
        case AW_ERR_EQ_TOO_MANY_FILTERS: return "Equalizer supports at most 64 filters";   // :110-111
        case AW_ERR_EQ_INVALID_FILTER: return "Filter is invalid";                          // :112-113
        case AW_ERR_EQ_NOT_FOLDABLE: return "Equalizer response too long to fold into the HRIR";
        default: return "unknown status";
    }
}

/* ---- context ------------------------------------------------------------------------------ */
static aw_status context_create_impl(int32_t device, void *ext_stream, bool use_ext, aw_context **out) {
    if (!out) return fail(AW_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(AW_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= count) return fail(AW_ERR_NO_DEVICE, "device ordinal out of range");
    AW_HIP_TRY(hipSetDevice(device));
    awr::Owner<aw_context> owner(new (std::nothrow) aw_context(), aw_context_destroy);
    aw_context *c = owner.get();
    if (!c) return fail(AW_ERR_OUT_OF_MEMORY, "context");
    c->device = device;
    if (use_ext) {
        c->stream = reinterpret_cast<hipStream_t>(ext_stream);
        c->owns_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { c->stream = nullptr; return awr::hip_fail(e, "hipStreamCreate"); }
        c->owns_stream = true;
    }
    hipError_t e = hipEventCreate(&c->t0);
    if (e == hipSuccess) e = hipEventCreate(&c->t1);
    if (e == hipSuccess) e = awk::prepare_kernels(&c->cfg);
    if (e == hipSuccess) e = awk::prepare_ola_kernels();
    if (e == hipSuccess) e = awk::prepare_lw_kernels();
    if (e == hipSuccess) e = awk::prepare_eq_kernels();
    if (e == hipSuccess) e = awk::prepare_prep_kernels();
    awh::Twiddles tw;
    awh::build_twiddles(tw);
    auto upload = [&](const std::vector<awk::cf> &v, awk::cf **d) -> hipError_t {
        hipError_t r = hipMalloc(reinterpret_cast<void **>(d), v.size() * sizeof(awk::cf));
        if (r != hipSuccess) return r;
        return hipMemcpy(*d, v.data(), v.size() * sizeof(awk::cf), hipMemcpyHostToDevice);
    };
    if (e == hipSuccess) e = upload(tw.tw1, &c->d_tw1);
    if (e == hipSuccess) e = upload(tw.twa, &c->d_twa);
    if (e == hipSuccess) e = upload(tw.twb, &c->d_twb);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->d_zeros), 4096);
    if (e == hipSuccess) e = hipMemset(c->d_zeros, 0, 4096);
    if (e != hipSuccess) return awr::hip_fail(e, "context setup");
    *out = owner.release();
    return AW_OK;
}

aw_status aw_context_create(int32_t device, aw_context **out) try { return context_create_impl(device, nullptr, false, out); } AW_NOEXCEPT_TAIL
aw_status aw_context_create_on_stream(int32_t device, void *hip_stream, aw_context **out) try {
    return context_create_impl(device, hip_stream, true, out);
} AW_NOEXCEPT_TAIL

void aw_context_destroy(aw_context *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->d_tw1) (void)hipFree(c->d_tw1);
    if (c->d_twa) (void)hipFree(c->d_twa);
    if (c->d_twb) (void)hipFree(c->d_twb);
    if (c->d_zeros) (void)hipFree(c->d_zeros);
    if (c->d_pool) (void)hipFree(c->d_pool);
    delete c->copy_pool;
    delete c->copy_pool_out;
    if (c->s_h2d) (void)hipStreamDestroy(c->s_h2d);
    if (c->s_d2h) (void)hipStreamDestroy(c->s_d2h);
    for (int i = 0; i < 2; ++i) {
        if (c->ev_h2d[i]) (void)hipEventDestroy(c->ev_h2d[i]);
        if (c->ev_run[i]) (void)hipEventDestroy(c->ev_run[i]);
        if (c->ev_d2h[i]) (void)hipEventDestroy(c->ev_d2h[i]);
    }
    if (c->t0) (void)hipEventDestroy(c->t0);
    if (c->t1) (void)hipEventDestroy(c->t1);
    if (c->owns_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

aw_status aw_context_synchronize(aw_context *c) try {
    if (!c) return fail(AW_ERR_INVALID_ARGUMENT, "ctx is NULL");
    AW_HIP_TRY(hipStreamSynchronize(c->stream));
    return AW_OK;
} AW_NOEXCEPT_TAIL

aw_status aw_context_set_resampler(aw_context *c, int32_t literal_vgenp) try {
    if (!c) return fail(AW_ERR_INVALID_ARGUMENT, "ctx is NULL");
    c->literal_resampler = literal_vgenp != 0;
    return AW_OK;
} AW_NOEXCEPT_TAIL

void *aw_context_stream(aw_context *c) { return c ? reinterpret_cast<void *>(c->stream) : nullptr; }

aw_status aw_context_timer_start(aw_context *c) try {
    if (!c) return fail(AW_ERR_INVALID_ARGUMENT, "ctx is NULL");
    AW_HIP_TRY(hipEventRecord(c->t0, c->stream));
    return AW_OK;
} AW_NOEXCEPT_TAIL

aw_status aw_context_timer_stop(aw_context *c, float *ms) try {
    if (!c || !ms) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
    AW_HIP_TRY(hipEventRecord(c->t1, c->stream));
    AW_HIP_TRY(hipEventSynchronize(c->t1));
    AW_HIP_TRY(hipEventElapsedTime(ms, c->t0, c->t1));
    return AW_OK;
} AW_NOEXCEPT_TAIL

// Grows the context's scratch pool to `need` complex elements (the caller holds launch_mu).  The pool is shared by every handle of the
// context, so a failed growth must not take it away from handles that reserved it earlier (round-5 advice): the old size is allocated
// again before the error is returned — if even that fails the pool is empty and the next process call of any handle allocates again (or
// reports the same error).  hipFree waits for the device: launches of other handles that still read the old buffer have finished.
static aw_status pool_grow(aw_context *c, size_t need) {
    const size_t old = c->pool_capacity;
    if (c->d_pool) AW_HIP_TRY(hipFree(c->d_pool));
    c->d_pool = nullptr; c->pool_capacity = 0;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&c->d_pool), need * sizeof(awk::cf));
    if (e != hipSuccess) {
        c->d_pool = nullptr;
        if (old > 0 && hipMalloc(reinterpret_cast<void **>(&c->d_pool), old * sizeof(awk::cf)) == hipSuccess) {
            c->device_allocs += 1;
            c->pool_capacity = old;
        } else {
            c->d_pool = nullptr;
        }
        return awr::hip_fail(e, "scratch pool");
    }
    c->device_allocs += 1;
    c->pool_capacity = need;
    return AW_OK;
}

/* The context's scratch pool (runtime.hpp), sized ahead of time: a host that knows its largest batch pays the one large hipMalloc at
 * start-up (its wall time is erratic on these boxes: 0.2 ms ... 3.7 s, profiles/round5_v1/alloc_probe.txt) instead of inside the first
 * aw_spatializer_reserve / process that needs it.  Grow-only; bytes the pool already holds are kept. */
aw_status aw_context_reserve_scratch(aw_context *c, size_t bytes) try {
    if (!c) return fail(AW_ERR_INVALID_ARGUMENT, "ctx is NULL");
    AW_HIP_TRY(hipSetDevice(c->device));
    std::lock_guard<std::mutex> lk(c->launch_mu);
    const size_t need = (bytes + sizeof(awk::cf) - 1) / sizeof(awk::cf);
    if (c->pool_capacity >= need) return AW_OK;
    return pool_grow(c, need);
} AW_NOEXCEPT_TAIL
size_t aw_context_scratch_bytes(const aw_context *c) { return c ? c->pool_capacity * sizeof(awk::cf) : 0; }

/* Measured ceilings of the device the context runs on ("confirm on the box and also quote a measured copy-kernel
 * ceiling"; bench.py's roofline.measured).  Own buffers, freed before returning; blocks; never on a process path. */
aw_status aw_context_bandwidth_probe(aw_context *c, size_t bytes, int32_t repetitions, double *read_gbs, double *write_gbs, double *copy_gbs) try {
    if (!c || !read_gbs || !write_gbs || !copy_gbs) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
    if (bytes < ((size_t)64 << 20) || repetitions < 1) return fail(AW_ERR_INVALID_ARGUMENT, "probe needs >= 64 MiB and >= 1 repetition");
    AW_HIP_TRY(hipSetDevice(c->device));
    void *a = nullptr, *b = nullptr; float *sink = nullptr;
    hipError_t e = hipMalloc(&a, bytes);
    if (e == hipSuccess) e = hipMalloc(&b, bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&sink), 64);
    // the source holds pseudo-random samples, not zeros (all-zero data draws less power and can read faster than real data does)
    if (e == hipSuccess) e = awk::launch_synth_fill(reinterpret_cast<float *>(a), 1, (long long)(bytes / sizeof(float)), 0xA17AEull, 0, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(b, 0, bytes, c->stream);
    double best[3] = {0.0, 0.0, 0.0};
    for (int what = 0; what < 3 && e == hipSuccess; ++what)
        for (int nt = 0; nt < 2 && e == hipSuccess; ++nt) {
            size_t moved = 0;
            e = awk::launch_bw_probe(what, nt != 0, a, b, bytes, c->cfg.cus, sink, c->stream, &moved);       // warm-up (clocks, TLB)
            for (int r = 0; r < repetitions && e == hipSuccess; ++r) {
                e = hipEventRecord(c->t0, c->stream);
                if (e == hipSuccess) e = awk::launch_bw_probe(what, nt != 0, a, b, bytes, c->cfg.cus, sink, c->stream, &moved);
                if (e == hipSuccess) e = hipEventRecord(c->t1, c->stream);
                if (e == hipSuccess) e = hipEventSynchronize(c->t1);
                float ms = 0.f;
                if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->t0, c->t1);
                if (e == hipSuccess && ms > 0.f) best[what] = std::max(best[what], (double)moved / (ms * 1e-3) / 1e9);
            }
        }
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (sink) (void)hipFree(sink);
    if (e != hipSuccess) return awr::hip_fail(e, "bandwidth probe");
    *read_gbs = best[0]; *write_gbs = best[1]; *copy_gbs = best[2];
    return AW_OK;
} AW_NOEXCEPT_TAIL

/* Host link: page-locked host memory to the device and back with hipMemcpyAsync, each way alone and both ways at once (two streams).
 * The yardstick of the host entry's PCIe-inclusive rate (bench.py's secondary_end_to_end). */
aw_status aw_context_pcie_probe(aw_context *c, size_t bytes, int32_t repetitions, double *h2d_gbs, double *d2h_gbs, double *duplex_gbs) try {
    if (!c || !h2d_gbs || !d2h_gbs || !duplex_gbs) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
    if (bytes < ((size_t)16 << 20) || repetitions < 1) return fail(AW_ERR_INVALID_ARGUMENT, "probe needs >= 16 MiB and >= 1 repetition");
    AW_HIP_TRY(hipSetDevice(c->device));
    void *h_in = nullptr, *h_out = nullptr, *d_in = nullptr, *d_out = nullptr;
    hipStream_t s2 = nullptr;
    hipError_t e = hipHostMalloc(&h_in, bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(&h_out, bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(&d_in, bytes);
    if (e == hipSuccess) e = hipMalloc(&d_out, bytes);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    if (e == hipSuccess) { std::memset(h_in, 0, bytes); std::memset(h_out, 0, bytes); e = hipMemsetAsync(d_out, 0, bytes, c->stream); }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    double best[3] = {0.0, 0.0, 0.0};
    for (int what = 0; what < 3 && e == hipSuccess; ++what)
        for (int r = 0; r <= repetitions && e == hipSuccess; ++r) {          // (r = 0: warm-up)
            const auto t0 = std::chrono::steady_clock::now();
            if (what != 1) e = hipMemcpyAsync(d_in, h_in, bytes, hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess && what != 0) e = hipMemcpyAsync(h_out, d_out, bytes, hipMemcpyDeviceToHost, what == 2 ? s2 : c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e == hipSuccess && what == 2) e = hipStreamSynchronize(s2);
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (r > 0 && sec > 0.0) best[what] = std::max(best[what], (double)bytes * (what == 2 ? 2.0 : 1.0) / sec / 1e9);
        }
    if (s2) (void)hipStreamDestroy(s2);
    if (h_in) (void)hipHostFree(h_in);
    if (h_out) (void)hipHostFree(h_out);
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return awr::hip_fail(e, "pcie probe");
    *h2d_gbs = best[0]; *d2h_gbs = best[1]; *duplex_gbs = best[2];
    return AW_OK;
} AW_NOEXCEPT_TAIL

aw_status aw_device_alloc(aw_context *c, size_t bytes, void **dptr) try {
    if (!c || !dptr) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
    AW_HIP_TRY(hipSetDevice(c->device));
    AW_HIP_TRY(hipMalloc(dptr, bytes ? bytes : 1));
    return AW_OK;
} AW_NOEXCEPT_TAIL
aw_status aw_device_free(aw_context *c, void *dptr) try {
    if (!c) return fail(AW_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (dptr) AW_HIP_TRY(hipFree(dptr));
    return AW_OK;
} AW_NOEXCEPT_TAIL
aw_status aw_memcpy_h2d(aw_context *c, void *dst, const void *src, size_t bytes) try {
    if (!c || (bytes && (!dst || !src))) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
    AW_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    AW_HIP_TRY(hipStreamSynchronize(c->stream));
    return AW_OK;
} AW_NOEXCEPT_TAIL
aw_status aw_memcpy_d2h(aw_context *c, void *dst, const void *src, size_t bytes) try {
    if (!c || (bytes && (!dst || !src))) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
    AW_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    AW_HIP_TRY(hipStreamSynchronize(c->stream));
    return AW_OK;
} AW_NOEXCEPT_TAIL

/* ---- HRIR set ------------------------------------------------------------------------------- */
aw_status aw_hrir_create(aw_context *ctx, const float *tracks, int32_t n_tracks, int32_t taps, double sample_rate,
                         aw_hrir **out) try {
    if (!out) return fail(AW_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (!ctx || !tracks) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_tracks <= 0) return fail(AW_ERR_INVALID_CHANNEL_COUNT, "HRIR needs at least one track");
    if (taps <= 0) return fail(AW_ERR_WAV_EMPTY_FILE, "HRIR has no taps");
    awr::Owner<aw_hrir> h(new (std::nothrow) aw_hrir(), aw_hrir_destroy);
    if (!h) return fail(AW_ERR_OUT_OF_MEMORY, "hrir");
    h->ctx = ctx; h->n_tracks = n_tracks; h->taps = taps; h->sample_rate = sample_rate;
    h->tracks.assign(tracks, tracks + (size_t)n_tracks * taps);
    *out = h.release();
    return AW_OK;
} AW_NOEXCEPT_TAIL
void aw_hrir_destroy(aw_hrir *h) { delete h; }
int32_t aw_hrir_track_count(const aw_hrir *h) { return h ? h->n_tracks : 0; }
int32_t aw_hrir_taps(const aw_hrir *h) { return h ? h->taps : 0; }
double aw_hrir_sample_rate(const aw_hrir *h) { return h ? h->sample_rate : 0.0; }

/* ---- spatializer ---------------------------------------------------------------------------- */
static aw_status sp_alloc_hist(aw_spatializer *sp) {
    // + 4 floats: whole-frame vector loads of layouts whose frames are not float4s run up to 3 floats past a frame
    const size_t n = (size_t)sp->n_streams * sp->hist_len * sp->n_channels + 4;
    for (int i = 0; i < 2; ++i) {
        AW_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sp->d_hist[i]), n * sizeof(float)));
        AW_HIP_TRY(hipMemsetAsync(sp->d_hist[i], 0, n * sizeof(float), sp->ctx->stream));
    }
    sp->hist_cur = 0;
    return AW_OK;
}

// Tiles start on 64-frame boundaries of the timeline: a shorter hop costs < 2 % more tiles and makes every tile's
// loads and stores start line-aligned (measured cfg 2: 1.555 -> 1.530 ms per call).  AW_HOP_ALIGN (read at aw_context_create) overrides (1 = off).
// Overlap-add tile or overlap-save tile for a path-0 spatializer?  Returns the block rows H (0: overlap-save).  `hist_len` = rows of the
// history buffer the spatializer will keep (N - the aligned overlap-save hop >= taps - 1): the block's tail must fit behind its hop.
// The overlap-add block is cheaper per transform (whole frames in registers, every input line read once, no boundary launch) but
// shorter than the overlap-save hop (512 H against 8193 - taps rounded down to 64), so it wins from a layout-specific ratio of the two
// on — measured, profiles/round6_v1/ola_sweep.txt (128 streams x 10 s, G frames/s overlap-save / overlap-add / long-window):
//   4 ch   3585 taps 92.2 / 87.3, 3969: 86.3 / 86.6, 4320: 79.5 / 79.5, 4609: 74.6 / 79.2            -> from hop_ola >= 0.94 hop_ols
//   6 ch   3585: 64.7 / 64.4, 3969: 60.9 / 65.2, 4098: 58.4 / 58.0, 4320: 56.5 / 57.8                 -> 0.90
//   7 ch   3000: 58.7 / 54.5, 3585: 53.3 / 54.2, 4320: 46.0 / 48.8, 4609: 42.9 (16384-frame tile) / 48.7   -> 0.875
//   8 ch   3585: 53.5 / 53.4, 3969: 49.4 / 53.3, 4320: 46.7 / 47.7, 5121: 34.6 / 41.7 / 39.0          -> 0.89
//   10 ch  2048: 45.6 / 42.7, 3000: 40.7 / 42.7, 4320: 32.4 / 38.7 / 30.0                             -> 0.74
//   12 ch  3000: 36.7 / 36.4, 3585: 34.1 / 36.2, 4320: 29.8 / 32.9 / 25.7                             -> 0.80
//   14 ch  3000: 29.7 / 27.5, 3585: 27.5 / 27.6, 4320: 23.5 / 27.7 / 23.8, 5121: 20.1 / 25.4 / 23.7   -> 0.89
//   16 ch  2048: 24.1 / 24.5, 4320: 17.6 / 22.7 / 20.6, 5121: 14.9 / 21.3 / 20.4                      -> 0.66
// odd wide layouts (profiles/round6_v5/ola_sweep_odd.txt): 9 ch 3000: 42.5 / 43.1, 4320: 33.9 / 38.8 / 31.6; 11 ch 4320: 29.3 / 33.4 / 28.1;
//   13 ch 4320: 24.1 / 28.6 / 24.9 -> 0.78 each; 15 ch 2048: 24.6 / 26.7, 4320: 17.7 / 24.2 / 21.7 -> 0.62; 5 channels: the 16384-frame
//   tile stays faster at every length (64.5 against 57.4 at 4320 taps): no kernel
// (tools/ola_sweep.py regenerates the table).  AW_OLA=0 never, AW_OLA=1 wherever a kernel exists.
static int ola_policy(int n_channels, int taps, int hist_len) {
    int mode = -1;
    if (const char *e = getenv("AW_OLA")) mode = atoi(e);
    if (mode == 0) return 0;
    int H = awk::fused_ola_rows(n_channels, taps);
    while (H >= 6 && hist_len > awk::kN - 512 * H) --H;          // (the aligned history may be a few rows longer than taps - 1)
    if (H < 6) return 0;
    if (mode > 0) return H;
    const int hop_ols = awk::kN - hist_len;
    double need;
    switch (n_channels) {
        case 4: need = 0.94; break;
        case 6: need = 0.90; break;
        case 7: need = 0.875; break;
        case 8: need = 0.89; break;
        case 9: need = 0.78; break;
        case 10: need = 0.74; break;
        case 11: need = 0.78; break;
        case 12: need = 0.80; break;
        case 13: need = 0.78; break;
        case 14: need = 0.89; break;
        case 15: need = 0.62; break;
        default: need = 0.66; break;      // 16 channels
    }
    return 512.0 * H >= need * hop_ols ? H : 0;
}
// Blocks per persistent workgroup from which a call runs the overlap-add tile: every run pays ceil(hist / hop) <= 2 warm-up blocks, which
// must stay below what the tile gains over the overlap-save tile — 2 - 10 % for layouts of up to eight channels, 10 - 40 % for wider ones.
static int ola_min_blocks(const aw_spatializer *sp) {
    const int forced = sp->ctx->cfg.ola_min_blocks_per_wg;       // AW_OLA_MIN_BLOCKS (read at context creation); < 0: per layout
    if (forced >= 0) return forced;
    return sp->n_channels <= 8 ? 48 : 16;
}

static int align_hop(const aw_context *ctx, int hop) {
    const int al = ctx->cfg.hop_align;          // read once at aw_context_create (LaunchCfg), like every other knob
    return (al > 1 && hop > 16 * al) ? hop - hop % al : hop;
}

aw_status aw_spatializer_create(aw_context *ctx, const aw_hrir *hrir, int32_t n_in, const int32_t *left_track,
                                const int32_t *right_track, int32_t n_streams, int32_t block_hint,
                                aw_spatializer **out) try {
    (void)block_hint;
    if (!out) return fail(AW_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (!ctx || !hrir || !left_track || !right_track) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_in <= 0 || n_streams <= 0) return fail(AW_ERR_INVALID_ARGUMENT, "channel and stream counts must be positive");
    // the per-speaker loop of activatePreset: skip unmapped, bounds-check, need >= 1 renderer
    int mapped = 0;
    for (int c = 0; c < n_in; ++c) {
        const int l = left_track[c], r = right_track[c];
        if (l < 0 || r < 0) continue;                                        // HRIRManager.swift:370-372
        if (l >= hrir->n_tracks || r >= hrir->n_tracks)                      // :375-379
            return fail(AW_ERR_INVALID_CHANNEL_MAPPING,
                        "HRIR indices (" + std::to_string(l) + ", " + std::to_string(r) + ") out of range for " +
                            std::to_string(hrir->n_tracks) + " channels");
        ++mapped;
    }
    if (mapped == 0) return fail(AW_ERR_CONVOLUTION_SETUP_FAILED, "No valid renderers created");   // :420-422
    AW_HIP_TRY(hipSetDevice(ctx->device));

    awr::Owner<aw_spatializer> owner(new (std::nothrow) aw_spatializer(), aw_spatializer_destroy);      // destroyed on every early return below
    aw_spatializer *sp = owner.get();
    if (!sp) return fail(AW_ERR_OUT_OF_MEMORY, "spatializer");
    sp->ctx = ctx; sp->n_channels = n_in; sp->n_pairs = (n_in + 1) / 2; sp->n_streams = n_streams;
    sp->taps = hrir->taps;
    const int N = awk::kN;
    // Path choice.  AW_WINDOW=8192|16384 forces the fused window, 4096 the partitioned path (tuning / A-B); default: see below.
    int window = 0;
    if (const char *e = getenv("AW_WINDOW")) window = atoi(e);
    const int hist2 = awh::poly_history_frames(hrir->taps);          // 16384-frame windows (tile_ols2.hpp)
    const bool fits1 = hrir->taps - 1 <= N - 2048, fits2 = hist2 <= awk::kN2 - 4096;
    // Path / window policy.  Every threshold below is a measured crossover (G stereo frames/s, 128 streams x 4 s unless stated) and
    // is regenerated in one command: `bash tools/regen_policy.sh` (round 2, after the marched partitioned path, the two-pass
    // wide kernels and the per-layout SLP choice moved most of them).
    // (1) Long end of the 16384-frame windows against the partitioned path (the hop shrinks to 4096 frames at 12 289 taps):
    //     1, 3 and 5 channels stay fused to the window's limit (mono 12 289 taps: 79 / 49), stereo too (12 289: 53 / 51);
    //     7 channels up to ~11 800 (11 000: 19.7 / 18.8; 12 289: 16.7 / 18.3); the even layouts leave early, their 16384-frame
    //     kernels carry ~100 spilled VGPRs: 4 channels up to ~7000 (6146: 36.8 / 35.1; 8000: 30.7 / 35.4), 6 up to ~8500
    //     (8000: 23.5 / 21.8; 9000: 20.2 / 21.1), 8 up to ~7800 (6146: 20.6 / 17.6; 8000: 18.2 / 18.2).  9+ channels have no
    //     16384-frame vector kernels: partitioned as soon as one 8192-frame window cannot hold the HRIR.
    //     Round 4, the 16384-frame kernels on the half-wave row transform (no spills, 4 / 6 / 8 channels +12 / +27 / +11 %), re-measured
    //     (profiles/round4_v2/policy_sweeps_final.txt, 16384 / partitioned): 4 channels up to ~10 500 (10 000: 36.7 / 33.2; 11 000: 32.7 / 35.8),
    //     6 to the window's limit (12 289: 21.0 / 20.7), 7 up to ~11 500 (11 000: 21.1 / 19.1; 11 800: 18.2 / 19.0), 8 up to ~11 200
    //     (11 000: 17.4 / 17.2; 11 800: 15.5 / 17.3).
    const int upto = (n_in <= 3 || n_in == 5 || n_in == 6) ? 12289 : n_in == 4 ? 10500 : n_in == 8 ? 11200 : n_in == 7 ? 11500 : 0;
    const bool fused2_ok = fits2 && hrir->taps <= upto;
    // (0) Round 6: the overlap-add form of the 8192-frame tile (device/tile_ola.hpp) where the library carries a kernel for the layout and
    //     the HRIR length (blocks of 512 H frames, H = 5 .. 8: 3585 .. 5633 taps and shorter ones on H = 8).  It reads every input line
    //     once and holds whole frames of every layout, so it is measured against BOTH overlap-save windows (ola_policy()); the choice is
    //     per call: calls with too few blocks per workgroup to amortise a run's warm-up blocks stay on the overlap-save tile.
    //     AW_OLA=0 never, AW_OLA=1 wherever a kernel exists.
    const int ola_h = (window == 0 || window == N) ? ola_policy(n_in, hrir->taps, N - align_hop(ctx, N - (hrir->taps - 1))) : 0;
    // batches of fewer than 48 streams keep the window policy below: their calls rarely have the blocks per workgroup the overlap-add tile
    // needs (ola_min_blocks), and then run the overlap-save tile of the window chosen here
    if (ola_h && window == 0 && n_streams >= 48) window = N;
    if (window == 0) {
        // (2) 8192- against 16384-frame windows where both can hold the HRIR (8192 / 16384): mono always 16384 (4320 taps 93 / 202),
        //     stereo from ~2000 taps (2048: 157 / 160; 4320: 105 / 143), 3 channels from ~1800 (1024: 120 / 113; 2048: 107 / 111;
        //     4320: 67 / 101), 5 from ~3200 (3000: 57.6 / 56.2; 3600: 52.4 / 56.9; 4320: 43 / 55) — both with the 8192-frame vector
        //     variants <3,2> and <5,3> of the end of round 2 (generic kernels before: crossovers at ~1000 / ~2000) —, 7 from ~4400 (4320: 35.9 / 35.6; 5000: 29.7 / 33.2); 4, 6 and 8 channels only at the very
        //     end of the 8192-frame window's range (6 and 8 from ~6100: 26.4 / 27.5 and 21.1 / 21.2 at 6145; 4 never: 38.2 / 36.9).
        //     Odd counts cross early: the 8192-frame kernels pad them to whole pairs, the polyphase view has 2C pseudo-channels.
        const int c = n_in;
        //     Round 4: both fused kernels moved to the half-wave row transform (tile_ols.hpp; the 16384-frame kernels with their tables in
        //     parts of four bins and all of them without SLP), re-measured (profiles/round4_v2/policy_sweeps_final.txt, 128 streams x 4 s):
        //     stereo from ~1800 (1500: 180 / 171; 1800: 171 / 172), 3 channels from ~1800 (1500: 123 / 119; 1800: 117 / 119), 5 from ~3400
        //     (3200: 63.7 / 60.5; 3500: 59.9 / 61.4), 7 from ~4400 (4320: 38.9 / 38.8; 4500: 36.4 / 36.8), 4 from ~5300 (5000: 59.9 / 58.5;
        //     5300: 57.9 / 58.0), 6 from ~4700 (4320: 47.2 / 45.9; 5000: 39.4 / 44.9), 8 from ~5500 (5300: 30.1 / 29.0; 5600: 27.4 / 28.4).
        const int from = c == 1 ? 0 : c == 2 ? 1800 : c == 3 ? 1800 : c == 5 ? 3400 : c == 7 ? 4400 : c == 4 ? 5300 : c == 6 ? 4700 : c == 8 ? 5500 : (1 << 30);
        window = (hrir->taps >= from && fused2_ok) ? awk::kN2 : AW_DEFAULT_WINDOW;
        // (3) small batches cannot fill 256 CUs with 16384-frame tiles (a 10 s stream is 40 of them): the 8192-frame kernels give
        //     three times the tiles.  Crossover in streams (tools/small_batch_sweep.py, 4320 taps, 10 s per stream): mono 8
        //     (4 streams 67 / 55, 8: 79 / 81), stereo 24 (16: 88 / 86, 24: 92 / 107), 3 channels 12, 5 channels 24; 16 elsewhere
        const int min_streams = c == 1 ? 8 : c == 2 ? 24 : c == 3 ? 12 : c == 5 ? 24 : 16;
        if (window == awk::kN2 && n_streams < min_streams && fits1) window = awk::kN;
    }
    // (4) 9+ channels near the end of the 8192-frame window's range (hop down to 2048 frames) against the partitioned path
    //     (8192 / partitioned, G frames/s at 6145 taps, tools/path_sweep.py): with the one-pass vector kernels of round 2 the fused
    //     kernels win to the end for 9-15 channels (9: 16.3 / 13.4, 12: 14.3 / 11.6, 14: 11.3 / 10.2, 15: 9.2 / 8.9); 16 channels
    //     cross at ~6000 taps (5800: 10.1 / 9.5; 6145: 9.1 / 9.4)
    const bool prefer_partitioned = getenv("AW_WINDOW") == nullptr && n_in >= 16 && hrir->taps >= 6000;
    const bool force_partitioned = window == 4096 || prefer_partitioned;       // AW_WINDOW=4096: the partitioned path (A/B)
    if (!force_partitioned && ((window == awk::kN2 && fits2) || (!fits1 && fused2_ok))) {
        sp->path = 0; sp->fused2 = true;
        sp->hop = align_hop(ctx, awk::kN2 - hist2);
        sp->hist_len = awk::kN2 - sp->hop;                           // even, >= 2 * floor(taps / 2)
        sp->partitions = 1;
        sp->n_pairs = (2 * n_in + 1) / 2;                            // pseudo-pairs of the half-rate 2C-channel view
    } else if (fits1 && !force_partitioned) {
        sp->path = 0;
        sp->hop = align_hop(ctx, N - (hrir->taps - 1));
        sp->hist_len = N - sp->hop;
        sp->partitions = 1;
        sp->ola_h = ola_h;
    } else {
        sp->path = 1;
        sp->hop = N / 2;
        sp->partitions = (hrir->taps + sp->hop - 1) / sp->hop;
        sp->hist_len = sp->partitions * sp->hop;
        // every tuning knob of this path is read here, never on the process path
        sp->cmac_group = sp->n_pairs > 8;                               // the marched kernel's lane groups hold up to 8 channel pairs
        if (const char *e = getenv("AW_PART_CMAC")) sp->cmac_group = sp->cmac_group || std::strcmp(e, "group") == 0;
        // forward kernel form: all pairs of a window in one workgroup up to 4 pairs (cfg 3: 11.2 against 12.1 ms); with more pairs
        // the four-channel batches of 56-byte-or-wider frames each re-read every line (14 channels: fabric reads 4.3x the input),
        // and one pair per workgroup — the pairs of a window side by side on one XCD — wins (27.0 -> 22.7 ms).  AW_PART_FWD=1|2 forces.
        sp->fwd_one_pair = sp->n_pairs > 4;
        if (const char *e = getenv("AW_PART_FWD")) sp->fwd_one_pair = atoi(e) == 1;
        if (const char *e = getenv("AW_PART_HERM")) sp->herm_ok = atoi(e) != 0;          // A/B: 0 stores the last pair's redundant half too
    }
    // Long calls may run on the long-window kernels (device/tile_lw.hpp) whatever the path above — chosen per call, see lw_choose();
    // they use the same history buffer.  AW_LW: 0 never, 32/64/128 force that window, unset = the measured policy.
    sp->lw_plans.reserve(kLwPlanSlots);                                                  // one entry per window length: pointers into it stay valid (static_assert at kLwRowChoices)
    if (const char *e = getenv("AW_LW")) sp->lw_mode = atoi(e);
    if (n_in > 16) sp->lw_mode = 0;                                                      // up to eight channel pairs
    // scratch budget per stream chunk: of the partitioned kernels and of the long-window ones, which also serve path-0 layouts
    if (const char *e = getenv("AW_SPEC_SCRATCH_MB")) sp->scratch_budget = (size_t)atoll(e) << 20;
    if (sp->lw_mode != 0) {
        sp->lw_tracks = hrir->tracks; sp->lw_n_tracks = hrir->n_tracks;
        sp->lw_left.assign(left_track, left_track + n_in); sp->lw_right.assign(right_track, right_track + n_in);
        if (n_in > 8) {                  // 32 floats for the wide split kernel's padded copy of a call's very last frame (tile_lw.hpp)
            hipError_t et = hipMalloc(reinterpret_cast<void **>(&sp->d_tail), 32 * sizeof(float));
            if (et == hipSuccess) et = hipMemsetAsync(sp->d_tail, 0, 32 * sizeof(float), ctx->stream);
            if (et != hipSuccess) return awr::hip_fail(et, "spatializer setup");
        }
    }
    // The tables of the short-call kernels (fused tiles, partitioned delay line) are built in float64 on the host, the analogue of the
    // partition FFTs of ConvolutionEngine.init (ConvolutionEngine.swift:141-175) — the partitions of a long HRIR side by side, one host
    // thread each (round 5; cfg 3: 8 partitions x 4 pairs of 8192-point transforms, 48 ms on one thread).  Nothing throws across the ABI.
    std::vector<awk::cf2> all;
    bool tables_ok = true;
    try {
        if (sp->fused2) {
            std::vector<awk::cf4> t4;
            tables_ok = awh::build_poly_tables(hrir->tracks.data(), hrir->n_tracks, hrir->taps, n_in, left_track, right_track, t4);
            all.resize(t4.size() * 2);
            std::memcpy(all.data(), t4.data(), t4.size() * sizeof(awk::cf4));
            all.resize(all.size() + 2 * (size_t)awk::kN, awk::cf2{awk::mk(0.f, 0.f), awk::mk(0.f, 0.f)});   // the zero pair
        } else {
            const int P = sp->partitions;
            std::vector<std::vector<awk::cf2>> tabs((size_t)P);
            tables_ok = awh::parallel_for(P, [&](int q) {
                const int off = sp->path == 0 ? 0 : q * sp->hop;
                const int cnt = sp->path == 0 ? hrir->taps : sp->hop;
                if (!awh::build_pair_tables(hrir->tracks.data(), hrir->n_tracks, hrir->taps, n_in, left_track, right_track, off, cnt, tabs[(size_t)q],
                                            /*pair_threads=*/P == 1))
                    throw std::bad_alloc();
            });
            for (int q = 0; q < P && tables_ok; ++q) all.insert(all.end(), tabs[(size_t)q].begin(), tabs[(size_t)q].end());
            // fused path: one all-zero pair after the last one — a phantom pair (odd pair count in the runtime-loop
            // kernels; stray lanes of non-float4 frames) multiplies it and contributes nothing
            if (sp->path == 0) all.resize(all.size() + awk::kN, awk::cf2{awk::mk(0.f, 0.f), awk::mk(0.f, 0.f)});
        }
    } catch (...) {
        tables_ok = false;
    }
    if (!tables_ok) return fail(AW_ERR_OUT_OF_MEMORY, "filter tables: host memory");
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&sp->d_tab), all.size() * sizeof(awk::cf2));
    if (e == hipSuccess) e = hipMemcpy(sp->d_tab, all.data(), all.size() * sizeof(awk::cf2), hipMemcpyHostToDevice);
    if (e != hipSuccess) return awr::hip_fail(e, "filter tables");
    aw_status st = sp_alloc_hist(sp);
    if (st != AW_OK) return st;
    e = hipEventCreate(&sp->k0);
    if (e == hipSuccess) e = hipEventCreate(&sp->k1);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return awr::hip_fail(e, "spatializer setup");
    *out = owner.release();
    return AW_OK;
} AW_NOEXCEPT_TAIL

void aw_spatializer_destroy(aw_spatializer *sp) {
    if (!sp) return;
    (void)hipSetDevice(sp->ctx->device);
    (void)hipStreamSynchronize(sp->ctx->stream);
    if (sp->d_tab) (void)hipFree(sp->d_tab);
    for (int i = 0; i < 2; ++i)
        if (sp->d_hist[i]) (void)hipFree(sp->d_hist[i]);
    if (sp->d_tail) (void)hipFree(sp->d_tail);
    if (sp->d_lw_tracks) (void)hipFree(sp->d_lw_tracks);
    if (sp->d_lw_left) (void)hipFree(sp->d_lw_left);
    if (sp->d_lw_right) (void)hipFree(sp->d_lw_right);
    for (auto &pl : sp->lw_plans) {
        if (pl.d_tab) (void)hipFree(pl.d_tab);
        if (pl.d_tab16) (void)hipFree(pl.d_tab16);
        if (pl.d_tw2) (void)hipFree(pl.d_tw2);
        if (pl.d_coarse) (void)hipFree(pl.d_coarse);
        if (pl.d_fine) (void)hipFree(pl.d_fine);
        if (pl.d_tw_r) (void)hipFree(pl.d_tw_r);
        if (pl.d_step) (void)hipFree(pl.d_step);
        if (pl.d_tw1m) (void)hipFree(pl.d_tw1m);
    }
    if (sp->d_dbg) (void)hipFree(sp->d_dbg);
    if (sp->d_stage_in) (void)hipFree(sp->d_stage_in);
    if (sp->d_stage_out) (void)hipFree(sp->d_stage_out);
    if (sp->h_pin_in) (void)hipHostFree(sp->h_pin_in);
    if (sp->h_pin_out) (void)hipHostFree(sp->h_pin_out);
    if (sp->h_bounce_in) (void)hipHostFree(sp->h_bounce_in);
    if (sp->h_bounce_out) (void)hipHostFree(sp->h_bounce_out);
    if (sp->k0) (void)hipEventDestroy(sp->k0);
    if (sp->k1) (void)hipEventDestroy(sp->k1);
    for (auto &pr : sp->pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto &r : sp->stage_pending) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto ev : sp->event_pool) (void)hipEventDestroy(ev);
    delete sp;
}

int32_t aw_spatializer_stream_count(const aw_spatializer *sp) { return sp ? sp->n_streams : 0; }
int32_t aw_spatializer_channel_count(const aw_spatializer *sp) { return sp ? sp->n_channels : 0; }
int64_t aw_spatializer_info(const aw_spatializer *sp, int32_t what) {
    if (!sp) return -1;
    switch (what) {
        case 0: return sp->fused2 ? awk::kN2 : awk::kN;
        case 1: return sp->hop;
        case 2: return sp->partitions;
        case 3: return sp->path;
        case 4: return sp->hist_len;
        case 5: return sp->dominant_frames;   // output frames covered by the launch aw_spatializer_kernel_time() times (last call)
        case 7: return sp->last_lw_R;         // long-window path: rows R of the last call's windows (N = R x 4096); 0 = the partitioned kernels ran
        case 8: return sp->last_lw_R2;        // rows of the last call's remainder window when it ran as two groups of windows (0: one group)
        case 9: return (int64_t)sp->lw_plans.size();   // long-window table sets built so far (one per window length; reserve builds those of its plan)
        case 6: return (int64_t)(sp->ctx->pool_capacity * sizeof(awk::cf) + (sp->stage_in_cap + sp->stage_out_cap) * sizeof(float));   // grow-only device buffers, bytes (the context's scratch pool + this handle's staging)
        case 10: return sp->reserve_tables_us;    // last aw_spatializer_reserve: float64 table build on host threads, microseconds
        case 11: return sp->reserve_upload_us;    //   table upload (hipMalloc + hipMemcpy)
        case 12: return sp->reserve_scratch_us;   //   scratch pool growth (hipMalloc)
        case 13: return sp->ctx->device_allocs;   // device / pinned allocations made so far on behalf of this context's handles
        case 14: return sp->ctx->sync_copies;     // blocking host-to-device table uploads likewise
        case 15: return sp->host_chunk_streams;   // streams per staged chunk of the host entry (0: the whole batch in one piece)
        case 16: return sp->last_ola_h;           // rows H of the overlap-add blocks (512 H frames) the last call ran on; 0: it ran another tile
        case 17: return sp->ola_h;                // rows H this spatializer's long-enough calls use (0: the overlap-add tile is not used)
        default: return -1;
    }
}

static void sp_drain_stages(aw_spatializer *sp);

aw_status aw_spatializer_set_profiling(aw_spatializer *sp, int32_t enabled) try {
    if (!sp) return fail(AW_ERR_INVALID_ARGUMENT, "sp is NULL");
    sp->profiling = enabled != 0;
    sp->kernel_ms_sum = 0.0;
    sp->kernel_launches = 0;
    sp_drain_stages(sp);
    sp->stage_stats.clear();
    return AW_OK;
} AW_NOEXCEPT_TAIL

static hipEvent_t sp_get_event(aw_spatializer *sp) {
    if (!sp->event_pool.empty()) {
        hipEvent_t e = sp->event_pool.back();
        sp->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

// HIP events around single launches on the context stream; drained into per-name sums by sp_drain_stages().
struct SpStageTimer final : awk::StageTimer {
    aw_spatializer *sp;
    hipEvent_t cur = nullptr;
    explicit SpStageTimer(aw_spatializer *s) : sp(s) {}
    void begin() override { cur = sp_get_event(sp); (void)hipEventRecord(cur, sp->ctx->stream); }
    void end(const char *name) override {
        hipEvent_t e1 = sp_get_event(sp);
        (void)hipEventRecord(e1, sp->ctx->stream);
        sp->stage_pending.push_back({name, cur, e1});
        cur = nullptr;
    }
};

static void sp_drain_stages(aw_spatializer *sp) {
    for (auto &r : sp->stage_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            auto it = std::find_if(sp->stage_stats.begin(), sp->stage_stats.end(), [&](const aw_spatializer::StageStat &s) { return std::strcmp(s.name, r.name) == 0; });
            if (it == sp->stage_stats.end()) sp->stage_stats.push_back({r.name, ms, 1});
            else { it->ms_sum += ms; it->launches += 1; }
        }
        sp->event_pool.push_back(r.e0);
        sp->event_pool.push_back(r.e1);
    }
    sp->stage_pending.clear();
}

int32_t aw_spatializer_stage_time(aw_spatializer *sp, int32_t index, const char **name, double *total_ms, int32_t *launches) {
    if (!sp || index < 0) return 0;
    sp_drain_stages(sp);
    if ((size_t)index >= sp->stage_stats.size()) return 0;
    const auto &s = sp->stage_stats[(size_t)index];
    if (name) *name = s.name;
    if (total_ms) *total_ms = s.ms_sum;
    if (launches) *launches = s.launches;
    return 1;
}

int32_t aw_spatializer_kernel_time(aw_spatializer *sp, double *avg_ms, const char **kernel_name) {
    if (!sp) return 0;
    for (auto &pr : sp->pending) {
        float ms = 0.f;
        if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
            sp->kernel_ms_sum += ms;
            sp->kernel_launches += 1;
        }
        sp->event_pool.push_back(pr.first);
        sp->event_pool.push_back(pr.second);
    }
    sp->pending.clear();
    if (avg_ms) *avg_ms = sp->kernel_launches ? sp->kernel_ms_sum / sp->kernel_launches : 0.0;
    if (kernel_name) *kernel_name = sp->last_lw_R ? "aw_lw_split_kernel + aw_lw_rows_kernel + aw_lw_merge_kernel" : sp->path == 0 ? (sp->fused2 ? awk::fused_ols2_kernel_name(sp->n_channels) : sp->last_ola_h ? awk::fused_ola_kernel_name(sp->n_channels, sp->last_ola_h) : awk::fused_ols_kernel_name(sp->n_channels)) : (sp->cmac_group ? "aw_part_forward_kernel + aw_part_cmac_kernel + aw_part_inverse_kernel" : "aw_part_forward_kernel + aw_part_march_kernel + aw_part_inverse_kernel");
    const int n = sp->kernel_launches;
    sp->kernel_ms_sum = 0.0;
    sp->kernel_launches = 0;
    return n;
}

aw_status aw_spatializer_debug_stamps(aw_spatializer *sp, uint64_t *host_out, int64_t capacity_words,
                                      int64_t *n_workgroups) try {
    if (!sp || !host_out || !n_workgroups) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
#if defined(AW_STAMPS) && AW_STAMPS
    const int64_t words = (int64_t)sp->dbg_nwg * awk::kStamps;
    if (!sp->d_dbg || words <= 0) return fail(AW_ERR_INVALID_ARGUMENT, "no launch recorded");
    if (capacity_words < words) return fail(AW_ERR_INVALID_ARGUMENT, "capacity too small");
    AW_HIP_TRY(hipStreamSynchronize(sp->ctx->stream));
    AW_HIP_TRY(hipMemcpy(host_out, sp->d_dbg, (size_t)words * sizeof(uint64_t), hipMemcpyDeviceToHost));
    *n_workgroups = sp->dbg_nwg;
    return AW_OK;
#else
    (void)capacity_words;
    *n_workgroups = 0;
    return fail(AW_ERR_INVALID_ARGUMENT, "library not built with AW_STAMPS");
#endif
} AW_NOEXCEPT_TAIL

static void sp_fill_cfg(const aw_spatializer *sp, awk::TileParams &p) {
    const awk::LaunchCfg &c = sp->ctx->cfg;
    p.persistent_wgs = c.persistent_wgs; p.wide_two_pass = c.wide_two_pass;
    p.debug_occupancy = c.debug_occupancy;
}

// Every sp_process_* below runs ONE call's kernels for the streams [s_base, s_base + ns_total) of the spatializer; `in` / `out` point at
// stream s_base's frames (the caller's whole batch with s_base = 0, or one staged chunk of the host-entry pipeline).
static aw_status sp_process_fused(aw_spatializer *sp, int s_base, int ns_total, const float *in, float *out, int64_t frames) {
    awk::TileParams p{};
    p.in = in; p.out = out; p.hist = sp->d_hist[sp->hist_cur] + (size_t)s_base * sp->hist_len * sp->n_channels;
    p.tab = sp->d_tab; p.tw1 = sp->ctx->d_tw1; p.twa = sp->ctx->d_twa; p.twb = sp->ctx->d_twb; p.zeros = sp->ctx->d_zeros;
    p.frames = frames; p.n_channels = sp->n_channels; p.n_pairs = sp->n_pairs;
    p.hop = sp->hop; p.hist_len = sp->hist_len;
    p.tiles_per_stream = (int)((frames + sp->hop - 1) / sp->hop);
    p.dbg = nullptr;
    p.stagger = sp->ctx->cfg.stamp_thread;
    sp_fill_cfg(sp, p);
#if defined(AW_STAMPS) && AW_STAMPS
    {
        const long long nwg = (long long)ns_total * p.tiles_per_stream;
        const size_t need = (size_t)nwg * awk::kStamps;
        if (sp->dbg_cap < need) {
            if (sp->d_dbg) AW_HIP_TRY(hipFree(sp->d_dbg));
            sp->d_dbg = nullptr; sp->dbg_cap = 0;
            AW_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sp->d_dbg), need * sizeof(unsigned long long)));
            sp->ctx->device_allocs += 1;
            sp->dbg_cap = need;
        }
        AW_HIP_TRY(hipMemsetAsync(sp->d_dbg, 0, need * sizeof(unsigned long long), sp->ctx->stream));
        sp->dbg_nwg = nwg;
        p.dbg = sp->d_dbg;
    }
#endif
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (sp->profiling) { e0 = sp_get_event(sp); e1 = sp_get_event(sp); }
    long long dom_tiles = 0;
    sp->last_ola_h = 0;
    if (sp->ola_h && !sp->fused2) {
        // Overlap-add tile: every workgroup walks ONE contiguous run of blocks and pays ceil(hist / hop) warm-up blocks per run and per
        // stream start; a call needs ola_min_blocks() blocks per workgroup for that to stay a few per cent (cfg 2: 67).  Shorter
        // calls, and streams too long for 32-bit byte offsets, run the overlap-save tile on the same tables and history.
        const int hop_ola = 512 * sp->ola_h;
        // (the CALL's block count — every stream of the spatializer — not this chunk's: the staged chunks of a host-entry call run the
        // very tile the device entry runs on the whole batch, so both entries give the same bits)
        const long long blocks = (long long)sp->n_streams * ((frames + hop_ola - 1) / hop_ola);
        const long long wgs = std::max(1, sp->ctx->cfg.persistent_wgs);
        if (blocks >= (long long)ola_min_blocks(sp) * wgs && frames * sp->n_channels * 4LL <= awk::kOlaMaxBytes) {
            p.hop = hop_ola;
            p.tiles_per_stream = (int)((frames + hop_ola - 1) / hop_ola);
            AW_HIP_TRY(awk::launch_fused_ola(p, sp->ola_h, ns_total, sp->ctx->stream, e0, e1));
            sp->last_ola_h = sp->ola_h;
            sp->dominant_frames = (long long)ns_total * frames;              // one launch covers the whole call
            if (sp->profiling) sp->pending.emplace_back(e0, e1);
            return AW_OK;
        }
    }
    if (sp->fused2) AW_HIP_TRY(awk::launch_fused_ols2(p, ns_total, sp->ctx->stream, e0, e1, &dom_tiles));
    else AW_HIP_TRY(awk::launch_fused_ols(p, ns_total, sp->ctx->stream, e0, e1, &dom_tiles));
    // output frames the timed launch produced (tiles x hop, the last tile of a stream may be short)
    sp->dominant_frames = std::min<long long>(dom_tiles * (long long)sp->hop, (long long)ns_total * frames);
    if (sp->profiling) sp->pending.emplace_back(e0, e1);
    return AW_OK;
}

// Scratch of the partitioned path for calls of `frames` frames: window spectra + accumulated W of one stream chunk.
// Grow-only; aw_spatializer_reserve() sizes it ahead of time so that process never allocates.
struct PartPlan { int n_blocks; long long n_windows; size_t per_stream, per_stream_w; long long chunk; size_t need; };

// held_elems > 0 (a call inside what aw_spatializer_reserve() sized): the stream chunk is clamped to the buffer that is already
// there, so that process never reallocates — a short call would otherwise pick a larger chunk whose rounding can exceed it.
static PartPlan part_plan(const aw_spatializer *sp, int64_t frames, size_t budget_bytes, size_t held_elems = 0) {
    PartPlan pl{};
    const int N = awk::kN, B = sp->hop, P = sp->partitions;
    pl.n_blocks = (int)((frames + B - 1) / B);
    pl.n_windows = (long long)pl.n_blocks + P - 1;
    pl.per_stream = (size_t)pl.n_windows * sp->n_pairs * N;          // complex elements of window spectra
    pl.per_stream_w = (size_t)pl.n_blocks * N;                       // complex elements of accumulated W
    pl.chunk = (long long)(budget_bytes / ((pl.per_stream + pl.per_stream_w) * sizeof(awk::cf)));
    const long long held_chunk = (long long)(held_elems / (pl.per_stream + pl.per_stream_w));
    if (held_chunk >= 1 && pl.chunk > held_chunk) pl.chunk = held_chunk;
    if (pl.chunk < 1) pl.chunk = 1;
    if (pl.chunk > sp->n_streams) pl.chunk = sp->n_streams;
    if (pl.chunk > 65535) pl.chunk = 65535;                          // grid.y / grid.z of the CMAC launches
    pl.need = (pl.per_stream + pl.per_stream_w) * (size_t)pl.chunk;
    return pl;
}

// Budget: AW_SPEC_SCRATCH_MB (read at create), else min(64 GiB, 40 % of the device memory free when first needed) —
// sized for 288 GB of HBM: cfg 3 (1024 streams x 10 s, 4 pairs, 8 partitions) needs 40 GB and runs as one chunk.
static size_t part_budget(aw_spatializer *sp) {
    if (sp->scratch_budget == 0) {
        size_t free_b = 0, total_b = 0;
        size_t b = (size_t)6 << 30;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) b = std::min<size_t>((size_t)64 << 30, free_b / 5 * 2);
        sp->scratch_budget = std::max<size_t>(b, (size_t)64 << 20);
    }
    return sp->scratch_budget;
}

// The scratch is the CONTEXT's pool (runtime.hpp): grow-only, the largest need of any spatializer created on the context.
static aw_status part_ensure_scratch(aw_spatializer *sp, size_t need) {
    aw_context *c = sp->ctx;
    if (c->pool_capacity >= need) return AW_OK;
    return pool_grow(c, need);
}
// the buffer a call inside what aw_spatializer_reserve() sized may count on (its stream chunk is clamped to it: never a reallocation)
static size_t sp_held_scratch(const aw_spatializer *sp, int64_t frames) { return frames <= sp->reserved_frames ? sp->ctx->pool_capacity : 0; }

static aw_status sp_process_partitioned(aw_spatializer *sp, int s_base, int ns_total, const float *in, float *out, int64_t frames) {
    const int N = awk::kN, B = sp->hop, P = sp->partitions;
    const PartPlan pl = part_plan(sp, frames, part_budget(sp), sp_held_scratch(sp, frames));
    const int n_blocks = pl.n_blocks;
    const long long chunk = pl.chunk;
    aw_status st = part_ensure_scratch(sp, pl.need);
    if (st != AW_OK) return st;
    awk::cf *const d_spec = sp->ctx->d_pool;
    sp->dominant_frames = 0;
    for (long long s0 = 0; s0 < ns_total; s0 += chunk) {
        const int ns = (int)std::min<long long>(chunk, ns_total - s0);
        awk::TileParams p{};
        p.in = in + (size_t)s0 * frames * sp->n_channels;
        p.out = out + (size_t)s0 * frames * 2;
        p.hist = sp->d_hist[sp->hist_cur] + (size_t)(s_base + s0) * sp->hist_len * sp->n_channels;
        p.tab = sp->d_tab; p.tw1 = sp->ctx->d_tw1; p.twa = sp->ctx->d_twa; p.twb = sp->ctx->d_twb; p.zeros = sp->ctx->d_zeros;
        p.frames = frames; p.n_channels = sp->n_channels; p.n_pairs = sp->n_pairs;
        p.hop = B; p.hist_len = sp->hist_len; p.tiles_per_stream = n_blocks;
        p.spec = d_spec; p.wspec = d_spec + pl.per_stream * (size_t)chunk;
        p.partitions = P; p.n_blocks = n_blocks; p.first_valid = N - B;
        p.stagger = 0; p.dbg = nullptr;
        sp_fill_cfg(sp, p);
        p.fwd_one_pair = sp->fwd_one_pair ? 1 : 0;
        p.herm_last = (!sp->cmac_group && sp->herm_ok && (sp->n_channels & 1)) ? 1 : 0;
        // the timed unit of this path is the whole three-kernel pipeline of one stream chunk
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (sp->profiling) {
            e0 = sp_get_event(sp); e1 = sp_get_event(sp);
            AW_HIP_TRY(hipEventRecord(e0, sp->ctx->stream));
        }
        SpStageTimer tm(sp);
        awk::StageTimer *tmp = sp->profiling ? &tm : nullptr;
        AW_HIP_TRY(awk::launch_part_forward(p, ns, sp->ctx->stream, tmp));
        if (sp->cmac_group) AW_HIP_TRY(awk::launch_part_cmac(p, ns, sp->ctx->stream, tmp));
        else AW_HIP_TRY(awk::launch_part_march(p, ns, sp->ctx->stream, tmp));
        AW_HIP_TRY(awk::launch_part_inverse(p, ns, sp->ctx->stream, tmp));
        if (sp->profiling) {
            AW_HIP_TRY(hipEventRecord(e1, sp->ctx->stream));
            sp->pending.emplace_back(e0, e1);
        }
        sp->dominant_frames = (long long)ns * frames;
    }
    return AW_OK;
}


/* ---- long-window path (device/tile_lw.hpp) ---------------------------------------------------- */
// Per call: windows of N = R x 4096 frames (R = 8 RA rows, lw_choose below), hop = N - hist_len; hist_len is the history the spatializer's
// other kernel set keeps too (path 1: P x 4096 >= taps; path 0: the fused window minus its hop, any value >= taps - 1), so both kernel
// sets serve the same spatializer and the choice is free per call.
// Cost model (fabric bytes): the long-window kernels move 12 C + 24 bytes per WINDOW frame (input + rows
// written, rows read + s1/s2 written, s1/s2 read + stereo out); what they are compared with: below.
// Path-0 spatializers (HRIRs one fused window can hold): HRIR length from which the long-window kernels measure faster than
// the fused 8192- / 16384-frame tiles on long calls (tools/lw_sweep.py, 128 streams x 10 s); 1 << 30 = never.
// Round 4 (16-point rows kernel; profiles/round4_v1/lw_sweep.txt), G frames/s fused / long-window:
//   C=1  8640 taps 147 / 113;   C=2  8640: 93 / 89;   C=3  6145: 85 / 70, 8640: 68 / 69;   C=4  4320: 70 / 60, 5300: 54 / 58;
//   C=5  6145: 52 / 49, 8640: 41 / 48;   C=6  4320: 47 / 43, 5300: 38 / 43;   C=7  4320: 40.6 / 39.9, 5300: 37 / 40;   C=8  4320: 40 / 36, 5300: 32 / 36;
//   C=9  4320: 30.2 / 29.4, 6145: 19 / 29;   C=10, 12  4320: 29 / 27, 26 / 24;   C=14  3000: 25.8 / 22.3, 4320: 20.5 / 22.3;   C=16  3000: 20.1 / 19.7, 4320: 16.4 / 19.6
// The long-window kernels' rate does not depend on the HRIR length; the fused tiles' hop shrinks with it.
static int lw_fused_crossover_taps(int channels, int ola_h) {
    if (ola_h) return 5122;      // Round 6, spatializers on the overlap-add tile (blocks of 8 / 7 / 6 rows up to 4097 / 4609 / 5121 taps): faster than the
                                 // long-window kernels on every layout and length it is chosen for (ola_policy(); profiles/round6_v1/ola_sweep.txt)
    switch (channels) {          // profiles/round4_v2/policy_sweeps_final.txt and lw_sweep.txt (both fused kernels on the half-wave row transform)
        case 1: return 10800;
        case 2: return 9300;
        case 3: return 9000;
        case 4: return 5200;
        case 5: return 7500;
        case 6: return 5500;
        case 7: return 4700;
        case 8: return 5200;
        case 13: return 4000;
        case 14: return 4300;
        case 15: case 16: return 3200;     // the two-pass layouts (15 channels at 3900 taps: 18.6 against 21.7)
        default: return 4900;       // 9 - 12 channels
    }
}

// A call runs as at most two groups of windows: `n` windows of R_a rows, then one window of R_b rows for what is left (either may be
// absent).  Window lengths R = 8 RA, RA = 4 .. 16 without 11 and 13: the padded length sum(N) stays close to frames + windows x
// history for every call length — the reference's cost per frame does not depend on the stream length (uniform partitions,
// ConvolutionEngine.swift:93,232-367), and with three window lengths and one length per call this path's did (x 1.49 at 11 s).
struct LwGroup { int R; int n_windows; long long frame0, frame_end; };
struct LwCallPlan { int n_groups; LwGroup g[2]; double cost; };

static const int kLwRowChoices[] = {32, 40, 48, 56, 64, 72, 80, 96, 112, 120, 128};
static_assert(sizeof(kLwRowChoices) / sizeof(kLwRowChoices[0]) <= kLwPlanSlots, "aw_spatializer::lw_plans holds one table set per window length and never reallocates (sp_process_longwin keeps pointers into it)");

static LwCallPlan lw_choose(const aw_spatializer *sp, int64_t frames, bool for_reserve = false) {
    LwCallPlan none{};
    if (sp->lw_mode == 0 || sp->n_channels > 16) return none;
    if (sp->path == 0 && sp->lw_mode < 0 && sp->taps < lw_fused_crossover_taps(sp->n_channels, sp->ola_h)) return none;
    // calls inside what aw_spatializer_reserve() sized never build tables: only window lengths whose tables exist are candidates
    const bool existing_only = !for_reserve && frames <= sp->reserved_frames;
    auto have = [&](int R) { for (const auto &pl : sp->lw_plans) if (pl.R == R) return true; return false; };
    auto usable = [&](int R) {
        const long long N = (long long)R * awk::kLwM, hop = N - sp->hist_len;
        if (hop < N / 4) return false;                                  // the window must be mostly new frames
        if (sp->lw_mode > 0 && sp->lw_mode != R) return false;
        if (existing_only && !have(R)) return false;
        return true;
    };
    const int C = sp->n_channels;
    const double per_frame = 12.0 * C + 24.0;
    // a second group costs three more launches and their ramps: ~40 us of the device's time, in the cost's currency (bytes at ~5 TB/s)
    const double group_penalty = 2.0e8 / (double)std::max(1, sp->n_streams);
    LwCallPlan best{};
    for (int Ra : kLwRowChoices) {
        if (!usable(Ra)) continue;
        const long long Na = (long long)Ra * awk::kLwM, hopa = Na - sp->hist_len;
        {   // one group
            const long long n = (frames + hopa - 1) / hopa;
            const double cost = (double)n * (double)Na * per_frame;
            if (!best.n_groups || cost < best.cost) { best = LwCallPlan{1, {{Ra, (int)n, 0, frames}, {}}, cost}; }
        }
        if (sp->lw_mode > 0) continue;                                  // a forced window length: one group
        for (int Rb : kLwRowChoices) {
            if (Rb >= Ra || !usable(Rb)) continue;
            const long long Nb = (long long)Rb * awk::kLwM, hopb = Nb - sp->hist_len;
            const long long n = frames > hopb ? (frames - hopb + hopa - 1) / hopa : 0;
            if (n < 1) continue;                                        // (the remainder window alone: the one-group case of Rb)
            const long long rest = frames - n * hopa;
            if (rest <= 0) continue;
            const double cost = ((double)n * (double)Na + (double)Nb) * per_frame + group_penalty;
            if (cost < best.cost) best = LwCallPlan{2, {{Ra, (int)n, 0, n * hopa}, {Rb, 1, n * hopa, frames}}, cost};
        }
    }
    if (!best.n_groups || sp->lw_mode > 0) return best;
    long long row_tiles = 0;
    for (int i = 0; i < best.n_groups; ++i) row_tiles += (long long)sp->n_streams * best.g[i].n_windows * (best.g[i].R / 2);
    if (row_tiles < 32) return none;     // (measured down to ONE stream x 10 s, 64 row tiles: 7 channels x 32768 taps 8.2 against 3.9 G frames/s partitioned)
    // path 0: the crossovers above were measured on batches that fill the chip; a small batch stays on the fused tiles, which serve
    // a single stream well (cfg 1: 9.7 G frames/s), until its row tiles cover the CUs twice
    if (sp->path == 0 && row_tiles < 2LL * sp->ctx->cfg.cus) return none;
    double other_cost;
    if (sp->path == 0) {
        // past the measured crossover (above) the long call only has to fill its windows to 80 %
        other_cost = 1.25 * (double)frames * per_frame;
    } else {
        // Partitioned kernels, in the same currency (fabric-byte equivalents at the rate both kernel sets reach): a long call costs
        // ~110 + 13 C bytes per output frame up to eight channels (measured 20-23 G frames/s for C = 7, 29 for C = 2; the one-pair
        // forward kernels of wider layouts ~30 C), 86 % of it per input WINDOW (forward transform + marched CMAC) — and a call of
        // n blocks transforms n + P - 1 windows, which is what makes short calls expensive there — and 14 % per output block.
        // tools/lw_calls_sweep.py (G frames/s, partitioned / long-window, 128 streams x 7 channels x 32768 taps): 16 384 frames
        // 6.2 / 5.9, 32 768: 9.9 / 10.9, 49 152: 11.5 / 15.6, 65 536: 13.1 / 20.0, 131 072: 15.8 / 21.0, 480 000: 20.0 / 35.8.
        const double b_part = C <= 8 ? 110.0 + 13.0 * C : 30.0 * C;
        const long long blocks = (frames + sp->hop - 1) / sp->hop;
        other_cost = (double)sp->hop * b_part * (0.86 * (double)(blocks + sp->partitions - 1) + 0.14 * (double)blocks);
    }
    return best.cost < other_cost ? best : none;
}

static aw_status lw_get_plan(aw_spatializer *sp, int R, const aw_spatializer::LwPlan **out) {
    for (const auto &pl : sp->lw_plans)
        if (pl.R == R) { *out = &pl; return AW_OK; }
    awh::LwTables t;
    const int form = sp->ctx->cfg.lw_rows_form;          // which rows kernel the tables are laid out for (read once per context)
    // The filter tables of the 16-point rows kernel are computed ON THE DEVICE (device/prep_kernels.hip: float64, the analogue of the
    // partition FFTs of ConvolutionEngine.init, ConvolutionEngine.swift:141-175); the host builds only the small twiddle tables then.
    // AW_LW_TABLES=host (read at context creation) or the 8-point kernel forms: the float64 host builder computes everything.
    const bool on_gpu = form == 16 && sp->ctx->cfg.lw_tables_on_gpu != 0;
    const auto t_build = std::chrono::steady_clock::now();
    if (!awh::build_lw_tables(sp->lw_tracks.data(), sp->lw_n_tracks, sp->taps, sp->n_channels, sp->lw_left.data(), sp->lw_right.data(), R, t, form, /*filters=*/!on_gpu))
        return fail(AW_ERR_OUT_OF_MEMORY, "long-window tables: host memory");      // (the caller falls back to the kernels that have always served the spatializer)
    auto t_up = std::chrono::steady_clock::now();
    sp->reserve_tables_us += std::chrono::duration_cast<std::chrono::microseconds>(t_up - t_build).count();
    aw_spatializer::LwPlan pl;
    pl.R = R;
    auto up = [&](const void *src, size_t bytes, void **d) -> hipError_t {
        hipError_t r = hipMalloc(d, bytes);
        if (r != hipSuccess) { *d = nullptr; return r; }
        sp->ctx->device_allocs += 1; sp->ctx->sync_copies += 1;
        r = hipMemcpy(*d, src, bytes, hipMemcpyHostToDevice);
        if (r != hipSuccess) { (void)hipFree(*d); *d = nullptr; }        // a buffer that was never written is not left behind as if it had been
        return r;
    };
    hipError_t e = hipSuccess;
    if (on_gpu) {
        const int n_pairs = (sp->n_channels + 1) / 2;
        if (!sp->d_lw_tracks || !sp->d_lw_left || !sp->d_lw_right) {     // the impulse responses and the channel map, once per spatializer
            // all three or none: a failure part way (out of memory) frees what was uploaded, so that the next long call uploads again
            // instead of handing the prep kernel a null or never-written channel map (round-5 advice)
            for (void **d : {reinterpret_cast<void **>(&sp->d_lw_tracks), reinterpret_cast<void **>(&sp->d_lw_left), reinterpret_cast<void **>(&sp->d_lw_right)})
                if (*d) { (void)hipFree(*d); *d = nullptr; }
            e = up(sp->lw_tracks.data(), sp->lw_tracks.size() * sizeof(float), reinterpret_cast<void **>(&sp->d_lw_tracks));
            if (e == hipSuccess) e = up(sp->lw_left.data(), sp->lw_left.size() * sizeof(int32_t), reinterpret_cast<void **>(&sp->d_lw_left));
            if (e == hipSuccess) e = up(sp->lw_right.data(), sp->lw_right.size() * sizeof(int32_t), reinterpret_cast<void **>(&sp->d_lw_right));
            if (e != hipSuccess)
                for (void **d : {reinterpret_cast<void **>(&sp->d_lw_tracks), reinterpret_cast<void **>(&sp->d_lw_left), reinterpret_cast<void **>(&sp->d_lw_right)})
                    if (*d) { (void)hipFree(*d); *d = nullptr; }
        }
        void *d_tmp = nullptr;
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&pl.d_tab16), (size_t)(R / 2) * n_pairs * 2 * awk::kLwM * sizeof(awk::LwTab2));
        if (e == hipSuccess) e = hipMalloc(&d_tmp, awk::lw_prep_scratch_bytes(sp->n_channels, R));
        if (e == hipSuccess) {
            sp->ctx->device_allocs += 2;
            e = awk::launch_lw_prep(sp->d_lw_tracks, sp->lw_n_tracks, sp->taps, sp->n_channels, sp->d_lw_left, sp->d_lw_right, R, d_tmp, pl.d_tab16, sp->ctx->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(sp->ctx->stream);
        if (d_tmp) (void)hipFree(d_tmp);
        const auto t_gpu = std::chrono::steady_clock::now();
        sp->reserve_tables_us += std::chrono::duration_cast<std::chrono::microseconds>(t_gpu - t_up).count();
        t_up = t_gpu;
        if (e == hipSuccess) e = up(t.tw2.data(), t.tw2.size() * sizeof(awk::cf), reinterpret_cast<void **>(&pl.d_tw2));
    } else if (form == 16) {
        e = up(t.tab16.data(), t.tab16.size() * sizeof(awk::LwTab2), reinterpret_cast<void **>(&pl.d_tab16));
        if (e == hipSuccess) e = up(t.tw2.data(), t.tw2.size() * sizeof(awk::cf), reinterpret_cast<void **>(&pl.d_tw2));
    } else {
        e = up(t.tab.data(), t.tab.size() * sizeof(awk::LwTab), reinterpret_cast<void **>(&pl.d_tab));
    }
    if (e == hipSuccess) e = up(t.coarse.data(), t.coarse.size() * sizeof(awk::cf), reinterpret_cast<void **>(&pl.d_coarse));
    if (e == hipSuccess) e = up(t.fine.data(), t.fine.size() * sizeof(awk::cf), reinterpret_cast<void **>(&pl.d_fine));
    if (e == hipSuccess) e = up(t.step.data(), t.step.size() * sizeof(awk::cf), reinterpret_cast<void **>(&pl.d_step));
    if (e == hipSuccess) e = up(t.tw_r.data(), t.tw_r.size() * sizeof(awk::cf), reinterpret_cast<void **>(&pl.d_tw_r));
    if (e == hipSuccess) e = up(t.tw1m.data(), t.tw1m.size() * sizeof(awk::cf), reinterpret_cast<void **>(&pl.d_tw1m));
    if (e != hipSuccess) {                  // nothing half-built stays behind: the next call tries again (or takes the partitioned kernels)
        for (void *d : {(void *)pl.d_tab, (void *)pl.d_tab16, (void *)pl.d_tw2, (void *)pl.d_coarse, (void *)pl.d_fine, (void *)pl.d_step, (void *)pl.d_tw_r, (void *)pl.d_tw1m})
            if (d) (void)hipFree(d);
        return awr::hip_fail(e, "long-window tables");
    }
    sp->reserve_upload_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_up).count();
    sp->lw_plans.push_back(pl);
    *out = &sp->lw_plans.back();
    return AW_OK;
}

// scratch of one stream chunk: rows of every (stream, window) + s1/s2
struct LwScratch { long long n_windows; size_t per_stream; long long chunk; size_t need; long long spec_per_sw; };
static LwScratch lw_scratch(const aw_spatializer *sp, int R, long long n_windows, size_t budget_bytes, size_t held_elems) {
    LwScratch r{};
    const long long N = (long long)R * awk::kLwM;
    const int real_last = sp->n_channels & 1;
    r.n_windows = n_windows;
    const int n_pairs = (sp->n_channels + 1) / 2;                         // (a fused2 spatializer's n_pairs counts pseudo-pairs)
    r.spec_per_sw = (long long)(n_pairs - real_last) * N + (real_last ? N / 2 : 0);
    r.per_stream = (size_t)r.n_windows * (size_t)(r.spec_per_sw + N);
    r.chunk = (long long)(budget_bytes / sizeof(awk::cf) / r.per_stream);
    const long long held_chunk = (long long)(held_elems / r.per_stream);       // see part_plan
    if (held_chunk >= 1 && r.chunk > held_chunk) r.chunk = held_chunk;
    if (r.chunk < 1) r.chunk = 1;
    if (r.chunk > sp->n_streams) r.chunk = sp->n_streams;
    r.need = r.per_stream * (size_t)r.chunk;
    return r;
}
// the stream chunk and scratch of a whole call plan: every group of the plan runs on the same chunk of streams
static LwScratch lw_plan_scratch(const aw_spatializer *sp, const LwCallPlan &plan, size_t budget_bytes, size_t held_elems, LwScratch (&per_group)[2]) {
    LwScratch all{};
    all.chunk = sp->n_streams;
    for (int i = 0; i < plan.n_groups; ++i) {
        per_group[i] = lw_scratch(sp, plan.g[i].R, plan.g[i].n_windows, budget_bytes, held_elems);
        all.chunk = std::min(all.chunk, per_group[i].chunk);
    }
    for (int i = 0; i < plan.n_groups; ++i) all.need = std::max(all.need, per_group[i].per_stream * (size_t)all.chunk);
    return all;
}

static aw_status sp_process_longwin(aw_spatializer *sp, const LwCallPlan &call, int s_base, int ns_total, const float *in, float *out, int64_t frames) {
    const aw_spatializer::LwPlan *tables[2] = {nullptr, nullptr};
    for (int i = 0; i < call.n_groups; ++i) {
        aw_status st = lw_get_plan(sp, call.g[i].R, &tables[i]);
        if (st != AW_OK) return st;
    }
    LwScratch per_group[2];
    const LwScratch sc = lw_plan_scratch(sp, call, part_budget(sp), sp_held_scratch(sp, frames), per_group);
    aw_status st = part_ensure_scratch(sp, sc.need);
    if (st != AW_OK) return st;
    awk::cf *const d_spec = sp->ctx->d_pool;
    sp->dominant_frames = 0;
    for (long long s0 = 0; s0 < ns_total; s0 += sc.chunk) {
        const int ns = (int)std::min<long long>(sc.chunk, ns_total - s0);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (sp->profiling) {
            e0 = sp_get_event(sp); e1 = sp_get_event(sp);
            AW_HIP_TRY(hipEventRecord(e0, sp->ctx->stream));
        }
        SpStageTimer tm(sp);
        awk::StageTimer *tmp = sp->profiling ? &tm : nullptr;
        if (sp->n_channels > 8)          // the wide split kernel reads the last frame of the last stream from a padded copy (allocated at create)
            AW_HIP_TRY(hipMemcpyAsync(sp->d_tail, in + ((size_t)(s0 + ns) * frames - 1) * sp->n_channels, sp->n_channels * sizeof(float),
                                      hipMemcpyDefault, sp->ctx->stream));          // (`in` is device memory, or the page-locked staging of a one-stream call)
        for (int gi = 0; gi < call.n_groups; ++gi) {
            const LwGroup &g = call.g[gi];
            const aw_spatializer::LwPlan *plan = tables[gi];
            const long long N = (long long)g.R * awk::kLwM;
            awk::LwParams p{};
            p.in = in + (size_t)s0 * frames * sp->n_channels;
            p.out = out + (size_t)s0 * frames * 2;
            p.hist = sp->d_hist[sp->hist_cur] + (size_t)(s_base + s0) * sp->hist_len * sp->n_channels;
            // the tail carry rides along in the split kernel of the LAST group: its last window spans the last hist_len frames of the call
            p.hist_out = gi == call.n_groups - 1 ? sp->d_hist[sp->hist_cur ^ 1] + (size_t)(s_base + s0) * sp->hist_len * sp->n_channels : nullptr;
            p.zeros = sp->ctx->d_zeros;
            p.frames = frames; p.frame0 = g.frame0; p.frame_end = g.frame_end;
            p.n_channels = sp->n_channels; p.n_pairs = (sp->n_channels + 1) / 2; p.real_last = sp->n_channels & 1;
            p.hist_len = sp->hist_len; p.hop = (int)(N - sp->hist_len); p.n_windows = g.n_windows;
            p.R = g.R; p.N = (int)N;
            p.spec = d_spec; p.spec_per_sw = per_group[gi].spec_per_sw;
            p.wrows = d_spec + (size_t)sc.chunk * g.n_windows * per_group[gi].spec_per_sw;
            p.tab = plan->d_tab; p.tw_coarse = plan->d_coarse; p.tw_fine = plan->d_fine; p.tw_step = plan->d_step; p.tw_r = plan->d_tw_r; p.tw1m = plan->d_tw1m;
            p.twa = sp->ctx->d_twa; p.twb = sp->ctx->d_twb;
            p.persistent_wgs = sp->ctx->cfg.persistent_wgs;
            p.rows_pairs_per_batch = sp->ctx->cfg.lw_rows_pb;
            p.rows_form = plan->d_tab16 ? 16 : 8; p.tab16 = plan->d_tab16; p.tw2 = plan->d_tw2; p.rows16_wgs = sp->ctx->cfg.lw_rows16_wgs;
            p.n_streams = ns;
            p.tail = sp->n_channels > 8 ? sp->d_tail : nullptr;
            AW_HIP_TRY(awk::launch_lw_split(p, ns, sp->ctx->stream, tmp));
            AW_HIP_TRY(awk::launch_lw_rows(p, ns, sp->ctx->stream, tmp));
            AW_HIP_TRY(awk::launch_lw_merge(p, ns, sp->ctx->stream, tmp));
        }
        if (sp->profiling) {
            AW_HIP_TRY(hipEventRecord(e1, sp->ctx->stream));
            sp->pending.emplace_back(e0, e1);
        }
        sp->dominant_frames = (long long)ns * frames;
    }
    return AW_OK;
}

static aw_status sp_grow(aw_spatializer *sp, float **buf, size_t *cap, size_t need) {
    if (*cap >= need) return AW_OK;
    if (*buf) AW_HIP_TRY(hipFree(*buf));
    *buf = nullptr; *cap = 0;
    AW_HIP_TRY(hipMalloc(reinterpret_cast<void **>(buf), need * sizeof(float)));
    sp->ctx->device_allocs += 1;
    *cap = need;
    return AW_OK;
}

static aw_status sp_grow_pinned(aw_spatializer *sp, float **buf, size_t *cap, size_t need) {
    if (*cap >= need) return AW_OK;
    if (*buf) AW_HIP_TRY(hipHostFree(*buf));
    *buf = nullptr; *cap = 0;
    AW_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(buf), need * sizeof(float), hipHostMallocDefault));
    sp->ctx->device_allocs += 1;
    *cap = need;
    return AW_OK;
}

// Callback-sized calls of a one-stream spatializer take the zero-copy staging (runtime.hpp) when aw_spatializer_reserve has made it;
// longer calls move by DMA (a fused tile reads every input line about twice: not over PCIe).
static constexpr int64_t kZeroCopyFrames = 16384;
static bool sp_zero_copy(const aw_spatializer *sp, int64_t frames) {
    return sp->n_streams == 1 && frames <= kZeroCopyFrames && sp->pin_in_cap >= (size_t)frames * sp->n_channels && sp->pin_out_cap >= (size_t)frames * 2;
}

// The longest call of at most max_frames frames that the policy (lw_choose with every window length available) leaves to the partitioned
// kernels (0: none; a path-0 spatializer never runs them).  A reserved spatializer only has the window lengths of its reserve() plan, so a
// call well short of max_frames may still come to the partitioned kernels beyond this length: it then runs in stream chunks clamped to the
// pool (part_plan), correct and a little slower — what reserve() buys is no allocation, and the full rate at the length it was asked for.  Scanned in steps of whole blocks; with more than 2048
// candidate lengths the scan strides and answers one stride longer, i.e. never too short.
static int64_t part_longest_call(const aw_spatializer *sp, int64_t max_frames) {
    if (sp->path != 1) return 0;
    const int64_t B = sp->hop, n_blocks = (max_frames + B - 1) / B;
    const int64_t stride = std::max<int64_t>(1, n_blocks / 2048);
    int64_t longest = 0;
    for (int64_t nb = 1;; nb += stride) {
        const int64_t f = std::min<int64_t>(std::min(nb, n_blocks) * B, max_frames);
        if (!lw_choose(sp, f, true).n_groups) longest = std::min<int64_t>(f + (stride - 1) * B, max_frames);
        if (nb >= n_blocks) break;
    }
    return longest;
}

// Sizes every grow-only device buffer for calls of up to max_frames frames, so that the process entries never
// allocate afterwards ("process must not allocate"; creation may block).
aw_status aw_spatializer_reserve(aw_spatializer *sp, int64_t max_frames) try {
    if (!sp) return fail(AW_ERR_INVALID_ARGUMENT, "sp is NULL");
    if (max_frames <= 0) return fail(AW_ERR_INVALID_ARGUMENT, "max_frames must be positive");
    AW_HIP_TRY(hipSetDevice(sp->ctx->device));
    std::lock_guard<std::mutex> lk(sp->ctx->launch_mu);
    sp->reserve_tables_us = sp->reserve_upload_us = sp->reserve_scratch_us = 0;
    const LwCallPlan lw_res = lw_choose(sp, max_frames, true);
    if (sp->path == 1 || lw_res.n_groups) {
        for (int i = 0; i < lw_res.n_groups; ++i) {
            const aw_spatializer::LwPlan *plan = nullptr;
            aw_status st = lw_get_plan(sp, lw_res.g[i].R, &plan);
            if (st != AW_OK) return st;
        }
        // From here on a call of up to max_frames frames chooses only among the window lengths whose tables exist (lw_choose).
        const int64_t reserved_before = sp->reserved_frames;
        sp->reserved_frames = std::max<int64_t>(sp->reserved_frames, max_frames);
        size_t need = 0;
        if (lw_res.n_groups) {
            LwScratch per_group[2];
            need = lw_plan_scratch(sp, lw_res, part_budget(sp), 0, per_group).need;
        }
        if (sp->path == 1) {
            // Shorter calls may still run on the partitioned kernels: size for the longest call lw_choose() leaves to them (cfg 3: 45 056
            // of 480 000 frames, 4 GB — round 4 sized for max_frames on BOTH kernel sets: 41.5 GB where the long-window kernels need 19),
            // and for ONE stream of max_frames whatever the policy says (a call inside the reserved size clamps its stream chunk to the
            // buffer that is there, part_plan / lw_scratch: it never reallocates).
            const int64_t longest = part_longest_call(sp, sp->reserved_frames);
            if (longest > 0) need = std::max(need, part_plan(sp, longest, part_budget(sp)).need);
            const PartPlan one = part_plan(sp, sp->reserved_frames, part_budget(sp));
            need = std::max(need, one.per_stream + one.per_stream_w);
        }
        const auto t0 = std::chrono::steady_clock::now();
        aw_status st = part_ensure_scratch(sp, need);
        if (st == AW_OK) {
            const hipError_t es = hipStreamSynchronize(sp->ctx->stream);
            if (es != hipSuccess) st = awr::hip_fail(es, "hipStreamSynchronize (reserve)");
        }
        sp->reserve_scratch_us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
        if (st != AW_OK) { sp->reserved_frames = reserved_before; return st; }
        sp->reserved_pool = std::max(sp->reserved_pool, need);
    }
    if (sp->n_streams == 1) {       // plug-in shaped use (host / planar entries): their staging buffers too
        const size_t n = (size_t)max_frames * std::max(sp->n_channels, 4);
        aw_status st = sp_grow(sp, &sp->d_stage_in, &sp->stage_in_cap, n);
        if (st == AW_OK) st = sp_grow(sp, &sp->d_stage_out, &sp->stage_out_cap, (size_t)max_frames * 4);
        // callback-sized calls: page-locked staging the kernels address directly
        const size_t zf = (size_t)std::min<int64_t>(max_frames, kZeroCopyFrames);
        if (st == AW_OK) st = sp_grow_pinned(sp, &sp->h_pin_in, &sp->pin_in_cap, zf * sp->n_channels);
        if (st == AW_OK) st = sp_grow_pinned(sp, &sp->h_pin_out, &sp->pin_out_cap, zf * 2);
        if (st != AW_OK) return st;
    }
    sp->reserved_frames = std::max<int64_t>(sp->reserved_frames, max_frames);
    return AW_OK;
} AW_NOEXCEPT_TAIL

// One call's launches for the streams [s_base, s_base + ns) (`in` / `out` at stream s_base), history carry included; the caller holds
// the context's launch lock and flips hist_cur once every stream of the call has been through here.
static aw_status sp_run_streams(aw_spatializer *sp, const LwCallPlan &lw, int s_base, int ns, const float *in, float *out, int64_t frames) {
    aw_status st = AW_OK;
    bool lw_ran = false;
    if (lw.n_groups && sp->last_lw_R) {
        st = sp_process_longwin(sp, lw, s_base, ns, in, out, frames);
        lw_ran = st == AW_OK;
        if (st == AW_ERR_OUT_OF_MEMORY) {
            // the long-window kernels are an optimisation: without memory for their tables or scratch the call takes the kernels
            // that have always served this spatializer (nothing has been launched yet: allocation comes first)
            (void)hipGetLastError();
            sp->last_lw_R = 0; sp->last_lw_R2 = 0;
            st = AW_OK;
        }
    }
    if (st == AW_OK && !lw_ran) st = sp->path == 0 ? sp_process_fused(sp, s_base, ns, in, out, frames) : sp_process_partitioned(sp, s_base, ns, in, out, frames);
    if (st != AW_OK) return st;
    // carry the convolution tail: next call's history = last hist_len frames of (history ++ input)
    if (!lw_ran) {    // (the long-window split kernel has written it on the way)
        const size_t off = (size_t)s_base * sp->hist_len * sp->n_channels;
        float *h_old = sp->d_hist[sp->hist_cur] + off, *h_new = sp->d_hist[sp->hist_cur ^ 1] + off;
        SpStageTimer tm(sp);
        if (sp->profiling) tm.begin();
        AW_HIP_TRY(awk::launch_hist_update(in, h_old, h_new, frames, sp->n_channels, sp->hist_len, ns, sp->ctx->stream));
        if (sp->profiling) tm.end("aw_hist_update_kernel");
    }
    return AW_OK;
}

static LwCallPlan sp_begin_call(aw_spatializer *sp, int64_t frames) {
    const LwCallPlan lw = lw_choose(sp, frames);
    sp->last_lw_R = lw.n_groups ? lw.g[0].R : 0;
    sp->last_lw_R2 = lw.n_groups > 1 ? lw.g[1].R : 0;
    return lw;
}

aw_status aw_spatializer_process(aw_spatializer *sp, const float *in, float *out, int64_t frames) try {
    if (!sp || !in || !out) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
    if (frames <= 0) return frames == 0 ? AW_OK : fail(AW_ERR_INVALID_ARGUMENT, "frames must be >= 0");
    AW_HIP_TRY(hipSetDevice(sp->ctx->device));
    std::lock_guard<std::mutex> lk(sp->ctx->launch_mu);       // this call's launches stay together on the stream (the scratch pool is the context's)
    const LwCallPlan lw = sp_begin_call(sp, frames);
    aw_status st = sp_run_streams(sp, lw, 0, sp->n_streams, in, out, frames);
    if (st != AW_OK) return st;
    sp->hist_cur ^= 1;
    return AW_OK;
} AW_NOEXCEPT_TAIL

/* ---- host entry ------------------------------------------------------------------------------------
 * The reference's callers hand over host buffers (AudioPipeline.swift:3-11: the four planar pointers of a render callback); an offline
 * batch host does too.  A multi-stream batch crosses PCIe in CHUNKS OF STREAMS (streams are independent: state is per stream), double
 * buffered on three HIP streams: H2D of chunk k+1 || kernels of chunk k || D2H of chunk k-1.  Page-locked caller buffers
 * (aw_host_alloc_pinned, hipHostMalloc, hipHostRegister) are read and written by the DMA engines directly; pageable ones are bounced
 * through page-locked chunks by the context's copy threads (below).  aw_spatializer_reserve_host() sizes the device-side chunk buffers
 * and the bounce chunks ahead of time; without it they grow on the first call. */
static int64_t host_chunk_streams(const aw_spatializer *sp, int64_t frames) {
    const size_t chunk_bytes = (size_t)sp->ctx->cfg.host_chunk_mb << 20;        // input bytes per staged chunk (LaunchCfg: read once per context)
    const size_t per_stream = (size_t)frames * sp->n_channels * sizeof(float);
    if (sp->n_streams < 4 || per_stream * sp->n_streams < 2 * chunk_bytes) return 0;           // small batches: one piece, serial (plug-in shaped calls)
    int64_t cs = (int64_t)std::max<size_t>(1, chunk_bytes / per_stream);
    cs = std::max<int64_t>(cs, 2);
    return std::min<int64_t>(cs, (sp->n_streams + 1) / 2);
}

// Page-locked (hipHostMalloc / hipHostRegister) memory moves by DMA as it is; anything else is bounced.
static bool host_ptr_is_pinned(const void *p) {
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }     // (plain malloc'ed memory: invalid value)
    return a.type == hipMemoryTypeHost;
}

static aw_status host_pipeline_objects(aw_context *c) {
    if (c->s_h2d && c->s_d2h && c->ev_d2h[1]) return AW_OK;           // (ev_d2h[1] is the last object made below)
    if (!c->copy_pool) {
        const unsigned hw = std::thread::hardware_concurrency();
        c->copy_pool = new (std::nothrow) aw_context::CopyPool((int)std::min(12u, std::max(1u, hw / 4)));
        if (!c->copy_pool) return fail(AW_ERR_OUT_OF_MEMORY, "copy threads");
        // the output direction moves a quarter or less of the input's bytes (8 against 4 C per frame): a few threads of its own
        c->copy_pool_out = new (std::nothrow) aw_context::CopyPool((int)std::min(4u, std::max(1u, hw / 8)));
        if (!c->copy_pool_out) return fail(AW_ERR_OUT_OF_MEMORY, "copy threads");
    }
    // (a failure half-way leaves what exists in place: the next call makes the rest, aw_context_destroy frees whatever is there)
    if (!c->s_h2d) AW_HIP_TRY(hipStreamCreateWithFlags(&c->s_h2d, hipStreamNonBlocking));
    if (!c->s_d2h) AW_HIP_TRY(hipStreamCreateWithFlags(&c->s_d2h, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        if (!c->ev_h2d[i]) AW_HIP_TRY(hipEventCreateWithFlags(&c->ev_h2d[i], hipEventDisableTiming));
        if (!c->ev_run[i]) AW_HIP_TRY(hipEventCreateWithFlags(&c->ev_run[i], hipEventDisableTiming));
        if (!c->ev_d2h[i]) AW_HIP_TRY(hipEventCreateWithFlags(&c->ev_d2h[i], hipEventDisableTiming));
    }
    return AW_OK;
}

static aw_status sp_grow_pinned(aw_spatializer *sp, float **buf, size_t *cap, size_t need);

// bounce: also the page-locked chunks that pageable caller buffers go through (aw_spatializer_reserve_host makes them; a call on pageable
// memory that was not reserved for makes them on its first use)
static aw_status host_stage_buffers(aw_spatializer *sp, int64_t frames, int64_t cs, bool bounce_in = false, bool bounce_out = false) {
    const size_t streams = cs > 0 ? 2 * (size_t)cs : (size_t)sp->n_streams;           // two chunks in flight each way, or the whole batch
    aw_status st = sp_grow(sp, &sp->d_stage_in, &sp->stage_in_cap, streams * frames * sp->n_channels);
    if (st == AW_OK) st = sp_grow(sp, &sp->d_stage_out, &sp->stage_out_cap, streams * frames * 2);
    if (st == AW_OK && cs > 0 && bounce_in) st = sp_grow_pinned(sp, &sp->h_bounce_in, &sp->bounce_in_cap, streams * frames * sp->n_channels);
    if (st == AW_OK && cs > 0 && bounce_out) st = sp_grow_pinned(sp, &sp->h_bounce_out, &sp->bounce_out_cap, streams * frames * 2);
    return st;
}

aw_status aw_spatializer_reserve_host(aw_spatializer *sp, int64_t max_frames) try {
    aw_status st = aw_spatializer_reserve(sp, max_frames);
    if (st != AW_OK) return st;
    std::lock_guard<std::mutex> lk(sp->ctx->launch_mu);
    const int64_t cs = host_chunk_streams(sp, max_frames);
    if (cs > 0) { st = host_pipeline_objects(sp->ctx); if (st != AW_OK) return st; }
    st = host_stage_buffers(sp, max_frames, cs, /*bounce_in=*/true, /*bounce_out=*/true);
    if (st == AW_OK) { sp->host_chunk_streams = cs; sp->host_chunk_reserved = cs; sp->host_reserved_frames = std::max(sp->host_reserved_frames, max_frames); }
    return st;
} AW_NOEXCEPT_TAIL

aw_status aw_spatializer_process_host(aw_spatializer *sp, const float *in, float *out, int64_t frames) try {
    if (!sp || !in || !out) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
    if (frames <= 0) return frames == 0 ? AW_OK : fail(AW_ERR_INVALID_ARGUMENT, "frames must be >= 0");
    AW_HIP_TRY(hipSetDevice(sp->ctx->device));
    aw_context *c = sp->ctx;
    std::lock_guard<std::mutex> lk(c->launch_mu);
    const size_t in_ps = (size_t)frames * sp->n_channels, out_ps = (size_t)frames * 2;          // floats per stream
    if (sp_zero_copy(sp, frames)) {      // one stream, a callback's worth of frames: the kernels read and write page-locked host memory themselves
        std::memcpy(sp->h_pin_in, in, in_ps * sizeof(float));
        const LwCallPlan lw0 = sp_begin_call(sp, frames);
        aw_status st0 = sp_run_streams(sp, lw0, 0, 1, sp->h_pin_in, sp->h_pin_out, frames);
        if (st0 != AW_OK) return st0;
        sp->hist_cur ^= 1;
        AW_HIP_TRY(hipStreamSynchronize(c->stream));
        std::memcpy(out, sp->h_pin_out, out_ps * sizeof(float));
        sp->host_chunk_streams = 0;
        return AW_OK;
    }
    int64_t cs = host_chunk_streams(sp, frames);
    // a reserved spatializer keeps the chunking its buffers were sized for (never a reallocation on this path)
    if (frames <= sp->host_reserved_frames) cs = sp->host_chunk_reserved > 0 ? (cs > 0 ? std::min(cs, sp->host_chunk_reserved) : sp->host_chunk_reserved) : 0;
    const bool page_in = cs > 0 && !host_ptr_is_pinned(in), page_out = cs > 0 && !host_ptr_is_pinned(out);
    aw_status st = host_stage_buffers(sp, frames, cs, page_in, page_out);
    if (st == AW_OK && cs > 0) st = host_pipeline_objects(c);
    if (st != AW_OK) return st;
    sp->host_chunk_streams = cs;
    const LwCallPlan lw = sp_begin_call(sp, frames);
    if (cs == 0) {                       // one piece: H2D -> kernels -> D2H on the context's stream
        AW_HIP_TRY(hipMemcpyAsync(sp->d_stage_in, in, in_ps * sp->n_streams * sizeof(float), hipMemcpyHostToDevice, c->stream));
        st = sp_run_streams(sp, lw, 0, sp->n_streams, sp->d_stage_in, sp->d_stage_out, frames);
        if (st != AW_OK) return st;
        sp->hist_cur ^= 1;
        AW_HIP_TRY(hipMemcpyAsync(out, sp->d_stage_out, out_ps * sp->n_streams * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        AW_HIP_TRY(hipStreamSynchronize(c->stream));
        return AW_OK;
    }
    // whatever the context's stream still holds (an earlier device-buffer call of this spatializer) comes first
    AW_HIP_TRY(hipEventRecord(c->ev_run[0], c->stream));
    AW_HIP_TRY(hipStreamWaitEvent(c->s_h2d, c->ev_run[0], 0));
    // Pageable caller memory is bounced through page-locked chunks by the context's copy threads: the copy INTO slot k & 1 runs while the
    // DMA engines move chunk k-1, the copy OUT of chunk k-1 while the kernels of chunk k run (round 5; before: hipMemcpyAsync on the
    // pageable pointers themselves, which the HIP runtime stages on one thread).
    int k = 0;
    int64_t prev_s0 = -1; int prev_ns = 0;
    hipError_t he = hipSuccess;
    // chunk k-1's output: wait for its D2H, then hand it to the output copy threads (they work while this thread copies chunk k+1 IN;
    // the bounce slot is waited for before a later D2H is queued into it, and once more at the end)
    auto drain_prev = [&](int slot_prev) {
        if (!page_out || prev_s0 < 0 || he != hipSuccess) return;
        he = hipEventSynchronize(c->ev_d2h[slot_prev]);
        if (he != hipSuccess) return;
        float *d = out + (size_t)prev_s0 * out_ps; const float *b = sp->h_bounce_out + (size_t)slot_prev * cs * out_ps;
        const size_t n = (size_t)prev_ns * out_ps * sizeof(float);
        if (c->cfg.host_out_async) c->copy_pool_out->start(d, b, n);
        else c->copy_pool->copy(d, b, n);            // AW_HOST_OUT_ASYNC=0 (A/B): round 5's form, on the driver thread
    };
    for (int64_t s0 = 0; s0 < sp->n_streams && he == hipSuccess; s0 += cs, ++k) {
        const int ns = (int)std::min<int64_t>(cs, sp->n_streams - s0), slot = k & 1;
        float *d_in = sp->d_stage_in + (size_t)slot * cs * in_ps, *d_out = sp->d_stage_out + (size_t)slot * cs * out_ps;
        const float *src = in + (size_t)s0 * in_ps;
        if (page_in) {
            float *b = sp->h_bounce_in + (size_t)slot * cs * in_ps;
            if (k >= 2) he = hipEventSynchronize(c->ev_h2d[slot]);                              // chunk k-2's H2D has read this bounce slot
            if (he != hipSuccess) break;
            c->copy_pool->copy(b, src, (size_t)ns * in_ps * sizeof(float));
            src = b;
        }
        if (k >= 2) he = hipStreamWaitEvent(c->s_h2d, c->ev_run[slot], 0);                      // chunk k-2's kernels have read this device slot
        if (he == hipSuccess) he = hipMemcpyAsync(d_in, src, (size_t)ns * in_ps * sizeof(float), hipMemcpyHostToDevice, c->s_h2d);
        if (he == hipSuccess) he = hipEventRecord(c->ev_h2d[slot], c->s_h2d);
        if (he == hipSuccess) he = hipStreamWaitEvent(c->stream, c->ev_h2d[slot], 0);
        if (he == hipSuccess && k >= 2) he = hipStreamWaitEvent(c->stream, c->ev_d2h[slot], 0); // chunk k-2's output has left this device slot
        if (he != hipSuccess) break;
        st = sp_run_streams(sp, lw, (int)s0, ns, d_in, d_out, frames);
        if (st != AW_OK) break;
        he = hipEventRecord(c->ev_run[slot], c->stream);
        if (he == hipSuccess) he = hipStreamWaitEvent(c->s_d2h, c->ev_run[slot], 0);
        if (page_out) c->copy_pool_out->wait();          // chunk k-2's copy-out has left this bounce slot (it had a whole chunk's time)
        float *dst = page_out ? sp->h_bounce_out + (size_t)slot * cs * out_ps : out + (size_t)s0 * out_ps;
        if (he == hipSuccess) he = hipMemcpyAsync(dst, d_out, (size_t)ns * out_ps * sizeof(float), hipMemcpyDeviceToHost, c->s_d2h);
        if (he == hipSuccess) he = hipEventRecord(c->ev_d2h[slot], c->s_d2h);
        drain_prev(slot ^ 1);
        prev_s0 = s0; prev_ns = ns;
    }
    const hipError_t e1 = hipStreamSynchronize(c->s_h2d), e2 = hipStreamSynchronize(c->stream), e3 = hipStreamSynchronize(c->s_d2h);
    if (page_out && (st != AW_OK || he != hipSuccess)) c->copy_pool_out->wait();       // no copy thread outlives the call
    if (st != AW_OK) return st;          // (a failed chunk: the streams are drained, the history has not been flipped)
    AW_HIP_TRY(he); AW_HIP_TRY(e1); AW_HIP_TRY(e2); AW_HIP_TRY(e3);
    drain_prev((k - 1) & 1);             // the last chunk's output
    if (page_out) c->copy_pool_out->wait();
    AW_HIP_TRY(he);
    sp->hist_cur ^= 1;
    return AW_OK;
} AW_NOEXCEPT_TAIL

/* Page-locked host memory for the host entries (hipHostMalloc): buffers from here cross PCIe by DMA without the runtime's staging copy. */
aw_status aw_host_alloc_pinned(aw_context *ctx, size_t bytes, void **ptr) try {
    if (!ctx || !ptr) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
    AW_HIP_TRY(hipSetDevice(ctx->device));
    AW_HIP_TRY(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
    ctx->device_allocs += 1;
    return AW_OK;
} AW_NOEXCEPT_TAIL
aw_status aw_host_free_pinned(aw_context *ctx, void *ptr) try {
    if (!ctx) return fail(AW_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (ptr) AW_HIP_TRY(hipHostFree(ptr));
    return AW_OK;
} AW_NOEXCEPT_TAIL

aw_status aw_spatializer_process_planar(aw_spatializer *sp, const float *in_l, const float *in_r, float *out_l,
                                        float *out_r, int32_t frames) try {
    if (!sp || !in_l || !out_l || !out_r) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
    if (sp->n_streams != 1 || sp->n_channels != 2)
        return fail(AW_ERR_INVALID_ARGUMENT, "planar entry needs a 1-stream, 2-channel spatializer");
    if (frames <= 0) return frames == 0 ? AW_OK : fail(AW_ERR_INVALID_ARGUMENT, "frameCount must be >= 0");
    AW_HIP_TRY(hipSetDevice(sp->ctx->device));
    hipStream_t s = sp->ctx->stream;
    if (sp_zero_copy(sp, frames)) {
        // The render-callback shape (AudioPipeline.swift:3-11).  Interleave on the way into the page-locked staging (mono duplication,
        // RealtimeAudioProcessor.swift:95-107), let the kernels read and write it in place over PCIe, deinterleave on the way out: two or
        // three kernel launches and one synchronisation per callback (round 4: four copies through the DMA engines and two more kernels).
        std::lock_guard<std::mutex> lk(sp->ctx->launch_mu);
        const float *r_src = in_r ? in_r : in_l;
        float *pi = sp->h_pin_in;
        for (int i = 0; i < frames; ++i) { pi[2 * (size_t)i] = in_l[i]; pi[2 * (size_t)i + 1] = r_src[i]; }
        const LwCallPlan lw0 = sp_begin_call(sp, frames);
        aw_status st0 = sp_run_streams(sp, lw0, 0, 1, sp->h_pin_in, sp->h_pin_out, frames);
        if (st0 != AW_OK) return st0;
        sp->hist_cur ^= 1;
        AW_HIP_TRY(hipStreamSynchronize(s));
        const float *po = sp->h_pin_out;
        for (int i = 0; i < frames; ++i) { out_l[i] = po[2 * (size_t)i]; out_r[i] = po[2 * (size_t)i + 1]; }
        return AW_OK;
    }
    // staging layout: [in interleaved 2F | planar L F | planar R F] and [out interleaved 2F | L F | R F]
    aw_status st = sp_grow(sp, &sp->d_stage_in, &sp->stage_in_cap, (size_t)frames * 4);
    if (st == AW_OK) st = sp_grow(sp, &sp->d_stage_out, &sp->stage_out_cap, (size_t)frames * 4);
    if (st != AW_OK) return st;
    float *d_il = sp->d_stage_in + 2 * (size_t)frames, *d_ir = d_il + frames;
    float *d_ol = sp->d_stage_out + 2 * (size_t)frames, *d_or = d_ol + frames;
    AW_HIP_TRY(hipMemcpyAsync(d_il, in_l, sizeof(float) * frames, hipMemcpyHostToDevice, s));
    AW_HIP_TRY(hipMemcpyAsync(d_ir, in_r ? in_r : in_l, sizeof(float) * frames, hipMemcpyHostToDevice, s));   // mono dup, RealtimeAudioProcessor.swift:95-107
    AW_HIP_TRY(awk::launch_interleave2(d_il, d_ir, sp->d_stage_in, frames, s));
    st = aw_spatializer_process(sp, sp->d_stage_in, sp->d_stage_out, frames);
    if (st != AW_OK) return st;
    AW_HIP_TRY(awk::launch_deinterleave2(sp->d_stage_out, d_ol, d_or, frames, s));
    AW_HIP_TRY(hipMemcpyAsync(out_l, d_ol, sizeof(float) * frames, hipMemcpyDeviceToHost, s));
    AW_HIP_TRY(hipMemcpyAsync(out_r, d_or, sizeof(float) * frames, hipMemcpyDeviceToHost, s));
    AW_HIP_TRY(hipStreamSynchronize(s));
    return AW_OK;
} AW_NOEXCEPT_TAIL

aw_status aw_spatializer_reset(aw_spatializer *sp) try {
    if (!sp) return fail(AW_ERR_INVALID_ARGUMENT, "sp is NULL");
    AW_HIP_TRY(hipSetDevice(sp->ctx->device));
    const size_t n = (size_t)sp->n_streams * sp->hist_len * sp->n_channels;
    for (int i = 0; i < 2; ++i)
        AW_HIP_TRY(hipMemsetAsync(sp->d_hist[i], 0, std::max<size_t>(n, 1) * sizeof(float), sp->ctx->stream));
    return AW_OK;
} AW_NOEXCEPT_TAIL

/* ---- mono engine (ConvolutionEngine) ------------------------------------------------------- */
aw_status aw_engine_create(aw_context *ctx, const float *hrir_samples, int32_t count, int32_t block_size,
                           aw_engine **out) try {
    if (!out) return fail(AW_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (!ctx || !hrir_samples) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
    if (count <= 0 || block_size <= 0) return fail(AW_ERR_CONVOLUTION_SETUP_FAILED, "empty HRIR or non-positive block size");
    awr::Owner<aw_engine> owner(new (std::nothrow) aw_engine(), aw_engine_destroy);
    aw_engine *e = owner.get();
    if (!e) return fail(AW_ERR_OUT_OF_MEMORY, "engine");
    e->ctx = ctx; e->block_size = block_size;
    aw_status st = aw_hrir_create(ctx, hrir_samples, 1, count, 0.0, &e->hrir);
    const int32_t zero = 0;
    if (st == AW_OK) st = aw_spatializer_create(ctx, e->hrir, 1, &zero, &zero, 1, block_size, &e->sp);
    // the engine processes exactly block_size frames per call: everything process needs is allocated here, like
    // ConvolutionEngine.init (ConvolutionEngine.swift:97-138) — process does not allocate
    if (st == AW_OK) st = aw_spatializer_reserve_host(e->sp, block_size);
    if (st != AW_OK) return st;
    e->tmp_out.assign((size_t)block_size * 2, 0.f);
    *out = owner.release();
    return AW_OK;
} AW_NOEXCEPT_TAIL

void aw_engine_destroy(aw_engine *e) {
    if (!e) return;
    aw_spatializer_destroy(e->sp);
    aw_hrir_destroy(e->hrir);
    delete e;
}

int32_t aw_engine_block_size(const aw_engine *e) { return e ? e->block_size : 0; }

aw_status aw_engine_process(aw_engine *e, const float *input, float *output) try {
    if (!e || !input || !output) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
    aw_status st = aw_spatializer_process_host(e->sp, input, e->tmp_out.data(), e->block_size);
    if (st != AW_OK) return st;
    for (int i = 0; i < e->block_size; ++i) output[i] = e->tmp_out[2 * (size_t)i];
    return AW_OK;
} AW_NOEXCEPT_TAIL

aw_status aw_engine_process_n(aw_engine *e, const float *input, float *output, int32_t frame_count) try {
    if (!e) return fail(AW_ERR_INVALID_ARGUMENT, "engine is NULL");
    if (frame_count != e->block_size)      // guard count == blockSize else { return }  ConvolutionEngine.swift:372
        return fail(AW_ERR_BLOCK_SIZE_MISMATCH, "frameCount != blockSize: block ignored");
    return aw_engine_process(e, input, output);
} AW_NOEXCEPT_TAIL

aw_status aw_engine_process_accumulate(aw_engine *e, const float *input, float *acc) try {
    if (!e || !input || !acc) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
    aw_status st = aw_spatializer_process_host(e->sp, input, e->tmp_out.data(), e->block_size);
    if (st != AW_OK) return st;
    for (int i = 0; i < e->block_size; ++i) acc[i] += e->tmp_out[2 * (size_t)i];    // vDSP_vadd, ConvolutionEngine.swift:393
    return AW_OK;
} AW_NOEXCEPT_TAIL

aw_status aw_engine_reset(aw_engine *e) try {
    if (!e) return fail(AW_ERR_INVALID_ARGUMENT, "engine is NULL");
    return aw_spatializer_reset(e->sp);
} AW_NOEXCEPT_TAIL

/* ---- callback-size adapter (RealtimeAudioProcessor) ---------------------------------------- */
aw_status aw_realtime_create(aw_context *ctx, const aw_hrir *hrir, int32_t n_renderers, const int32_t *left_track,
                             const int32_t *right_track, int32_t block_size, int32_t max_frames, aw_realtime **out) try {
    if (!out) return fail(AW_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (!ctx || !hrir) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
    if (block_size <= 0 || max_frames <= 0)      // precondition(blockSize > 0), (maxFramesPerCallback > 0)  :35-36
        return fail(AW_ERR_INVALID_ARGUMENT, "blockSize and maxFramesPerCallback must be positive");
    if (n_renderers < 0 || (n_renderers > 0 && (!left_track || !right_track)))
        return fail(AW_ERR_INVALID_ARGUMENT, "bad renderer list");
    awr::Owner<aw_realtime> owner(new (std::nothrow) aw_realtime(), aw_realtime_destroy);
    aw_realtime *p = owner.get();
    if (!p) return fail(AW_ERR_OUT_OF_MEMORY, "realtime");
    p->ctx = ctx; p->block_size = block_size; p->max_frames = max_frames; p->n_renderers = n_renderers;
    p->fifo_capacity = max_frames + block_size;                         // :41
    p->pending.assign((size_t)block_size * 2, 0.f);
    p->fifo_left.assign((size_t)p->fifo_capacity, 0.f);
    p->fifo_right.assign((size_t)p->fifo_capacity, 0.f);
    // everything a callback can need is allocated here, like the reference's init (RealtimeAudioProcessor.swift:30-62: pending, block and
    // FIFO buffers): one callback completes at most (block_size - 1 + max_frames) / block_size blocks = fewer than fifo_capacity frames
    p->ready_in.reserve((size_t)p->fifo_capacity * 2);
    p->ready_out.reserve((size_t)p->fifo_capacity * 2);
    const int used = std::min(n_renderers, 2);                          // min(renderers.count, 2)  :145
    if (used > 0) {
        int32_t lt[2] = {-1, -1}, rt[2] = {-1, -1};
        for (int r = 0; r < used; ++r) { lt[r] = left_track[r]; rt[r] = right_track[r]; }
        aw_status st = aw_spatializer_create(ctx, hrir, 2, lt, rt, 1, block_size, &p->sp);
        // device side: kernel scratch and the host entry's staging for the longest device call a callback can make (process must not allocate)
        if (st == AW_OK) st = aw_spatializer_reserve_host(p->sp, p->fifo_capacity);
        if (st != AW_OK) return st;
    }
    *out = owner.release();
    return AW_OK;
} AW_NOEXCEPT_TAIL

/* Introspection for the "process must not allocate" contract test: 0 bytes of host buffer capacity held (pending, ready, FIFO),
 * 1 bytes of grow-only device buffers (aw_spatializer_info 6), 2 device allocations made so far on the context (aw_spatializer_info 13). */
int64_t aw_realtime_info(const aw_realtime *p, int32_t what) {
    if (!p) return -1;
    switch (what) {
        case 0: return (int64_t)((p->pending.capacity() + p->ready_in.capacity() + p->ready_out.capacity() + p->fifo_left.capacity() + p->fifo_right.capacity()) * sizeof(float));
        case 1: return p->sp ? aw_spatializer_info(p->sp, 6) : 0;
        case 2: return p->ctx->device_allocs;
        default: return -1;
    }
}

void aw_realtime_destroy(aw_realtime *p) {
    if (!p) return;
    aw_spatializer_destroy(p->sp);
    delete p;
}

aw_status aw_realtime_process(aw_realtime *p, const float *in_l, const float *in_r, float *out_l, float *out_r,
                              int32_t frame_count) try {
    if (!p) return fail(AW_ERR_INVALID_ARGUMENT, "processor is NULL");
    if (frame_count <= 0) return AW_OK;                                 // guard frameCount > 0 else { return }  :84
    if (!in_l || !out_l || !out_r) return fail(AW_ERR_INVALID_ARGUMENT, "NULL buffer");
    if (frame_count > p->max_frames)                                    // precondition  :85
        return fail(AW_ERR_INVALID_ARGUMENT, "frameCount exceeds maxFramesPerCallback");
    const int B = p->block_size;
    p->ready_in.clear();
    int off = 0;
    while (off < frame_count) {                                         // :88-116
        const int copy = std::min(B - p->pending_count, frame_count - off);
        for (int i = 0; i < copy; ++i) {
            p->pending[2 * (size_t)(p->pending_count + i)] = in_l[off + i];
            p->pending[2 * (size_t)(p->pending_count + i) + 1] = (in_r ? in_r : in_l)[off + i];   // mono dup :95-107
        }
        p->pending_count += copy;
        off += copy;
        if (p->pending_count == B) {
            // processPendingBlock (:141-172) — blocks completed inside one callback are convolved
            // together in one device call below; the result is block-size independent.
            p->ready_in.insert(p->ready_in.end(), p->pending.begin(), p->pending.end());
            p->pending_count = 0;
        }
    }
    const int ready_frames = (int)(p->ready_in.size() / 2);
    if (ready_frames > 0) {
        p->ready_out.assign((size_t)ready_frames * 2, 0.f);             // memset blockLeft/blockRight :142-143
        if (p->sp) {
            aw_status st = aw_spatializer_process_host(p->sp, p->ready_in.data(), p->ready_out.data(), ready_frames);
            if (st != AW_OK) return st;
        }
        for (int i = 0; i < ready_frames; ++i) {                        // FIFO push :166-171
            const int w = (p->fifo_read_index + p->fifo_count) % p->fifo_capacity;
            p->fifo_left[(size_t)w] = p->ready_out[2 * (size_t)i];
            p->fifo_right[(size_t)w] = p->ready_out[2 * (size_t)i + 1];
            p->fifo_count += 1;
        }
    }
    for (int i = 0; i < frame_count; ++i) {                             // drain :174-190
        if (p->fifo_count > 0) {
            out_l[i] = p->fifo_left[(size_t)p->fifo_read_index];
            out_r[i] = p->fifo_right[(size_t)p->fifo_read_index];
            p->fifo_read_index = (p->fifo_read_index + 1) % p->fifo_capacity;
            p->fifo_count -= 1;
        } else {
            out_l[i] = 0.f;
            out_r[i] = 0.f;
        }
    }
    return AW_OK;
} AW_NOEXCEPT_TAIL

aw_status aw_realtime_reset(aw_realtime *p) try {                           // :121-139
    if (!p) return fail(AW_ERR_INVALID_ARGUMENT, "processor is NULL");
    if (p->sp) {
        aw_status st = aw_spatializer_reset(p->sp);
        if (st != AW_OK) return st;
    }
    std::fill(p->pending.begin(), p->pending.end(), 0.f);
    std::fill(p->fifo_left.begin(), p->fifo_left.end(), 0.f);
    std::fill(p->fifo_right.begin(), p->fifo_right.end(), 0.f);
    p->pending_count = 0; p->fifo_read_index = 0; p->fifo_count = 0;
    return AW_OK;
} AW_NOEXCEPT_TAIL

/* ---- synthetic input --------------------------------------------------------------------------- */
aw_status aw_synth_fill(aw_context *ctx, float *dst, int32_t n_streams, int64_t frames, int32_t n_channels,
                        uint64_t seed, uint64_t first_stream) try {
    if (!ctx || !dst) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_streams <= 0 || frames <= 0 || n_channels <= 0) return fail(AW_ERR_INVALID_ARGUMENT, "non-positive size");
    AW_HIP_TRY(hipSetDevice(ctx->device));
    AW_HIP_TRY(awk::launch_synth_fill(dst, n_streams, frames * n_channels, seed, first_stream, ctx->stream));
    return AW_OK;
} AW_NOEXCEPT_TAIL

}  // extern "C"
