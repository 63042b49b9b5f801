// eq_runtime.cpp — C ABI of the parametric EQ row: definitions/parser (host), prepared states and
// the crossfading processor (device).  Mirrors Airwave/ParametricEqualizerProcessor.swift; the
// render/control split of the reference collapses to one owner thread per handle here (like every
// other handle of this library), the state machine and its observable behaviour are kept.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>

#include "device/eq_kernels.hpp"
#include "device/kernels.hpp"
#include "host/eq.hpp"
#include "runtime.hpp"

using awr::fail;

struct aw_eq_definition {
    awh::EqDefinition def;
};

namespace {

// ParametricEqualizerState for n_streams streams: tables + per-stream Float64 histories in HBM.
struct EqState {
    aw_context *ctx = nullptr;
    int n_streams = 0;
    double sample_rate = 0;
    awk::EqTables t{};
    double *d_tables = nullptr;
    double *d_z = nullptr;
    size_t z_count = 0;
    ~EqState() {
        if (d_tables) (void)hipFree(d_tables);
        if (d_z) (void)hipFree(d_z);
    }
};
using EqStatePtr = std::shared_ptr<EqState>;

const char *biquad_error_text(int kind) {   // BiquadCoefficientError.errorDescription  BiquadCoefficientBuilder.swift:18-26
    switch (kind) {
        case 1: return "Sample rate must be finite and positive.";
        case 2: return "Frequency must be finite, positive, and below Nyquist.";
        case 3: return "Q must be finite and positive.";
        case 4: return "Filter parameters must be finite.";
        default: return "Filter coefficients must be finite.";
    }
}

aw_status prepare_state(aw_context *ctx, const awh::EqDefinition *def, double sample_rate, int n_streams, EqStatePtr &out) {
    awh::EqPrepared prep;
    int bad_index = 0, bad_kind = 0;
    switch (awh::eq_prepare(def, sample_rate, prep, &bad_index, &bad_kind)) {
        case awh::kEqPrepInvalidSampleRate: return fail(AW_ERR_EQ_INVALID_SAMPLE_RATE, "Sample rate must be finite and positive.");
        case awh::kEqPrepNonFinitePreamp: return fail(AW_ERR_EQ_NON_FINITE_PREAMP, "Preamp must produce a finite linear gain.");
        case awh::kEqPrepTooManyFilters:
            return fail(AW_ERR_EQ_TOO_MANY_FILTERS, "Equalizer supports at most 64 filters; received " + std::to_string(bad_index) + ".");
        case awh::kEqPrepInvalidFilter: {
            // "Filter N is invalid: ..." (:112) plus the source line EqualizerRuntimeEffect.map looks up (EqualizerRuntimeEffect.swift:85-89)
            int line = 0, seen = 0;
            for (const auto &f : def->filters)
                if (f.enabled && seen++ == bad_index) { line = f.source_line; break; }
            return awr::fail_eq_filter(bad_index, bad_kind, line, "Filter " + std::to_string(bad_index + 1) + " is invalid: " + biquad_error_text(bad_kind) +
                                                      " [kind " + std::to_string(bad_kind) + ", line " + std::to_string(line) + "]");
        }
        default: break;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(AW_ERR_NO_DEVICE, "hipSetDevice failed");
    auto st = std::make_shared<EqState>();
    st->ctx = ctx; st->n_streams = n_streams; st->sample_rate = sample_rate;
    const int K = prep.n_filters;
    const size_t n_tab = prep.tab.size(), n_pl = prep.plane.size();
    AW_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&st->d_tables), std::max<size_t>(n_tab + n_pl, 1) * sizeof(double)));
    if (K > 0) {
        std::vector<double> all(prep.tab);
        all.insert(all.end(), prep.plane.begin(), prep.plane.end());
        AW_HIP_TRY(hipMemcpy(st->d_tables, all.data(), all.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    st->t.tab = st->d_tables;
    st->t.plane = st->d_tables + n_tab;
    st->t.preamp = prep.preamp;
    st->t.n_filters = K;
    st->z_count = std::max<size_t>((size_t)n_streams * K * 4, 1);
    AW_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&st->d_z), st->z_count * sizeof(double)));
    AW_HIP_TRY(hipMemsetAsync(st->d_z, 0, st->z_count * sizeof(double), ctx->stream));
    out = std::move(st);
    return AW_OK;
}

aw_status state_reset(EqState &s) {
    AW_HIP_TRY(hipMemsetAsync(s.d_z, 0, s.z_count * sizeof(double), s.ctx->stream));
    return AW_OK;
}

// One prepared state over `frames` frames of every stream: chunk-parallel body + sequential tail.
aw_status state_process(EqState &s, const float *in, long long in_stride, float *out, long long out_stride, long long frames) {
    if (frames <= 0) return AW_OK;
    // the kernels take one stride for both sides; distinct strides go through an in-place pass on out
    if (in_stride != out_stride) {
        AW_HIP_TRY(awk::launch_eq_copy(in, in_stride, out, out_stride, s.n_streams, frames, s.ctx->stream));
        in = out;
    }
    awk::EqParams p{};
    p.z = s.d_z; p.t = s.t; p.stride_frames = out_stride;
    p.cus = s.ctx->cfg.cus; p.ear_split = s.ctx->cfg.eq_ear_split;
    const long long body = frames - frames % awk::kEqChunk;
    if (body > 0) {
        p.in = in; p.out = out; p.frames = body;
        AW_HIP_TRY(awk::launch_eq_cascade(p, s.n_streams, s.ctx->stream));
    }
    if (frames > body) {
        p.in = in + body * 2; p.out = out + body * 2; p.frames = frames - body;
        AW_HIP_TRY(awk::launch_eq_sequential(p, s.n_streams, s.ctx->stream));
    }
    return AW_OK;
}

}  // namespace

struct aw_eq_state {
    EqStatePtr s;
};

struct aw_eq {
    aw_context *ctx = nullptr;
    int n_streams = 0, max_frames = 0;
    double sample_rate = 0;
    long long transition_length = 1, transition_frame = 0;
    // render-thread state (touched by aw_eq_process only)
    EqStatePtr unity, active, from, to, pending_target, observed, audio_target, pending_retirement;
    // shared with the control thread, each behind its own lock (targetLock, retirementLock, resetLock :129-131)
    std::mutex target_lock, retirement_lock, reset_lock;
    EqStatePtr published, retired;
    bool reset_requested = false;
    float *d_old = nullptr, *d_new = nullptr;   // [stream][transition_length][2] crossfade scratch (oldScratch/newScratch :137-140)
    float *d_stage = nullptr;                   // planar host entry staging
    float *h_pin = nullptr;                     // ... and its page-locked form for callback-sized calls: [2 F] interleaved, processed IN PLACE by the kernels over PCIe
    size_t pin_cap = 0;                         // floats
    size_t stage_cap = 0;
};

extern "C" {

aw_status aw_biquad_make(int32_t type, double gain_db, double f, double q, double fs, double out[5], int32_t *error_kind) try {
    if (!out || type < 0 || type > 2) return fail(AW_ERR_INVALID_ARGUMENT, "bad argument");
    awh::Biquad c{};
    const int kind = awh::biquad_make(type, gain_db, f, q, fs, &c);
    if (error_kind) *error_kind = kind;
    if (kind) return fail(AW_ERR_EQ_INVALID_FILTER, biquad_error_text(kind));
    out[0] = c.b0; out[1] = c.b1; out[2] = c.b2; out[3] = c.a1; out[4] = c.a2;
    return AW_OK;
} AW_NOEXCEPT_TAIL

aw_status aw_eq_definition_create(double preamp_db, aw_eq_definition **out) try {
    if (!out) return fail(AW_ERR_INVALID_ARGUMENT, "out is NULL");
    auto *d = new (std::nothrow) aw_eq_definition();
    if (!d) return fail(AW_ERR_OUT_OF_MEMORY, "allocation failed");
    d->def.preamp_db = preamp_db;
    *out = d;
    return AW_OK;
} AW_NOEXCEPT_TAIL
aw_status aw_eq_definition_add_filter(aw_eq_definition *d, int32_t enabled, int32_t type, double f, double g, double q) try {
    if (!d || type < 0 || type > 2) return fail(AW_ERR_INVALID_ARGUMENT, "bad argument");
    awh::EqFilter fl;
    fl.source_line = (int)d->def.filters.size() + 1;
    fl.enabled = enabled != 0; fl.type = type; fl.frequency_hz = f; fl.gain_db = g; fl.q = q;
    d->def.filters.push_back(fl);
    return AW_OK;
} AW_NOEXCEPT_TAIL
aw_status aw_eq_definition_set_source(aw_eq_definition *d, int32_t i, int32_t line, int64_t number) try {
    if (!d || i < 0 || i >= (int32_t)d->def.filters.size()) return fail(AW_ERR_INVALID_ARGUMENT, "filter index out of range");
    d->def.filters[i].source_line = line;
    d->def.filters[i].source_number = number;
    return AW_OK;
} AW_NOEXCEPT_TAIL
void aw_eq_definition_destroy(aw_eq_definition *d) { delete d; }
double aw_eq_definition_preamp_db(const aw_eq_definition *d) { return d ? d->def.preamp_db : 0.0; }
int32_t aw_eq_definition_filter_count(const aw_eq_definition *d) { return d ? (int32_t)d->def.filters.size() : 0; }
aw_status aw_eq_definition_filter(const aw_eq_definition *d, int32_t i, int32_t *line, int64_t *number, int32_t *enabled,
                                  int32_t *type, double *f, double *g, double *q) try {
    if (!d || i < 0 || i >= (int32_t)d->def.filters.size()) return fail(AW_ERR_INVALID_ARGUMENT, "filter index out of range");
    const auto &fl = d->def.filters[i];
    if (line) *line = fl.source_line;
    if (number) *number = fl.source_number;
    if (enabled) *enabled = fl.enabled;
    if (type) *type = fl.type;
    if (f) *f = fl.frequency_hz;
    if (g) *g = fl.gain_db;
    if (q) *q = fl.q;
    return AW_OK;
} AW_NOEXCEPT_TAIL

aw_status aw_eq_parse(const void *data, size_t size, aw_eq_definition **out, char *issues_out, size_t cap) try {
    if (!out || (!data && size)) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    if (issues_out && cap) issues_out[0] = 0;
    auto d = std::unique_ptr<aw_eq_definition>(new (std::nothrow) aw_eq_definition());
    if (!d) return fail(AW_ERR_OUT_OF_MEMORY, "allocation failed");
    std::vector<awh::EqIssue> issues;
    if (!awh::eq_parse(data, size, d->def, issues)) {
        std::string text;
        for (size_t i = 0; i < issues.size(); ++i) {   // errorDescription :12-20
            if (i) text += "; ";
            if (issues[i].line) text += "line " + std::to_string(issues[i].line) + ": ";
            text += issues[i].reason;
        }
        if (issues_out && cap) {
            std::strncpy(issues_out, text.c_str(), cap - 1);
            issues_out[cap - 1] = 0;
        }
        return fail(AW_ERR_EQ_PARSE, text);
    }
    *out = d.release();
    return AW_OK;
} AW_NOEXCEPT_TAIL

/* ---- prepared state ------------------------------------------------------------------------ */
aw_status aw_eq_state_create(aw_context *ctx, const aw_eq_definition *def, double sample_rate, int32_t n_streams,
                             aw_eq_state **out) try {
    if (!out) return fail(AW_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (!ctx) return fail(AW_ERR_NO_DEVICE, "no context (no HIP device): there is no CPU fallback");
    if (n_streams <= 0) return fail(AW_ERR_INVALID_ARGUMENT, "n_streams must be > 0");
    EqStatePtr s;
    const aw_status st = prepare_state(ctx, def ? &def->def : nullptr, sample_rate, n_streams, s);
    if (st != AW_OK) return st;
    auto *h = new (std::nothrow) aw_eq_state();
    if (!h) return fail(AW_ERR_OUT_OF_MEMORY, "allocation failed");
    h->s = std::move(s);
    *out = h;
    return AW_OK;
} AW_NOEXCEPT_TAIL
void aw_eq_state_destroy(aw_eq_state *s) {
    if (s && s->s) (void)hipSetDevice(s->s->ctx->device);
    delete s;
}
aw_status aw_eq_state_reset(aw_eq_state *s) try {
    if (!s) return fail(AW_ERR_INVALID_ARGUMENT, "state is NULL");
    AW_HIP_TRY(hipSetDevice(s->s->ctx->device));
    return state_reset(*s->s);
} AW_NOEXCEPT_TAIL
aw_status aw_eq_state_process(aw_eq_state *s, const float *in, float *out, int64_t frames) try {
    if (!s || !in || !out) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
    if (frames < 0) return fail(AW_ERR_INVALID_ARGUMENT, "frames must be >= 0");
    AW_HIP_TRY(hipSetDevice(s->s->ctx->device));
    return state_process(*s->s, in, frames, out, frames, frames);
} AW_NOEXCEPT_TAIL
int32_t aw_eq_state_filter_count(const aw_eq_state *s) { return s ? s->s->t.n_filters : 0; }
double aw_eq_state_preamp_linear(const aw_eq_state *s) { return s ? s->s->t.preamp : 0.0; }

/* ---- the equalizer folded into the HRIR (host only) ---------------------------------------------------------- */
aw_status aw_eq_fold_hrir(const aw_eq_definition *def, double sample_rate, const float *tracks, int32_t n_tracks, int32_t taps,
                          double tail_tolerance, int32_t max_taps, float *out_tracks, int32_t *out_taps, int32_t *response_taps,
                          double *tail_bound) try {
    if (!tracks || !out_taps) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_tracks <= 0 || taps <= 0 || max_taps < taps) return fail(AW_ERR_INVALID_ARGUMENT, "n_tracks, taps must be positive and max_taps >= taps");
    if (!(tail_tolerance > 0.0) || !(tail_tolerance < 1.0)) return fail(AW_ERR_INVALID_ARGUMENT, "tail_tolerance must lie in (0, 1)");
    awh::EqFold info;
    int bad_index = 0, bad_kind = 0;
    std::vector<float> folded;
    const awh::EqDefinition *d = def ? &def->def : nullptr;
    switch (awh::eq_fold_tracks(d, sample_rate, tracks, n_tracks, taps, tail_tolerance, max_taps, out_tracks ? &folded : nullptr, info, &bad_index, &bad_kind)) {
        case awh::kEqPrepInvalidSampleRate: return fail(AW_ERR_EQ_INVALID_SAMPLE_RATE, "Sample rate must be finite and positive.");
        case awh::kEqPrepNonFinitePreamp: return fail(AW_ERR_EQ_NON_FINITE_PREAMP, "Preamp must produce a finite linear gain.");
        case awh::kEqPrepTooManyFilters:
            return fail(AW_ERR_EQ_TOO_MANY_FILTERS, "Equalizer supports at most 64 filters; received " + std::to_string(bad_index) + ".");
        case awh::kEqPrepInvalidFilter: {
            int line = 0, seen = 0;
            for (const auto &f : d->filters)
                if (f.enabled && seen++ == bad_index) { line = f.source_line; break; }
            return awr::fail_eq_filter(bad_index, bad_kind, line, "Filter " + std::to_string(bad_index + 1) + " is invalid: " + biquad_error_text(bad_kind) +
                                                      " [kind " + std::to_string(bad_kind) + ", line " + std::to_string(line) + "]");
        }
        case awh::kEqFoldTooLong:
            return fail(AW_ERR_EQ_NOT_FOLDABLE, "the equalizer's impulse response does not decay to " + std::to_string(tail_tolerance) +
                                                    " of its peak within " + std::to_string((long long)max_taps - taps + 1) +
                                                    " frames: run the cascade after the spatializer");
        default: break;
    }
    *out_taps = info.out_taps;
    if (response_taps) *response_taps = info.response_taps;
    if (tail_bound) *tail_bound = info.tail_bound;
    if (out_tracks) std::memcpy(out_tracks, folded.data(), folded.size() * sizeof(float));
    return AW_OK;
} AW_NOEXCEPT_TAIL

/* ---- processor ----------------------------------------------------------------------------- */
aw_status aw_eq_create(aw_context *ctx, double sample_rate, int32_t n_streams, int32_t max_frames, aw_eq **out) try {
    if (!out) return fail(AW_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (!ctx) return fail(AW_ERR_NO_DEVICE, "no context (no HIP device): there is no CPU fallback");
    if (n_streams <= 0) return fail(AW_ERR_INVALID_ARGUMENT, "n_streams must be > 0");
    if (!std::isfinite(sample_rate) || !(sample_rate > 0))                                  // :145-147
        return fail(AW_ERR_EQ_INVALID_SAMPLE_RATE, "Sample rate must be finite and positive.");
    if (max_frames < 0 || max_frames > 4096)                                                // :148-150 (0 = batch, no cap)
        return fail(AW_ERR_EQ_TOO_MANY_FILTERS, "maxFramesPerCallback must be in 1...4096 (0 = unlimited)");
    auto eq = std::unique_ptr<aw_eq>(new (std::nothrow) aw_eq());
    if (!eq) return fail(AW_ERR_OUT_OF_MEMORY, "allocation failed");
    eq->ctx = ctx; eq->n_streams = n_streams; eq->max_frames = max_frames; eq->sample_rate = sample_rate;
    const aw_status st = prepare_state(ctx, nullptr, sample_rate, n_streams, eq->unity);   // :154
    if (st != AW_OK) return st;
    eq->active = eq->unity;
    eq->transition_length = std::max<long long>(1, (long long)std::round(sample_rate * 0.020));   // :155 (.rounded(): half away from zero)
    const size_t n = (size_t)n_streams * eq->transition_length * 2;
    hipError_t he = hipMalloc(reinterpret_cast<void **>(&eq->d_old), n * sizeof(float));
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void **>(&eq->d_new), n * sizeof(float));
    // the planar (plug-in) entry's staging buffer for the largest callback: process never allocates afterwards
    if (he == hipSuccess && n_streams == 1 && max_frames > 0) {
        he = hipMalloc(reinterpret_cast<void **>(&eq->d_stage), (size_t)max_frames * 4 * sizeof(float));
        if (he == hipSuccess) eq->stage_cap = (size_t)max_frames * 4;
        if (he == hipSuccess) he = hipHostMalloc(reinterpret_cast<void **>(&eq->h_pin), (size_t)max_frames * 2 * sizeof(float), hipHostMallocDefault);
        if (he == hipSuccess) eq->pin_cap = (size_t)max_frames * 2;
    }
    if (he != hipSuccess) {                      // a failed later allocation must not leak the earlier ones: destroy frees whatever exists
        aw_eq_destroy(eq.release());
        return awr::hip_fail(he, "crossfade / staging scratch");
    }
    *out = eq.release();
    return AW_OK;
} AW_NOEXCEPT_TAIL

void aw_eq_destroy(aw_eq *eq) {
    if (!eq) return;
    (void)hipSetDevice(eq->ctx->device);
    (void)hipStreamSynchronize(eq->ctx->stream);
    if (eq->d_old) (void)hipFree(eq->d_old);
    if (eq->d_new) (void)hipFree(eq->d_new);
    if (eq->d_stage) (void)hipFree(eq->d_stage);
    if (eq->h_pin) (void)hipHostFree(eq->h_pin);
    delete eq;
}

aw_status aw_eq_set_target(aw_eq *eq, const aw_eq_definition *def) try {                        // :226-228
    if (!eq) return fail(AW_ERR_INVALID_ARGUMENT, "eq is NULL");
    EqStatePtr s;
    const aw_status st = prepare_state(eq->ctx, def ? &def->def : nullptr, eq->sample_rate, eq->n_streams, s);
    if (st != AW_OK) return st;
    std::lock_guard<std::mutex> g(eq->target_lock);                                         // publish :219-227
    eq->published = std::move(s);
    return AW_OK;
} AW_NOEXCEPT_TAIL
aw_status aw_eq_reset(aw_eq *eq) try {                                                          // :230-234
    if (!eq) return fail(AW_ERR_INVALID_ARGUMENT, "eq is NULL");
    std::lock_guard<std::mutex> g(eq->reset_lock);
    eq->reset_requested = true;
    return AW_OK;
} AW_NOEXCEPT_TAIL
aw_status aw_eq_drain_retired(aw_eq *eq) try {                                                  // :237-241
    if (!eq) return fail(AW_ERR_INVALID_ARGUMENT, "eq is NULL");
    EqStatePtr gone;
    {
        std::lock_guard<std::mutex> g(eq->retirement_lock);
        gone = std::move(eq->retired);
        eq->retired.reset();
    }
    if (gone) {
        // the state's buffers may still be read by queued kernels: let the stream finish before freeing
        AW_HIP_TRY(hipSetDevice(eq->ctx->device));
        AW_HIP_TRY(hipStreamSynchronize(eq->ctx->stream));
        gone.reset();
    }
    return AW_OK;
} AW_NOEXCEPT_TAIL

static void eq_begin_transition(aw_eq *eq, const EqStatePtr &target) {                      // :349-354
    if (target == eq->active) return;
    eq->from = eq->active;
    eq->to = target;
    eq->transition_frame = 0;
}
static bool eq_retire(aw_eq *eq, const EqStatePtr &state) {                                 // :373-386
    if (eq->pending_retirement) return false;
    bool parked = false;
    if (eq->retirement_lock.try_lock()) {                                                   // withLockIfAvailable :380
        if (!eq->retired) { eq->retired = state; parked = true; }
        eq->retirement_lock.unlock();
    }
    if (parked) return true;
    eq->pending_retirement = state;
    return false;
}
static void eq_start_pending(aw_eq *eq) {
    if (eq->pending_target) {
        EqStatePtr p = std::move(eq->pending_target);
        eq->pending_target.reset();
        if (p != eq->active) eq_begin_transition(eq, p);
    }
}
static void eq_finish_transition(aw_eq *eq) {                                               // :356-371
    EqStatePtr from = std::move(eq->from);
    eq->active = std::move(eq->to);
    eq->from.reset(); eq->to.reset();
    eq->transition_frame = 0;
    if (!eq_retire(eq, from)) return;
    eq_start_pending(eq);
}
static void eq_observe(aw_eq *eq) {                                                         // :311-333
    if (eq->target_lock.try_lock()) {                                                       // withLockIfAvailable :322: contended -> keep the prior target
        if (eq->published) eq->audio_target = eq->published;
        eq->target_lock.unlock();
    }
    const EqStatePtr &t = eq->audio_target;
    if (!t || t == eq->observed) return;
    eq->observed = t;
    if (eq->to) {
        if (t != eq->to) eq->pending_target = t;
    } else if (eq->pending_retirement) {
        eq->pending_target = t;
    } else if (t != eq->active) {
        eq_begin_transition(eq, t);
    }
}
static void eq_flush_pending_retirement(aw_eq *eq) {                                        // :388-406
    if (!eq->pending_retirement) return;
    bool parked = false;
    if (eq->retirement_lock.try_lock()) {                                                   // :393
        if (!eq->retired) { eq->retired = eq->pending_retirement; parked = true; }
        eq->retirement_lock.unlock();
    }
    if (!parked) return;
    eq->pending_retirement.reset();
    eq_start_pending(eq);
}

aw_status aw_eq_process(aw_eq *eq, const float *in, float *out, int64_t frames) try {           // :253-309
    if (!eq || !in || !out) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
    if (frames <= 0) return frames == 0 ? AW_OK : fail(AW_ERR_INVALID_ARGUMENT, "frames must be >= 0");
    if (eq->max_frames > 0 && frames > eq->max_frames)                                      // precondition :261
        return fail(AW_ERR_INVALID_ARGUMENT, "frameCount exceeds maxFramesPerCallback");
    AW_HIP_TRY(hipSetDevice(eq->ctx->device));
    eq_observe(eq);
    eq_flush_pending_retirement(eq);
    bool do_reset = false;
    if (eq->reset_lock.try_lock()) {                                                        // applyPendingReset :335-347 (:342: contended -> next call)
        do_reset = eq->reset_requested;
        eq->reset_requested = false;
        eq->reset_lock.unlock();
    }
    if (do_reset) {
        for (const EqStatePtr *s : {&eq->active, &eq->from, &eq->to})
            if (*s) {
                const aw_status st = state_reset(**s);
                if (st != AW_OK) return st;
            }
    }
    hipStream_t stream = eq->ctx->stream;
    long long offset = 0;
    while (offset < frames) {
        if (!eq->from || !eq->to)
            return state_process(*eq->active, in + offset * 2, frames, out + offset * 2, frames, frames - offset);
        const long long seg = std::min<long long>(eq->transition_length - eq->transition_frame, frames - offset);
        aw_status st = state_process(*eq->from, in + offset * 2, frames, eq->d_old, seg, seg);
        if (st == AW_OK) st = state_process(*eq->to, in + offset * 2, frames, eq->d_new, seg, seg);
        if (st != AW_OK) return st;
        AW_HIP_TRY(awk::launch_eq_blend(eq->d_old, eq->d_new, out + offset * 2, eq->n_streams, seg, frames, eq->transition_frame,
                                        eq->transition_length, stream));
        eq->transition_frame += seg;
        offset += seg;
        if (eq->transition_frame == eq->transition_length) eq_finish_transition(eq);
    }
    return AW_OK;
} AW_NOEXCEPT_TAIL

aw_status aw_eq_process_planar(aw_eq *eq, const float *in_l, const float *in_r, float *out_l, float *out_r, int32_t frames) try {
    if (!eq || !in_l || !out_l || !out_r) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
    if (eq->n_streams != 1) return fail(AW_ERR_INVALID_ARGUMENT, "planar entry needs a 1-stream equalizer");
    if (frames <= 0) return frames == 0 ? AW_OK : fail(AW_ERR_INVALID_ARGUMENT, "frameCount must be >= 0");
    AW_HIP_TRY(hipSetDevice(eq->ctx->device));
    hipStream_t s = eq->ctx->stream;
    if (eq->pin_cap >= (size_t)frames * 2) {
        // the render-callback shape: interleave into the page-locked staging (left duplicated when there is no right, :68), let the
        // kernels filter it in place over PCIe, deinterleave on the way out — no copy engine, no (de)interleave kernels
        const float *r_src = in_r ? in_r : in_l;
        float *px = eq->h_pin;
        for (int i = 0; i < frames; ++i) { px[2 * (size_t)i] = in_l[i]; px[2 * (size_t)i + 1] = r_src[i]; }
        const aw_status st0 = aw_eq_process(eq, px, px, frames);
        if (st0 != AW_OK) return st0;
        AW_HIP_TRY(hipStreamSynchronize(s));
        for (int i = 0; i < frames; ++i) { out_l[i] = px[2 * (size_t)i]; out_r[i] = px[2 * (size_t)i + 1]; }
        return AW_OK;
    }
    const size_t need = (size_t)frames * 4;   // [interleaved 2F | planar L F | planar R F]
    if (eq->stage_cap < need) {
        if (eq->d_stage) AW_HIP_TRY(hipFree(eq->d_stage));
        eq->d_stage = nullptr; eq->stage_cap = 0;
        AW_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&eq->d_stage), need * sizeof(float)));
        eq->stage_cap = need;
    }
    float *d_x = eq->d_stage, *d_l = d_x + 2 * (size_t)frames, *d_r = d_l + frames;
    AW_HIP_TRY(hipMemcpyAsync(d_l, in_l, sizeof(float) * frames, hipMemcpyHostToDevice, s));
    AW_HIP_TRY(hipMemcpyAsync(d_r, in_r ? in_r : in_l, sizeof(float) * frames, hipMemcpyHostToDevice, s));   // :68
    AW_HIP_TRY(awk::launch_interleave2(d_l, d_r, d_x, frames, s));
    const aw_status st = aw_eq_process(eq, d_x, d_x, frames);
    if (st != AW_OK) return st;
    AW_HIP_TRY(awk::launch_deinterleave2(d_x, d_l, d_r, frames, s));
    AW_HIP_TRY(hipMemcpyAsync(out_l, d_l, sizeof(float) * frames, hipMemcpyDeviceToHost, s));
    AW_HIP_TRY(hipMemcpyAsync(out_r, d_r, sizeof(float) * frames, hipMemcpyDeviceToHost, s));
    AW_HIP_TRY(hipStreamSynchronize(s));
    return AW_OK;
} AW_NOEXCEPT_TAIL

// withPublicationLockForTesting (:229-233, DEBUG builds of the reference): hold = 1 takes the publication lock, 0 releases it
// (same thread).  Lets a test show that a render call under contention keeps its prior target.
aw_status aw_eq_debug_hold_publication_lock(aw_eq *eq, int32_t hold) try {
    if (!eq) return fail(AW_ERR_INVALID_ARGUMENT, "eq is NULL");
    if (hold) eq->target_lock.lock(); else eq->target_lock.unlock();
    return AW_OK;
} AW_NOEXCEPT_TAIL

int32_t aw_eq_transition_length(const aw_eq *eq) { return eq ? (int32_t)eq->transition_length : 0; }
int32_t aw_eq_is_transitioning(const aw_eq *eq) { return eq && eq->from && eq->to ? 1 : 0; }

}  // extern "C"
