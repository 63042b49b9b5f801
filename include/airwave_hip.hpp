// airwave_hip.hpp — header-only C++ mirror of the reference's Swift types over the C ABI
// (include/airwave_hip.h).  Same names and call shapes as ConvolutionEngine.swift /
// RealtimeAudioProcessor.swift / HRIRManager.swift so that C++ hosts (and the ABI replay harness in
// tests/harness/) read like the reference.  Errors surface as aw::Error (status + message);
// the failable initialiser `init?` becomes `ConvolutionEngine::make` returning nullptr.
#pragma once
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "airwave_hip.h"

namespace aw {

struct Error : std::runtime_error {
    aw_status status;
    Error(aw_status s, const std::string &m) : std::runtime_error(std::string(aw_status_string(s)) + ": " + m), status(s) {}
};
inline void check(aw_status s) {
    if (s != AW_OK) throw Error(s, aw_last_error_message());
}

class Context {
  public:
    explicit Context(int device = 0) { check(aw_context_create(device, &h_)); }
    Context(int device, void *hip_stream) { check(aw_context_create_on_stream(device, hip_stream, &h_)); }
    ~Context() { aw_context_destroy(h_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    aw_context *get() const { return h_; }
    void synchronize() { check(aw_context_synchronize(h_)); }
    void reserveScratch(size_t bytes) { check(aw_context_reserve_scratch(h_, bytes)); }      // the pool its spatializers share, sized at start-up
    size_t scratchBytes() const { return aw_context_scratch_bytes(h_); }
  private:
    aw_context *h_ = nullptr;
};

// ConvolutionEngine.swift:14-408
class ConvolutionEngine {
  public:
    // init?(hrirSamples:blockSize:)  :68 — nullptr instead of nil
    static std::unique_ptr<ConvolutionEngine> make(Context &ctx, const std::vector<float> &hrirSamples, int blockSize = 512) {
        aw_engine *e = nullptr;
        if (aw_engine_create(ctx.get(), hrirSamples.data(), (int32_t)hrirSamples.size(), blockSize, &e) != AW_OK) return nullptr;
        return std::unique_ptr<ConvolutionEngine>(new ConvolutionEngine(e));
    }
    ~ConvolutionEngine() { aw_engine_destroy(h_); }
    void process(const float *input, float *output) { check(aw_engine_process(h_, input, output)); }                  // :232
    // process(input:output:frameCount:) :370 — silently ignores a wrong count, like the reference
    void process(const std::vector<float> &input, std::vector<float> &output, int frameCount = -1) {
        const int count = frameCount < 0 ? blockSize() : frameCount;
        const aw_status s = aw_engine_process_n(h_, input.data(), output.data(), count);
        if (s != AW_OK && s != AW_ERR_BLOCK_SIZE_MISMATCH) check(s);
    }
    void processAndAccumulate(const float *input, float *outputAccumulator) {                                      // :388
        check(aw_engine_process_accumulate(h_, input, outputAccumulator));
    }
    void reset() { check(aw_engine_reset(h_)); }                                                                   // :397
    int blockSize() const { return aw_engine_block_size(h_); }
  private:
    explicit ConvolutionEngine(aw_engine *e) : h_(e) {}
    aw_engine *h_;
};

class HRIR {
  public:
    HRIR(Context &ctx, const float *tracks, int nTracks, int taps, double sampleRate) {
        check(aw_hrir_create(ctx.get(), tracks, nTracks, taps, sampleRate, &h_));
    }
    ~HRIR() { aw_hrir_destroy(h_); }
    HRIR(const HRIR &) = delete;
    aw_hrir *get() const { return h_; }
  private:
    aw_hrir *h_ = nullptr;
};

// The batch engine network (HRIRManager.RendererState + processPendingBlock, generalised to N speakers)
class Spatializer {
  public:
    Spatializer(Context &ctx, const HRIR &hrir, const std::vector<int32_t> &leftTrack, const std::vector<int32_t> &rightTrack,
                int nStreams = 1) {
        check(aw_spatializer_create(ctx.get(), hrir.get(), (int32_t)leftTrack.size(), leftTrack.data(), rightTrack.data(),
                                    nStreams, 0, &h_));
    }
    explicit Spatializer(aw_spatializer *adopt) : h_(adopt) {}
    ~Spatializer() { aw_spatializer_destroy(h_); }
    Spatializer(const Spatializer &) = delete;
    void processDevice(const float *in, float *out, int64_t frames) { check(aw_spatializer_process(h_, in, out, frames)); }
    void processHost(const float *in, float *out, int64_t frames) { check(aw_spatializer_process_host(h_, in, out, frames)); }
    // StereoAudioProcessing.process  AudioPipeline.swift:3-11
    void process(const float *inputLeft, const float *inputRight, float *outputLeft, float *outputRight, int frameCount) {
        check(aw_spatializer_process_planar(h_, inputLeft, inputRight, outputLeft, outputRight, frameCount));
    }
    void reserve(int64_t maxFrames) { check(aw_spatializer_reserve(h_, maxFrames)); }
    void reserveHost(int64_t maxFrames) { check(aw_spatializer_reserve_host(h_, maxFrames)); }     // + the host entry's device-side staging
    int64_t info(int32_t what) const { return aw_spatializer_info(h_, what); }
    void reset() { check(aw_spatializer_reset(h_)); }
    aw_spatializer *get() const { return h_; }
  private:
    aw_spatializer *h_ = nullptr;
};

// RealtimeAudioProcessor.swift:11-191
class RealtimeAudioProcessor {
  public:
    RealtimeAudioProcessor(Context &ctx, const HRIR &hrir, const std::vector<int32_t> &leftTrack,
                           const std::vector<int32_t> &rightTrack, int blockSize = 512, int maxFramesPerCallback = 4096) {
        check(aw_realtime_create(ctx.get(), hrir.get(), (int32_t)leftTrack.size(), leftTrack.data(), rightTrack.data(), blockSize,
                                 maxFramesPerCallback, &h_));
    }
    ~RealtimeAudioProcessor() { aw_realtime_destroy(h_); }
    void process(const float *inputLeft, const float *inputRight, float *leftOutput, float *rightOutput, int frameCount) {   // :77
        check(aw_realtime_process(h_, inputLeft, inputRight, leftOutput, rightOutput, frameCount));
    }
    void reset() { check(aw_realtime_reset(h_)); }                                                                            // :121
    int64_t info(int32_t what) const { return aw_realtime_info(h_, what); }          // 0 host bytes, 1 device bytes, 2 device allocations: constant across process()
  private:
    aw_realtime *h_ = nullptr;
};

// ---- parametric EQ (EqualizerPreset.swift, EqualizerAPOParser.swift, ParametricEqualizerProcessor.swift) ----
enum class EqualizerFilterType : int32_t { peaking = 0, lowShelf = 1, highShelf = 2 };

class EqualizerDefinition {
  public:
    explicit EqualizerDefinition(double preampDB = 0.0) { check(aw_eq_definition_create(preampDB, &h_)); }
    // EqualizerAPOParser.parse(data:filename:) — throws aw::Error(AW_ERR_EQ_PARSE, "line N: reason; ...")
    static EqualizerDefinition parse(const void *data, size_t size) {
        aw_eq_definition *d = nullptr;
        char issues[4096];
        const aw_status s = aw_eq_parse(data, size, &d, issues, sizeof(issues));
        if (s != AW_OK) throw Error(s, issues);
        return EqualizerDefinition(d);
    }
    EqualizerDefinition(EqualizerDefinition &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    EqualizerDefinition(const EqualizerDefinition &) = delete;
    ~EqualizerDefinition() { aw_eq_definition_destroy(h_); }
    EqualizerDefinition &addFilter(EqualizerFilterType type, double frequencyHz, double gainDB, double q, bool isEnabled = true) {
        check(aw_eq_definition_add_filter(h_, isEnabled ? 1 : 0, (int32_t)type, frequencyHz, gainDB, q));
        return *this;
    }
    double preampDB() const { return aw_eq_definition_preamp_db(h_); }
    int filterCount() const { return aw_eq_definition_filter_count(h_); }
    const aw_eq_definition *get() const { return h_; }
    // aw_eq_fold_hrir: this equalizer folded into HRIR tracks ([nTracks][taps] planar at sampleRate) — EQ(x * h) = x * (h * g), the spatial
    // effect and the equalizer that follows it (AudioEffectGraph.swift:195-211) in one pass.  Returns [nTracks][outTaps]; throws
    // aw::Error(AW_ERR_EQ_NOT_FOLDABLE) when the response does not decay within maxTaps - taps + 1 frames (run the cascade then).
    struct Folded { std::vector<float> tracks; int taps; int responseTaps; double tailBound; };
    Folded foldInto(const float *tracks, int nTracks, int taps, double sampleRate, double tailTolerance = 1e-7, int maxTaps = 65536) const {
        Folded f{};
        int32_t outTaps = 0, response = 0;
        check(aw_eq_fold_hrir(h_, sampleRate, tracks, nTracks, taps, tailTolerance, maxTaps, nullptr, &outTaps, &response, &f.tailBound));
        f.tracks.resize((size_t)nTracks * (size_t)outTaps);
        check(aw_eq_fold_hrir(h_, sampleRate, tracks, nTracks, taps, tailTolerance, maxTaps, f.tracks.data(), &outTaps, &response, &f.tailBound));
        f.taps = outTaps; f.responseTaps = response;
        return f;
    }
  private:
    explicit EqualizerDefinition(aw_eq_definition *d) : h_(d) {}
    aw_eq_definition *h_ = nullptr;
};

// ParametricEqualizerProcessor :116-408 (one stream, planar host buffers: the StereoAudioProcessing surface)
class ParametricEqualizerProcessor {
  public:
    ParametricEqualizerProcessor(Context &ctx, double sampleRate, int maxFramesPerCallback = 4096) {
        check(aw_eq_create(ctx.get(), sampleRate, 1, maxFramesPerCallback, &h_));
    }
    ~ParametricEqualizerProcessor() { aw_eq_destroy(h_); }
    ParametricEqualizerProcessor(const ParametricEqualizerProcessor &) = delete;
    void setTarget(const EqualizerDefinition *definition) { check(aw_eq_set_target(h_, definition ? definition->get() : nullptr)); }   // :226
    void reset() { check(aw_eq_reset(h_)); }                                                                                          // :230
    template <class F> void withPublicationLockForTesting(F body) {                                                                   // :229-233
        check(aw_eq_debug_hold_publication_lock(h_, 1));
        body();
        check(aw_eq_debug_hold_publication_lock(h_, 0));
    }
    void drainRetiredStates() { check(aw_eq_drain_retired(h_)); }                                                                     // :237
    void process(const float *inputLeft, const float *inputRight, float *leftOutput, float *rightOutput, int frameCount) {          // :253
        check(aw_eq_process_planar(h_, inputLeft, inputRight, leftOutput, rightOutput, frameCount));
    }
    int transitionLength() const { return aw_eq_transition_length(h_); }
  private:
    aw_eq *h_ = nullptr;
};

}  // namespace aw
