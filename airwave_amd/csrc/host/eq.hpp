// eq.hpp — host side of the parametric EQ: coefficient builder, Equalizer APO
// text parser, and the double-precision tables the cascade kernel consumes (device/eq_cascade.hpp).
// Reference: Airwave/BiquadCoefficientBuilder.swift, Airwave/EqualizerAPOParser.swift,
// Airwave/EqualizerPreset.swift, ParametricEqualizerProcessor.prepare (ParametricEqualizerProcessor.swift:168-212).
#pragma once
#include <cstddef>
#include <string>
#include <vector>

namespace awh {

struct Biquad {
    double b0, b1, b2, a1, a2;
};
// BiquadCoefficientBuilder.make :29-107.  type 0 peaking, 1 lowShelf, 2 highShelf.  Returns 0 or the
// BiquadCoefficientError kind: 1 invalidSampleRate, 2 invalidFrequency, 3 invalidQ, 4 nonFiniteInput,
// 5 nonFiniteCoefficients.
int biquad_make(int type, double gain_db, double frequency_hz, double q, double sample_rate, Biquad *out);

struct EqFilter {   // EqualizerFilter  EqualizerPreset.swift:9-17
    int source_line = 0;
    long long source_number = -1;   // -1 = nil
    bool enabled = true;
    int type = 0;
    double frequency_hz = 0, gain_db = 0, q = 0;
};
struct EqDefinition {   // EqualizerDefinition  EqualizerPreset.swift:19-27
    double preamp_db = 0;
    std::vector<EqFilter> filters;
};
struct EqIssue {   // EqualizerParseIssue  EqualizerAPOParser.swift:3-6
    int line = 0;   // 0 = nil
    std::string reason;
};
// EqualizerAPOParser.parse :36-151.  Returns true and fills def, or false with >= 1 issue.
bool eq_parse(const void *data, size_t len, EqDefinition &def, std::vector<EqIssue> &issues);

// ParametricEqualizerProcessor.prepare :168-212 plus the kernel's scan tables.
struct EqPrepared {
    std::vector<double> tab;    // [K][65]: coef 5 | zero-input rows chunk x 2 | P powers steps x 4  (device/eq_cascade.hpp EqTables)
    std::vector<double> plane;  // [K][64][4]  P^(m+1)
    double preamp = 1.0;
    int n_filters = 0;
};
enum { kEqPrepOk = 0, kEqPrepInvalidSampleRate = 1, kEqPrepInvalidFilter = 2, kEqPrepTooManyFilters = 3, kEqPrepNonFinitePreamp = 4 };
// def may be NULL (unity).  On kEqPrepInvalidFilter *bad_index = index among ENABLED filters and
// *bad_kind = the BiquadCoefficientError kind; on kEqPrepTooManyFilters *bad_index = the count.
int eq_prepare(const EqDefinition *def, double sample_rate, EqPrepared &out, int *bad_index, int *bad_kind);

// The equalizer FOLDED INTO THE IMPULSE RESPONSES (round 6).  Between two setTarget calls the reference's equalizer is a linear
// time-invariant filter (preamp x cascade of biquads, ParametricEqualizerProcessor.swift:58-91) that follows the spatializer in the graph
// (AudioEffectGraph.swift:195-211), so  EQ(x * h) = x * (h * g)  with g its impulse response: a batch host that knows the definition
// convolves every HRIR track with g ONCE, at activation, and the convolution kernels apply both effects in one pass — no second pass over
// the stereo output, no Float64 recurrence per frame.  g is infinite; it is cut where what is left of it cannot matter:
//   response_taps = the smallest L with  sum_{n >= L} |g[n]|  <=  tail_tolerance x max |g|
// (sum measured over 4 x the allowed length plus a geometric bound on the rest), and the folded track is the first taps + L - 1 samples of
// h * g, computed by running the reference's own Float64 recurrence over the zero-extended track.  The output then differs from the
// reference's by at most tail_bound x (peak of the spatializer's output), tail_bound <= tail_tolerance, on top of the convolution's
// own float32 rounding.  An equalizer whose response does not decay within max_taps - taps + 1 frames (a narrow band at a few Hz) is NOT
// folded: kEqFoldTooLong, and the host runs the cascade kernel after the spatializer as before.
struct EqFold {
    int response_taps = 0;      // L: samples of g that are kept
    int out_taps = 0;           // taps + L - 1
    double tail_bound = 0.0;    // (sum of |g| past L, incl. the bound on what was not simulated) / max |g|
};
enum { kEqFoldTooLong = 5 };
// Same validation and codes as eq_prepare.  out == nullptr: only `info` (the length) is computed.  def may be NULL (unity: g = delta).
int eq_fold_tracks(const EqDefinition *def, double sample_rate, const float *tracks, int n_tracks, int taps, double tail_tolerance, int max_taps,
                   std::vector<float> *out, EqFold &info, int *bad_index, int *bad_kind);

}  // namespace awh
