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

}  // namespace awh
