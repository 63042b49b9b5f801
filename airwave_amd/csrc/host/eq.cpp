// eq.cpp — see eq.hpp.
#include "eq.hpp"

#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "../device/eq_cascade.hpp"

namespace awh {

int biquad_make(int type, double gain_db, double f, double q, double fs, Biquad *out) {
    if (!std::isfinite(fs) || !(fs > 0)) return 1;                                        // :36-38
    if (!std::isfinite(gain_db) || !std::isfinite(f) || !std::isfinite(q)) return 4;      // :39-41
    if (!(f > 0) || !(f < fs / 2)) return 2;                                              // :42-44
    if (!(q > 0)) return 3;                                                               // :45-47
    const double A = std::pow(10.0, gain_db / 40.0);                                      // :49
    const double omega = 2.0 * M_PI * f / fs;
    const double sn = std::sin(omega), cs = std::cos(omega);
    const double alpha = sn / (2.0 * q);
    const double beta = 2.0 * std::sqrt(A) * alpha;
    double b0, b1, b2, a0, a1, a2;
    switch (type) {
        case 0:   // peaking :57-65
            b0 = 1 + alpha * A; b1 = -2 * cs; b2 = 1 - alpha * A;
            a0 = 1 + alpha / A; a1 = -2 * cs; a2 = 1 - alpha / A;
            break;
        case 1:   // lowShelf :66-74
            b0 = A * ((A + 1) - (A - 1) * cs + beta);
            b1 = 2 * A * ((A - 1) - (A + 1) * cs);
            b2 = A * ((A + 1) - (A - 1) * cs - beta);
            a0 = (A + 1) + (A - 1) * cs + beta;
            a1 = -2 * ((A - 1) + (A + 1) * cs);
            a2 = (A + 1) + (A - 1) * cs - beta;
            break;
        default:  // highShelf :75-83
            b0 = A * ((A + 1) + (A - 1) * cs + beta);
            b1 = -2 * A * ((A - 1) + (A + 1) * cs);
            b2 = A * ((A + 1) + (A - 1) * cs - beta);
            a0 = (A + 1) - (A - 1) * cs + beta;
            a1 = 2 * ((A - 1) - (A + 1) * cs);
            a2 = (A + 1) - (A - 1) * cs - beta;
            break;
    }
    if (!std::isfinite(a0) || a0 == 0) return 5;                                          // :86-88
    const Biquad r{b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0};
    if (!std::isfinite(r.b0) || !std::isfinite(r.b1) || !std::isfinite(r.b2) || !std::isfinite(r.a1) || !std::isfinite(r.a2))
        return 5;                                                                         // :97-104
    *out = r;
    return 0;
}

// ---- Equalizer APO text -------------------------------------------------------------------------
namespace {

// Decodes one UTF-8 scalar; returns its length or 0 when the bytes are not valid UTF-8.
int utf8_next(const unsigned char *s, size_t n, unsigned &cp) {
    if (n == 0) return 0;
    const unsigned c = s[0];
    if (c < 0x80) { cp = c; return 1; }
    int len; unsigned min;
    if ((c & 0xE0) == 0xC0) { len = 2; cp = c & 0x1F; min = 0x80; }
    else if ((c & 0xF0) == 0xE0) { len = 3; cp = c & 0x0F; min = 0x800; }
    else if ((c & 0xF8) == 0xF0) { len = 4; cp = c & 0x07; min = 0x10000; }
    else return 0;
    if ((size_t)len > n) return 0;
    for (int i = 1; i < len; ++i) {
        if ((s[i] & 0xC0) != 0x80) return 0;
        cp = (cp << 6) | (s[i] & 0x3F);
    }
    if (cp < min || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) return 0;
    return len;
}
bool is_newline(unsigned cp) {   // CharacterSet.newlines: U+000A-000D, U+0085, U+2028, U+2029
    return (cp >= 0x0A && cp <= 0x0D) || cp == 0x85 || cp == 0x2028 || cp == 0x2029;
}
bool is_space(unsigned cp) {     // CharacterSet.whitespaces / ICU \s  (Zs + TAB) plus the newlines
    return cp == 0x09 || cp == 0x20 || cp == 0xA0 || cp == 0x1680 || (cp >= 0x2000 && cp <= 0x200A) || cp == 0x202F ||
           cp == 0x205F || cp == 0x3000 || is_newline(cp);
}

struct Cursor {
    std::vector<unsigned> s;   // the trimmed line as scalars
    size_t i = 0;
    bool end() const { return i >= s.size(); }
    size_t skip_ws() {
        size_t n = 0;
        while (!end() && is_space(s[i])) { ++i; ++n; }
        return n;
    }
    // case-insensitive keyword at the cursor, as ICU's .caseInsensitive compares: simple case foldings, i.e. the ASCII pairs plus
    // U+212A KELVIN SIGN -> k and U+017F LATIN SMALL LETTER LONG S -> s
    bool keyword(const char *kw) {
        const size_t n = std::strlen(kw);
        if (i + n > s.size()) return false;
        for (size_t k = 0; k < n; ++k) {
            unsigned c = s[i + k];
            if (c >= 'A' && c <= 'Z') c += 32;
            else if (c == 0x212A) c = 'k';
            else if (c == 0x17F) c = 's';
            unsigned e = (unsigned char)kw[k];
            if (e >= 'A' && e <= 'Z') e += 32;
            if (c != e) return false;
        }
        i += n;
        return true;
    }
    // maximal run of non-space scalars (the regex's \S+ followed by \s or end); ASCII only is kept
    // verbatim, anything else makes the token non-numeric
    bool token(std::string &out) {
        out.clear();
        const size_t b = i;
        while (!end() && !is_space(s[i])) {
            out.push_back(s[i] < 0x80 ? (char)s[i] : '?');
            ++i;
        }
        return i > b;
    }
};

// Swift's Double(String) accepts what strtod accepts without surrounding whitespace; non-finite
// results are rejected (finiteDouble :153-156).
bool finite_double(const std::string &t, double &v) {
    if (t.empty() || std::isspace((unsigned char)t[0])) return false;
    errno = 0;
    char *endp = nullptr;
    v = std::strtod(t.c_str(), &endp);
    if (endp != t.c_str() + t.size()) return false;
    return std::isfinite(v);
}

bool starts_with_ci(const std::vector<unsigned> &s, const char *kw) {
    Cursor c;
    c.s = s;
    return c.keyword(kw);
}

// ^Preamp\s*:\s*(\S+)\s+dB$   :27-30
bool match_preamp(const std::vector<unsigned> &line, std::string &value) {
    Cursor c;
    c.s = line;
    if (!c.keyword("preamp")) return false;
    c.skip_ws();
    if (c.end() || c.s[c.i] != ':') return false;
    ++c.i;
    c.skip_ws();
    if (!c.token(value)) return false;
    if (c.skip_ws() == 0) return false;
    if (!c.keyword("db")) return false;
    return c.end();
}

// ^Filter(?:\s+([0-9]+))?\s*:\s+(ON|OFF)\s+(PK|LSC|HSC)\s+Fc\s+(\S+)\s+Hz\s+Gain\s+(\S+)\s+dB\s+Q\s+(\S+)$   :31-34
bool match_filter(const std::vector<unsigned> &line, std::string &number, bool &on, int &type, std::string &fc,
                  std::string &gain, std::string &q) {
    Cursor c;
    c.s = line;
    if (!c.keyword("filter")) return false;
    number.clear();
    {
        Cursor t = c;
        if (t.skip_ws() > 0) {
            std::string digits;
            while (!t.end() && t.s[t.i] >= '0' && t.s[t.i] <= '9') digits.push_back((char)t.s[t.i++]);
            if (!digits.empty()) {
                Cursor u = t;
                u.skip_ws();
                if (!u.end() && u.s[u.i] == ':') { number = digits; c = t; }
            }
        }
    }
    c.skip_ws();
    if (c.end() || c.s[c.i] != ':') return false;
    ++c.i;
    if (c.skip_ws() == 0) return false;
    if (c.keyword("on")) on = true;
    else if (c.keyword("off")) on = false;
    else return false;
    if (c.skip_ws() == 0) return false;
    const size_t type_at = c.i;
    if (c.keyword("pk")) type = c.s[type_at + 1] == 0x212A ? -1 : 0;      // :91-99: "P\u212A".uppercased() is not "PK" -> unsupported filter type
    else if (c.keyword("lsc")) type = 1;                                    //         (a long s uppercases to S and is accepted)
    else if (c.keyword("hsc")) type = 2;
    else return false;
    if (c.skip_ws() == 0 || !c.keyword("fc") || c.skip_ws() == 0 || !c.token(fc)) return false;
    if (c.skip_ws() == 0 || !c.keyword("hz") || c.skip_ws() == 0 || !c.keyword("gain")) return false;
    if (c.skip_ws() == 0 || !c.token(gain)) return false;
    if (c.skip_ws() == 0 || !c.keyword("db") || c.skip_ws() == 0 || !c.keyword("q")) return false;
    if (c.skip_ws() == 0 || !c.token(q)) return false;
    return c.end();
}

}  // namespace

bool eq_parse(const void *data, size_t len, EqDefinition &def, std::vector<EqIssue> &issues) {
    issues.clear();
    def = EqDefinition{};
    if (len > 1048576) {                                                                  // :37-42
        issues.push_back({0, "file exceeds the 1 MiB limit"});
        return false;
    }
    const unsigned char *p = static_cast<const unsigned char *>(data);
    std::vector<unsigned> src;
    src.reserve(len);
    for (size_t i = 0; i < len;) {                                                        // :43-48
        unsigned cp;
        const int n = utf8_next(p + i, len - i, cp);
        if (n == 0) {
            issues.push_back({0, "file is not valid UTF-8"});
            return false;
        }
        src.push_back(cp);
        i += n;
    }
    size_t pos = (!src.empty() && src[0] == 0xFEFF) ? 1 : 0;                              // :49-51
    bool has_preamp = false;
    int declarations = 0, line_number = 0;
    // components(separatedBy: .newlines) splits at EVERY newline scalar, so CR LF yields an empty
    // component between the two and advances the line number twice (:59-60) — kept as is.
    while (pos <= src.size()) {
        size_t e = pos;
        while (e < src.size() && !is_newline(src[e])) ++e;
        ++line_number;
        size_t b = pos, t = e;
        while (b < t && is_space(src[b])) ++b;                                            // :61
        while (t > b && is_space(src[t - 1])) --t;
        std::vector<unsigned> line(src.begin() + b, src.begin() + t);
        pos = e + 1;
        if (line.empty() || line[0] == '#') continue;                                     // :62

        std::string v;
        if (match_preamp(line, v)) {                                                      // :64-76
            double d;
            if (has_preamp) issues.push_back({line_number, "duplicate Preamp directive"});
            else if (!finite_double(v, d)) issues.push_back({line_number, "Preamp must be a finite number"});
            else { def.preamp_db = d; has_preamp = true; }
            continue;
        }
        if (starts_with_ci(line, "filter")) {                                             // :78-136
            if (++declarations > 64) {
                issues.push_back({line_number, "more than 64 filter declarations are not allowed"});
                continue;
            }
            std::string number, fc, gain, q;
            bool on = false;
            int type = 0;
            if (!match_filter(line, number, on, type, fc, gain, q)) {
                issues.push_back({line_number, "malformed Filter directive"});
                continue;
            }
            if (type < 0) {
                issues.push_back({line_number, "unsupported filter type"});
                continue;
            }
            double f = 0, g = 0, qq = 0;
            const bool hf = finite_double(fc, f), hg = finite_double(gain, g), hq = finite_double(q, qq);
            bool bad = false;
            if (hf) { if (f <= 0) { issues.push_back({line_number, "frequency must be positive"}); bad = true; } }
            else { issues.push_back({line_number, "frequency must be a finite number"}); bad = true; }
            if (!hg) { issues.push_back({line_number, "gain must be a finite number"}); bad = true; }
            if (hq) { if (qq <= 0) { issues.push_back({line_number, "Q must be positive"}); bad = true; } }
            else { issues.push_back({line_number, "Q must be a finite number"}); bad = true; }
            if (bad) continue;
            EqFilter fl;
            fl.source_line = line_number;
            if (!number.empty() && number.size() <= 18) fl.source_number = std::strtoll(number.c_str(), nullptr, 10);
            fl.enabled = on; fl.type = type; fl.frequency_hz = f; fl.gain_db = g; fl.q = qq;
            def.filters.push_back(fl);
            continue;
        }
        issues.push_back({line_number, starts_with_ci(line, "preamp") ? "malformed Preamp directive" : "unsupported directive"});   // :138-142
    }
    bool any_enabled = false;
    for (const auto &f : def.filters) any_enabled |= f.enabled;
    if (issues.empty() && def.preamp_db == 0 && !any_enabled)                             // :145-147
        issues.push_back({0, "effective configuration must contain a non-zero preamp or an enabled supported filter"});
    return issues.empty();
}

// ---- prepare ------------------------------------------------------------------------------------
// The kernel's tables are powers of M up to M^(64 chunk) = M^2048.  A low-frequency section's M is nearly defective
// (a double pole next to z = 1: 20 Hz at 96 kHz is an angle of 1.3e-3), and products formed in double lose
// n eps / angle^2 of their entries — 1e-7 at n = 2048, which reached the Float32 output (4.6 ulp of the peak on one of 12 000
// randomised 64-filter scripts, tools/fuzz_eq.py).  The powers are therefore formed in double-double arithmetic
// (error-free two_sum / two_prod, ~32 digits) and rounded once.
struct DD { double hi, lo; };
static inline DD dd_norm(double s, double e) { const double hi = s + e; return {hi, e - (hi - s)}; }
static inline DD dd_add(DD x, DD y) {
    const double s = x.hi + y.hi, bb = s - x.hi;
    const double e = (x.hi - (s - bb)) + (y.hi - bb) + x.lo + y.lo;
    return dd_norm(s, e);
}
static inline DD dd_mul(DD x, DD y) {
    const double p = x.hi * y.hi;
    const double e = std::fma(x.hi, y.hi, -p) + (x.hi * y.lo + x.lo * y.hi);
    return dd_norm(p, e);
}
struct Mat2 { DD m[4]; };
static Mat2 mat2_mul(const Mat2 &a, const Mat2 &b) {
    Mat2 c;
    c.m[0] = dd_add(dd_mul(a.m[0], b.m[0]), dd_mul(a.m[1], b.m[2]));
    c.m[1] = dd_add(dd_mul(a.m[0], b.m[1]), dd_mul(a.m[1], b.m[3]));
    c.m[2] = dd_add(dd_mul(a.m[2], b.m[0]), dd_mul(a.m[3], b.m[2]));
    c.m[3] = dd_add(dd_mul(a.m[2], b.m[1]), dd_mul(a.m[3], b.m[3]));
    return c;
}
static void mat2_store(const Mat2 &a, double *dst) { for (int i = 0; i < 4; ++i) dst[i] = a.m[i].hi + a.m[i].lo; }

int eq_prepare(const EqDefinition *def, double fs, EqPrepared &out, int *bad_index, int *bad_kind) {
    if (!std::isfinite(fs) || !(fs > 0)) return kEqPrepInvalidSampleRate;                 // :172-174
    const double preamp_db = def ? def->preamp_db : 0.0;
    const double preamp = std::pow(10.0, preamp_db / 20.0);                               // :176-179
    if (!std::isfinite(preamp_db) || !std::isfinite(preamp)) return kEqPrepNonFinitePreamp;
    std::vector<const EqFilter *> enabled;
    if (def)
        for (const auto &f : def->filters)
            if (f.enabled) enabled.push_back(&f);                                         // :181
    if ((int)enabled.size() > awk::kEqMaxFilters) {                                       // :182-184
        if (bad_index) *bad_index = (int)enabled.size();
        return kEqPrepTooManyFilters;
    }
    const int K = (int)enabled.size();
    out = EqPrepared{};
    out.preamp = preamp;
    out.n_filters = K;
    out.tab.resize((size_t)K * awk::kEqTabDoubles);
    out.plane.resize((size_t)K * 64 * 4);
    for (int k = 0; k < K; ++k) {
        Biquad c;
        const int kind = biquad_make(enabled[k]->type, enabled[k]->gain_db, enabled[k]->frequency_hz, enabled[k]->q, fs, &c);
        if (kind) {                                                                       // :188-202
            if (bad_index) *bad_index = k;
            if (bad_kind) *bad_kind = kind;
            return kEqPrepInvalidFilter;
        }
        double *cf = &out.tab[(size_t)k * awk::kEqTabDoubles];
        double *zir = cf + 5, *ppow = zir + awk::kEqChunk * 2;
        cf[0] = c.b0; cf[1] = c.b1; cf[2] = c.b2; cf[3] = c.a1; cf[4] = c.a2;
        // zero-input state matrix of the transposed direct form II section (x = 0 in :71-77):
        //   z1' = -a1 z1 + z2,  z2' = -a2 z1,  y = z1
        const Mat2 M{{{-c.a1, 0.0}, {1.0, 0.0}, {-c.a2, 0.0}, {0.0, 0.0}}};
        Mat2 Mj{{{1.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}, {1.0, 0.0}}};
        for (int j = 0; j < awk::kEqChunk; ++j) {
            zir[j * 2] = Mj.m[0].hi + Mj.m[0].lo;
            zir[j * 2 + 1] = Mj.m[1].hi + Mj.m[1].lo;
            Mj = mat2_mul(M, Mj);
        }
        Mat2 P = Mj;                                  // M^chunk
        Mat2 Pl{{{1.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}, {1.0, 0.0}}};
        for (int l = 0; l < 64; ++l) {          // entry m holds P^(m+1)
            Pl = mat2_mul(P, Pl);
            mat2_store(Pl, &out.plane[((size_t)k * 64 + l) * 4]);
        }
        for (int s = 0; s < awk::kEqScanSteps; ++s) {
            mat2_store(P, &ppow[s * 4]);
            P = mat2_mul(P, P);
        }
    }
    return kEqPrepOk;
}

// ---- the equalizer folded into impulse responses (eq.hpp) ------------------------------------------------------------------------
namespace {
// ParametricEqualizerState.process :58-91 for one channel: Float64, transposed direct form II, the reference's subnormal flush
struct Cascade {
    std::vector<Biquad> c;
    std::vector<double> z;      // [filter][2]
    double preamp = 1.0;
    void reset() { z.assign(c.size() * 2, 0.0); }
    static double flush(double v) { return std::fabs(v) < 1e-30 ? 0.0 : v; }          // :93-96
    double step(double x) {
        double v = x * preamp;
        for (size_t k = 0; k < c.size(); ++k) {
            const Biquad &q = c[k];
            const double y = q.b0 * v + z[2 * k];
            const double z1 = q.b1 * v - q.a1 * y + z[2 * k + 1];
            const double z2 = q.b2 * v - q.a2 * y;
            z[2 * k] = flush(z1);
            z[2 * k + 1] = flush(z2);
            v = y;
        }
        return v;
    }
};
}  // namespace

int eq_fold_tracks(const EqDefinition *def, double fs, const float *tracks, int n_tracks, int taps, double tol, int max_taps,
                   std::vector<float> *out, EqFold &info, int *bad_index, int *bad_kind) {
    info = EqFold{};
    if (!std::isfinite(fs) || !(fs > 0)) return kEqPrepInvalidSampleRate;
    const double preamp_db = def ? def->preamp_db : 0.0;
    Cascade eq;
    eq.preamp = std::pow(10.0, preamp_db / 20.0);
    if (!std::isfinite(preamp_db) || !std::isfinite(eq.preamp)) return kEqPrepNonFinitePreamp;
    std::vector<const EqFilter *> enabled;
    if (def)
        for (const auto &f : def->filters)
            if (f.enabled) enabled.push_back(&f);
    if ((int)enabled.size() > awk::kEqMaxFilters) {
        if (bad_index) *bad_index = (int)enabled.size();
        return kEqPrepTooManyFilters;
    }
    for (size_t k = 0; k < enabled.size(); ++k) {
        Biquad b;
        const int kind = biquad_make(enabled[k]->type, enabled[k]->gain_db, enabled[k]->frequency_hz, enabled[k]->q, fs, &b);
        if (kind) {
            if (bad_index) *bad_index = (int)k;
            if (bad_kind) *bad_kind = kind;
            return kEqPrepInvalidFilter;
        }
        eq.c.push_back(b);
    }
    const long long cap = (long long)max_taps - taps + 1;         // longest response that still fits
    if (cap < 1) return kEqFoldTooLong;
    // impulse response over 4 x the allowed length: the tail sums below need what lies past the cut
    const long long M = eq.c.empty() ? 1 : 4 * cap;
    std::vector<double> g((size_t)M);
    eq.reset();
    for (long long n = 0; n < M; ++n) g[(size_t)n] = eq.step(n == 0 ? 1.0 : 0.0);
    double peak = 0.0;
    for (double v : g) peak = std::max(peak, std::fabs(v));
    if (!(peak > 0.0) || !std::isfinite(peak)) {                   // a muted equalizer (preamp underflow): every folded track is zero
        info.response_taps = 1; info.out_taps = taps; info.tail_bound = 0.0;
        if (out) out->assign((size_t)n_tracks * taps, 0.0f);
        return kEqPrepOk;
    }
    // what the simulation did not see: the response decays geometrically in the long run; bound it from its last two quarters
    double rest = 0.0;
    if (M >= 8) {
        double b1 = 0.0, b2 = 0.0;
        for (long long n = M / 2; n < 3 * M / 4; ++n) b1 += std::fabs(g[(size_t)n]);
        for (long long n = 3 * M / 4; n < M; ++n) b2 += std::fabs(g[(size_t)n]);
        // (a last quarter far below the tolerance is the recurrence's noise floor, not response: the reference's subnormal flush at 1e-30
        // keeps a zero-input limit cycle of ~1e-29 alive for ever — 500 dB below anything a float32 output can show)
        if (b2 > 1e-3 * tol * peak) {
            const double rho = b1 > 0.0 ? b2 / b1 : 1.0;
            if (!(rho < 0.999)) return kEqFoldTooLong;              // not decaying over the simulated span
            rest = b2 * rho / (1.0 - rho);
        }
    }
    long long L = M;
    double tail = rest;
    for (long long n = M - 1; n >= 1; --n) {                       // tail(n) = sum_{m >= n} |g[m]| + rest
        const double with_n = tail + std::fabs(g[(size_t)n]);
        if (with_n > tol * peak) break;
        tail = with_n;
        L = n;
    }
    if (L > cap) return kEqFoldTooLong;
    info.response_taps = (int)L;
    info.out_taps = (int)(taps + L - 1);
    info.tail_bound = tail / peak;
    if (!out) return kEqPrepOk;
    out->assign((size_t)n_tracks * info.out_taps, 0.0f);
    for (int t = 0; t < n_tracks; ++t) {
        eq.reset();
        float *dst = out->data() + (size_t)t * info.out_taps;
        const float *src = tracks + (size_t)t * taps;
        for (int n = 0; n < info.out_taps; ++n) dst[n] = (float)eq.step(n < taps ? (double)src[n] : 0.0);
    }
    return kEqPrepOk;
}

}  // namespace awh
