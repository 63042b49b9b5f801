// host_api.cpp — host-side data model of the hot path behind the C ABI (no GPU needed):
// WAVLoader, InputLayout, HRIRChannelMap, Resampler and the activatePreset assembly.
// Native C++ equivalents of the reference's Swift types so the engine drops in.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "../runtime.hpp"

using awr::fail;

struct aw_wav {
    double sample_rate = 0.0;
    int channels = 0, frames = 0;
    std::vector<float> planar;   // [channels][frames]
};

struct aw_layout {
    std::vector<std::string> speakers;
    std::string name;
};

struct aw_channel_map {
    // insertion-ordered speaker -> (leftEar, rightEar); later setMapping overwrites (VirtualSpeaker.swift:110-112)
    std::vector<std::string> keys;
    std::map<std::string, std::pair<int, int>> idx;
    void set(const std::string &s, int l, int r) {
        if (!idx.count(s)) keys.push_back(s);
        idx[s] = {l, r};
    }
};

namespace {

uint32_t rd32(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

const char *const kLeftSide[] = {"FL", "BL", "SL", "TFL", "TBL", "FLC"};     // VirtualSpeaker.swift:143
const char *const kRightSide[] = {"FR", "BR", "SR", "TFR", "TBR", "FRC"};    // :148
bool in_set(const std::string &s, const char *const *set, int n) {
    for (int i = 0; i < n; ++i)
        if (s == set[i]) return true;
    return false;
}

// CharacterSet.whitespaces = general category Zs + TAB, as UTF-8 byte sequences; returns the length of the one at s[i] (0: none)
size_t ws_at(const std::string &s, size_t i) {
    const unsigned char c = (unsigned char)s[i];
    if (c == ' ' || c == '\t') return 1;
    if (c == 0xC2 && i + 1 < s.size() && (unsigned char)s[i + 1] == 0xA0) return 2;                      // U+00A0
    if (i + 2 >= s.size()) return 0;
    const unsigned char d = (unsigned char)s[i + 1], e = (unsigned char)s[i + 2];
    if (c == 0xE1 && d == 0x9A && e == 0x80) return 3;                                                     // U+1680
    if (c == 0xE2 && d == 0x80 && ((e >= 0x80 && e <= 0x8A) || e == 0xAF)) return 3;                      // U+2000-200A, U+202F
    if (c == 0xE2 && d == 0x81 && e == 0x9F) return 3;                                                     // U+205F
    if (c == 0xE3 && d == 0x80 && e == 0x80) return 3;                                                     // U+3000
    return 0;
}
std::string trim_ws(const std::string &s) {   // trimmingCharacters(in: .whitespaces)
    size_t a = 0, b = s.size();
    for (size_t n; a < b && (n = ws_at(s, a)) > 0 && a + n <= b;) a += n;
    for (bool again = true; again && b > a;) {
        again = false;
        for (size_t n = 1; n <= 3 && n <= b - a; ++n)
            if (ws_at(s, b - n) == n) { b -= n; again = true; break; }
    }
    return s.substr(a, b - a);
}
// CharacterSet.newlines = U+000A-000D, U+0085, U+2028, U+2029: components(separatedBy:) splits at EVERY such scalar
std::vector<std::string> split_newlines(const std::string &t) {
    std::vector<std::string> out(1);
    for (size_t i = 0; i < t.size(); ++i) {
        const unsigned char c = (unsigned char)t[i];
        size_t n = 0;
        if (c >= 0x0A && c <= 0x0D) n = 1;
        else if (c == 0xC2 && i + 1 < t.size() && (unsigned char)t[i + 1] == 0x85) n = 2;
        else if (c == 0xE2 && i + 2 < t.size() && (unsigned char)t[i + 1] == 0x80 && ((unsigned char)t[i + 2] == 0xA8 || (unsigned char)t[i + 2] == 0xA9)) n = 3;
        if (n) { out.emplace_back(); i += n - 1; }
        else out.back().push_back(t[i]);
    }
    return out;
}
// String.uppercased() as far as it can reach the ASCII alias table: ASCII letters, U+017F LONG S -> S, U+FB02 "fl" ligature -> FL
std::string alias_upper(const std::string &s) {
    std::string u;
    for (size_t i = 0; i < s.size(); ++i) {
        const unsigned char c = (unsigned char)s[i];
        if (c == 0xC5 && i + 1 < s.size() && (unsigned char)s[i + 1] == 0xBF) { u.push_back('S'); ++i; }
        else if (c == 0xEF && i + 2 < s.size() && (unsigned char)s[i + 1] == 0xAC && (unsigned char)s[i + 2] == 0x82) { u += "FL"; i += 2; }
        else u.push_back(c < 0x80 ? (char)std::toupper(c) : (char)c);
    }
    return u;
}

// Swift Int(String): optional sign + ASCII digits only, nil on Int64 overflow.  The ABI carries indices as
// int32: values beyond that range are kept as INT32_MAX / INT32_MIN — still integers, still out of range for
// any HRIR, so activation reports invalidChannelMapping exactly where the reference does.
bool swift_int(const std::string &t, int *out) {
    size_t i = 0;
    if (i < t.size() && (t[i] == '+' || t[i] == '-')) ++i;
    if (i >= t.size()) return false;
    const bool neg = t[0] == '-';
    unsigned long long v = 0;
    const unsigned long long limit = neg ? 9223372036854775808ULL : 9223372036854775807ULL;
    for (size_t j = i; j < t.size(); ++j) {
        if (t[j] < '0' || t[j] > '9') return false;
        const unsigned d = (unsigned)(t[j] - '0');
        if (v > (limit - d) / 10) return false;          // Int64 overflow -> nil
        v = v * 10 + d;
    }
    if (neg) *out = v > 2147483648ULL ? (-2147483647 - 1) : (int)(-(long long)v);
    else *out = v > 2147483647ULL ? 2147483647 : (int)v;
    return true;
}

std::vector<std::string> split(const std::string &s, char sep) {
    std::vector<std::string> out;
    size_t start = 0;
    for (;;) {
        size_t p = s.find(sep, start);
        if (p == std::string::npos) { out.push_back(s.substr(start)); break; }
        out.push_back(s.substr(start, p - start));
        start = p + 1;
    }
    return out;
}

}  // namespace

extern "C" {

/* ---- WAVLoader.load (WAVLoader.swift:26-99) ------------------------------------------------------ */
aw_status aw_wav_load(const char *path, aw_wav **out) try {
    if (!out) return fail(AW_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (!path) return fail(AW_ERR_INVALID_ARGUMENT, "path is NULL");
    FILE *f = std::fopen(path, "rb");
    if (!f) return fail(AW_ERR_WAV_FILE_READ, std::string("Failed to open WAV file: ") + path);
    std::vector<unsigned char> b;
    unsigned char chunk[65536];
    size_t n;
    while ((n = std::fread(chunk, 1, sizeof(chunk), f)) > 0) b.insert(b.end(), chunk, chunk + n);
    std::fclose(f);
    if (b.size() < 12 || std::memcmp(b.data(), "RIFF", 4) != 0 || std::memcmp(b.data() + 8, "WAVE", 4) != 0)
        return fail(AW_ERR_WAV_FILE_READ, "Failed to open WAV file: not a RIFF/WAVE file");
    const unsigned char *fmt = nullptr, *data = nullptr;
    size_t fmt_len = 0, data_len = 0, pos = 12;
    while (pos + 8 <= b.size()) {
        const uint32_t sz = rd32(&b[pos + 4]);
        const size_t avail = std::min<size_t>(sz, b.size() - pos - 8);
        if (std::memcmp(&b[pos], "fmt ", 4) == 0) { fmt = &b[pos + 8]; fmt_len = avail; }
        else if (std::memcmp(&b[pos], "data", 4) == 0) { data = &b[pos + 8]; data_len = avail; }
        pos += 8 + (size_t)sz + (sz & 1u);
    }
    if (!fmt || !data || fmt_len < 16) return fail(AW_ERR_WAV_FILE_READ, "Failed to read audio data: missing fmt/data chunk");
    int tag = rd16(fmt);
    const int ch = rd16(fmt + 2);
    const uint32_t rate = rd32(fmt + 4);
    const int bits = rd16(fmt + 14);
    if (tag == 0xFFFE && fmt_len >= 26) tag = rd16(fmt + 24);     // WAVE_FORMAT_EXTENSIBLE: SubFormat GUID's first word
    if (ch <= 0) return fail(AW_ERR_INVALID_CHANNEL_COUNT, "Invalid channel count: " + std::to_string(ch));   // :40-42
    const int bps = bits / 8;
    const size_t frames = bps > 0 ? data_len / ((size_t)bps * ch) : 0;
    if (frames == 0) return fail(AW_ERR_WAV_EMPTY_FILE, "WAV file is empty (0 frames)");                     // :44-46
    const bool is_float = tag == 3, is_pcm = tag == 1;
    if (!((is_float && (bits == 32 || bits == 64)) || (is_pcm && (bits == 8 || bits == 16 || bits == 24 || bits == 32))))
        return fail(AW_ERR_WAV_UNSUPPORTED_FORMAT, "Unsupported WAV format");                                // :89-91
    awr::Owner<aw_wav> w(new (std::nothrow) aw_wav(), aw_wav_destroy);
    if (!w) return fail(AW_ERR_OUT_OF_MEMORY, "Failed to allocate audio buffer");                            // :49-54
    w->sample_rate = (double)rate; w->channels = ch; w->frames = (int)frames;
    w->planar.resize((size_t)ch * frames);
    for (size_t i = 0; i < frames; ++i) {
        for (int c = 0; c < ch; ++c) {
            const unsigned char *p = data + (i * ch + c) * bps;
            float v;
            if (is_float && bits == 32) { uint32_t u = rd32(p); std::memcpy(&v, &u, 4); }
            else if (is_float) { uint64_t u = (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); double d; std::memcpy(&d, &u, 8); v = (float)d; }
            else if (bits == 16) v = (float)(int16_t)rd16(p) / 32768.0f;                                      // :77
            else if (bits == 32) v = (float)((double)(int32_t)rd32(p) / 2147483648.0);                        // :86
            else if (bits == 24) { int32_t s = (int32_t)(p[0] | (p[1] << 8) | (p[2] << 16)); if (s & 0x800000) s -= 0x1000000; v = (float)((double)s / 8388608.0); }
            else v = ((float)p[0] - 128.0f) / 128.0f;
            w->planar[(size_t)c * frames + i] = v;
        }
    }
    *out = w.release();
    return AW_OK;
} AW_NOEXCEPT_TAIL
void aw_wav_destroy(aw_wav *w) { delete w; }
double aw_wav_sample_rate(const aw_wav *w) { return w ? w->sample_rate : 0.0; }
int32_t aw_wav_channel_count(const aw_wav *w) { return w ? w->channels : 0; }
int32_t aw_wav_frame_count(const aw_wav *w) { return w ? w->frames : 0; }
const float *aw_wav_channel(const aw_wav *w, int32_t c) {
    return (w && c >= 0 && c < w->channels) ? w->planar.data() + (size_t)c * w->frames : nullptr;
}
const float *aw_wav_planar(const aw_wav *w) { return w ? w->planar.data() : nullptr; }

/* ---- InputLayout (VirtualSpeaker.swift:59-100) ---------------------------------------------------- */
aw_status aw_layout_detect(int32_t n, aw_layout **out) try {
    if (!out) return fail(AW_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (n < 0) return fail(AW_ERR_INVALID_ARGUMENT, "negative channel count");
    awr::Owner<aw_layout> l(new (std::nothrow) aw_layout(), aw_layout_destroy);
    if (!l) return fail(AW_ERR_OUT_OF_MEMORY, "layout");
    switch (n) {                                                               // :88-99
        case 2: l->speakers = {"FL", "FR"}; l->name = "Stereo"; break;
        case 6: l->speakers = {"FL", "FR", "FC", "LFE", "BL", "BR"}; l->name = "5.1 Surround"; break;
        case 8: l->speakers = {"FL", "FR", "FC", "LFE", "BL", "BR", "SL", "SR"}; l->name = "7.1 Surround"; break;
        case 12: l->speakers = {"FL", "FR", "FC", "LFE", "BL", "BR", "SL", "SR", "TFL", "TFR", "TBL", "TBR"}; l->name = "7.1.4 Atmos"; break;
        default:
            for (int i = 0; i < n; ++i) l->speakers.push_back("Ch" + std::to_string(i));
            l->name = std::to_string(n) + " Channel";
    }
    *out = l.release();
    return AW_OK;
} AW_NOEXCEPT_TAIL
aw_status aw_layout_create(const char *const *names, int32_t count, const char *name, aw_layout **out) try {
    if (!out) return fail(AW_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (count < 0 || (count > 0 && !names)) return fail(AW_ERR_INVALID_ARGUMENT, "bad speaker list");
    awr::Owner<aw_layout> l(new (std::nothrow) aw_layout(), aw_layout_destroy);
    if (!l) return fail(AW_ERR_OUT_OF_MEMORY, "layout");
    for (int i = 0; i < count; ++i) {
        if (!names[i]) return fail(AW_ERR_INVALID_ARGUMENT, "NULL speaker name");
        l->speakers.emplace_back(names[i]);
    }
    l->name = name ? name : "";
    *out = l.release();
    return AW_OK;
} AW_NOEXCEPT_TAIL
void aw_layout_destroy(aw_layout *l) { delete l; }
int32_t aw_layout_count(const aw_layout *l) { return l ? (int32_t)l->speakers.size() : 0; }
const char *aw_layout_speaker(const aw_layout *l, int32_t i) {
    return (l && i >= 0 && i < (int32_t)l->speakers.size()) ? l->speakers[(size_t)i].c_str() : nullptr;
}
const char *aw_layout_name(const aw_layout *l) { return l ? l->name.c_str() : nullptr; }

/* ---- HRIRChannelMap (VirtualSpeaker.swift:103-347) -------------------------------------------------- */
static aw_status map_new(aw_channel_map **out, aw_channel_map **m) {
    if (!out) return fail(AW_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    *m = new (std::nothrow) aw_channel_map();
    if (!*m) return fail(AW_ERR_OUT_OF_MEMORY, "channel map");
    return AW_OK;
}

aw_status aw_map_hesuvi14(const aw_layout *spk, aw_channel_map **out) try {       // :270-297
    aw_channel_map *m;
    aw_status st = map_new(out, &m);
    if (st != AW_OK) return st;
    awr::Owner<aw_channel_map> owner(m, aw_map_destroy);
    if (!spk) return fail(AW_ERR_INVALID_ARGUMENT, "speakers is NULL");
    static const struct { const char *s; int l, r; } T[] = {{"FL", 0, 1}, {"FR", 8, 7}, {"FC", 6, 13}, {"LFE", 6, 13},
                                                             {"BL", 4, 5}, {"BR", 12, 11}, {"SL", 2, 3}, {"SR", 10, 9}};
    for (const auto &s : spk->speakers)
        for (const auto &t : T)
            if (s == t.s) m->set(s, t.l, t.r);
    *out = owner.release();
    return AW_OK;
} AW_NOEXCEPT_TAIL

aw_status aw_map_hesuvi7(const aw_layout *spk, aw_channel_map **out) try {        // :224-250
    aw_channel_map *m;
    aw_status st = map_new(out, &m);
    if (st != AW_OK) return st;
    awr::Owner<aw_channel_map> owner(m, aw_map_destroy);
    if (!spk) return fail(AW_ERR_INVALID_ARGUMENT, "speakers is NULL");
    static const struct { const char *s; int l, r; } T[] = {{"FL", 0, 1}, {"FR", 1, 0}, {"FC", 2, 2}, {"LFE", 2, 2},
                                                             {"BL", 3, 4}, {"BR", 4, 3}, {"SL", 5, 6}, {"SR", 6, 5}};
    for (const auto &s : spk->speakers)
        for (const auto &t : T)
            if (s == t.s) m->set(s, t.l, t.r);
    *out = owner.release();
    return AW_OK;
} AW_NOEXCEPT_TAIL

aw_status aw_map_interleaved_pairs(const aw_layout *spk, aw_channel_map **out) try {   // :126-159
    aw_channel_map *m;
    aw_status st = map_new(out, &m);
    if (st != AW_OK) return st;
    awr::Owner<aw_channel_map> owner(m, aw_map_destroy);
    if (!spk) return fail(AW_ERR_INVALID_ARGUMENT, "speakers is NULL");
    for (size_t i = 0; i < spk->speakers.size(); ++i) {
        const int base = (int)i * 2;
        const std::string &s = spk->speakers[i];
        if (in_set(s, kRightSide, 6)) m->set(s, base + 1, base);               // right side: pair swapped
        else m->set(s, base, base + 1);                                          // left side and centre speakers
        (void)kLeftSide;
    }
    *out = owner.release();
    return AW_OK;
} AW_NOEXCEPT_TAIL

aw_status aw_map_split_blocks(const aw_layout *spk, aw_channel_map **out) try {    // :200-209
    aw_channel_map *m;
    aw_status st = map_new(out, &m);
    if (st != AW_OK) return st;
    awr::Owner<aw_channel_map> owner(m, aw_map_destroy);
    if (!spk) return fail(AW_ERR_INVALID_ARGUMENT, "speakers is NULL");
    const int n = (int)spk->speakers.size();
    for (int i = 0; i < n; ++i) m->set(spk->speakers[(size_t)i], i, i + n);
    *out = owner.release();
    return AW_OK;
} AW_NOEXCEPT_TAIL

aw_status aw_map_parse_text(const char *text, aw_channel_map **out) try {          // parseHeSuViFormat :301-346
    aw_channel_map *m;
    aw_status st = map_new(out, &m);
    if (st != AW_OK) return st;
    awr::Owner<aw_channel_map> owner(m, aw_map_destroy);
    if (!text) return fail(AW_ERR_INVALID_ARGUMENT, "text is NULL");
    static const struct { const char *alias; const char *spk; } A[] = {
        {"FL", "FL"}, {"L", "FL"}, {"FR", "FR"}, {"R", "FR"}, {"FC", "FC"}, {"C", "FC"}, {"LFE", "LFE"}, {"SUB", "LFE"},
        {"BL", "BL"}, {"RL", "BL"}, {"BR", "BR"}, {"RR", "BR"}, {"SL", "SL"}, {"SR", "SR"}, {"TFL", "TFL"},
        {"TFR", "TFR"}, {"TBL", "TBL"}, {"TBR", "TBR"}};
    for (const std::string &raw : split_newlines(text)) {
        const std::string line = trim_ws(raw);
        if (line.empty() || line[0] == '#' || line[0] == ';') continue;        // :308-311
        const auto parts = split(line, '=');
        if (parts.size() != 2) continue;                                         // :315
        const std::string name = trim_ws(parts[0]);
        std::vector<int> idx;
        for (const std::string &tok : split(trim_ws(parts[1]), ',')) {
            int v;
            if (swift_int(trim_ws(tok), &v)) idx.push_back(v);                    // compactMap { Int(...) }  :320
        }
        if (idx.size() != 2) continue;                                           // :322
        const std::string upper = alias_upper(name);                             // :325
        std::string speaker = name;                                              // default: .custom(speakerName)  :339
        for (const auto &a : A)
            if (upper == a.alias) { speaker = a.spk; break; }
        m->set(speaker, idx[0], idx[1]);
    }
    *out = owner.release();
    return AW_OK;
} AW_NOEXCEPT_TAIL

void aw_map_destroy(aw_channel_map *m) { delete m; }
int32_t aw_map_count(const aw_channel_map *m) { return m ? (int32_t)m->keys.size() : 0; }
int32_t aw_map_get(const aw_channel_map *m, const char *speaker, int32_t *l, int32_t *r) {
    if (!m || !speaker) return 0;
    auto it = m->idx.find(speaker);
    if (it == m->idx.end()) return 0;
    if (l) *l = it->second.first;
    if (r) *r = it->second.second;
    return 1;
}

aw_status aw_map_resolve(const aw_channel_map *m, const aw_layout *layout, int32_t n_tracks, int32_t *left,
                         int32_t *right) try {
    if (!m || !layout || !left || !right) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
    int mapped = 0;
    for (size_t i = 0; i < layout->speakers.size(); ++i) {
        left[i] = right[i] = -1;
        auto it = m->idx.find(layout->speakers[i]);
        if (it == m->idx.end()) continue;                                        // HRIRManager.swift:370-372
        const int l = it->second.first, r = it->second.second;
        if (!(l < n_tracks && r < n_tracks) || l < 0 || r < 0)                   // :375-379
            return fail(AW_ERR_INVALID_CHANNEL_MAPPING, "HRIR indices (" + std::to_string(l) + ", " + std::to_string(r) +
                                                            ") out of range for " + std::to_string(n_tracks) + " channels");
        left[i] = l; right[i] = r;
        ++mapped;
    }
    if (mapped == 0) return fail(AW_ERR_CONVOLUTION_SETUP_FAILED, "No valid renderers created");   // :420-422
    return AW_OK;
} AW_NOEXCEPT_TAIL

/* ---- Resampler (Resampler.swift:31-68) -------------------------------------------------------------- */
int32_t aw_resample_output_count(int32_t count, double from_rate, double to_rate) {
    if (std::fabs(from_rate - to_rate) < 0.01) return count;                   // :33-35
    const double stride = from_rate / to_rate;                                  // :37
    return (int32_t)((double)count / stride);                                   // :38
}

aw_status aw_resample(const float *input, int32_t count, double from_rate, double to_rate, float *output,
                      int32_t capacity, int32_t *output_count) try {
    if (!input || !output || !output_count || count < 0) return fail(AW_ERR_INVALID_ARGUMENT, "bad argument");
    const int n_out = aw_resample_output_count(count, from_rate, to_rate);
    *output_count = n_out > 0 ? n_out : 0;
    if (n_out <= 0) return AW_OK;                                               // guard outputCount > 0 else { return [] }
    if (capacity < n_out) return fail(AW_ERR_INVALID_ARGUMENT, "output capacity too small");
    if (std::fabs(from_rate - to_rate) < 0.01) { std::memcpy(output, input, sizeof(float) * (size_t)count); return AW_OK; }
    const float step = (float)(from_rate / to_rate);                            // var step: Float = Float(stride)  :55
    for (int i = 0; i < n_out; ++i) {
        const float pos = (float)i * step;                                      // vDSP_vramp(start 0, step)  :56
        long long i0 = (long long)std::floor(pos);
        const float frac = pos - (float)i0;
        long long a = std::min<long long>(i0, count - 1), b = std::min<long long>(i0 + 1, count - 1);
        output[i] = input[a] + frac * (input[b] - input[a]);
    }
    return AW_OK;
} AW_NOEXCEPT_TAIL

// vDSP_vgenp as documented, on the control ramp of Resampler.swift:53-56 (float32, C[m] = m * step).  The ramp the
// reference allocates has only outputCount entries while vgenp is told M = input.count knots; every evaluation point
// n < outputCount is bracketed by knots below outputCount when stride > 1, so the entries past the array are never needed.
aw_status aw_resample_vgenp(const float *input, int32_t count, double from_rate, double to_rate, float *output,
                            int32_t capacity, int32_t *output_count) try {
    if (!input || !output || !output_count || count < 0) return fail(AW_ERR_INVALID_ARGUMENT, "bad argument");
    const int n_out = aw_resample_output_count(count, from_rate, to_rate);
    *output_count = n_out > 0 ? n_out : 0;
    if (n_out <= 0) return AW_OK;
    if (capacity < n_out) return fail(AW_ERR_INVALID_ARGUMENT, "output capacity too small");
    if (std::fabs(from_rate - to_rate) < 0.01) { std::memcpy(output, input, sizeof(float) * (size_t)count); return AW_OK; }
    const float step = (float)(from_rate / to_rate);
    auto knot = [&](long long m) { return (float)m * step; };          // B[m]
    const long long M = count;
    long long m = 0;                                                    // invariant: B[m] < n (for n >= 1)
    for (int n = 0; n < n_out; ++n) {
        const float x = (float)n;
        if (x <= knot(0)) { output[n] = input[0]; continue; }
        if (x > knot(M - 1)) { output[n] = input[M - 1]; continue; }
        while (m + 1 < M - 1 && knot(m + 1) < x) ++m;                   // B[m] < n <= B[m+1]
        const float b0 = knot(m), b1 = knot(m + 1);
        output[n] = input[m] + (input[m + 1] - input[m]) * ((x - b0) / (b1 - b0));
    }
    return AW_OK;
} AW_NOEXCEPT_TAIL

/* ---- HRIRManager.activatePreset (HRIRManager.swift:347-446) ------------------------------------------ */
aw_status aw_preset_activate(aw_context *ctx, const char *wav_path, double target_rate, const aw_layout *layout,
                             const aw_channel_map *custom_map, int32_t n_streams, aw_spatializer **sp_out,
                             aw_hrir **hrir_out) try {
    if (!sp_out) return fail(AW_ERR_INVALID_ARGUMENT, "spatializer_out is NULL");
    *sp_out = nullptr;
    if (hrir_out) *hrir_out = nullptr;
    if (!ctx || !wav_path || !layout) return fail(AW_ERR_INVALID_ARGUMENT, "NULL argument");
    aw_wav *wav_raw = nullptr;
    aw_status st = aw_wav_load(wav_path, &wav_raw);                              // :349
    if (st != AW_OK) return st;
    const awr::Owner<aw_wav> wav(wav_raw, aw_wav_destroy);
    awr::Owner<aw_channel_map> own(nullptr, aw_map_destroy);
    const aw_channel_map *map = custom_map;
    if (!map) {                                                                   // :355-360
        aw_channel_map *made = nullptr;
        st = wav->channels == 7 ? aw_map_hesuvi7(layout, &made) : aw_map_hesuvi14(layout, &made);
        if (st != AW_OK) return st;
        own.reset(made);
        map = made;
    }
    const int C = (int)layout->speakers.size();
    std::vector<int32_t> lt((size_t)std::max(C, 1)), rt((size_t)std::max(C, 1));
    st = aw_map_resolve(map, layout, wav->channels, lt.data(), rt.data());        // :366-379, :420-422
    if (st != AW_OK) return st;
    std::vector<float> tracks;
    int taps = wav->frames;
    if (target_rate > 0.0 && std::fabs(wav->sample_rate - target_rate) > 0.01) {  // :389-403
        taps = aw_resample_output_count(wav->frames, wav->sample_rate, target_rate);
        if (taps <= 0) return fail(AW_ERR_CONVOLUTION_SETUP_FAILED, "resampled HRIR is empty");
        tracks.resize((size_t)wav->channels * taps);
        for (int c = 0; c < wav->channels; ++c) {
            int n = 0;
            st = (awr::context_literal_resampler(ctx) ? aw_resample_vgenp : aw_resample)(aw_wav_channel(wav.get(), c), wav->frames, wav->sample_rate, target_rate,
                                                                                            tracks.data() + (size_t)c * taps, taps, &n);
            if (st != AW_OK) return st;
        }
    } else {
        tracks = wav->planar;
    }
    aw_hrir *hrir_raw = nullptr;
    st = aw_hrir_create(ctx, tracks.data(), wav->channels, taps, target_rate > 0.0 ? target_rate : wav->sample_rate, &hrir_raw);
    if (st != AW_OK) return st;
    awr::Owner<aw_hrir> hrir(hrir_raw, aw_hrir_destroy);
    st = aw_spatializer_create(ctx, hrir.get(), C, lt.data(), rt.data(), n_streams, 0, sp_out);   // :406-418
    if (st == AW_OK && hrir_out) *hrir_out = hrir.release();
    return st;
} AW_NOEXCEPT_TAIL

}  // extern "C"
