#include "tables.hpp"

#include <cmath>
#include <complex>
#include <algorithm>
#include <atomic>
#include <functional>
#include <thread>

namespace awh {

using cd = std::complex<double>;

static void fft_inplace(std::vector<cd> &a) {
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * M_PI / (double)len;
        for (size_t i = 0; i < n; i += len) {
            for (size_t k = 0; k < len / 2; ++k) {
                const cd w(std::cos(ang * (double)k), std::sin(ang * (double)k));
                const cd u = a[i + k], v = a[i + k + len / 2] * w;
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
        }
    }
}

// fn(0) .. fn(n - 1) on up to n host threads (fn(0) on the calling one).  Nothing throws out of here — the library is a C ABI: a worker
// records its failure, every started thread is joined on every path, and a thread that cannot be started has its index run inline.
// false = some fn threw (std::bad_alloc in practice).
bool parallel_for(int n, const std::function<void(int)> &fn) {
    std::atomic<bool> failed{false};
    auto guarded = [&](int i) {
        try { fn(i); } catch (...) { failed = true; }
    };
    {
        struct Joiner {
            std::vector<std::thread> t;
            ~Joiner() { for (auto &w : t) if (w.joinable()) w.join(); }
        } workers;
        try { workers.t.reserve((size_t)std::max(0, n - 1)); } catch (...) { failed = true; }
        for (int i = 1; i < n; ++i) {
            try { workers.t.emplace_back(guarded, i); }
            catch (...) { guarded(i); }
        }
        if (n > 0) guarded(0);
    }
    return !failed.load();
}

static awk::cf unit(double num, double den) {   // exp(-2 pi i num / den)
    const double a = -2.0 * M_PI * num / den;
    return awk::mk((float)std::cos(a), (float)std::sin(a));
}

void build_twiddles(Twiddles &tw) {
    tw.tw1.resize(512);
    for (int t = 0; t < 512; ++t) tw.tw1[t] = unit((double)t, awk::kN);
    tw.twa.resize(awk::kTwaElems);
    for (int ka = 0; ka < 8; ++ka)
        for (int l = 0; l < 64; ++l) tw.twa[ka * 64 + l] = unit((double)l * ka, 512.0);
    tw.twb.resize(awk::kTwbElems);
    for (int kb = 0; kb < 8; ++kb)
        for (int l0 = 0; l0 < 8; ++l0) tw.twb[kb * 8 + l0] = unit((double)l0 * kb, 64.0);
}

bool build_pair_tables(const float *tracks, int n_tracks, int taps, int n_channels,
                       const int32_t *left_track, const int32_t *right_track, int tap_offset,
                       int tap_count, std::vector<awk::cf2> &out, bool pair_threads) {
    const int N = awk::kN;
    const int n_pairs = (n_channels + 1) / 2;
    out.assign((size_t)n_pairs * N, awk::cf2{awk::mk(0, 0), awk::mk(0, 0)});
    const double scale = 1.0 / (2.0 * (double)N);
    auto tap = [&](int track, int i) -> double {
        if (track < 0 || track >= n_tracks) return 0.0;
        const int idx = tap_offset + i;
        return (idx < taps && i < tap_count) ? (double)tracks[(size_t)track * taps + idx] : 0.0;
    };
    // the pairs write disjoint table entries: one host thread each when the caller allows it (aw_spatializer_create runs the PARTITIONS of a
    // long HRIR side by side instead, one thread per partition)
    auto one_pair = [&](int p) {
        std::vector<cd> zl(N), zr(N);
        const int a = 2 * p, b = 2 * p + 1;
        const int la = left_track[a], ra = right_track[a];
        const int lb = b < n_channels ? left_track[b] : -1, rb = b < n_channels ? right_track[b] : -1;
        // a channel is rendered only when BOTH ears are mapped (the reference looks up a pair or skips)
        const bool use_a = la >= 0 && ra >= 0, use_b = lb >= 0 && rb >= 0;
        for (int i = 0; i < N; ++i) {
            zl[i] = cd(use_a ? tap(la, i) : 0.0, use_b ? tap(lb, i) : 0.0);   // h_aL + i h_bL
            zr[i] = cd(use_a ? tap(ra, i) : 0.0, use_b ? tap(rb, i) : 0.0);   // h_aR + i h_bR
        }
        fft_inplace(zl);
        fft_inplace(zr);
        const cd I(0.0, 1.0);
        for (int k = 0; k < N; ++k) {
            const int nk = (N - k) & (N - 1);
            const cd g1l = std::conj(zl[nk]) * scale, g1r = std::conj(zr[nk]) * scale;   // (H_a - i H_b)/(2N)
            const cd g2l = zl[k] * scale, g2r = zr[k] * scale;                             // (H_a + i H_b)/(2N)
            const cd A = g1l + I * g1r, B = g2l + I * g2r;
            const int k1 = k & 15, k2 = k >> 4;
            awk::cf2 &e = out[(size_t)p * N + (size_t)k1 * awk::kSub + k2];
            e.a = awk::mk((float)A.real(), (float)A.imag());
            e.b = awk::mk((float)B.real(), (float)B.imag());
        }
    };
    if (pair_threads && n_pairs > 1) return parallel_for(n_pairs, one_pair);
    for (int p = 0; p < n_pairs; ++p) one_pair(p);
    return true;
}

bool build_poly_tables(const float *tracks, int n_tracks, int taps, int n_channels, const int32_t *left_track,
                       const int32_t *right_track, std::vector<awk::cf4> &out) {
    const int N = awk::kN;
    const int Lh = taps / 2 + 1;
    // polyphase bank [track][3][Lh]: 0: h[2j]   1: h[2j+1]   2: h[2j-1] (j >= 1)
    std::vector<float> bank((size_t)n_tracks * 3 * Lh, 0.f);
    for (int tr = 0; tr < n_tracks; ++tr) {
        const float *h = tracks + (size_t)tr * taps;
        float *he = &bank[((size_t)tr * 3 + 0) * Lh], *ho = he + Lh, *hd = ho + Lh;
        for (int j = 0; j < Lh; ++j) {
            if (2 * j < taps) he[j] = h[2 * j];
            if (2 * j + 1 < taps) ho[j] = h[2 * j + 1];
            if (j >= 1 && 2 * j - 1 < taps) hd[j] = h[2 * j - 1];
        }
    }
    const int C2 = 2 * n_channels, n_pairs = (C2 + 1) / 2;
    std::vector<awk::cf2> tab[2];
    for (int o = 0; o < 2; ++o) {
        std::vector<int32_t> l2(C2), r2(C2);
        for (int cp = 0; cp < C2; ++cp) {
            const int par = cp >= n_channels ? 1 : 0, c = cp - par * n_channels;
            const int sel = o == 0 ? (par == 0 ? 0 : 2) : (par == 0 ? 1 : 0);
            l2[cp] = left_track[c] < 0 || left_track[c] >= n_tracks ? -1 : left_track[c] * 3 + sel;
            r2[cp] = right_track[c] < 0 || right_track[c] >= n_tracks ? -1 : right_track[c] * 3 + sel;
        }
        if (!build_pair_tables(bank.data(), n_tracks * 3, Lh, C2, l2.data(), r2.data(), 0, Lh, tab[o], /*pair_threads=*/true)) return false;
    }
    out.resize((size_t)n_pairs * N);
    for (size_t i = 0; i < out.size(); ++i) { out[i].e = tab[0][i]; out[i].o = tab[1][i]; }
    return true;
}

// FFT with a precomputed twiddle table tw[k] = exp(-2 pi i k / n), k < n/2 (every entry one cos/sin in double)
static void fft_inplace_tw(std::vector<cd> &a, const std::vector<cd> &tw) {
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const size_t stride = n / len;
        for (size_t i = 0; i < n; i += len) {
            for (size_t k = 0; k < len / 2; ++k) {
                const cd u = a[i + k], v = a[i + k + len / 2] * tw[k * stride];
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
        }
    }
}

// DFT of any length n = 2^a q, q in {1, 3, 5, 7} (the window lengths N = 8 RA x 4096 of the long-window path): q interleaved
// power-of-two transforms, then X[k] = sum_{n1 < q} w_n^{n1 k} Sub_{n1}[k mod (n / q)].  unit_n[j] = exp(-2 pi i j / n), j < n.
static void fft_any(std::vector<cd> &a, const std::vector<cd> &unit_n) {
    const size_t n = a.size();
    size_t q = n;
    while (q % 2 == 0) q /= 2;
    const size_t P = n / q;
    std::vector<cd> twP(P / 2);
    for (size_t k = 0; k < P / 2; ++k) twP[k] = unit_n[k * q];
    if (q == 1) { fft_inplace_tw(a, twP); return; }
    std::vector<std::vector<cd>> sub(q, std::vector<cd>(P));
    for (size_t n1 = 0; n1 < q; ++n1) {
        for (size_t m = 0; m < P; ++m) sub[n1][m] = a[q * m + n1];
        fft_inplace_tw(sub[n1], twP);
    }
    for (size_t k = 0; k < n; ++k) {
        cd acc = sub[0][k % P];
        for (size_t n1 = 1; n1 < q; ++n1) acc += unit_n[(n1 * k) % n] * sub[n1][k % P];
        a[k] = acc;
    }
}

// false: host memory ran out (a worker's or this thread's std::bad_alloc, or no thread could be started) — nothing throws across the
// C ABI, every worker is joined on every path, and the caller maps the failure to AW_ERR_OUT_OF_MEMORY.
static void build_lw_tables_impl(const float *tracks, int n_tracks, int taps, int n_channels, const int32_t *left_track,
                                 const int32_t *right_track, int R, LwTables &out, int rows_form, bool filters, std::atomic<bool> &failed);

bool build_lw_tables(const float *tracks, int n_tracks, int taps, int n_channels, const int32_t *left_track,
                     const int32_t *right_track, int R, LwTables &out, int rows_form, bool filters) {
    std::atomic<bool> failed{false};
    try {
        build_lw_tables_impl(tracks, n_tracks, taps, n_channels, left_track, right_track, R, out, rows_form, filters, failed);
    } catch (...) {
        failed = true;
    }
    return !failed.load();
}

static void build_lw_tables_impl(const float *tracks, int n_tracks, int taps, int n_channels, const int32_t *left_track,
                                 const int32_t *right_track, int R, LwTables &out, int rows_form, bool filters, std::atomic<bool> &failed) {
    const int M = awk::kLwM;
    const size_t N = (size_t)R * M;
    const int n_pairs = (n_channels + 1) / 2;
    const bool real_last = (n_channels & 1) != 0;
    // twiddles
    const int RA = R / 8;
    out.coarse.resize((size_t)RA * 64);
    out.fine.resize((size_t)RA * 64);
    for (int ka = 0; ka < RA; ++ka)
        for (int i = 0; i < 64; ++i) {
            out.coarse[(size_t)ka * 64 + i] = unit(64.0 * i * (2 * ka + 1), 2.0 * (double)N);
            out.fine[(size_t)ka * 64 + i] = unit((double)i * (2 * ka + 1), 2.0 * (double)N);
        }
    out.step.resize((size_t)3 * M);
    for (int m = 1; m <= 3; ++m)
        for (int t = 0; t < M; ++t) out.step[(size_t)(m - 1) * M + t] = unit(2.0 * RA * m * (double)t, 2.0 * (double)N);
    out.tw_r.resize((size_t)RA * 8);
    for (int ka = 0; ka < RA; ++ka)
        for (int j1 = 0; j1 < 8; ++j1) out.tw_r[(size_t)ka * 8 + j1] = unit((double)j1 * (2 * ka + 1), 2.0 * R);
    out.tw1m.resize(512);
    for (int t = 0; t < 512; ++t) out.tw1m[t] = unit((double)t, (double)M);
    out.tab.clear(); out.tab16.clear(); out.tw2.clear();
    if (rows_form == 16) {
        out.tw2.resize(256);
        for (int m0 = 0; m0 < 16; ++m0)
            for (int a = 0; a < 16; ++a) out.tw2[(size_t)m0 * 16 + a] = unit((double)a * m0, 256.0);
    }
    if (!filters) return;
    // filter tables
    std::vector<cd> unit_n(N), mod(N);
    for (size_t k = 0; k < N; ++k) { const double a = -2.0 * M_PI * (double)k / (double)N; unit_n[k] = cd(std::cos(a), std::sin(a)); }
    for (size_t n = 0; n < N; ++n) { const double a = -M_PI * (double)n / (double)N; mod[n] = cd(std::cos(a), std::sin(a)); }     // w_N^{n/2}
    const bool form16 = rows_form == 16;
    if (form16) {
        out.tab16.assign((size_t)(R / 2) * n_pairs * 2 * M, awk::LwTab2{awk::mk(0, 0), awk::mk(0, 0)});
    } else {
        out.tab.assign((size_t)(R / 2) * n_pairs * M, awk::LwTab{awk::mk(0, 0), awk::mk(0, 0), awk::mk(0, 0), awk::mk(0, 0)});
    }
    std::vector<int> pos16(M, 0);                  // row bin k2 -> m1 * 256 + thread
    for (int m1 = 0; m1 < 16; ++m1)
        for (int th = 0; th < awk::kR16Threads; ++th) pos16[awk::r16_bin(th, m1)] = m1 * awk::kR16Threads + th;
    const double scale = 1.0 / (2.0 * (double)N);
    auto tap = [&](int track, size_t i) -> double {
        if (track < 0 || track >= n_tracks) return 0.0;
        return i < (size_t)taps ? (double)tracks[(size_t)track * taps + i] : 0.0;
    };
    const cd I(0.0, 1.0);
    auto c32 = [](cd v) { return awk::mk((float)v.real(), (float)v.imag()); };
    // one host thread per channel pair (two N-point transforms each; the pairs write disjoint table entries): the analogue of the
    // per-engine partition FFTs of ConvolutionEngine.init (Airwave/ConvolutionEngine.swift:143-182), which the reference also runs off
    // the audio thread
    auto build_pair = [&](int p) {
        std::vector<cd> zl(N), zr(N);
        const int a = 2 * p, b = 2 * p + 1;
        const int la = left_track[a], ra_ = right_track[a];
        const int lb = b < n_channels ? left_track[b] : -1, rb_ = b < n_channels ? right_track[b] : -1;
        const bool use_a = la >= 0 && ra_ >= 0, use_b = lb >= 0 && rb_ >= 0;       // both ears mapped, like build_pair_tables
        for (size_t i = 0; i < N; ++i) {
            zl[i] = i < (size_t)taps ? cd(use_a ? tap(la, i) : 0.0, use_b ? tap(lb, i) : 0.0) * mod[i] : cd(0.0, 0.0);
            zr[i] = i < (size_t)taps ? cd(use_a ? tap(ra_, i) : 0.0, use_b ? tap(rb_, i) : 0.0) * mod[i] : cd(0.0, 0.0);
        }
        fft_any(zl, unit_n);
        fft_any(zr, unit_n);
        const bool fold = real_last && p == n_pairs - 1;
        for (int rp = 0; rp < R / 2; ++rp) {
            for (int k2 = 0; k2 < M; ++k2) {
                const size_t k = (size_t)rp + (size_t)R * k2, kp = N - 1 - k;
                const cd Ak = (std::conj(zl[kp]) + I * std::conj(zr[kp])) * scale, Bk = (zl[k] + I * zr[k]) * scale;
                const cd Akp = (std::conj(zl[k]) + I * std::conj(zr[k])) * scale, Bkp = (zl[kp] + I * zr[kp]) * scale;
                cd t0 = Ak, t1 = Bk, t2 = std::conj(Akp), t3 = std::conj(Bkp);
                if (fold) { t0 += t1; t3 += t2; t1 = cd(0, 0); t2 = cd(0, 0); }
                if (form16) {
                    const size_t base = ((size_t)rp * n_pairs + p) * 2 * M + pos16[k2];
                    out.tab16[base] = awk::LwTab2{c32(t0), c32(t3)};
                    out.tab16[base + M] = awk::LwTab2{c32(t1), c32(t2)};
                } else {
                    const int q1 = k2 & 7, q2 = k2 >> 3;
                    awk::LwTab &e = out.tab[(((size_t)rp * n_pairs + p) * awk::kLwInner + q1) * awk::kSub + q2];
                    e.t0 = c32(t0); e.t1 = c32(t1); e.t2 = c32(t2); e.t3 = c32(t3);
                }
            }
        }
    };
    // A worker never lets an exception escape (std::terminate under a C ABI): it records the failure instead.  The guard joins
    // whatever was started on every path — also when build_pair(0) on this thread, or starting a thread, throws.
    auto guarded = [&](int p) {
        try { build_pair(p); } catch (...) { failed = true; }
    };
    struct Joiner {
        std::vector<std::thread> t;
        ~Joiner() { for (auto &w : t) if (w.joinable()) w.join(); }
    } workers;
    workers.t.reserve((size_t)std::max(0, n_pairs - 1));
    for (int p = 1; p < n_pairs; ++p) {
        try { workers.t.emplace_back(guarded, p); }
        catch (...) { guarded(p); }                  // no thread to be had (std::system_error): build that pair here
    }
    guarded(0);
}

}  // namespace awh
