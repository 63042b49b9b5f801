// eq_cascade.hpp — parametric EQ biquad cascade for batches of stereo streams.
//
// Replaces ParametricEqualizerState.process (Airwave/ParametricEqualizerProcessor.swift:58-91): a
// cascade of up to 64 transposed-direct-form-II biquads in Float64 behind a linear preamp.  The
// reference walks one stereo stream sample by sample; a recurrence of 480 k dependent steps per
// stream is a non-starter on a GPU, so the time axis is parallelised exactly (in exact arithmetic):
//
//   * one workgroup per stream walks the call in spans of kEqThreads x kEqChunk frames;
//   * every thread owns one chunk of kEqChunk consecutive frames of BOTH ears in registers and, per
//     filter, (1) runs the biquad from a ZERO state over its chunk (zero-state response) and keeps
//     the end state e[c]; (2) the true state entering chunk c obeys s[c+1] = P s[c] + e[c] with
//     P = M^kEqChunk (M = the filter's zero-input state matrix [[-a1, 1], [-a2, 0]]), a linear
//     recurrence over chunks that is solved with a two-level scan: Hillis-Steele inside each wave
//     with the precomputed powers P^(2^j) (wave-private LDS slots, no workgroup barrier), then the
//     four wave totals are chained with P^64 and applied per lane with P^lane (one barrier);
//     (3) the zero-input response of s[c] is added to the chunk: y[j] += (M^j s[c])_1;
//   * the state after the last chunk is carried to the next span / the next call.
//
// All arithmetic is Float64 like the reference's; results differ from the sequential recurrence
// only by reassociation (~1e-15 relative before the Float32 store).  Calls shorter than one chunk
// and the < kEqChunk frame tail of a call run eq_sequential, the reference's recurrence verbatim.
//
// Shared between hipcc (kernels.hip) and the CPU thread-emulation harness (tests/emu).
#pragma once
#include "cplx.hpp"

// Timing ablations (wrong results; tools/eq_ablate.sh only): bit 0 no recurrence, 1 no in-wave scan,
// 2 no wave chaining, 3 no zero-input correction.
#ifndef AW_EQ_ABL
#define AW_EQ_ABL 0
#endif

namespace awk {

constexpr int kEqThreads = 256;
#ifndef AW_EQ_CHUNK
#define AW_EQ_CHUNK 16
#endif
constexpr int kEqChunk = AW_EQ_CHUNK;            // frames per thread and span (power of two)
constexpr int kEqSpan = kEqThreads * kEqChunk;   // 4096 frames
constexpr int kEqMaxFilters = 64;                // ParametricEqualizerState.maximumFilterCount :17
constexpr int kEqScanSteps = 7;                  // P^(2^s), s = 0 .. 6 (6 = one whole wave of chunks)
// LDS map (bytes)
#ifndef AW_EQ_STAGE_EARS
#define AW_EQ_STAGE_EARS 2       // 1: stage sized for the per-ear workgroups only (tuning builds with bigger chunks)
#endif
constexpr int kEqStageBytes = ((kEqThreads * (kEqChunk * AW_EQ_STAGE_EARS + (AW_EQ_STAGE_EARS == 2 ? 4 : 1)) * 4 + 15) / 16) * 16;   // 36,864: [chunk][32 + 4] floats (both ears)
constexpr int kEqScanBytes = 4 * kEqThreads * 8;                          // [4][thread] double (component-major: conflict-free 8-B accesses)
constexpr int kEqTotalsBytes = 2 * (kEqThreads / 64) * 4 * 8;             // ping-pong [wave][4]
constexpr int kEqCarryBytes = 2 * kEqMaxFilters * 4 * 8;                  // ping-pong [filter][4]
constexpr int kEqLdsBytes = kEqStageBytes + kEqScanBytes + kEqTotalsBytes + kEqCarryBytes;   // 49,408
constexpr int kEqTabDoubles = 5 + kEqChunk * 2 + kEqScanSteps * 4;        // 65 per filter

// Per-state tables, built on the host in double (host/eq.cpp).
struct EqTables {
    // [K][kEqTabDoubles] wave-uniform entries, read straight from global memory: the index is uniform and the
    // kernel takes the pointer as a `const __restrict__` argument, so hipcc issues scalar loads (s_load) and the
    // values reach the FMAs as SGPR operands — no LDS or VGPR traffic for them:
    //   [0,5)   b0 b1 b2 a1 a2 (normalised by a0)
    //   [5,37)  row 0 of M^j, j = 0 .. kEqChunk-1          (zero-input response)
    //   [37,65) P^(2^s), s = 0 .. 6, P = M^kEqChunk, row-major 2x2
    const double *tab;
    const double *plane;  // [K][64][4]  P^lane, lane = 0 .. 63 (per-lane: vector loads)
    double preamp;        // 10^(dB/20)
    int n_filters;
};

struct EqParams {
    const float *in;      // [stream][stride_frames][2] interleaved L,R
    float *out;           // same layout; may alias `in`
    double *z;            // [stream][K][4]  lz1 lz2 rz1 rz2
    EqTables t;
    long long frames;         // frames this launch processes per stream
    long long stride_frames;  // distance between streams, in frames
    int cus;                  // compute units of the context's device (LaunchCfg); 0 = 256
    int ear_split;            // LaunchCfg::eq_ear_split: -1 automatic, 0 / 1 forced
};

AW_HD double eq_flush(double v) { return (v < 0 ? -v : v) < 1e-30 ? 0.0 : v; }   // flushSubnormal :95-97

// st += P * q for the (z1, z2) state pairs of E ears (st, q: [2 E])
template <int E> AW_HD void eq_apply(const double *P, const double (&q)[2 * E], double (&st)[2 * E]) {
    const double p0 = P[0], p1 = P[1], p2 = P[2], p3 = P[3];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const double q0 = q[2 * e], q1 = q[2 * e + 1];
        st[2 * e] = __builtin_fma(p1, q1, __builtin_fma(p0, q0, st[2 * e]));
        st[2 * e + 1] = __builtin_fma(p3, q1, __builtin_fma(p2, q0, st[2 * e + 1]));
    }
}

// scan slots: component c of thread t at scan[c * kEqThreads + t] (component-major: conflict-free 8-B accesses)
template <int E> AW_HD void eq_put(double *scan, int t, const double (&v)[2 * E]) {
#pragma unroll
    for (int c = 0; c < 2 * E; ++c) scan[c * kEqThreads + t] = v[c];
}
template <int E> AW_HD void eq_get(const double *scan, int t, double (&q)[2 * E]) {
#pragma unroll
    for (int c = 0; c < 2 * E; ++c) q[c] = scan[c * kEqThreads + t];
}

// One workgroup walks one stream's timeline for E ears starting at ear0: E = 2 (both ears in every thread)
// or E = 1 (a workgroup per (stream, ear): twice the waves in flight for small batches; the kernel is
// latency bound — VALU 40 % busy at 2 waves/SIMD, rocprofv3 PMC — so occupancy is what it needs).
// p.frames must be a multiple of kEqChunk.
template <class Ctx, int E> AW_HD void eq_cascade_stream(Ctx &ctx, const EqParams &p, long long stream, int ear0) {
    constexpr int S = 2 * E;                     // state doubles per filter handled here
    const int tid = ctx.tid();
    const int lane = tid & 63;
    char *lds = reinterpret_cast<char *>(ctx.lds());
    float *stage = reinterpret_cast<float *>(lds);                         // [chunk][kEqChunk * E + pad] floats
    double *scan = reinterpret_cast<double *>(lds + kEqStageBytes);
    double *totals = reinterpret_cast<double *>(lds + kEqStageBytes + kEqScanBytes);
    double *carry = reinterpret_cast<double *>(lds + kEqStageBytes + kEqScanBytes + kEqTotalsBytes);
    constexpr int kStride = kEqChunk * E + (E == 2 ? 4 : 1);               // floats per chunk row: 36 (16-B units) / 17
    const int wave = ctx.wave();
    const int K = p.t.n_filters;
    double *zs = p.z + stream * (long long)K * 4 + ear0 * 2;               // [K][4]: this workgroup's S of every 4
    const float *in = p.in + stream * p.stride_frames * 2 + ear0;
    float *out = p.out + stream * p.stride_frames * 2 + ear0;
    const double preamp = p.t.preamp;

    for (int i = tid; i < K * S; i += kEqThreads) carry[i] = zs[(i / S) * 4 + (i % S)];
    int par = 0;

    for (long long base = 0; base < p.frames; base += kEqSpan) {
        const long long rem = p.frames - base;
        const int nfr = rem < kEqSpan ? (int)rem : kEqSpan;
        const int nchunks = nfr / kEqChunk;
        ctx.barrier();   // stage reads of the previous span and the carry writes are complete
        // HBM -> LDS coalesced, LDS -> registers transposed to one chunk per thread
#pragma unroll
        for (int j = 0; j < kEqChunk; ++j) {
            const int f = j * kEqThreads + tid;
            float *dst = stage + (f / kEqChunk) * kStride + (f % kEqChunk) * E;
            if constexpr (E == 2) {
                cf v = mk(0.f, 0.f);
                if (f < nfr) v = *reinterpret_cast<const cf *>(in + (base + f) * 2);
                *reinterpret_cast<cf *>(dst) = v;
            } else {
                *dst = f < nfr ? in[(base + f) * 2] : 0.f;
            }
        }
        ctx.barrier();
        double x[E][kEqChunk];
#pragma unroll
        for (int j = 0; j < kEqChunk; ++j)
#pragma unroll
            for (int e = 0; e < E; ++e) x[e][j] = (double)stage[tid * kStride + j * E + e] * preamp;   // :67-69

        for (int k = 0; k < K; ++k) {
            const double *c = p.t.tab + (long long)k * kEqTabDoubles;      // uniform: scalar loads
            const double b0 = c[0], b1 = c[1], b2 = c[2], na1 = -c[3], na2 = -c[4];
            // this lane's power of P for step (2c): issued now, consumed after the chunk's recurrence
            const double *plp = p.t.plane + ((long long)k * 64 + lane) * 4;
            const double pl[4] = {plp[0], plp[1], plp[2], plp[3]};
            // (1) zero-state response of this chunk, in place (:71-87 with z = 0)
            double st[S];
#pragma unroll
            for (int i = 0; i < S; ++i) st[i] = 0.0;
#pragma unroll
            for (int j = 0; j < ((AW_EQ_ABL & 1) ? 1 : kEqChunk); ++j)
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const double lo = __builtin_fma(b0, x[e][j], st[2 * e]);
                    st[2 * e] = __builtin_fma(na1, lo, __builtin_fma(b1, x[e][j], st[2 * e + 1]));
                    st[2 * e + 1] = __builtin_fma(na2, lo, b2 * x[e][j]);
                    x[e][j] = lo;
                }
            const double *pp = c + 5 + kEqChunk * 2;
            double end[S];
#pragma unroll
            for (int i = 0; i < S; ++i) end[i] = st[i];
            // (2a) inclusive Hillis-Steele scan INSIDE each wave (d = 1 .. 32) through wave-private slots
#pragma unroll
            for (int s = 0; s < ((AW_EQ_ABL & 2) ? 1 : 6); ++s) {
                const int d = 1 << s;
                eq_put<E>(scan, tid, st);
                ctx.wave_sync();
                if (lane >= d) {
                    double q[S];
                    eq_get<E>(scan, tid - d, q);
                    eq_apply<E>(pp + s * 4, q, st);
                }
                ctx.wave_sync();
            }
            // wave totals -> LDS (ping-pong by filter parity), exclusive in-wave value from lane - 1
            double *tot = totals + (k & 1) * (kEqThreads / 64) * 4;
            eq_put<E>(scan, tid, st);
            if (lane == 63) {
#pragma unroll
                for (int i = 0; i < S; ++i) tot[wave * 4 + i] = st[i];
            }
            if (!(AW_EQ_ABL & 4)) ctx.barrier();
            // (2b) state entering this wave: W_0 = carried state, W_w = P^64 W_{w-1} + T_{w-1}
            const double *cin = carry + par * kEqMaxFilters * 4 + k * S;
            double w[S];
#pragma unroll
            for (int i = 0; i < S; ++i) w[i] = cin[i];
            for (int i = 0; i < ((AW_EQ_ABL & 4) ? 0 : wave); ++i) {
                double n[S];
#pragma unroll
                for (int m = 0; m < S; ++m) n[m] = tot[i * 4 + m];
                eq_apply<E>(pp + 6 * 4, w, n);
#pragma unroll
                for (int m = 0; m < S; ++m) w[m] = n[m];
            }
            // (2c) state entering this chunk: in-wave exclusive prefix + P^lane W_w
            double sin_[S];
#pragma unroll
            for (int i = 0; i < S; ++i) sin_[i] = 0.0;
            if (lane > 0) eq_get<E>(scan, tid - 1, sin_);
            eq_apply<E>(pl, w, sin_);
            if (tid == nchunks - 1) {   // state after the last active chunk -> next span / next call
                double *cout = carry + (par ^ 1) * kEqMaxFilters * 4 + k * S;
                eq_apply<E>(pp, sin_, end);
#pragma unroll
                for (int i = 0; i < S; ++i) cout[i] = end[i];
            }
            ctx.wave_sync();   // the exclusive reads above precede the next filter's slot writes
            // (3) zero-input response of the entering state
            const double *g = c + 5;
#pragma unroll
            for (int j = 0; j < ((AW_EQ_ABL & 8) ? 1 : kEqChunk); ++j) {
                const double g0 = g[2 * j], g1 = g[2 * j + 1];
#pragma unroll
                for (int e = 0; e < E; ++e) x[e][j] = __builtin_fma(g1, sin_[2 * e + 1], __builtin_fma(g0, sin_[2 * e], x[e][j]));
            }
        }
        par = K ? par ^ 1 : par;

#pragma unroll
        for (int j = 0; j < kEqChunk; ++j)
#pragma unroll
            for (int e = 0; e < E; ++e) stage[tid * kStride + j * E + e] = (float)x[e][j];   // :88-89
        ctx.barrier();
#pragma unroll
        for (int j = 0; j < kEqChunk; ++j) {
            const int f = j * kEqThreads + tid;
            const float *src = stage + (f / kEqChunk) * kStride + (f % kEqChunk) * E;
            if (f < nfr) {
                if constexpr (E == 2) *reinterpret_cast<cf *>(out + (base + f) * 2) = *reinterpret_cast<const cf *>(src);
                else out[(base + f) * 2] = *src;
            }
        }
    }
    ctx.barrier();
    for (int i = tid; i < K * S; i += kEqThreads) zs[(i / S) * 4 + (i % S)] = eq_flush(carry[par * kEqMaxFilters * 4 + i]);
}

// The reference's recurrence verbatim, one thread per (stream, ear): calls shorter than a chunk and
// the tail of a call.  Non-contracted Float64 (bit-exact with the sequential definition).
AW_HD void eq_sequential(const EqParams &p, long long stream, int ear) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const int K = p.t.n_filters;
    double *zs = p.z + stream * (long long)K * 4 + ear * 2;
    const float *in = p.in + (stream * p.stride_frames) * 2 + ear;
    float *out = p.out + (stream * p.stride_frames) * 2 + ear;
    for (long long f = 0; f < p.frames; ++f) {
        double v = (double)in[2 * f] * p.t.preamp;
        for (int k = 0; k < K; ++k) {
            const double *c = p.t.tab + (long long)k * kEqTabDoubles;
            const double lo = c[0] * v + zs[k * 4];
            const double z1 = c[1] * v - c[3] * lo + zs[k * 4 + 1];
            const double z2 = c[2] * v - c[4] * lo;
            zs[k * 4] = eq_flush(z1);
            zs[k * 4 + 1] = eq_flush(z2);
            v = lo;
        }
        out[2 * f] = (float)v;
    }
}

}  // namespace awk
