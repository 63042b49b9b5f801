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
constexpr int kEqChunk = 16;
constexpr int kEqSpan = kEqThreads * kEqChunk;   // 4096 frames
constexpr int kEqMaxFilters = 64;                // ParametricEqualizerState.maximumFilterCount :17
constexpr int kEqScanSteps = 7;                  // P^(2^s), s = 0 .. 6 (6 = one whole wave of chunks)
constexpr int kEqStageStride = kEqChunk + 2;     // float2 units; +16 B per chunk spreads the banks
// LDS map (bytes)
constexpr int kEqStageBytes = kEqThreads * kEqStageStride * 8;            // 36,864
constexpr int kEqScanBytes = kEqThreads * 4 * 8;                          // [thread][4] double, wave-private slots
constexpr int kEqTotalsBytes = 2 * (kEqThreads / 64) * 4 * 8;             // ping-pong [wave][4]
constexpr int kEqCarryBytes = 2 * kEqMaxFilters * 4 * 8;                  // ping-pong [filter][4]
constexpr int kEqLdsFixedBytes = kEqStageBytes + kEqScanBytes + kEqTotalsBytes + kEqCarryBytes;   // 49,408
// + the uniform per-filter tables (coef | zir | ppow), staged once per launch
constexpr int kEqTabDoubles = 5 + kEqChunk * 2 + kEqScanSteps * 4;        // 65 per filter
AW_HD int eq_lds_bytes(int n_filters) { return kEqLdsFixedBytes + n_filters * kEqTabDoubles * 8; }

// Per-state tables, built on the host in double (host/eq.cpp).
struct EqTables {
    const double *coef;   // [K][5]  b0 b1 b2 a1 a2 (normalised by a0)
    const double *zir;    // [K][kEqChunk][2]   row 0 of M^j, j = 0 .. kEqChunk-1
    const double *ppow;   // [K][kEqScanSteps][4]   P^(2^s), P = M^kEqChunk, row-major 2x2
    const double *plane;  // [K][64][4]             P^lane, lane = 0 .. 63
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
};

AW_HD double eq_flush(double v) { return (v < 0 ? -v : v) < 1e-30 ? 0.0 : v; }   // flushSubnormal :95-97

// acc += P * q for the (z1, z2) pairs of both ears
AW_HD void eq_apply(const double *P, const double *q, double &l1, double &l2, double &r1, double &r2) {
    const double p0 = P[0], p1 = P[1], p2 = P[2], p3 = P[3];
    const double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    l1 = __builtin_fma(p1, q1, __builtin_fma(p0, q0, l1));
    l2 = __builtin_fma(p3, q1, __builtin_fma(p2, q0, l2));
    r1 = __builtin_fma(p1, q3, __builtin_fma(p0, q2, r1));
    r2 = __builtin_fma(p3, q3, __builtin_fma(p2, q2, r2));
}

// One workgroup, one stream; p.frames must be a multiple of kEqChunk.
template <class Ctx> AW_HD void eq_cascade_stream(Ctx &ctx, const EqParams &p, long long stream) {
    const int tid = ctx.tid();
    const int lane = tid & 63;
    char *lds = reinterpret_cast<char *>(ctx.lds());
    cf *stage = reinterpret_cast<cf *>(lds);
    double *scan = reinterpret_cast<double *>(lds + kEqStageBytes);
    double *totals = reinterpret_cast<double *>(lds + kEqStageBytes + kEqScanBytes);
    double *carry = reinterpret_cast<double *>(lds + kEqStageBytes + kEqScanBytes + kEqTotalsBytes);
    double *tab = reinterpret_cast<double *>(lds + kEqLdsFixedBytes);   // [K][65]: coef 5 | zir 32 | ppow 28
    const int wave = ctx.wave();
    const int K = p.t.n_filters;
    double *zs = p.z + stream * (long long)K * 4;
    const cf *in = reinterpret_cast<const cf *>(p.in) + stream * p.stride_frames;
    cf *out = reinterpret_cast<cf *>(p.out) + stream * p.stride_frames;
    const double preamp = p.t.preamp;

    for (int i = tid; i < K * 4; i += kEqThreads) carry[i] = zs[i];
    for (int i = tid; i < K * kEqTabDoubles; i += kEqThreads) {
        const int k = i / kEqTabDoubles, r = i - k * kEqTabDoubles;
        tab[i] = r < 5 ? p.t.coef[k * 5 + r]
                       : r < 5 + kEqChunk * 2 ? p.t.zir[k * kEqChunk * 2 + (r - 5)] : p.t.ppow[k * kEqScanSteps * 4 + (r - 5 - kEqChunk * 2)];
    }
    int par = 0;

    for (long long base = 0; base < p.frames; base += kEqSpan) {
        const long long rem = p.frames - base;
        const int nfr = rem < kEqSpan ? (int)rem : kEqSpan;
        const int nchunks = nfr / kEqChunk;
        ctx.barrier();   // stage reads of the previous span and the carry writes are complete
#pragma unroll
        for (int j = 0; j < kEqChunk; ++j) {
            const int f = j * kEqThreads + tid;
            cf v = mk(0.f, 0.f);
            if (f < nfr) v = in[base + f];
            stage[(f >> 4) * kEqStageStride + (f & 15)] = v;
        }
        ctx.barrier();
        double xl[kEqChunk], xr[kEqChunk];
#pragma unroll
        for (int j = 0; j < kEqChunk; ++j) {
            const cf v = stage[tid * kEqStageStride + j];
            xl[j] = (double)v.x * preamp;   // :67-69
            xr[j] = (double)v.y * preamp;
        }

        for (int k = 0; k < K; ++k) {
            const double *c = tab + k * kEqTabDoubles;
            const double b0 = c[0], b1 = c[1], b2 = c[2], na1 = -c[3], na2 = -c[4];
            // this lane's power of P for step (2c): issued now, consumed after the chunk's recurrence
            const double *plp = p.t.plane + ((long long)k * 64 + lane) * 4;
            const double pl[4] = {plp[0], plp[1], plp[2], plp[3]};
            // (1) zero-state response of this chunk, in place (:71-87 with z = 0)
            double l1 = 0, l2 = 0, r1 = 0, r2 = 0;
#pragma unroll
            for (int j = 0; j < ((AW_EQ_ABL & 1) ? 1 : kEqChunk); ++j) {
                const double lo = __builtin_fma(b0, xl[j], l1);
                l1 = __builtin_fma(na1, lo, __builtin_fma(b1, xl[j], l2));
                l2 = __builtin_fma(na2, lo, b2 * xl[j]);
                xl[j] = lo;
                const double ro = __builtin_fma(b0, xr[j], r1);
                r1 = __builtin_fma(na1, ro, __builtin_fma(b1, xr[j], r2));
                r2 = __builtin_fma(na2, ro, b2 * xr[j]);
                xr[j] = ro;
            }
            const double *pp = c + 5 + kEqChunk * 2;
            const double e0 = l1, e1 = l2, e2 = r1, e3 = r2;
            // (2a) inclusive Hillis-Steele scan INSIDE each wave (d = 1 .. 32) through wave-private slots
#pragma unroll
            for (int s = 0; s < ((AW_EQ_ABL & 2) ? 1 : 6); ++s) {
                const int d = 1 << s;
                double *w = scan + tid * 4;
                w[0] = l1; w[1] = l2; w[2] = r1; w[3] = r2;
                ctx.wave_sync();
                if (lane >= d) eq_apply(pp + s * 4, scan + (tid - d) * 4, l1, l2, r1, r2);
                ctx.wave_sync();
            }
            // wave totals -> LDS (ping-pong by filter parity), exclusive in-wave value from lane - 1
            double *tot = totals + (k & 1) * (kEqThreads / 64) * 4;
            {
                double *w = scan + tid * 4;
                w[0] = l1; w[1] = l2; w[2] = r1; w[3] = r2;
                if (lane == 63) {
                    double *t = tot + wave * 4;
                    t[0] = l1; t[1] = l2; t[2] = r1; t[3] = r2;
                }
            }
            if (!(AW_EQ_ABL & 4)) ctx.barrier();
            // (2b) state entering this wave: W_0 = carried state, W_w = P^64 W_{w-1} + T_{w-1}
            const double *cin = carry + par * kEqMaxFilters * 4 + k * 4;
            double w0 = cin[0], w1 = cin[1], w2 = cin[2], w3 = cin[3];
            for (int i = 0; i < ((AW_EQ_ABL & 4) ? 0 : wave); ++i) {
                const double *t = tot + i * 4;
                double n0 = t[0], n1 = t[1], n2 = t[2], n3 = t[3];
                const double q[4] = {w0, w1, w2, w3};
                eq_apply(pp + 6 * 4, q, n0, n1, n2, n3);
                w0 = n0; w1 = n1; w2 = n2; w3 = n3;
            }
            // (2c) state entering this chunk: in-wave exclusive prefix + P^lane W_w
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            if (lane > 0) {
                const double *q = scan + (tid - 1) * 4;
                s0 = q[0]; s1 = q[1]; s2 = q[2]; s3 = q[3];
            }
            {
                const double q[4] = {w0, w1, w2, w3};
                eq_apply(pl, q, s0, s1, s2, s3);
            }
            if (tid == nchunks - 1) {   // state after the last active chunk -> next span / next call
                double *cout = carry + (par ^ 1) * kEqMaxFilters * 4 + k * 4;
                const double q[4] = {s0, s1, s2, s3};
                double c0 = e0, c1 = e1, c2 = e2, c3 = e3;
                eq_apply(pp, q, c0, c1, c2, c3);
                cout[0] = c0; cout[1] = c1; cout[2] = c2; cout[3] = c3;
            }
            ctx.wave_sync();   // the exclusive reads above precede the next filter's slot writes
            // (3) zero-input response of the entering state
            const double *g = c + 5;
#pragma unroll
            for (int j = 0; j < ((AW_EQ_ABL & 8) ? 1 : kEqChunk); ++j) {
                const double g0 = g[2 * j], g1 = g[2 * j + 1];
                xl[j] = __builtin_fma(g1, s1, __builtin_fma(g0, s0, xl[j]));
                xr[j] = __builtin_fma(g1, s3, __builtin_fma(g0, s2, xr[j]));
            }
        }
        par = K ? par ^ 1 : par;

#pragma unroll
        for (int j = 0; j < kEqChunk; ++j) stage[tid * kEqStageStride + j] = mk((float)xl[j], (float)xr[j]);   // :88-89
        ctx.barrier();
#pragma unroll
        for (int j = 0; j < kEqChunk; ++j) {
            const int f = j * kEqThreads + tid;
            if (f < nfr) out[base + f] = stage[(f >> 4) * kEqStageStride + (f & 15)];
        }
    }
    ctx.barrier();
    for (int i = tid; i < K * 4; i += kEqThreads) zs[i] = eq_flush(carry[par * kEqMaxFilters * 4 + i]);
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
            const double *c = p.t.coef + k * 5;
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
