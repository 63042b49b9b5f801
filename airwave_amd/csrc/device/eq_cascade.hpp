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
//     recurrence over chunks that is solved with a two-level scan: inside each wave in registers
//     (DPP moves: rows of 16 lanes with the precomputed powers P^1..P^8, then the row totals with the
//     per-lane powers P^(m+1); no LDS, no branches), then the four wave totals are chained with P^64
//     through LDS and applied per lane with P^(lane+1) (one barrier);
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

// Timing ablations (wrong results; tools/archive/eq_ablate.sh only): bit 0 no recurrence, 1 no in-wave scan,
// 2 no wave chaining, 3 no zero-input correction.
#ifndef AW_EQ_ABL
#define AW_EQ_ABL 0
#endif

namespace awk {

constexpr int kEqThreads = 256;
#ifndef AW_EQ_CHUNK
#define AW_EQ_CHUNK 32
#endif
constexpr int kEqChunk = AW_EQ_CHUNK;            // frames per thread and span (power of two)
constexpr int kEqSpan = kEqThreads * kEqChunk;   // 8192 frames
#ifndef AW_EQ_COEF_AHEAD
#define AW_EQ_COEF_AHEAD 0                       // 1: coefficient tables fetched one filter ahead (50 more SGPRs: spills, 3 % slower)
#endif
#ifndef AW_EQ_PREFETCH
#define AW_EQ_PREFETCH 0                         // 1: the next span's loads are issued before the filter loop (kEqChunk x E more VGPRs; no faster at 16, spills at 32)
#endif
constexpr int kEqMaxFilters = 64;                // ParametricEqualizerState.maximumFilterCount :17
constexpr int kEqScanSteps = 7;                  // P^(2^s), s = 0 .. 6 (6 = one whole wave of chunks)
// LDS map (bytes)
constexpr int eq_stage_bytes(int ears) { return ((kEqThreads * (kEqChunk * ears + (ears == 2 ? 4 : 1)) * 4 + 15) / 16) * 16; }   // [chunk][64 + 4] floats (both ears: 69,632) or [chunk][32 + 1] (one: 33,792)
constexpr int kEqTotalsBytes = 2 * (kEqThreads / 64) * 4 * 8;             // ping-pong [wave][4]
constexpr int kEqCarryBytes = 2 * kEqMaxFilters * 4 * 8;                  // ping-pong [filter][4]
constexpr int eq_lds_bytes(int ears) { return eq_stage_bytes(ears) + kEqTotalsBytes + kEqCarryBytes; }   // 73,984 / 38,144: two workgroups per CU / four
constexpr int kEqLdsBytes = eq_lds_bytes(2);
constexpr int kEqTabDoubles = 5 + kEqChunk * 2 + kEqScanSteps * 4;        // 97 per filter

// Per-state tables, built on the host in double (host/eq.cpp).
struct EqTables {
    // [K][kEqTabDoubles] wave-uniform entries, read straight from global memory: the index is uniform and the
    // kernel takes the pointer as a `const __restrict__` argument, so hipcc issues scalar loads (s_load) and the
    // values reach the FMAs as SGPR operands — no LDS or VGPR traffic for them:
    //   [0,5)   b0 b1 b2 a1 a2 (normalised by a0)
    //   [5,69)  row 0 of M^j, j = 0 .. kEqChunk-1          (zero-input response)
    //   [69,97) P^(2^s), s = 0 .. 6, P = M^kEqChunk, row-major 2x2 (the kernel uses s = 0 .. 3 and 6)
    const double *tab;
    const double *plane;  // [K][64][4]  P^(m+1), m = 0 .. 63 (per-lane: vector loads)
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

// What one lane moves per load / store of a span: two whole stereo frames (16 B at dword alignment: a stream's first frame is
// only 8-byte aligned when the stride is odd) or one ear of one frame (4 B).
struct __attribute__((packed, aligned(4))) EqF4 { float a, b, c, d; };
struct alignas(16) EqF4A { float a, b, c, d; };          // the stage rows are 16-byte aligned: one ds_read/write_b128
template <int E> struct EqRaw;
template <> struct EqRaw<2> {
    typedef EqF4 type;
    static constexpr int kFrames = 2;
    static AW_HD EqF4 zero() { return EqF4{0.f, 0.f, 0.f, 0.f}; }
    static AW_HD EqF4 load(const float *q) { return *reinterpret_cast<const EqF4 *>(q); }
    static AW_HD void store(float *q, EqF4 v) { *reinterpret_cast<EqF4 *>(q) = v; }
    // the span's last pair of frames may be half inside
    static AW_HD EqF4 load_partial(const float *q) { return EqF4{q[0], q[1], 0.f, 0.f}; }
    static AW_HD void store_partial(float *q, EqF4 v) { q[0] = v.a; q[1] = v.b; }
    static AW_HD EqF4 lds_load(const float *q) { const EqF4A v = *reinterpret_cast<const EqF4A *>(q); return EqF4{v.a, v.b, v.c, v.d}; }
    static AW_HD void lds_store(float *q, EqF4 v) { *reinterpret_cast<EqF4A *>(q) = EqF4A{v.a, v.b, v.c, v.d}; }
};
template <> struct EqRaw<1> {
    typedef float type;
    static constexpr int kFrames = 1;
    static AW_HD float zero() { return 0.f; }
    static AW_HD float load(const float *q) { return *q; }
    static AW_HD void store(float *q, float v) { *q = v; }
    static AW_HD float load_partial(const float *q) { return *q; }
    static AW_HD void store_partial(float *q, float v) { *q = v; }
    static AW_HD float lds_load(const float *q) { return *q; }
    static AW_HD void lds_store(float *q, float v) { *q = v; }
};

// One workgroup walks one stream's timeline for E ears starting at ear0: E = 2 (both ears in every thread)
// or E = 1 (a workgroup per (stream, ear): twice the waves in flight for small batches).
// p.frames must be a multiple of kEqChunk.
//
// Memory: the span's frames are fetched with kEqChunk independent coalesced loads per lane that are all in flight at
// once (a first version guarded and awaited every load separately: 16 serial round trips per span, 40 % of the kernel's
// time); the second workgroup of the CU computes meanwhile.  The last, partial span takes the guarded form.
// Scan: inside a wave the chunk recurrence is scanned in registers with DPP moves — row_shr:1/2/4/8 (rows of 16 lanes,
// uniform powers P^1..P^8), then row_bcast:15 and row_bcast:31 with the per-lane powers P^(m+1) — no LDS round trips,
// no branches; only the four wave totals go through LDS (one barrier per filter).
template <class Ctx, int E> AW_HD void eq_cascade_stream(Ctx &ctx, const EqParams &p, long long stream, int ear0) {
    constexpr int S = 2 * E;                     // state doubles per filter handled here
    typedef EqRaw<E> Raw;
    typedef typename Raw::type raw_t;
    const int tid = ctx.tid();
    const int lane = tid & 63;
    char *lds = reinterpret_cast<char *>(ctx.lds());
    float *stage = reinterpret_cast<float *>(lds);                         // [chunk][kEqChunk * E + pad] floats
    double *totals = reinterpret_cast<double *>(lds + eq_stage_bytes(E));
    double *carry = reinterpret_cast<double *>(lds + eq_stage_bytes(E) + kEqTotalsBytes);
    constexpr int kStride = kEqChunk * E + (E == 2 ? 4 : 1);               // floats per chunk row: 68 (16-B units) / 33
    const int wave = ctx.wave();
    const int K = p.t.n_filters;
    double *zs = p.z + stream * (long long)K * 4 + ear0 * 2;               // [K][4]: this workgroup's S of every 4
    const float *in = p.in + stream * p.stride_frames * 2 + ear0;
    float *out = p.out + stream * p.stride_frames * 2 + ear0;
    const double preamp = p.t.preamp;

    for (int i = tid; i < K * S; i += kEqThreads) carry[i] = zs[(i / S) * 4 + (i % S)];
    int par = 0;

    constexpr int kPer = Raw::kFrames, kLoads = kEqChunk / kPer;      // frames per element, elements per lane and span
    raw_t raw[kLoads];
    auto fetch = [&](long long base, int nfr) {
        if (nfr == kEqSpan) {
#pragma unroll
            for (int j = 0; j < kLoads; ++j) raw[j] = Raw::load(in + (base + (long long)(j * kEqThreads + tid) * kPer) * 2);
        } else {
#pragma unroll
            for (int j = 0; j < kLoads; ++j) {
                const int f = (j * kEqThreads + tid) * kPer;
                raw[j] = Raw::zero();
                if (f + kPer <= nfr) raw[j] = Raw::load(in + (base + f) * 2);
                else if (f < nfr) raw[j] = Raw::load_partial(in + (base + f) * 2);
            }
        }
    };
    auto span_frames = [&](long long base) { const long long rem = p.frames - base; return rem < kEqSpan ? (int)rem : kEqSpan; };
    if (AW_EQ_PREFETCH && p.frames > 0) fetch(0, span_frames(0));

    for (long long base = 0; base < p.frames; base += kEqSpan) {
        const int nfr = span_frames(base);
        const int nchunks = nfr / kEqChunk;
        if (!AW_EQ_PREFETCH) fetch(base, nfr);
        ctx.barrier();   // stage reads of the previous span and the carry writes are complete
        // registers -> LDS in frame order, LDS -> registers transposed to one chunk per thread
#pragma unroll
        for (int j = 0; j < kLoads; ++j) {
            const int f = (j * kEqThreads + tid) * kPer;      // kPer frames never straddle a chunk (kEqChunk is even)
            Raw::lds_store(stage + (f / kEqChunk) * kStride + (f % kEqChunk) * E, raw[j]);
        }
        ctx.barrier();
        double x[E][kEqChunk];
#pragma unroll
        for (int j = 0; j < kEqChunk; ++j)
#pragma unroll
            for (int e = 0; e < E; ++e) x[e][j] = (double)stage[tid * kStride + j * E + e] * preamp;   // :67-69
        if (AW_EQ_PREFETCH && base + kEqSpan < p.frames) fetch(base + kEqSpan, span_frames(base + kEqSpan));              // lands during the filter loop

        // the filter's coefficients and scan powers (25 wave-uniform doubles) come in ONE batch of scalar loads at the top of
        // its iteration (left to itself hipcc loads the scan powers where they are used: two more stalls per filter)
        double ck[5], pk[20];
        auto fetch_coefs = [&](int k) {
            const double *c = p.t.tab + (long long)k * kEqTabDoubles;
#pragma unroll
            for (int i = 0; i < 5; ++i) ck[i] = c[i];
#pragma unroll
            for (int i = 0; i < 16; ++i) pk[i] = c[5 + kEqChunk * 2 + i];
#pragma unroll
            for (int i = 0; i < 4; ++i) pk[16 + i] = c[5 + kEqChunk * 2 + 6 * 4 + i];
        };
        if (AW_EQ_COEF_AHEAD && K > 0) fetch_coefs(0);
        for (int k = 0; k < K; ++k) {
            const double *c = p.t.tab + (long long)k * kEqTabDoubles;      // uniform: scalar loads
            if (!AW_EQ_COEF_AHEAD) fetch_coefs(k);
            const double b0 = ck[0], b1 = ck[1], b2 = ck[2], na1 = -ck[3], na2 = -ck[4];
            double pp[20];
#pragma unroll
            for (int i = 0; i < 20; ++i) pp[i] = pk[i];
            if (AW_EQ_COEF_AHEAD) fetch_coefs(k + 1 < K ? k + 1 : k);
            // this lane's powers of P: D[m] = P^(m+1).  Issued now, consumed after the chunk's recurrence
            const double *dk = p.t.plane + (long long)k * 64 * 4;
            double d16[4], d32[4], d64[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { d16[i] = dk[(lane & 15) * 4 + i]; d32[i] = dk[(lane & 31) * 4 + i]; d64[i] = dk[lane * 4 + i]; }
            // (1) zero-state response of this chunk, in place (:71-87 with z = 0)
            double st[S];
#pragma unroll
            for (int i = 0; i < S; ++i) st[i] = 0.0;
#pragma unroll
            for (int j = 0; j < ((AW_EQ_ABL & 1) ? 1 : kEqChunk); ++j)
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    // the output is the LAST reader of the input and takes its register (no copies at the loop's back edge)
                    const double t1 = __builtin_fma(b1, x[e][j], st[2 * e + 1]), t2 = b2 * x[e][j];
                    ctx.fma_in_place(x[e][j], b0, st[2 * e], t1, t2);      // x = b0 x + st, after t1 and t2
                    st[2 * e] = __builtin_fma(na1, x[e][j], t1);
                    st[2 * e + 1] = __builtin_fma(na2, x[e][j], t2);
                }
            // (2a) inclusive scan of s[c+1] = P s[c] + e[c] inside the wave, in registers.  Rows of 16 lanes first
            // (lanes that have no partner d lanes below in their row receive zeros) ...
            double q[S];
            if (!(AW_EQ_ABL & 2)) {
#pragma unroll
                for (int i = 0; i < S; ++i) q[i] = ctx.template row_shr<1>(st[i]);
                eq_apply<E>(pp + 0, q, st);
#pragma unroll
                for (int i = 0; i < S; ++i) q[i] = ctx.template row_shr<2>(st[i]);
                eq_apply<E>(pp + 4, q, st);
#pragma unroll
                for (int i = 0; i < S; ++i) q[i] = ctx.template row_shr<4>(st[i]);
                eq_apply<E>(pp + 8, q, st);
#pragma unroll
                for (int i = 0; i < S; ++i) q[i] = ctx.template row_shr<8>(st[i]);
                eq_apply<E>(pp + 12, q, st);
                // ... then rows 1 and 3 take the total of the row below (its lane 15) times P^(lane % 16 + 1), and rows 2
                // and 3 the running total at lane 31 times P^(lane - 31); the other rows receive zeros
#pragma unroll
                for (int i = 0; i < S; ++i) q[i] = ctx.row_bcast15(st[i]);
                eq_apply<E>(d16, q, st);
#pragma unroll
                for (int i = 0; i < S; ++i) q[i] = ctx.row_bcast31(st[i]);
                eq_apply<E>(d32, q, st);
            }
            // wave totals -> LDS (ping-pong by filter parity)
            double *tot = totals + (k & 1) * (kEqThreads / 64) * 4;
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
            if (!(AW_EQ_ABL & 4)) {
                double tw[kEqThreads / 64 - 1][S];     // all totals are read at once (one LDS latency), then chained
#pragma unroll
                for (int i = 0; i < kEqThreads / 64 - 1; ++i)
#pragma unroll
                    for (int m = 0; m < S; ++m) tw[i][m] = tot[i * 4 + m];
#pragma unroll
                for (int i = 0; i < kEqThreads / 64 - 1; ++i)
                    if (i < wave) {                    // wave-uniform
                        eq_apply<E>(pp + 16, w, tw[i]);
#pragma unroll
                        for (int m = 0; m < S; ++m) w[m] = tw[i][m];
                    }
            }
            // (2c) state leaving this chunk = in-wave inclusive prefix + P^(lane + 1) W_w; the state entering it is the
            // one leaving the lane below (lane 0: W_w itself)
            eq_apply<E>(d64, w, st);
            if (tid == nchunks - 1) {   // state after the last active chunk -> next span / next call
                double *cout = carry + (par ^ 1) * kEqMaxFilters * 4 + k * S;
#pragma unroll
                for (int i = 0; i < S; ++i) cout[i] = st[i];
            }
            double sin_[S];
#pragma unroll
            for (int i = 0; i < S; ++i) sin_[i] = ctx.wave_shr1(st[i], w[i]);
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

        // each thread rewrites its own row of the stage (it was the only reader of it), then frame order -> HBM
#pragma unroll
        for (int j = 0; j < kEqChunk; ++j)
#pragma unroll
            for (int e = 0; e < E; ++e) stage[tid * kStride + j * E + e] = (float)x[e][j];   // :88-89
        ctx.barrier();
        if (nfr == kEqSpan) {
#pragma unroll
            for (int j = 0; j < kLoads; ++j) {
                const int f = (j * kEqThreads + tid) * kPer;
                Raw::store(out + (base + f) * 2, Raw::lds_load(stage + (f / kEqChunk) * kStride + (f % kEqChunk) * E));
            }
        } else {
#pragma unroll
            for (int j = 0; j < kLoads; ++j) {
                const int f = (j * kEqThreads + tid) * kPer;
                const raw_t v = Raw::lds_load(stage + (f / kEqChunk) * kStride + (f % kEqChunk) * E);
                if (f + kPer <= nfr) Raw::store(out + (base + f) * 2, v);
                else if (f < nfr) Raw::store_partial(out + (base + f) * 2, v);
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
