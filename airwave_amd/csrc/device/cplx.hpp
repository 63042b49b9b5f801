// cplx.hpp — complex helpers and in-register radix-4/8/16 butterflies.
//
// Shared verbatim between the HIP kernels (hipcc, gfx950) and the CPU thread-emulation
// harness under tests/emu/ (g++), so index math can be debugged without a GPU.
// Everything is fully unrolled with compile-time indices so arrays stay in VGPRs.
#pragma once

// Timing-ablation macros (AW_ABL_*, AW_LW_ABL_*, AW_ABL2, AW_EQ_ABL != 0) remove work from the kernels to time what is left: they produce
// WRONG RESULTS (or races).  They may only be compiled into a library that says so: build.py passes -DAW_ABLATION_BUILD=1 together with a
// library suffix (libairwave_hip_<suffix>.so, never loaded unless AIRWAVE_HIP_LIBRARY points at it); in any other build they are an error,
// so a stray define or AW_EXTRA_HIPCC_FLAGS on a build box cannot yield a wrong-result libairwave_hip.so (tests/test_build_provenance.py).
#if !defined(AW_ABLATION_BUILD)
#if defined(AW_ABL_NOLOAD) || defined(AW_ABL_NOTAB) || defined(AW_ABL_NOCMAC) || defined(AW_ABL_NOPARTNER) || defined(AW_ABL_NOSTORE) || \
    defined(AW_ABL_FWD_NOSTORE) || defined(AW_ABL2) || defined(AW_ABL_NOFFT) || defined(AW_ABL_NOBARRIER) || defined(AW_ABL_MARCH_NOSTORE) || \
    defined(AW_ABL_MARCH_NOLOAD) || defined(AW_LW_ABL_SPLIT_NOLOAD) || defined(AW_LW_ABL_SPLIT_NOSTORE) || defined(AW_LW_ABL_ROWS_NOLOAD) || \
    defined(AW_LW_ABL_ROWS_NOTAB) || defined(AW_LW_ABL_ROWS_NOSTORE) || defined(AW_ABL_OLA_NOLOAD) || defined(AW_ABL_OLA_NOSTORE) || defined(AW_ABL_OLA_NOPARTNER) || defined(AW_ABL_NOTWLDS) || defined(AW_ABL_NOXCHG) || \
    (defined(AW_EQ_ABL) && AW_EQ_ABL != 0)
#error "timing-ablation macro in a product build: ablations give wrong results; build them with -DAW_ABLATION_BUILD=1 and a library suffix (airwave_amd/build.py)"
#endif
#endif

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define AW_HD __host__ __device__ __forceinline__
#else
#define AW_HD inline __attribute__((always_inline))
#endif

#ifndef AW_PK_CMUL
#define AW_PK_CMUL 0     // measured: -23 % VALU instructions (3872 -> 2995 per tile and wave), +2.5x v_mov, 10 spills: 1.56 -> 1.67 ms.  Off.
#endif

namespace awk {

struct alignas(8) cf {
    float x, y;
};

AW_HD cf mk(float x, float y) { cf r; r.x = x; r.y = y; return r; }
AW_HD cf operator+(cf a, cf b) { return mk(a.x + b.x, a.y + b.y); }
AW_HD cf operator-(cf a, cf b) { return mk(a.x - b.x, a.y - b.y); }
#if defined(__HIP_DEVICE_COMPILE__) && AW_PK_CMUL
// Complex multiplies as two packed-FP32 instructions instead of four scalar ones: v_pk_mul/fma_f32 take a
// register PAIR per operand and can pick the low or high half of each source per result half (op_sel /
// op_sel_hi) and negate per half (neg_lo / neg_hi) — exactly the swizzles of (re, im) arithmetic.  hipcc packs
// complex adds by itself but not these.
typedef float aw_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ aw_f2 aw_v2(cf a) { aw_f2 r; r.x = a.x; r.y = a.y; return r; }
__device__ __forceinline__ cf aw_c(aw_f2 a) { return mk(a.x, a.y); }
// t = (a.x b.x, a.y b.x) [+ acc]
__device__ __forceinline__ aw_f2 aw_pk_mul_bx(aw_f2 a, aw_f2 b) {
    aw_f2 t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));
    return t;
}
__device__ __forceinline__ aw_f2 aw_pk_fma_bx(aw_f2 a, aw_f2 b, aw_f2 c) {
    aw_f2 t;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(t) : "v"(a), "v"(b), "v"(c));
    return t;
}
// r = (t.x - a.y b.y, t.y + a.x b.y)          (a * b, second half)
__device__ __forceinline__ aw_f2 aw_pk_fma_by(aw_f2 a, aw_f2 b, aw_f2 t) {
    aw_f2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
// r = (t.x + a.y b.y, t.y - a.x b.y)          (a * conj(b) and conj(a) * b share it up to which operand is conjugated)
__device__ __forceinline__ aw_f2 aw_pk_fma_by_c(aw_f2 a, aw_f2 b, aw_f2 t) {
    aw_f2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
AW_HD cf cmul(cf a, cf b) { const aw_f2 A = aw_v2(a), B = aw_v2(b); return aw_c(aw_pk_fma_by(A, B, aw_pk_mul_bx(A, B))); }
// a * conj(b) = (a.x b.x + a.y b.y, a.y b.x - a.x b.y)
AW_HD cf cmulc(cf a, cf b) { const aw_f2 A = aw_v2(a), B = aw_v2(b); return aw_c(aw_pk_fma_by_c(A, B, aw_pk_mul_bx(A, B))); }
AW_HD cf conj(cf a) { return mk(a.x, -a.y); }
// acc + a*b
AW_HD cf cfma(cf a, cf b, cf acc) { const aw_f2 A = aw_v2(a), B = aw_v2(b); return aw_c(aw_pk_fma_by(A, B, aw_pk_fma_bx(A, B, aw_v2(acc)))); }
// acc + conj(a)*b = acc + (a.x b.x + a.y b.y, a.x b.y - a.y b.x) = acc + b * conj(a)
AW_HD cf cfmac(cf a, cf b, cf acc) { const aw_f2 A = aw_v2(a), B = aw_v2(b); return aw_c(aw_pk_fma_by_c(B, A, aw_pk_fma_bx(B, A, aw_v2(acc)))); }
#else
AW_HD cf cmul(cf a, cf b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// a * conj(b)
AW_HD cf cmulc(cf a, cf b) { return mk(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
AW_HD cf conj(cf a) { return mk(a.x, -a.y); }
// acc + a*b
AW_HD cf cfma(cf a, cf b, cf acc) {
    return mk(acc.x + a.x * b.x - a.y * b.y, acc.y + a.x * b.y + a.y * b.x);
}
// acc + conj(a)*b
AW_HD cf cfmac(cf a, cf b, cf acc) {
    return mk(acc.x + a.x * b.x + a.y * b.y, acc.y + a.x * b.y - a.y * b.x);
}
#endif
// multiply by -i (forward quarter turn) or +i
AW_HD cf mul_mi(cf a) { return mk(a.y, -a.x); }
AW_HD cf mul_pi(cf a) { return mk(-a.y, a.x); }
template <bool INV> AW_HD cf rot90(cf a) { return INV ? mul_pi(a) : mul_mi(a); }   // * W4^1
// twiddle multiply: forward uses w, inverse uses conj(w)
template <bool INV> AW_HD cf twmul(cf a, cf w) { return INV ? cmulc(a, w) : cmul(a, w); }

constexpr float kS2 = 0.70710678118654752440f;   // cos(pi/4)
constexpr float kC8 = 0.92387953251128675613f;   // cos(pi/8)
constexpr float kS8 = 0.38268343236508977173f;   // sin(pi/8)

// 4-point DFT, natural order in/out.  forward kernel e^{-2 pi i nk/4}.
template <bool INV> AW_HD void fft4(cf &a0, cf &a1, cf &a2, cf &a3) {
    const cf t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = rot90<INV>(a1 - a3);
    a0 = t0 + t2; a1 = t1 + t3; a2 = t0 - t2; a3 = t1 - t3;
}

// multiply by W8^1 = (1 - i)/sqrt2 (forward) or its conjugate
template <bool INV> AW_HD cf mul_w8_1(cf a) {
    return INV ? mk((a.x - a.y) * kS2, (a.x + a.y) * kS2) : mk((a.x + a.y) * kS2, (a.y - a.x) * kS2);
}
// multiply by W8^3 = (-1 - i)/sqrt2 (forward) or its conjugate
template <bool INV> AW_HD cf mul_w8_3(cf a) {
    return INV ? mk((-a.x - a.y) * kS2, (a.x - a.y) * kS2) : mk((a.y - a.x) * kS2, (-a.x - a.y) * kS2);
}

// 8-point DFT, natural order in/out.
template <bool INV> AW_HD void fft8(cf (&v)[8]) {
    cf e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    cf o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    fft4<INV>(e0, e1, e2, e3);
    fft4<INV>(o0, o1, o2, o3);
    o1 = mul_w8_1<INV>(o1);
    o2 = rot90<INV>(o2);
    o3 = mul_w8_3<INV>(o3);
    v[0] = e0 + o0; v[4] = e0 - o0;
    v[1] = e1 + o1; v[5] = e1 - o1;
    v[2] = e2 + o2; v[6] = e2 - o2;
    v[3] = e3 + o3; v[7] = e3 - o3;
}

// multiply by W16^m (forward) / conj (inverse), m compile-time in {0,1,2,3,4,6,9}
template <bool INV, int M> AW_HD cf mul_w16(cf a) {
    if constexpr (M == 0) return a;
    else if constexpr (M == 4) return rot90<INV>(a);
    else if constexpr (M == 2) return mul_w8_1<INV>(a);
    else if constexpr (M == 6) return mul_w8_3<INV>(a);
    else {
        // W16^1 = c - i s, W16^3 = s - i c, W16^9 = -c + i s   (c = cos(pi/8), s = sin(pi/8))
        constexpr float wr = (M == 1) ? kC8 : (M == 3) ? kS8 : -kC8;
        constexpr float wi_f = (M == 1) ? -kS8 : (M == 3) ? -kC8 : kS8;
        const float wi = INV ? -wi_f : wi_f;
        return mk(a.x * wr - a.y * wi, a.x * wi + a.y * wr);
    }
}

// ---- 16-point DFT, natural order in/out (4 x 4) ------------------------------------------------------------------------------
// Second-layer groups with their constant twiddles folded into the butterflies (the vector unit issues an FMA at the price of an
// add): x0 + W x2 and x0 - W x2 for an eighth-turn W are two FMAs on one unscaled sum, W_a x1 +- W_b x3 is one explicit product,
// one FMA-form sum and 2 (W_a x1) - sum.  148 instead of 160 instructions per transform.
#ifndef AW_FFT16_PLAIN
#define AW_FFT16_PLAIN 0        // 1: the plain form (twiddle multiplies, then fft4) for A/B
#endif

// unscaled W8^M x / kS2 (forward) or conj(W8^M) x / kS2, M in {1, 3}: two adds
template <bool INV, int M> AW_HD cf w8_sum(cf a) {
    if constexpr (M == 1) return INV ? mk(a.x - a.y, a.x + a.y) : mk(a.x + a.y, a.y - a.x);
    else return INV ? mk(-a.x - a.y, a.x - a.y) : mk(a.y - a.x, -a.x - a.y);
}
// W16^M as (re, im) of the forward kernel, M in {1, 3, 9}
template <int M> struct W16c {
    static constexpr float re = (M == 1) ? kC8 : (M == 3) ? kS8 : -kC8;
    static constexpr float im = (M == 1) ? -kS8 : (M == 3) ? -kC8 : kS8;
};
// fft4 of (x0, W^K x1, W^2K x2, W^3K x3), W = W16, K in {1, 3}: the general pair (W^K, W^3K) and the eighth turn W^2K
template <bool INV, int K> AW_HD void fft4_w16(cf &x0, cf &x1, cf &x2, cf &x3) {
    constexpr float ar = W16c<K>::re, ai = INV ? -W16c<K>::im : W16c<K>::im;
    constexpr float br = W16c<(3 * K) % 16 == 9 ? 9 : 3>::re, bi = INV ? -W16c<(3 * K) % 16 == 9 ? 9 : 3>::im : W16c<(3 * K) % 16 == 9 ? 9 : 3>::im;
    static_assert(K == 1 || K == 3, "W16^K with K = 1 or 3");
    const cf s = w8_sum<INV, K == 1 ? 1 : 3>(x2);                        // W^2K x2 = kS2 s
    const cf t0 = mk(x0.x + kS2 * s.x, x0.y + kS2 * s.y), t1 = mk(x0.x - kS2 * s.x, x0.y - kS2 * s.y);
    const cf a1 = mk(x1.x * ar - x1.y * ai, x1.x * ai + x1.y * ar);       // W^K x1
    const cf t2 = mk(a1.x + x3.x * br - x3.y * bi, a1.y + x3.x * bi + x3.y * br);      // + W^3K x3
    const cf d = mk(2.0f * a1.x - t2.x, 2.0f * a1.y - t2.y);              // W^K x1 - W^3K x3
    const cf t3 = rot90<INV>(d);
    x0 = t0 + t2; x1 = t1 + t3; x2 = t0 - t2; x3 = t1 - t3;
}
// fft4 of (x0, W8^1 x1, W4^1 x2, W8^3 x3)
template <bool INV> AW_HD void fft4_w8(cf &x0, cf &x1, cf &x2, cf &x3) {
    const cf r = rot90<INV>(x2);
    const cf t0 = x0 + r, t1 = x0 - r;
    const cf s1 = w8_sum<INV, 1>(x1), s3 = w8_sum<INV, 3>(x3);
    const cf u = s1 + s3, w = rot90<INV>(s1 - s3);                        // t2 = kS2 u, t3 = kS2 w
    x0 = mk(t0.x + kS2 * u.x, t0.y + kS2 * u.y); x2 = mk(t0.x - kS2 * u.x, t0.y - kS2 * u.y);
    x1 = mk(t1.x + kS2 * w.x, t1.y + kS2 * w.y); x3 = mk(t1.x - kS2 * w.x, t1.y - kS2 * w.y);
}

template <bool INV> AW_HD void fft16(cf (&v)[16]) {
#ifdef AW_ABL_NOFFT      // timing ablation only (wrong results): butterflies removed, data flow kept
    v[0] = v[0] + v[15];
    return;
#endif
    // F_{n0}[k0] = fft4 over n1 of v[4 n1 + n0]
    fft4<INV>(v[0], v[4], v[8], v[12]);
    fft4<INV>(v[1], v[5], v[9], v[13]);
    fft4<INV>(v[2], v[6], v[10], v[14]);
    fft4<INV>(v[3], v[7], v[11], v[15]);
    // now v[4 k0 + n0] holds F_{n0}[k0]; X[k0 + 4 k1] = fft4 over n0 of W16^{n0 k0} v[4 k0 + n0]  -> result index k1 lands in slot 4 k0 + k1
#if AW_FFT16_PLAIN
    v[5] = mul_w16<INV, 1>(v[5]);  v[6] = mul_w16<INV, 2>(v[6]);   v[7] = mul_w16<INV, 3>(v[7]);
    v[9] = mul_w16<INV, 2>(v[9]);  v[10] = mul_w16<INV, 4>(v[10]); v[11] = mul_w16<INV, 6>(v[11]);
    v[13] = mul_w16<INV, 3>(v[13]); v[14] = mul_w16<INV, 6>(v[14]); v[15] = mul_w16<INV, 9>(v[15]);
    fft4<INV>(v[0], v[1], v[2], v[3]);
    fft4<INV>(v[4], v[5], v[6], v[7]);
    fft4<INV>(v[8], v[9], v[10], v[11]);
    fft4<INV>(v[12], v[13], v[14], v[15]);
#else
    fft4<INV>(v[0], v[1], v[2], v[3]);
    fft4_w16<INV, 1>(v[4], v[5], v[6], v[7]);
    fft4_w8<INV>(v[8], v[9], v[10], v[11]);
    fft4_w16<INV, 3>(v[12], v[13], v[14], v[15]);
#endif
    // slot 4 k0 + k1 holds X[k0 + 4 k1]: transpose to natural order
    cf t;
    t = v[1]; v[1] = v[4]; v[4] = t;
    t = v[2]; v[2] = v[8]; v[8] = t;
    t = v[3]; v[3] = v[12]; v[12] = t;
    t = v[6]; v[6] = v[9]; v[9] = t;
    t = v[7]; v[7] = v[13]; v[13] = t;
    t = v[11]; v[11] = v[14]; v[14] = t;
}

// v[k] *= w^k, k = 1..15; the powers by a depth-4 product tree, each used as soon as it exists (about seven live at a time)
AW_HD void r16_pow_apply(cf (&v)[16], cf w) {

    const cf p2 = cmul(w, w), p4 = cmul(p2, p2), p8 = cmul(p4, p4);
    v[1] = cmul(v[1], w);    v[9] = cmul(v[9], cmul(p8, w));
    v[2] = cmul(v[2], p2);   v[10] = cmul(v[10], cmul(p8, p2));
    const cf p3 = cmul(p2, w);
    v[3] = cmul(v[3], p3);   v[11] = cmul(v[11], cmul(p8, p3));
    v[4] = cmul(v[4], p4);   v[12] = cmul(v[12], cmul(p8, p4));
    const cf p5 = cmul(p4, w);
    v[5] = cmul(v[5], p5);   v[13] = cmul(v[13], cmul(p8, p5));
    const cf p6 = cmul(p4, p2);
    v[6] = cmul(v[6], p6);   v[14] = cmul(v[14], cmul(p8, p6));
    const cf p7 = cmul(p4, p3);
    v[7] = cmul(v[7], p7);   v[15] = cmul(v[15], cmul(p8, p7));
    v[8] = cmul(v[8], p8);
}

}  // namespace awk
