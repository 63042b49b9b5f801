// prep_kernels.hip — the HRIR-prep kernels of the long-window path: the frequency-domain filter tables of one window length, computed
// on the GPU in float64 ("HRIR-prep kernel (partition + FFT)").
//
// Reference semantics: ConvolutionEngine.init transforms every partition of the impulse response once, at engine creation
// (Airwave/ConvolutionEngine.swift:141-175: zero-pad, vDSP_ctoz, forward fft_zrip into H[p]).  Here one window holds the whole response,
// and what the rows kernel multiplies by (tile_lw.hpp / tile_lw16.hpp) is, per channel pair p = (a, b) and bin k of the ODD-frequency
// transform of length N = R x 4096 (k' = N - 1 - k its Hermitian partner):
//     zl = odd_DFT_N(h[left(a)] + i h[left(b)]),   zr = odd_DFT_N(h[right(a)] + i h[right(b)])
//     A[k] = (conj(zl[k']) + i conj(zr[k'])) / 2N,   B[k] = (zl[k] + i zr[k]) / 2N
//     T0 = A[k], T1 = B[k], T2 = conj(A[k']), T3 = conj(B[k'])        (a real last channel: T0 += T1, T3 += T2, T1 = T2 = 0)
// stored for the 16-point rows kernel as {T0, T3} (row ra) and {T1, T2} (row rb) at [row pair][pair][row][m1][thread]
// (host/tables.hpp; the float64 host builder build_lw_tables computes the same values and remains the reference the CPU emulation
// tests run against).
//
// Kernel 1 (aw_lw_prep_rows_kernel), one 256-thread workgroup per (sequence zl | zr of a pair, row k1 of the four-step split N = R x 4096):
//     d[t] = sum_j x[4096 j + t] w_2N^{(4096 j + t)(2 k1 + 1)}      (j < ceil(taps / 4096): the response is short against the window; the odd
//                                                                    frequency offset, the size-R DFT and the four-step twiddle are ONE phase)
//     X[k1 + R k2] = FFT_4096(d)[k2]  as 64 x 64: two passes of direct 64-point DFTs through LDS (float64; twiddles by sincospi on exactly
//     reduced integer phases, so every twiddle is correct to an ulp whatever the length)
// Kernel 2 (aw_lw_prep_assemble_kernel): the table entries from X at k and k', rounded to float32 once.
// Cost for cfg 3 (N = 524 288, 4 pairs): 8 x 128 workgroups x 2 passes of 4096 x 64 complex multiply-adds = 4.3 G FP64 FMAs.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace awk {

struct cdbl { double x, y; };
__device__ __forceinline__ cdbl dmul(cdbl a, cdbl b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cdbl dfma(cdbl a, cdbl b, cdbl c) { return {fma(a.x, b.x, fma(-a.y, b.y, c.x)), fma(a.x, b.y, fma(a.y, b.x, c.y))}; }
// exp(-2 pi i num / den), num >= 0: the phase is reduced exactly in integers first
__device__ __forceinline__ cdbl dunit(long long num, long long den) {
    double s, c;
    sincospi(-2.0 * (double)(num % den) / (double)den, &s, &c);
    return {c, s};
}

constexpr int kPrepThreads = 256;
constexpr int kPrepPitch = 65;                                  // E[a][b'] at b' + 65 a: both passes touch 16 consecutive 16-byte slots per 16-lane group
constexpr int kPrepLdsBytes = (64 * kPrepPitch + 64) * (int)sizeof(cdbl);

__global__ void __launch_bounds__(kPrepThreads) aw_lw_prep_rows_kernel(const float *__restrict__ tracks, int n_tracks, int taps, int n_channels,
                                                                       const int *__restrict__ left_track, const int *__restrict__ right_track,
                                                                       int R, cdbl *__restrict__ Z) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cdbl *buf = reinterpret_cast<cdbl *>(smem);
    cdbl *w64 = buf + 64 * kPrepPitch;
    const int tid = (int)threadIdx.x, lane = tid & 63, grp = tid >> 6;
    const int k1 = (int)blockIdx.x, seq = (int)blockIdx.y, p = seq >> 1, ear = seq & 1;
    const long long N = (long long)R * kLwM;
    const int a = 2 * p, b = 2 * p + 1;
    const int la = left_track[a], ra = right_track[a];
    const int lb = b < n_channels ? left_track[b] : -1, rb = b < n_channels ? right_track[b] : -1;
    // a channel is rendered only when BOTH ears are mapped (HRIRManager.swift:370-372: the speaker is skipped otherwise)
    const bool use_a = la >= 0 && ra >= 0 && la < n_tracks && ra < n_tracks, use_b = lb >= 0 && rb >= 0 && lb < n_tracks && rb < n_tracks;
    const float *ha = use_a ? tracks + (size_t)(ear ? ra : la) * taps : nullptr;
    const float *hb = use_b ? tracks + (size_t)(ear ? rb : lb) * taps : nullptr;
    if (tid < 64) w64[tid] = dunit(tid, 64);
    // d[t], t = lane + 64 bb (bb = grp + 4 i): stored at bb + 65 lane, the layout pass 1 reads column-wise
    const int J = (taps + kLwM - 1) / kLwM;
#pragma unroll 1
    for (int i = 0; i < 16; ++i) {
        const int bb = grp + 4 * i, t = lane + 64 * bb;
        cdbl acc = {0.0, 0.0};
        for (int j = 0; j < J; ++j) {
            const long long n = (long long)kLwM * j + t;
            if (n >= taps) break;
            const cdbl x = {ha ? (double)ha[n] : 0.0, hb ? (double)hb[n] : 0.0};
            acc = dfma(x, dunit(n * (2 * k1 + 1), 2 * N), acc);
        }
        buf[bb + kPrepPitch * lane] = acc;
    }
    __syncthreads();
    // pass 1: E[a][b'] = w_4096^{a b'} sum_b d[a + 64 b] w_64^{b b'};  thread: a = lane, b' = grp + 4 i (wave-uniform)
    cdbl e[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) e[i] = {0.0, 0.0};
#pragma unroll 1
    for (int bb = 0; bb < 64; ++bb) {
        const cdbl v = buf[bb + kPrepPitch * lane];
#pragma unroll
        for (int i = 0; i < 16; ++i) e[i] = dfma(v, w64[(bb * (grp + 4 * i)) & 63], e[i]);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int bp = grp + 4 * i;
        buf[bp + kPrepPitch * lane] = dmul(e[i], dunit((long long)lane * bp, kLwM));
    }
    __syncthreads();
    // pass 2: X[64 a' + b'] = sum_a E[a][b'] w_64^{a a'};  thread: b' = lane, a' = grp + 4 i
#pragma unroll
    for (int i = 0; i < 16; ++i) e[i] = {0.0, 0.0};
#pragma unroll 1
    for (int aa = 0; aa < 64; ++aa) {
        const cdbl v = buf[lane + kPrepPitch * aa];
#pragma unroll
        for (int i = 0; i < 16; ++i) e[i] = dfma(v, w64[(aa * (grp + 4 * i)) & 63], e[i]);
    }
    cdbl *dst = Z + ((size_t)seq * R + k1) * kLwM;
#pragma unroll
    for (int i = 0; i < 16; ++i) dst[64 * (grp + 4 * i) + lane] = e[i];
}

// one thread per (row pair rp, channel pair p, row bin k2)
__global__ void __launch_bounds__(256) aw_lw_prep_assemble_kernel(const cdbl *__restrict__ Z, int R, int n_pairs, int real_last, LwTab2 *__restrict__ tab16) {
    const int k2 = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int rp = (int)blockIdx.y, p = (int)blockIdx.z;
    if (k2 >= kLwM) return;
    const double scale = 1.0 / (2.0 * (double)R * (double)kLwM);
    const cdbl *zl = Z + (size_t)(2 * p) * R * kLwM, *zr = Z + (size_t)(2 * p + 1) * R * kLwM;
    const size_t ik = (size_t)rp * kLwM + k2, ikp = (size_t)(R - 1 - rp) * kLwM + (kLwM - 1 - k2);       // k = rp + R k2,  k' = N - 1 - k
    const cdbl lk = zl[ik], lkp = zl[ikp], rk = zr[ik], rkp = zr[ikp];
    // conj(l) + i conj(r) = (l.x + r.y) + i (r.x - l.y);   l + i r = (l.x - r.y) + i (l.y + r.x)
    cdbl t0 = {(lkp.x + rkp.y) * scale, (rkp.x - lkp.y) * scale};             // A[k]
    cdbl t1 = {(lk.x - rk.y) * scale, (lk.y + rk.x) * scale};                 // B[k]
    cdbl t2 = {(lk.x + rk.y) * scale, -(rk.x - lk.y) * scale};                // conj(A[k'])
    cdbl t3 = {(lkp.x - rkp.y) * scale, -(lkp.y + rkp.x) * scale};            // conj(B[k'])
    if (real_last && p == n_pairs - 1) { t0.x += t1.x; t0.y += t1.y; t3.x += t2.x; t3.y += t2.y; t1 = {0.0, 0.0}; t2 = {0.0, 0.0}; }
    // bin k2 = r16_bin(thread, m1) = (thread >> 4) + 16 (thread & 15) + 256 m1
    const int m1 = k2 >> 8, rem = k2 & 255, th = ((rem & 15) << 4) | (rem >> 4);
    LwTab2 *base = tab16 + (((size_t)rp * n_pairs + p) * 2) * kLwM + (size_t)m1 * kR16Threads + th;
    base[0] = LwTab2{mk((float)t0.x, (float)t0.y), mk((float)t3.x, (float)t3.y)};
    base[kLwM] = LwTab2{mk((float)t1.x, (float)t1.y), mk((float)t2.x, (float)t2.y)};
}

size_t lw_prep_scratch_bytes(int n_channels, int R) { return (size_t)2 * ((n_channels + 1) / 2) * R * kLwM * sizeof(cdbl); }

hipError_t prepare_prep_kernels() {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_lw_prep_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kPrepLdsBytes);
}

hipError_t launch_lw_prep(const float *d_tracks, int n_tracks, int taps, int n_channels, const int *d_left, const int *d_right, int R,
                          void *d_scratch, LwTab2 *d_tab16, hipStream_t stream) {
    const int n_pairs = (n_channels + 1) / 2;
    if (n_pairs < 1 || n_pairs > 8 || R < 8 || taps < 1) return hipErrorInvalidValue;
    cdbl *Z = reinterpret_cast<cdbl *>(d_scratch);
    hipLaunchKernelGGL(aw_lw_prep_rows_kernel, dim3((unsigned)R, (unsigned)(2 * n_pairs)), dim3(kPrepThreads), kPrepLdsBytes, stream,
                       d_tracks, n_tracks, taps, n_channels, d_left, d_right, R, Z);
    hipLaunchKernelGGL(aw_lw_prep_assemble_kernel, dim3(kLwM / 256, (unsigned)(R / 2), (unsigned)n_pairs), dim3(256), 0, stream,
                       Z, R, n_pairs, n_channels & 1, d_tab16);
    return hipGetLastError();
}

}  // namespace awk
