// eq_kernels.hip — gfx950 kernels of the parametric EQ row (device code in eq_cascade.hpp).
#include "eq_kernels.hpp"

#include <cstdlib>

#ifndef AW_EQ_WAVES
#define AW_EQ_WAVES 2
#endif
namespace awk {

namespace {

struct EqGpuCtx {
    cf *lds_;
    __device__ __forceinline__ int tid() const { return (int)threadIdx.x; }
    __device__ __forceinline__ int wave() const { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
    __device__ __forceinline__ cf *lds() const { return lds_; }
    __device__ __forceinline__ void barrier() const { __syncthreads(); }
    // Cross-lane moves of the in-register scan (two DPP v_mov_b32 per double).  Lanes without a source receive zero
    // (row_shr: bound_ctrl; row_bcast: the rows outside the row mask keep the zero `old`).
    template <int CTRL, int ROW_MASK, bool BOUND>
    static __device__ __forceinline__ double dpp(double old, double v) {
        const unsigned long long o = __builtin_bit_cast(unsigned long long, old), u = __builtin_bit_cast(unsigned long long, v);
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)o, (int)(unsigned)u, CTRL, ROW_MASK, 0xf, BOUND);
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(o >> 32), (int)(unsigned)(u >> 32), CTRL, ROW_MASK, 0xf, BOUND);
        return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
    }
    // x = a x + c in x's own register (a tied operand): the filter loop carries the 2 x 16 samples in fixed registers, without
    // it hipcc renames them in every filter and copies all 32 back at the loop's back edge.  `after0/1` only order the
    // statement behind the other readers of the old x.
    __device__ __forceinline__ void fma_in_place(double &x, double a, double c, double after0, double after1) const {
        asm("v_fma_f64 %0, %1, %0, %2" : "+v"(x) : "s"(a), "v"(c), "v"(after0), "v"(after1));
    }
    template <int D> __device__ __forceinline__ double row_shr(double v) const { return dpp<0x110 + D, 0xf, true>(0.0, v); }   // lane - D of the same 16-lane row
    __device__ __forceinline__ double row_bcast15(double v) const { return dpp<0x142, 0xa, false>(0.0, v); }   // rows 1, 3 <- lane 15 of the row below
    __device__ __forceinline__ double row_bcast31(double v) const { return dpp<0x143, 0xc, false>(0.0, v); }   // rows 2, 3 <- lane 31
    __device__ __forceinline__ double wave_shr1(double v, double fill) const { return dpp<0x138, 0xf, false>(fill, v); }   // lane - 1; lane 0 <- fill
};

}  // namespace

// The tables come in as `const __restrict__` kernel arguments (not inside the struct): with a wave-uniform
// index that is what lets hipcc read them with scalar loads.  in/out may alias (in place) and are not restrict.
// E = 2: one workgroup per stream, two waves per SIMD.  E = 1: one per (stream, ear), three waves per SIMD (<= 168 VGPRs,
// 38 KB of LDS); workgroups b and b + 8 — the same XCD under the round-robin dispatch — take the two ears of one stream,
// so the frames both of them load cross the fabric once.
template <int E>
__global__ void __launch_bounds__(kEqThreads, E == 2 ? AW_EQ_WAVES : (kEqChunk <= 16 ? 4 : kEqChunk <= 32 ? 3 : 2)) aw_eq_cascade_kernel(EqParams p, const double *__restrict__ tab,
                                                                                      const double *__restrict__ plane, int n_streams) {
    extern __shared__ __align__(16) unsigned char eq_lds[];
    EqGpuCtx ctx{reinterpret_cast<cf *>(eq_lds)};
    p.t.tab = tab;
    p.t.plane = plane;
    if constexpr (E == 2) eq_cascade_stream<EqGpuCtx, 2>(ctx, p, (long long)blockIdx.x, 0);
    else {
        const int b = (int)blockIdx.x, stream = (b >> 4) * 8 + (b & 7);
        if (stream < n_streams) eq_cascade_stream<EqGpuCtx, 1>(ctx, p, (long long)stream, (b >> 3) & 1);
    }
}

__global__ void aw_eq_sequential_kernel(EqParams p, int n_streams) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < n_streams * 2) eq_sequential(p, i >> 1, i & 1);
}

__global__ void aw_eq_blend_kernel(const float *__restrict__ old_seg, const float *__restrict__ new_seg, float *__restrict__ out,
                                   long long seg_frames, long long out_stride, long long t_frame, long long length) {
#pragma clang fp contract(off)
    const long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long s = blockIdx.y;
    if (f >= seg_frames) return;
    const double progress = (double)(t_frame + f + 1) / (double)length;
    const double inverse = 1 - progress;
    const float2 o = reinterpret_cast<const float2 *>(old_seg)[s * seg_frames + f];
    const float2 n = reinterpret_cast<const float2 *>(new_seg)[s * seg_frames + f];
    float2 r;
    r.x = (float)((double)o.x * inverse + (double)n.x * progress);
    r.y = (float)((double)o.y * inverse + (double)n.y * progress);
    reinterpret_cast<float2 *>(out)[s * out_stride + f] = r;
}

__global__ void aw_eq_copy_kernel(const float *__restrict__ src, long long src_stride, float *__restrict__ dst,
                                  long long dst_stride, long long frames) {
    const long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long s = blockIdx.y;
    if (f < frames) reinterpret_cast<float2 *>(dst)[s * dst_stride + f] = reinterpret_cast<const float2 *>(src)[s * src_stride + f];
}

// 74 KB (both ears) / 38 KB of dynamic LDS: above the 64 KB a kernel gets without asking.  Called from aw_context_create.
hipError_t prepare_eq_kernels() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_eq_cascade_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, eq_lds_bytes(2));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_eq_cascade_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, eq_lds_bytes(1));
    return e;
}

hipError_t launch_eq_cascade(const EqParams &p, int n_streams, hipStream_t stream) {
    if (n_streams <= 0 || p.frames <= 0) return hipSuccess;
    // split the ears over two workgroups while the per-ear grid still fits the CUs in one round (three workgroups per CU):
    // 128 streams 2.34 -> 1.76 ms, 384 streams 3.80 -> 3.42 ms; 448 streams 3.92 vs 4.53 ms and 512 streams 4.14 vs 4.61 ms
    // the other way (tools/archive/ab_eq_split.sh)
    const int cus = p.cus > 0 ? p.cus : 256, force = p.ear_split;       // from the context (read once at its creation)
    const bool split = force >= 0 ? force != 0 : 2 * n_streams <= 3 * cus;
    if (split)
        hipLaunchKernelGGL(aw_eq_cascade_kernel<1>, dim3((unsigned)((n_streams + 7) / 8) * 16), dim3(kEqThreads), eq_lds_bytes(1), stream, p, p.t.tab, p.t.plane, n_streams);
    else
        hipLaunchKernelGGL(aw_eq_cascade_kernel<2>, dim3((unsigned)n_streams), dim3(kEqThreads), eq_lds_bytes(2), stream, p, p.t.tab, p.t.plane, n_streams);
    return hipGetLastError();
}

hipError_t launch_eq_sequential(const EqParams &p, int n_streams, hipStream_t stream) {
    if (n_streams <= 0 || p.frames <= 0) return hipSuccess;
    const int n = n_streams * 2;
    hipLaunchKernelGGL(aw_eq_sequential_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, p, n_streams);
    return hipGetLastError();
}

hipError_t launch_eq_blend(const float *old_seg, const float *new_seg, float *out, int n_streams, long long seg_frames,
                           long long out_stride, long long t_frame, long long length, hipStream_t stream) {
    if (n_streams <= 0 || seg_frames <= 0) return hipSuccess;
    hipLaunchKernelGGL(aw_eq_blend_kernel, dim3((unsigned)((seg_frames + 255) / 256), (unsigned)n_streams), dim3(256), 0, stream,
                       old_seg, new_seg, out, seg_frames, out_stride, t_frame, length);
    return hipGetLastError();
}

hipError_t launch_eq_copy(const float *src, long long src_stride, float *dst, long long dst_stride, int n_streams,
                          long long frames, hipStream_t stream) {
    if (n_streams <= 0 || frames <= 0) return hipSuccess;
    hipLaunchKernelGGL(aw_eq_copy_kernel, dim3((unsigned)((frames + 255) / 256), (unsigned)n_streams), dim3(256), 0, stream, src,
                       src_stride, dst, dst_stride, frames);
    return hipGetLastError();
}

}  // namespace awk
