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
    // exchanges inside one wave: LDS instructions of a wave execute in issue order, the fences only
    // pin the compiler's ordering
    __device__ __forceinline__ void wave_sync() const {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};

}  // namespace

// The tables come in as `const __restrict__` kernel arguments (not inside the struct): with a wave-uniform
// index that is what lets hipcc read them with scalar loads.  in/out may alias (in place) and are not restrict.
// E = 2: one workgroup per stream; E = 1: one per (stream, ear).
template <int E>
__global__ void __launch_bounds__(kEqThreads, AW_EQ_WAVES) aw_eq_cascade_kernel(EqParams p, const double *__restrict__ tab,
                                                                        const double *__restrict__ plane) {
    extern __shared__ __align__(16) unsigned char eq_lds[];
    EqGpuCtx ctx{reinterpret_cast<cf *>(eq_lds)};
    p.t.tab = tab;
    p.t.plane = plane;
    if constexpr (E == 2) eq_cascade_stream<EqGpuCtx, 2>(ctx, p, (long long)blockIdx.x, 0);
    else eq_cascade_stream<EqGpuCtx, 1>(ctx, p, (long long)(blockIdx.x >> 1), (int)(blockIdx.x & 1));
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

hipError_t launch_eq_cascade(const EqParams &p, int n_streams, hipStream_t stream) {
    if (n_streams <= 0 || p.frames <= 0) return hipSuccess;
    // split the ears over two workgroups while one workgroup per stream leaves CUs idle (measured: 128 streams
    // 2.84 -> 2.25 ms; at 512 streams the unsplit kernel wins, 7.1 vs 9.8 ms: the split reads every line twice
    // and halves each thread's independent FMA chains)
    const int cus = p.cus > 0 ? p.cus : 256, force = p.ear_split;       // from the context (read once at its creation)
    const bool split = force >= 0 ? force != 0 : 2 * n_streams < 3 * cus;
    if (split)
        hipLaunchKernelGGL(aw_eq_cascade_kernel<1>, dim3((unsigned)n_streams * 2), dim3(kEqThreads), kEqLdsBytes, stream, p, p.t.tab, p.t.plane);
    else
        hipLaunchKernelGGL(aw_eq_cascade_kernel<2>, dim3((unsigned)n_streams), dim3(kEqThreads), kEqLdsBytes, stream, p, p.t.tab, p.t.plane);
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
