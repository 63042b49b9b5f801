// probe_kernels.hip — the measured-ceiling probe behind aw_context_bandwidth_probe ("confirm on the box and also quote a
// measured copy-kernel ceiling"): a read-only, a write-only and a copy kernel over one large buffer, 16 bytes per lane, four
// independent accesses in flight per lane, plain and non-temporal forms (the caller keeps the faster one).  Diagnostic only: nothing
// on a process path launches these.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace awk {

typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ v4f probe_ld(const v4f *p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT> __device__ __forceinline__ void probe_st(v4f *p, v4f v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// n4 is a multiple of 4 * gridDim.x * blockDim.x (the launcher rounds the byte count down): no tail
template <bool NT> __global__ void __launch_bounds__(256) aw_probe_read_kernel(const v4f *src, size_t n4, float *sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += 4 * stride) {
        const v4f a = probe_ld<NT>(src + i), b = probe_ld<NT>(src + i + stride), c = probe_ld<NT>(src + i + 2 * stride), d = probe_ld<NT>(src + i + 3 * stride);
        acc += (a + b) + (c + d);
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345e-30f) *sink = acc.x;       // keeps the loads alive; never true for the zero-filled buffer
}
template <bool NT> __global__ void __launch_bounds__(256) aw_probe_write_kernel(v4f *dst, size_t n4, float v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const v4f x = {v, v + 1.f, v + 2.f, v + 3.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += 4 * stride) {
        probe_st<NT>(dst + i, x); probe_st<NT>(dst + i + stride, x); probe_st<NT>(dst + i + 2 * stride, x); probe_st<NT>(dst + i + 3 * stride, x);
    }
}
template <bool NT> __global__ void __launch_bounds__(256) aw_probe_copy_kernel(const v4f *src, v4f *dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += 4 * stride) {
        const v4f a = probe_ld<NT>(src + i), b = probe_ld<NT>(src + i + stride), c = probe_ld<NT>(src + i + 2 * stride), d = probe_ld<NT>(src + i + 3 * stride);
        probe_st<NT>(dst + i, a); probe_st<NT>(dst + i + stride, b); probe_st<NT>(dst + i + 2 * stride, c); probe_st<NT>(dst + i + 3 * stride, d);
    }
}

// what: 0 read, 1 write, 2 copy; nt: non-temporal accesses.  Returns the bytes one launch moves through `*bytes_moved` (0: nothing to do).
hipError_t launch_bw_probe(int what, bool nt, const void *src, void *dst, size_t bytes, int cus, float *sink, hipStream_t stream, size_t *bytes_moved) {
    const unsigned grid = (unsigned)(cus > 0 ? cus : 256) * 8u;
    const size_t quantum = (size_t)grid * 256u * 4u;                 // 16-byte elements per sweep of the unrolled loop
    const size_t n4 = bytes / 16 / quantum * quantum;
    *bytes_moved = n4 * 16 * (what == 2 ? 2 : 1);
    if (n4 == 0) return hipSuccess;
    const v4f *s = reinterpret_cast<const v4f *>(src);
    v4f *d = reinterpret_cast<v4f *>(dst);
    switch (what) {
        case 0:
            if (nt) hipLaunchKernelGGL(aw_probe_read_kernel<true>, dim3(grid), dim3(256), 0, stream, s, n4, sink);
            else hipLaunchKernelGGL(aw_probe_read_kernel<false>, dim3(grid), dim3(256), 0, stream, s, n4, sink);
            break;
        case 1:
            if (nt) hipLaunchKernelGGL(aw_probe_write_kernel<true>, dim3(grid), dim3(256), 0, stream, d, n4, 1.0f);
            else hipLaunchKernelGGL(aw_probe_write_kernel<false>, dim3(grid), dim3(256), 0, stream, d, n4, 1.0f);
            break;
        default:
            if (nt) hipLaunchKernelGGL(aw_probe_copy_kernel<true>, dim3(grid), dim3(256), 0, stream, s, d, n4);
            else hipLaunchKernelGGL(aw_probe_copy_kernel<false>, dim3(grid), dim3(256), 0, stream, s, d, n4);
            break;
    }
    return hipGetLastError();
}

}  // namespace awk
