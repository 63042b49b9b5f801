// lw_split_impl.hpp — definitions behind lw_split_inst.hpp; included by lw_split_a .. e.hip only, which instantiate their RA values.
#pragma once
#include "kernels.hpp"
#include "gpu_ctx.hpp"
#include "tile_lw.hpp"

namespace awk {

// Tile id = (stream, window) * 64 + t-chunk: workgroups that run at the same time read and write neighbouring 64-frame
// pieces of the same R strided sub-sequences (whole DRAM pages between them).
template <int RA, int CS>
__global__ void __launch_bounds__(kThreads, RA > 8 ? 2 : 4) aw_lw_split_kernel(LwParams p, long long n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GpuCtx ctx{reinterpret_cast<cf *>(smem), nullptr};
    lw_split_tiles<GpuCtx, RA, CS>(ctx, p, (long long)blockIdx.x, (long long)gridDim.x, n_tiles);
}

template <int RA, int CS1>
__global__ void __launch_bounds__(kThreads, RA > 8 ? 2 : 4) aw_lw_split_wide_kernel(LwParams p, long long n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GpuCtx ctx{reinterpret_cast<cf *>(smem), nullptr};
    lw_split_wide_tiles<GpuCtx, RA, CS1>(ctx, p, (long long)blockIdx.x, (long long)gridDim.x, n_tiles);
}

template <int RA> constexpr int lw_split_lds_bytes() { return lw_split_lds_elems<RA>() * (int)sizeof(cf); }

#define AW_LW_FOR_CS(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)

template <int RA> hipError_t lw_split_prepare() {
    hipError_t e = hipSuccess;
#define AW_SET(CS)                                                                                                       \
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_lw_split_kernel<RA, CS>),            \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, lw_split_lds_bytes<RA>()); \
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_lw_split_wide_kernel<RA, CS>),       \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, lw_split_lds_bytes<RA>());
    AW_LW_FOR_CS(AW_SET)
#undef AW_SET
    return e;
}

template <int RA> hipError_t lw_split_launch(const LwParams &p, bool wide, int cs, dim3 grid, hipStream_t stream, long long n_tiles) {
    switch (cs) {
#define AW_CASE(CS)                                                                                                                          \
        case CS:                                                                                                                             \
            if (wide) hipLaunchKernelGGL((aw_lw_split_wide_kernel<RA, CS>), grid, dim3(kThreads), lw_split_lds_bytes<RA>(), stream, p, n_tiles); \
            else hipLaunchKernelGGL((aw_lw_split_kernel<RA, CS>), grid, dim3(kThreads), lw_split_lds_bytes<RA>(), stream, p, n_tiles);           \
            break;
        AW_LW_FOR_CS(AW_CASE)
#undef AW_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

#define AW_LW_SPLIT_INSTANTIATE(RA)                        \
    template hipError_t lw_split_prepare<RA>();            \
    template hipError_t lw_split_launch<RA>(const LwParams &, bool, int, dim3, hipStream_t, long long);

}  // namespace awk
