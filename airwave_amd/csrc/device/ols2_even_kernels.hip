// ols2_even_kernels.hip — the 16384-frame-window kernels of 4, 6 and 8 channels, in a translation unit of their
// own: through round 3 they kept hipcc's SLP vectoriser (see ols2_kernel.hpp; since the half-wave row transform of round 4 they are
// built without it like everything else, build.py), and the unit builds beside kernels.hip.  Everything else about them is in kernels.hip.
#include "ols2_kernel.hpp"

namespace awk {

#define AW_FOR_EACH_VEC2_EVEN(X) X(4, 2) X(6, 3) X(8, 4)

hipError_t prepare_ols2_even() {
    hipError_t e = hipSuccess;
#define AW_SET(CS, NB)                                                                                \
    if (e == hipSuccess)                                                                              \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_fused_ols2_kernel<CS, NB, true>),  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);               \
    if (e == hipSuccess)                                                                              \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_fused_ols2_kernel<CS, NB, false>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    AW_FOR_EACH_VEC2_EVEN(AW_SET)
#undef AW_SET
    return e;
}

void launch_ols2_even(const TileParams &p, bool interior, long long n_tiles, dim3 grid, hipStream_t stream) {
    const dim3 block(kThreads);
    if (interior) {
        switch (p.n_channels) {
#define AW_CASE(CS, NB) case CS: hipLaunchKernelGGL((aw_fused_ols2_kernel<CS, NB, true>), grid, block, kLdsBytes, stream, p, n_tiles); break;
            AW_FOR_EACH_VEC2_EVEN(AW_CASE)
#undef AW_CASE
            default: break;
        }
    } else {
        switch (p.n_channels) {
#define AW_CASE(CS, NB) case CS: hipLaunchKernelGGL((aw_fused_ols2_kernel<CS, NB, false>), grid, block, kLdsBytes, stream, p, n_tiles); break;
            AW_FOR_EACH_VEC2_EVEN(AW_CASE)
#undef AW_CASE
            default: break;
        }
    }
}

}  // namespace awk
