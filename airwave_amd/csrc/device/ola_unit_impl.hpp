// ola_unit_impl.hpp — body of one ola_kernels_<x>.hip unit: define AW_OLA_UNIT_LIST, AW_OLA_UNIT_LAUNCH and AW_OLA_UNIT_PREPARE, then include.
#include "ola_inst.hpp"
#include "ola_kernel.hpp"

namespace awk {

hipError_t AW_OLA_UNIT_PREPARE() {
    hipError_t e = hipSuccess;
#define AW_SET(CS, H)                                                                                               \
    if (e == hipSuccess)                                                                                            \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&aw_fused_ola_kernel<CS, (CS + 1) / 2, H>),          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    AW_OLA_UNIT_LIST(AW_SET)
#undef AW_SET
    return e;
}

bool AW_OLA_UNIT_LAUNCH(const TileParams &p, int H, dim3 grid, long long n_tiles, hipStream_t stream) {
#define AW_CASE(CS, HH)                                                                                                              \
    if (p.n_channels == CS && H == HH) {                                                                                             \
        hipLaunchKernelGGL((aw_fused_ola_kernel<CS, (CS + 1) / 2, HH>), grid, dim3(kThreads), kLdsBytes, stream, p, n_tiles);        \
        return true;                                                                                                                 \
    }
    AW_OLA_UNIT_LIST(AW_CASE)
#undef AW_CASE
    return false;
}

}  // namespace awk
