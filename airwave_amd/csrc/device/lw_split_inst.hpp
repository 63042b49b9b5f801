// lw_split_inst.hpp — the split kernels of the long-window path (tile_lw.hpp, kernel 1), one set per window length R = 8 RA.
// The eleven RA values x eight channel counts x {narrow, wide} are 176 kernels: they are instantiated in five translation units
// (lw_split_a .. e.hip) that build side by side; lw_kernels.hip only sees the two entry points per RA.
#pragma once
#include "kernels.hpp"
#include "tile_lw.hpp"

namespace awk {

template <int RA> hipError_t lw_split_prepare();          // dynamic-LDS attribute of every kernel of this RA
// cs: channels of a narrow layout (1-8), or channels beyond the first eight of a wide one (1-8)
template <int RA> hipError_t lw_split_launch(const LwParams &p, bool wide, int cs, dim3 grid, hipStream_t stream, long long n_tiles);

#define AW_LW_SPLIT_EXTERN(RA)                             \
    extern template hipError_t lw_split_prepare<RA>();     \
    extern template hipError_t lw_split_launch<RA>(const LwParams &, bool, int, dim3, hipStream_t, long long);
AW_LW_SPLIT_EXTERN(4) AW_LW_SPLIT_EXTERN(5) AW_LW_SPLIT_EXTERN(6) AW_LW_SPLIT_EXTERN(7) AW_LW_SPLIT_EXTERN(8) AW_LW_SPLIT_EXTERN(9) AW_LW_SPLIT_EXTERN(10)
AW_LW_SPLIT_EXTERN(12) AW_LW_SPLIT_EXTERN(14) AW_LW_SPLIT_EXTERN(15) AW_LW_SPLIT_EXTERN(16)
#undef AW_LW_SPLIT_EXTERN

}  // namespace awk
