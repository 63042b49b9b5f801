// lw_split_c.hip — split kernels of the long-window path for RA = 10, 12 (window lengths R = 8 RA rows of 4096 frames);
// see lw_split_inst.hpp.  Built with -fno-slp-vectorize like the other tile kernels (airwave_amd/build.py).
#include "lw_split_impl.hpp"

namespace awk {
AW_LW_SPLIT_INSTANTIATE(10)
AW_LW_SPLIT_INSTANTIATE(12)
}  // namespace awk
