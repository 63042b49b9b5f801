// lw_split_d.hip — split kernels of the long-window path for RA = 14 (window lengths R = 8 RA rows of 4096 frames);
// see lw_split_inst.hpp.  Built with -fno-slp-vectorize like the other tile kernels (airwave_amd/build.py).
#include "lw_split_impl.hpp"

namespace awk {
AW_LW_SPLIT_INSTANTIATE(14)
}  // namespace awk
