// ola_kernels_f.hip — one build unit of the overlap-add tile kernels (ola_inst.hpp lists which); units exist to let hipcc work side by side.
#define AW_OLA_UNIT_LIST AW_OLA_LAYOUTS_F
#define AW_OLA_UNIT_LAUNCH launch_ola_f
#define AW_OLA_UNIT_PREPARE prepare_ola_f
#include "ola_unit_impl.hpp"
