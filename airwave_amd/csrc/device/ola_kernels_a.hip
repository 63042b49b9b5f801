// ola_kernels_a.hip — one build unit of the overlap-add tile kernels (ola_inst.hpp lists which); units exist to let hipcc work side by side.
#define AW_OLA_UNIT_LIST AW_OLA_LAYOUTS_A
#define AW_OLA_UNIT_LAUNCH launch_ola_a
#define AW_OLA_UNIT_PREPARE prepare_ola_a
#include "ola_unit_impl.hpp"
