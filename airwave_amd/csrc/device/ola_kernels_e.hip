// ola_kernels_e.hip — one build unit of the overlap-add tile kernels (ola_inst.hpp lists which); units exist to let hipcc work side by side.
#define AW_OLA_UNIT_LIST AW_OLA_LAYOUTS_E
#define AW_OLA_UNIT_LAUNCH launch_ola_e
#define AW_OLA_UNIT_PREPARE prepare_ola_e
#include "ola_unit_impl.hpp"
