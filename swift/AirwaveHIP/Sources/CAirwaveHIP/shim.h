#include "airwave_hip.h"
