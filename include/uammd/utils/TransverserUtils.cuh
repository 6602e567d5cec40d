// Forwarding header: same include path as the reference's src/utils/TransverserUtils.cuh.
// The whole host interface of the MI355X build lives in uammd.h; the Transverser concept's optional-member dispatch is in device/Transverser.hip.hpp (hipcc translation units).
#pragma once
#include "../uammd.h"
#if defined(__HIPCC__)
#include "../device/Transverser.hip.hpp"
#endif
