// Forwarding header: same include path as the reference's src/Interactor/NeighbourList/common.cuh.
// The whole host interface of the MI355X build lives in uammd.h; transverseWithNeighbourContainer is in device/Transverser.hip.hpp (hipcc translation units).
#pragma once
#include "../../uammd.h"
#if defined(__HIPCC__)
#include "../../device/Transverser.hip.hpp"
#endif
