// Forwarding header: same include path as the reference's src/Interactor/NeighbourList/VerletList/NeighbourContainer.cuh (the Verlet list's
// NeighbourContainer: device/Transverser.hip.hpp under hipcc).
#pragma once
#if defined(DOUBLE_PRECISION)
#error "NeighbourContainer.cuh: this module has a single-precision backend only on MI355X (uammd.h, PRECISION): build without -DDOUBLE_PRECISION"
#endif
#include "../../../uammd.h"
#if defined(__HIPCC__)
#include "../../../device/Transverser.hip.hpp"
#endif
