// Forwarding header: same include path as the reference's src/Interactor/Potential/ParameterHandler.cuh (BasicParameterHandler<Functor>, :8-66:
// the type-pair table of a Radial potential; device code of the user's translation unit, so hipcc).
#pragma once
#if defined(DOUBLE_PRECISION)
#error "ParameterHandler.cuh: this module has a single-precision backend only on MI355X (uammd.h, PRECISION): build without -DDOUBLE_PRECISION"
#endif
#include "../../uammd.h"
#if defined(__HIPCC__)
#include "../../device/PairForces.hip.hpp"
#endif
