// Forwarding header: same include path as the reference's src/Integrator/BDHI/FIB.cuh (BDHI::FIB).
#pragma once
#if defined(DOUBLE_PRECISION)
#error "FIB.cuh: this module has a single-precision backend only on MI355X (uammd.h, PRECISION): build without -DDOUBLE_PRECISION"
#endif
#include "../../uammd.h"
