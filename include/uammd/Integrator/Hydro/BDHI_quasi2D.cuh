// Forwarding header: same include path as the reference's src/Integrator/Hydro/BDHI_quasi2D.cuh (BDHI::True2D, BDHI::Quasi2D).
#pragma once
#if defined(DOUBLE_PRECISION)
#error "BDHI_quasi2D.cuh: this module has a single-precision backend only on MI355X (uammd.h, PRECISION): build without -DDOUBLE_PRECISION"
#endif
#include "../../uammd.h"
