// Forwarding header: same include path as the reference's src/Integrator/VerletNVT.cuh.
// The whole host interface of the MI355X build lives in uammd.h (C++14, no device code).
#pragma once
#if defined(DOUBLE_PRECISION)
#error "VerletNVT.cuh: this module has a single-precision backend only on MI355X (uammd.h, PRECISION): build without -DDOUBLE_PRECISION"
#endif
#include "../uammd.h"
