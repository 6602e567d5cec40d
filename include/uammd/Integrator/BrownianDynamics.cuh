// Forwarding header: same include path as the reference's src/Integrator/BrownianDynamics.cuh.
// The whole host interface of the MI355X build lives in uammd.h (C++14, no device code).  Both precisions (uammd.h, PRECISION).
#pragma once
#include "../uammd.h"
