// Forwarding header: same include path as the reference's src/Integrator/BDHI/BDHI_Lanczos.cuh (both precisions).
// The whole host interface of the MI355X build lives in uammd.h (C++14, no device code).
#pragma once
#include "../../uammd.h"
