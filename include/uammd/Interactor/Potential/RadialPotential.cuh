// Forwarding header: same include path as the reference's src/Interactor/Potential/RadialPotential.cuh.
// The whole host interface of the MI355X build lives in uammd.h; Potential::Radial<Functor> is device code of the user's translation unit (device/PairForces.hip.hpp).
#pragma once
#include "../../uammd.h"
#if defined(__HIPCC__)
#include "../../device/PairForces.hip.hpp"
#endif
