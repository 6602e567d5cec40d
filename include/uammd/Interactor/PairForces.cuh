// Forwarding header: same include path as the reference's src/Interactor/PairForces.cuh.
// uammd.h holds the host interface (C++14, no device code) with PairForces<Potential::LJ, NeighbourList> on the library's fused
// Lennard-Jones path.  A translation unit compiled by hipcc also gets the GENERIC PairForces<MyPotential, NeighbourList> and
// Potential::Radial<Functor> (device/PairForces.hip.hpp): a user's potential is a device functor, as in the reference.
#pragma once
#if defined(DOUBLE_PRECISION)
#error "PairForces.cuh: this module has a single-precision backend only on MI355X (uammd.h, PRECISION): build without -DDOUBLE_PRECISION"
#endif
#include "../uammd.h"
#if defined(__HIPCC__)
#include "../device/PairForces.hip.hpp"
#endif
