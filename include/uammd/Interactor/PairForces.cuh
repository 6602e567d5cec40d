// Forwarding header: same include path as the reference's src/Interactor/PairForces.cuh.
// uammd.h holds the host interface (C++14, no device code) with PairForces<Potential::LJ, NeighbourList> on the library's fused
// Lennard-Jones path.  A translation unit compiled by hipcc also gets the GENERIC PairForces<MyPotential, NeighbourList> and
// Potential::Radial<Functor> (device/PairForces.hip.hpp): a user's potential is a device functor, as in the reference.
#pragma once
#include "../uammd.h"
#if defined(__HIPCC__)
#include "../device/PairForces.hip.hpp"
#endif
