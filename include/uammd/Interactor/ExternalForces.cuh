// Same include path as the reference's src/Interactor/ExternalForces.cuh.  ExternalForces<Functor> takes a DEVICE functor: the template
// lives in device/ExternalForces.hip.hpp and needs hipcc (as it needs nvcc in the reference); a plain C++ translation unit gets the host
// interface alone.
#pragma once
#include "../uammd.h"
#if defined(__HIPCC__)
#include "../device/ExternalForces.hip.hpp"
#endif
