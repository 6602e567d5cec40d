// Forwarding header: same include path as the reference's src/Interactor/NeighbourList/BasicList/BasicListBase.cuh.
// The whole host interface of the MI355X build lives in uammd.h (C++14; under hipcc it also brings the device-side NeighbourContainer).
#pragma once
#include "../../../uammd.h"
