// Forwarding header: same include path as the reference's src/misc/IBM.cuh (IBM<Kernel, Grid, Index3D>, IBM_ns::LinearIndex3D live in uammd.h).
#pragma once
#include "../uammd.h"
