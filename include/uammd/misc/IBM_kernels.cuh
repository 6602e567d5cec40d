// Forwarding header: same include path as the reference's src/misc/IBM_kernels.cuh (namespace IBM_kernels lives in uammd.h).
#pragma once
#include "../uammd.h"
