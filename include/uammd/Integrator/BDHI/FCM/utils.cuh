// Forwarding header: same include path as the reference's src/Integrator/BDHI/FCM/utils.cuh (BDHI::cached_vector, :15; the Fourier-space
// helpers of that file are kernels of the library here: csrc/fcm.hip).
#pragma once
#include "../../../uammd.h"
