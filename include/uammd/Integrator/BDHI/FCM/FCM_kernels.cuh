// Forwarding header: same include path as the reference's src/Integrator/BDHI/FCM/FCM_kernels.cuh (BDHI::FCM_ns::Kernels::Gaussian,
// BarnettMagland, Peskin::threePoint / fourPoint, GaussianTorque: classes with a __host__ __device__ phi in uammd.h).
#pragma once
#include "../../../uammd.h"
