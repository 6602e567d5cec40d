// Forwarding header: same include path as the reference's src/Integrator/BDHI/BDHI_Cholesky.cuh (both precisions).
#pragma once
#include "../../uammd.h"
