// Forwarding header: same include path as the reference's src/Integrator/BDHI/BDHI.cuh (BDHI::Parameters, :13-24, lives in uammd.h).
#pragma once
#include "../../uammd.h"
