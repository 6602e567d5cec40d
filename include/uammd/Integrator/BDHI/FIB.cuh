// Forwarding header: same include path as the reference's src/Integrator/BDHI/FIB.cuh (BDHI::FIB).
#pragma once
#include "../../uammd.h"
