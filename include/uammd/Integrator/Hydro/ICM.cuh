// Forwarding header: same include path as the reference's src/Integrator/Hydro/ICM.cuh (Hydro::ICM).
#pragma once
#include "../../uammd.h"
