// Forwarding header: same include path as the reference's src/Integrator/Hydro/BDHI_quasi2D.cuh (BDHI::True2D, BDHI::Quasi2D; both precisions).
#pragma once
#include "../../uammd.h"
