// Forwarding header: same include path as the reference's src/utils/ForceEnergyVirial.cuh (struct ForceEnergyVirial and its operators).
#pragma once
#include "../uammd.h"
#include "../device/ForceEnergyVirial.hpp"
