// Forwarding header: same include path as the reference's src/utils/checkpoint.h (saveParticleData / restoreParticleData).
#pragma once
#include "../uammd.h"
