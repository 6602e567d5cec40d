// Forwarding header: same include path as the reference's src/ParticleData/ParticleGroup.cuh.
// The whole host interface of the MI355X build lives in uammd.h (ParticleGroup, its selectors and iterators).
#pragma once
#include "../uammd.h"
