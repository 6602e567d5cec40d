// Forwarding header: same include path as the reference's src/ParticleData/Property.cuh.
// The whole host interface of the MI355X build lives in uammd.h (Property, property_ptr, access).
#pragma once
#include "../uammd.h"
