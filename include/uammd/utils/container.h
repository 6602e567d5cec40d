// Forwarding header: same include path as the reference's src/utils/container.h.
// The whole host interface of the MI355X build lives in uammd.h.
#pragma once
#include "../uammd.h"
