// Forwarding header: same include path as the reference's src/misc/ParameterUpdatable.h (class ParameterUpdatable lives in uammd.h).
#pragma once
#include "../uammd.h"
