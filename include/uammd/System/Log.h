// Forwarding header: same include path as the reference's src/System/Log.h (the log levels and System::log<level> live in uammd.h).
#pragma once
#include "../uammd.h"
