// Forwarding header: same include path as the reference's src/misc/LanczosAlgorithm/MatrixDot.h (lanczos::MatrixDot, :7-25, lives in uammd.h).
#pragma once
#include "../../uammd.h"
