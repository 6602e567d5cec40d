// Forwarding header: same include path as the reference's src/utils/Grid.cuh (struct Grid, nextFFTWiseSize3D live in uammd.h).
#pragma once
#include "../uammd.h"
