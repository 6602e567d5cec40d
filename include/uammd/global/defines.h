// global/defines.h (reference: src/global/defines.h:33-44) — the scalar and small-vector types of the host interface.
// As in the reference, real2 / real3 / real4 and int2 / int3 ARE the runtime's vector types (float2 ... int3 of <hip/hip_vector_types.h>, a
// header any C++14 host compiler takes): user code that mixes `float3` and `real3`, or says `int3` after `using namespace uammd`, means
// one type.  `real` is float unless the program is compiled with -DDOUBLE_PRECISION (global/defines.h:9-11,33-44 — the reference's own
// unit tests are: test/CMakeLists.txt:9), which makes real / real2 / real3 / real4 the double types and routes the host classes that have a
// double-precision build in the library (IBM, FCM_impl, BDHI::FCM, BDHI::PSE, lanczos::Solver, ParticleData, ParticleSorter; the list and the
// reason are at the top of uammd.h) to the `_f64` entry points of the C ABI.  The tuned hot path is single precision: UAMMD's default.
// The runtime's types bring + - * / (vector and scalar operands), the compound forms, unary minus and ==; utils/vector.cuh adds the
// rest of what UAMMD code uses (make_realN conversions, dot, cross, length, ...).
#ifndef UAMMD_MI355X_GLOBAL_DEFINES_H
#define UAMMD_MI355X_GLOBAL_DEFINES_H

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define UAMMD_HD inline __host__ __device__
#define UAMMD_HOSTDEV __host__ __device__
#else
#include <hip/hip_vector_types.h>
#define UAMMD_HD inline
#define UAMMD_HOSTDEV
#endif

// cospi / sinpi on the HOST: CUDA's math headers have them for host code (the reference's tests call cospi from host functions,
// test/misc/ibm/test_ibm.cu:33); the HIP headers define the device functions only and glibc gains them with C23.  Reduced to [0, 1/2]
// before the multiplication by pi, so that the half-integers give exact zeros.
#include <cmath>
#if defined(__HIPCC__)
#define UAMMD_HOST_ONLY inline __host__
#else
#define UAMMD_HOST_ONLY inline
#endif
#if defined(__GLIBC_PREREQ)
#if __GLIBC_PREREQ(2, 41)
#define UAMMD_LIBM_HAS_COSPI   // (glibc 2.41 declares the C23 functions itself)
#endif
#endif
#if !defined(UAMMD_LIBM_HAS_COSPI)
UAMMD_HOST_ONLY double cospi(double x) {
  x = std::fmod(std::fabs(x), 2.0);
  if (x > 1.0) x = 2.0 - x;
  if (x == 0.5) return 0.0;
  const double pi = 3.14159265358979323846;
  return x < 0.25 ? std::cos(pi * x) : (x > 0.75 ? -std::cos(pi * (1.0 - x)) : std::sin(pi * (0.5 - x)));
}
UAMMD_HOST_ONLY double sinpi(double x) { return cospi(x - 0.5); }
UAMMD_HOST_ONLY float cospif(float x) { return (float)cospi((double)x); }
UAMMD_HOST_ONLY float sinpif(float x) { return (float)sinpi((double)x); }
#endif

#define UAMMD_VERSION "3.0.0"
#if !defined(DOUBLE_PRECISION) && !defined(SINGLE_PRECISION)
#define SINGLE_PRECISION
#endif
// the loop shorthands UAMMD programs use (global/defines.h:13-14; examples/misc/benchmark.cu:129)
#define fori(x, y) for (int i = x; i < int(y); i++)
#define forj(x, y) for (int j = x; j < int(y); j++)

namespace uammd {

#if defined(DOUBLE_PRECISION)
using real = double;
using real2 = ::double2;
using real3 = ::double3;
using real4 = ::double4;
#else
using real = float;
using real2 = ::float2;
using real3 = ::float3;
using real4 = ::float4;
#endif
using int2 = ::int2;
using int3 = ::int3;
using uint = unsigned int;
using ullint = unsigned long long;

}  // namespace uammd
#endif
