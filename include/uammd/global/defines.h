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
