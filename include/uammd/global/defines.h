// global/defines.h (reference: src/global/defines.h:33-44) — the scalar and small-vector types of the host interface.
// As in the reference, real2 / real3 / real4 and int2 / int3 ARE the runtime's vector types (float2 ... int3 of <hip/hip_vector_types.h>, a
// header any C++14 host compiler takes): user code that mixes `float3` and `real3`, or says `int3` after `using namespace uammd`, means
// one type.  `real` is float (the reference's default build; the DOUBLE_PRECISION entry points of the library are the _f64 C ABI).
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
#define SINGLE_PRECISION
// the loop shorthands UAMMD programs use (global/defines.h:13-14; examples/misc/benchmark.cu:129)
#define fori(x, y) for (int i = x; i < int(y); i++)
#define forj(x, y) for (int j = x; j < int(y); j++)

namespace uammd {

using real = float;
using real2 = ::float2;
using real3 = ::float3;
using real4 = ::float4;
using int2 = ::int2;
using int3 = ::int3;
using uint = unsigned int;
using ullint = unsigned long long;

}  // namespace uammd
#endif
