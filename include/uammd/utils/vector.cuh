// utils/vector.cuh (reference: src/utils/vector.cuh) — what UAMMD code uses on the small vector types beyond the arithmetic the runtime's
// float2 / float3 / float4 / double2 / double3 / double4 / int2 / int3 already carry (component-wise + - * / with vector and scalar operands,
// compound forms, unary minus, ==): the make_realN / make_doubleN / make_intN conversions between the kinds, dot, cross, length, normalize,
// floorf, sqrt, abs (examples/basic_concepts/12-your-first-integrator.cu:135,161, 13-your-first-interactor.cu:121, utils/Grid.cuh:36-41 are
// written on them).  Both precisions are always there — `real` picks one (global/defines.h) — in namespace uammd, and visible at global
// scope too (where the reference defines them) through using-declarations, so one function answers a call from either scope.
#ifndef UAMMD_MI355X_UTILS_VECTOR_CUH
#define UAMMD_MI355X_UTILS_VECTOR_CUH

#include "../global/defines.h"

#include <algorithm>
#include <cmath>
#include <istream>
#include <ostream>

// text form of the vector types: components separated by one blank (utils/printOverloads.h; `out << pos[i]` in
// examples/basic_concepts/8-interacting_particles.cu:75).  At global scope, where the types live.
inline std::ostream &operator<<(std::ostream &out, const ::float2 &f) { return out << f.x << " " << f.y; }
inline std::ostream &operator<<(std::ostream &out, const ::float3 &f) { return out << f.x << " " << f.y << " " << f.z; }
inline std::ostream &operator<<(std::ostream &out, const ::float4 &f) { return out << f.x << " " << f.y << " " << f.z << " " << f.w; }
inline std::ostream &operator<<(std::ostream &out, const ::double2 &f) { return out << f.x << " " << f.y; }
inline std::ostream &operator<<(std::ostream &out, const ::double3 &f) { return out << f.x << " " << f.y << " " << f.z; }
inline std::ostream &operator<<(std::ostream &out, const ::double4 &f) { return out << f.x << " " << f.y << " " << f.z << " " << f.w; }
inline std::ostream &operator<<(std::ostream &out, const ::int3 &f) { return out << f.x << " " << f.y << " " << f.z; }
inline std::istream &operator>>(std::istream &in, uammd::real2 &f) { return in >> f.x >> f.y; }
inline std::istream &operator>>(std::istream &in, uammd::real3 &f) { return in >> f.x >> f.y >> f.z; }
inline std::istream &operator>>(std::istream &in, uammd::real4 &f) { return in >> f.x >> f.y >> f.z >> f.w; }

namespace uammd {

// ---- construction / conversion: to the working precision from scalars and from vectors of either precision -------------------------------
UAMMD_HD real2 make_real2(real x, real y) { return real2(x, y); }
UAMMD_HD real2 make_real2(real v) { return real2(v, v); }
UAMMD_HD real2 make_real2(::float2 a) { return real2(real(a.x), real(a.y)); }
UAMMD_HD real2 make_real2(::double2 a) { return real2(real(a.x), real(a.y)); }
UAMMD_HD real2 make_real2(::float3 a) { return real2(real(a.x), real(a.y)); }
UAMMD_HD real2 make_real2(::double3 a) { return real2(real(a.x), real(a.y)); }
UAMMD_HD real2 make_real2(::float4 a) { return real2(real(a.x), real(a.y)); }
UAMMD_HD real2 make_real2(::double4 a) { return real2(real(a.x), real(a.y)); }
UAMMD_HD real2 make_real2(int2 a) { return real2(real(a.x), real(a.y)); }
UAMMD_HD real3 make_real3(real x, real y, real z) { return real3(x, y, z); }
UAMMD_HD real3 make_real3(real v) { return real3(v, v, v); }
UAMMD_HD real3 make_real3(::float3 a) { return real3(real(a.x), real(a.y), real(a.z)); }
UAMMD_HD real3 make_real3(::double3 a) { return real3(real(a.x), real(a.y), real(a.z)); }   // (System::rng().uniform3 hands out double3, utils/utils.h:70)
UAMMD_HD real3 make_real3(::float4 a) { return real3(real(a.x), real(a.y), real(a.z)); }
UAMMD_HD real3 make_real3(::double4 a) { return real3(real(a.x), real(a.y), real(a.z)); }
UAMMD_HD real3 make_real3(::float2 a, real z) { return real3(real(a.x), real(a.y), z); }     // (make_real3(rng.gf(0, 1), rng.gf(0, 1).x): Saru::gf is float2)
UAMMD_HD real3 make_real3(::double2 a, real z) { return real3(real(a.x), real(a.y), z); }
UAMMD_HD real3 make_real3(real x, ::float2 yz) { return real3(x, real(yz.x), real(yz.y)); }
UAMMD_HD real3 make_real3(real x, ::double2 yz) { return real3(x, real(yz.x), real(yz.y)); }
UAMMD_HD real3 make_real3(int3 a) { return real3(real(a.x), real(a.y), real(a.z)); }
UAMMD_HD real4 make_real4(real x, real y, real z, real w) { return real4(x, y, z, w); }
UAMMD_HD real4 make_real4(real v) { return real4(v, v, v, v); }
UAMMD_HD real4 make_real4(::float4 a) { return real4(real(a.x), real(a.y), real(a.z), real(a.w)); }
UAMMD_HD real4 make_real4(::double4 a) { return real4(real(a.x), real(a.y), real(a.z), real(a.w)); }
UAMMD_HD real4 make_real4(::float3 a) { return real4(real(a.x), real(a.y), real(a.z), real(0)); }
UAMMD_HD real4 make_real4(::double3 a) { return real4(real(a.x), real(a.y), real(a.z), real(0)); }
UAMMD_HD real4 make_real4(::float3 a, real w) { return real4(real(a.x), real(a.y), real(a.z), w); }
UAMMD_HD real4 make_real4(::double3 a, real w) { return real4(real(a.x), real(a.y), real(a.z), w); }
UAMMD_HD real4 make_real4(::float2 a, ::float2 b) { return real4(real(a.x), real(a.y), real(b.x), real(b.y)); }
UAMMD_HD real4 make_real4(::double2 a, ::double2 b) { return real4(real(a.x), real(a.y), real(b.x), real(b.y)); }
// to double whatever the working precision (the reference's own test programs accumulate in double3: test/BDHI/FCM/FCM.cu:88-101;
// utils/vector.cuh:410-777)
using ::make_double3;   // (of three scalars: the runtime's)
using ::make_double4;
UAMMD_HD ::double3 make_double3(double a) { return ::double3(a, a, a); }
UAMMD_HD ::double3 make_double3(::float3 a) { return ::double3(a.x, a.y, a.z); }
UAMMD_HD ::double3 make_double3(::float4 a) { return ::double3(a.x, a.y, a.z); }
UAMMD_HD ::double3 make_double3(::double3 a) { return a; }
UAMMD_HD ::double3 make_double3(::double4 a) { return ::double3(a.x, a.y, a.z); }
UAMMD_HD ::double3 make_double3(::double2 xy, double z) { return ::double3(xy.x, xy.y, z); }
UAMMD_HD ::double3 make_double3(double x, ::double2 yz) { return ::double3(x, yz.x, yz.y); }
UAMMD_HD ::double3 make_double3(int3 a) { return ::double3(a.x, a.y, a.z); }
UAMMD_HD ::double4 make_double4(::double3 a) { return ::double4(a.x, a.y, a.z, 0.0); }
UAMMD_HD ::double4 make_double4(::double3 a, double w) { return ::double4(a.x, a.y, a.z, w); }
UAMMD_HD ::double4 make_double4(::float4 a) { return ::double4(a.x, a.y, a.z, a.w); }
UAMMD_HD ::double4 make_double4(::double4 a) { return a; }
// (make_int2 / make_int3 of three scalars are the runtime's; these are the conversions it does not have)
using ::make_int2;
using ::make_int3;
UAMMD_HD int2 make_int2(int3 a) { return int2(a.x, a.y); }
UAMMD_HD int2 make_int2(::float2 a) { return int2(int(a.x), int(a.y)); }
UAMMD_HD int2 make_int2(::double2 a) { return int2(int(a.x), int(a.y)); }
UAMMD_HD int3 make_int3(int v) { return int3(v, v, v); }
UAMMD_HD int3 make_int3(int3 a) { return a; }
UAMMD_HD int3 make_int3(::float3 a) { return int3(int(a.x), int(a.y), int(a.z)); }   // (truncation, as a C cast does: Grid::getCell)
UAMMD_HD int3 make_int3(::double3 a) { return int3(int(a.x), int(a.y), int(a.z)); }
UAMMD_HD int3 make_int3(::float4 a) { return int3(int(a.x), int(a.y), int(a.z)); }
UAMMD_HD int3 make_int3(::double4 a) { return int3(int(a.x), int(a.y), int(a.z)); }
UAMMD_HD int3 make_int3(int2 a, int z) { return int3(a.x, a.y, z); }

// ---- products, norms, element-wise functions -------------------------------------------------------------------------------------------
// (dot products as one chain of fused multiply-adds, last component outermost: what nvcc's default contraction makes of the reference's
// a.x * b.x + a.y * b.y + a.z * b.z and what the library's kernels do explicitly (floating-point contract) — a user functor
// that tests dot(r12, r12) against a cut-off then takes the same decision as the library for a pair an ulp from it)
UAMMD_HD float dot(const ::float2 &a, const ::float2 &b) { return ::fmaf(a.y, b.y, a.x * b.x); }
UAMMD_HD float dot(const ::float3 &a, const ::float3 &b) { return ::fmaf(a.z, b.z, ::fmaf(a.y, b.y, a.x * b.x)); }
UAMMD_HD float dot(const ::float4 &a, const ::float4 &b) { return ::fmaf(a.w, b.w, ::fmaf(a.z, b.z, ::fmaf(a.y, b.y, a.x * b.x))); }
UAMMD_HD double dot(const ::double2 &a, const ::double2 &b) { return ::fma(a.y, b.y, a.x * b.x); }
UAMMD_HD double dot(const ::double3 &a, const ::double3 &b) { return ::fma(a.z, b.z, ::fma(a.y, b.y, a.x * b.x)); }
UAMMD_HD double dot(const ::double4 &a, const ::double4 &b) { return ::fma(a.w, b.w, ::fma(a.z, b.z, ::fma(a.y, b.y, a.x * b.x))); }
UAMMD_HD int dot(const int2 &a, const int2 &b) { return a.x * b.x + a.y * b.y; }
UAMMD_HD int dot(const int3 &a, const int3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
UAMMD_HD ::float3 cross(const ::float3 &a, const ::float3 &b) { return ::float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
UAMMD_HD ::double3 cross(const ::double3 &a, const ::double3 &b) { return ::double3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
UAMMD_HD float length(const ::float2 &v) { return std::sqrt(dot(v, v)); }
UAMMD_HD float length(const ::float3 &v) { return std::sqrt(dot(v, v)); }
UAMMD_HD float length(const ::float4 &v) { return std::sqrt(dot(v, v)); }
UAMMD_HD double length(const ::double2 &v) { return std::sqrt(dot(v, v)); }
UAMMD_HD double length(const ::double3 &v) { return std::sqrt(dot(v, v)); }
UAMMD_HD double length(const ::double4 &v) { return std::sqrt(dot(v, v)); }
UAMMD_HD ::float3 normalize(const ::float3 &v) { return v * (1.0f / length(v)); }
UAMMD_HD ::float4 normalize(const ::float4 &v) { return v * (1.0f / length(v)); }
UAMMD_HD ::double3 normalize(const ::double3 &v) { return v * (1.0 / length(v)); }
UAMMD_HD ::double4 normalize(const ::double4 &v) { return v * (1.0 / length(v)); }
// (the scalar functions of the same names stay visible inside the namespace)
using ::floorf;
using ::sqrt;
using ::abs;
// uammd::max / uammd::min of two scalars (examples/advanced/customPotentials.cu:133): the runtime's in device code, the standard
// library's on the host
#if defined(__HIPCC__)
using ::max;
using ::min;
#else
using std::max;
using std::min;
#endif
UAMMD_HD ::float2 floorf(const ::float2 &a) { return ::float2(std::floor(a.x), std::floor(a.y)); }
UAMMD_HD ::float3 floorf(const ::float3 &a) { return ::float3(std::floor(a.x), std::floor(a.y), std::floor(a.z)); }
UAMMD_HD ::float4 floorf(const ::float4 &a) { return ::float4(std::floor(a.x), std::floor(a.y), std::floor(a.z), std::floor(a.w)); }
UAMMD_HD ::double2 floorf(const ::double2 &a) { return ::double2(std::floor(a.x), std::floor(a.y)); }
UAMMD_HD ::double3 floorf(const ::double3 &a) { return ::double3(std::floor(a.x), std::floor(a.y), std::floor(a.z)); }
UAMMD_HD ::double4 floorf(const ::double4 &a) { return ::double4(std::floor(a.x), std::floor(a.y), std::floor(a.z), std::floor(a.w)); }
UAMMD_HD ::float3 sqrt(const ::float3 &a) { return ::float3(std::sqrt(a.x), std::sqrt(a.y), std::sqrt(a.z)); }
UAMMD_HD ::double3 sqrt(const ::double3 &a) { return ::double3(std::sqrt(a.x), std::sqrt(a.y), std::sqrt(a.z)); }
UAMMD_HD ::float3 abs(const ::float3 &a) { return ::float3(std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)); }
UAMMD_HD ::double3 abs(const ::double3 &a) { return ::double3(std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)); }

}  // namespace uammd

// ---- the same functions at global scope, where the reference defines them (utils/vector.cuh): `make_real3(...)`, `dot(a, b)` in a program
// that never says `using namespace uammd` ------------------------------------------------------------------------------------------------
using uammd::make_real2;
using uammd::make_real3;
using uammd::make_real4;
using uammd::make_double3;
using uammd::make_double4;
using uammd::dot;
using uammd::cross;
using uammd::length;
using uammd::normalize;
using uammd::make_int2;
using uammd::make_int3;
using uammd::floorf;
using uammd::abs;
using uammd::sqrt;
#endif
