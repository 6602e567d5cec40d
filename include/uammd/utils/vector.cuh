// utils/vector.cuh (reference: src/utils/vector.cuh) — what UAMMD code uses on real2 / real3 / real4 / int2 / int3 beyond the arithmetic the
// runtime's vector types already carry (component-wise + - * / with vector and scalar operands, compound forms, unary minus, ==):
// the make_realN / make_intN conversions between the kinds, dot, cross, length, normalize, floorf, sqrt, abs
// (examples/basic_concepts/12-your-first-integrator.cu:135,161, 13-your-first-interactor.cu:121, utils/Grid.cuh:36-41 are written on them).
#ifndef UAMMD_MI355X_UTILS_VECTOR_CUH
#define UAMMD_MI355X_UTILS_VECTOR_CUH

#include "../global/defines.h"

#include <algorithm>
#include <cmath>
#include <istream>
#include <ostream>

// text form of the vector types: components separated by one blank (utils/printOverloads.h; `out << pos[i]` in
// examples/basic_concepts/8-interacting_particles.cu:75).  At global scope, where the types live.
inline std::ostream &operator<<(std::ostream &out, const uammd::real2 &f) { return out << f.x << " " << f.y; }
inline std::ostream &operator<<(std::ostream &out, const uammd::real3 &f) { return out << f.x << " " << f.y << " " << f.z; }
inline std::ostream &operator<<(std::ostream &out, const uammd::real4 &f) { return out << f.x << " " << f.y << " " << f.z << " " << f.w; }
inline std::ostream &operator<<(std::ostream &out, const uammd::int3 &f) { return out << f.x << " " << f.y << " " << f.z; }
inline std::istream &operator>>(std::istream &in, uammd::real2 &f) { return in >> f.x >> f.y; }
inline std::istream &operator>>(std::istream &in, uammd::real3 &f) { return in >> f.x >> f.y >> f.z; }
inline std::istream &operator>>(std::istream &in, uammd::real4 &f) { return in >> f.x >> f.y >> f.z >> f.w; }

namespace uammd {

// ---- construction / conversion ------------------------------------------------------------------------------------------------------
UAMMD_HD real2 make_real2(real x, real y) { return real2(x, y); }
UAMMD_HD real2 make_real2(real v) { return real2(v, v); }
UAMMD_HD real2 make_real2(real2 a) { return a; }
UAMMD_HD real2 make_real2(real3 a) { return real2(a.x, a.y); }
UAMMD_HD real2 make_real2(real4 a) { return real2(a.x, a.y); }
UAMMD_HD real2 make_real2(int2 a) { return real2(real(a.x), real(a.y)); }
UAMMD_HD real3 make_real3(real x, real y, real z) { return real3(x, y, z); }
UAMMD_HD real3 make_real3(real v) { return real3(v, v, v); }
UAMMD_HD real3 make_real3(real3 a) { return a; }
UAMMD_HD real3 make_real3(real4 a) { return real3(a.x, a.y, a.z); }
UAMMD_HD real3 make_real3(real2 a, real z) { return real3(a.x, a.y, z); }
UAMMD_HD real3 make_real3(real x, real2 yz) { return real3(x, yz.x, yz.y); }
UAMMD_HD real3 make_real3(int3 a) { return real3(real(a.x), real(a.y), real(a.z)); }
UAMMD_HD real4 make_real4(real x, real y, real z, real w) { return real4(x, y, z, w); }
UAMMD_HD real4 make_real4(real v) { return real4(v, v, v, v); }
UAMMD_HD real4 make_real4(real4 a) { return a; }
UAMMD_HD real4 make_real4(real3 a) { return real4(a.x, a.y, a.z, real(0)); }
UAMMD_HD real4 make_real4(real3 a, real w) { return real4(a.x, a.y, a.z, w); }
UAMMD_HD real4 make_real4(real2 a, real2 b) { return real4(a.x, a.y, b.x, b.y); }
// from the double-precision vectors (System::rng().uniform3 / gaussian3 hand out double3, utils/utils.h:70,97; the reference's own test
// programs accumulate in double3: test/BDHI/FCM/FCM.cu:88-101) — utils/vector.cuh:341,371
UAMMD_HD real3 make_real3(::double3 a) { return real3(real(a.x), real(a.y), real(a.z)); }
UAMMD_HD real4 make_real4(::double3 a, real w) { return real4(real(a.x), real(a.y), real(a.z), w); }
UAMMD_HD real4 make_real4(::double4 a) { return real4(real(a.x), real(a.y), real(a.z), real(a.w)); }
UAMMD_HD real2 make_real2(::double2 a) { return real2(real(a.x), real(a.y)); }
// (make_int2 / make_int3 of three scalars are the runtime's; these are the conversions it does not have)
UAMMD_HD int2 make_int2(int3 a) { return int2(a.x, a.y); }
UAMMD_HD int2 make_int2(real2 a) { return int2(int(a.x), int(a.y)); }
UAMMD_HD int3 make_int3(int v) { return int3(v, v, v); }
UAMMD_HD int3 make_int3(int3 a) { return a; }
UAMMD_HD int3 make_int3(real3 a) { return int3(int(a.x), int(a.y), int(a.z)); }   // (truncation, as a C cast does: Grid::getCell)
UAMMD_HD int3 make_int3(real4 a) { return int3(int(a.x), int(a.y), int(a.z)); }
UAMMD_HD int3 make_int3(int2 a, int z) { return int3(a.x, a.y, z); }
using ::make_int2;
using ::make_int3;

// ---- products, norms, element-wise functions -------------------------------------------------------------------------------------------
// (dot products as one chain of fused multiply-adds, last component outermost: what nvcc's default contraction makes of the reference's
// a.x * b.x + a.y * b.y + a.z * b.z and what the library's kernels do explicitly (floating-point contract) — a user functor
// that tests dot(r12, r12) against a cut-off then takes the same decision as the library for a pair an ulp from it)
UAMMD_HD real dot(const real2 &a, const real2 &b) { return fmaf(a.y, b.y, a.x * b.x); }
UAMMD_HD real dot(const real3 &a, const real3 &b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
UAMMD_HD real dot(const real4 &a, const real4 &b) { return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x))); }
UAMMD_HD int dot(const int2 &a, const int2 &b) { return a.x * b.x + a.y * b.y; }
UAMMD_HD int dot(const int3 &a, const int3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
UAMMD_HD real3 cross(const real3 &a, const real3 &b) { return real3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
UAMMD_HD real length(const real2 &v) { return std::sqrt(dot(v, v)); }
UAMMD_HD real length(const real3 &v) { return std::sqrt(dot(v, v)); }
UAMMD_HD real length(const real4 &v) { return std::sqrt(dot(v, v)); }
UAMMD_HD real3 normalize(const real3 &v) { return v * (real(1) / length(v)); }
UAMMD_HD real4 normalize(const real4 &v) { return v * (real(1) / length(v)); }
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
UAMMD_HD real2 floorf(const real2 &a) { return real2(std::floor(a.x), std::floor(a.y)); }
UAMMD_HD real3 floorf(const real3 &a) { return real3(std::floor(a.x), std::floor(a.y), std::floor(a.z)); }
UAMMD_HD real4 floorf(const real4 &a) { return real4(std::floor(a.x), std::floor(a.y), std::floor(a.z), std::floor(a.w)); }
UAMMD_HD real3 sqrt(const real3 &a) { return real3(std::sqrt(a.x), std::sqrt(a.y), std::sqrt(a.z)); }
UAMMD_HD real3 abs(const real3 &a) { return real3(std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)); }

}  // namespace uammd

// ---- double3 / double4 beside the runtime's own arithmetic on them (utils/vector.cuh:410-777, at global scope as there) -------------------
UAMMD_HD ::double3 make_double3(uammd::real3 a) { return ::double3(a.x, a.y, a.z); }
UAMMD_HD ::double3 make_double3(uammd::real4 a) { return ::double3(a.x, a.y, a.z); }
UAMMD_HD ::double3 make_double3(double a) { return ::double3(a, a, a); }
UAMMD_HD ::double3 make_double3(::double3 a) { return a; }
UAMMD_HD ::double3 make_double3(::double2 xy, double z) { return ::double3(xy.x, xy.y, z); }
UAMMD_HD ::double3 make_double3(double x, ::double2 yz) { return ::double3(x, yz.x, yz.y); }
UAMMD_HD ::double3 make_double3(uammd::int3 a) { return ::double3(a.x, a.y, a.z); }
UAMMD_HD ::double3 make_double3(::double4 a) { return ::double3(a.x, a.y, a.z); }
UAMMD_HD ::double4 make_double4(::double3 a) { return ::double4(a.x, a.y, a.z, 0.0); }
UAMMD_HD ::double4 make_double4(::double3 a, double w) { return ::double4(a.x, a.y, a.z, w); }
UAMMD_HD ::double4 make_double4(uammd::real4 a) { return ::double4(a.x, a.y, a.z, a.w); }
UAMMD_HD uammd::int3 make_int3(::double3 a) { return uammd::int3(int(a.x), int(a.y), int(a.z)); }
UAMMD_HD double dot(const ::double3 &a, const ::double3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
UAMMD_HD double dot(const ::double4 &a, const ::double4 &b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
UAMMD_HD double length(const ::double3 &v) { return std::sqrt(dot(v, v)); }
UAMMD_HD ::double3 normalize(const ::double3 &v) { return v * (1.0 / length(v)); }
UAMMD_HD ::double3 cross(const ::double3 &a, const ::double3 &b) { return ::double3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
UAMMD_HD ::double3 floorf(const ::double3 &v) { return ::double3(std::floor(v.x), std::floor(v.y), std::floor(v.z)); }
UAMMD_HD ::double3 abs(const ::double3 &a) { return ::double3(std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)); }
inline std::ostream &operator<<(std::ostream &out, const ::double3 &f) { return out << f.x << " " << f.y << " " << f.z; }
#endif
