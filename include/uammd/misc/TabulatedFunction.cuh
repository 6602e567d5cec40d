// misc/TabulatedFunction.cuh — a function of one variable sampled by the host on an even mesh over [rmin, rmax] and evaluated in device code
// by interpolation between neighbouring samples (the user-facing counterpart of the reference's src/misc/TabulatedFunction.cuh; the
// library's own RPY table for the PSE near field is csrc/pse.hip:table_get and evaluates a sample pair with the same arithmetic).
//
//   TabulatedFunction<real2> f(d_samples, N, rmin, rmax, [](double r) { return real2(...); });   // N samples into the CALLER's device memory
//   TabulatedFunction<real>  g(N, rmin, rmax, foo);                                               // samples in pooled memory of the table's own
//   kernel<<<...>>>(g, ...);   ...   g(r) or g.get(r) inside the kernel
// Values: T() at and beyond rmax, the first sample at and below rmin, otherwise the chord between the two samples that bracket r.
//
// Design: the object IS the device-side view (samples pointer + the mesh), so it travels into a kernel by value; a table that allocated its
// samples shares them between all of its host copies through a count of owners (the last owner returns the block to the pool of
// temporary device memory), so copies — including the one a kernel launch makes — are ordinary values and nothing is released twice.
#ifndef UAMMD_MI355X_MISC_TABULATEDFUNCTION_CUH
#define UAMMD_MI355X_MISC_TABULATEDFUNCTION_CUH
#include "../uammd.h"
#include <vector>

namespace uammd {

// v0 + t (v1 - v0), component by component, as two fused multiply-adds (v0 - t v0, then + t v1): exact at t = 0 and independent of the
// compiler's contraction setting
namespace tabulated_ns {
UAMMD_HD float chord(float a, float b, float t) { return ::fmaf(t, b, ::fmaf(-t, a, a)); }
UAMMD_HD double chord(double a, double b, double t) { return ::fma(t, b, ::fma(-t, a, a)); }
}  // namespace tabulated_ns
template <class S> UAMMD_HD float lerp(float a, float b, S t) { return tabulated_ns::chord(a, b, float(t)); }
template <class S> UAMMD_HD double lerp(double a, double b, S t) { return tabulated_ns::chord(a, b, double(t)); }
template <class S> UAMMD_HD real2 lerp(const real2 &a, const real2 &b, S t) { return real2(lerp(a.x, b.x, t), lerp(a.y, b.y, t)); }
template <class S> UAMMD_HD real3 lerp(const real3 &a, const real3 &b, S t) { return real3(lerp(a.x, b.x, t), lerp(a.y, b.y, t), lerp(a.z, b.z, t)); }
template <class S> UAMMD_HD real4 lerp(const real4 &a, const real4 &b, S t) {
  return real4(lerp(a.x, b.x, t), lerp(a.y, b.y, t), lerp(a.z, b.z, t), lerp(a.w, b.w, t));
}

// The interpolation rule: samples[0 .. cells] cover the unit interval in `cells` steps of `step`; u in (0, 1) is the argument mapped onto it
struct LinearInterpolation {
  template <class Samples> UAMMD_HD auto operator()(const Samples &samples, int cells, real step, real u) const -> decltype(samples[0] + samples[0]) {
    const int k = int(u * real(cells));
    return lerp(samples[k], samples[k + 1], (u - real(k) * step) * real(cells));
  }
};

template <class T, class Interpolation = LinearInterpolation> class TabulatedFunction {
  const T *samples = nullptr;
  int cells = 0;                 // number of intervals: samples[0 .. cells]
  real lower = 0, upper = 0;     // the sampled range
  real toUnit = 0, step = 0;     // 1 / (upper - lower), 1 / cells
  Interpolation rule;
  int *owners = nullptr;         // host counter shared by the copies of a table that allocated its samples; null: the samples are the caller's

  UAMMD_HOSTDEV void share(const TabulatedFunction &o) {
    samples = o.samples; cells = o.cells; lower = o.lower; upper = o.upper; toUnit = o.toUnit; step = o.step; rule = o.rule; owners = o.owners;
#if !defined(__HIP_DEVICE_COMPILE__)
    if (owners) ++*owners;
#endif
  }
  UAMMD_HOSTDEV void drop() {
#if !defined(__HIP_DEVICE_COMPILE__)
    if (owners && --*owners == 0) {
      detail::DevicePool::instance().deallocate(const_cast<T *>(samples));
      delete owners;
    }
#endif
    owners = nullptr;
    samples = nullptr;
  }
  template <class Functor> void fill(T *d_samples, int N, Functor &&f) {
    if (N < 2 || !(upper > lower)) throw std::invalid_argument("TabulatedFunction: needs at least two samples on a range rmin < rmax");
    cells = N - 1;
    toUnit = real(1.0 / (double(upper) - double(lower)));
    step = real(1.0) / real(cells);
    std::vector<T> host((size_t)N);
    for (int k = 0; k < N; ++k) host[k] = f((double(k) / double(cells)) * (double(upper) - double(lower)) + double(lower));  // mesh point k, in double
    detail::hipCheck(hipMemcpy(d_samples, host.data(), sizeof(T) * (size_t)N, hipMemcpyHostToDevice), "hipMemcpy");
    samples = d_samples;
  }
public:
  UAMMD_HOSTDEV TabulatedFunction() {}
  // N samples of foo written to d_samples (device memory of the caller's, which outlives the table)
  template <class Functor> TabulatedFunction(T *d_samples, int N, real rmin, real rmax, Functor foo) : lower(rmin), upper(rmax) { fill(d_samples, N, foo); }
  // ... to memory the table takes from the pool and gives back with its last copy
  template <class Functor> TabulatedFunction(int N, real rmin, real rmax, Functor foo) : lower(rmin), upper(rmax) {
    T *block = static_cast<T *>(detail::DevicePool::instance().allocate(sizeof(T) * (size_t)std::max(N, 0)));
    try { fill(block, N, foo); }
    catch (...) { detail::DevicePool::instance().deallocate(block); throw; }
    owners = new int(1);
  }
  UAMMD_HOSTDEV TabulatedFunction(const TabulatedFunction &o) { share(o); }
  UAMMD_HOSTDEV TabulatedFunction &operator=(const TabulatedFunction &o) {
    if (this != &o) { drop(); share(o); }
    return *this;
  }
  UAMMD_HOSTDEV ~TabulatedFunction() { drop(); }

  UAMMD_HOSTDEV T operator()(real r) const {   // (reads device memory: for device code)
    if (r >= upper) return T();
    const real u = (r - lower) * toUnit;
    if (u <= real(0.0)) return samples[0];
    return rule(samples, cells, step, u);
  }
  UAMMD_HOSTDEV T get(real r) const { return (*this)(r); }
  UAMMD_HOSTDEV int size() const { return cells + 1; }
  UAMMD_HOSTDEV const T *data() const { return samples; }
};
}  // namespace uammd
#endif
