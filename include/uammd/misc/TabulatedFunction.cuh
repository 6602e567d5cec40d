// misc/TabulatedFunction.cuh (reference: src/misc/TabulatedFunction.cuh:28-162) — a function sampled on N points of [rmin, rmax) by the host
// and read back on the device with linear interpolation: what the PSE near field reads its RPY coefficients from (NearField.cuh:65-99; the
// library's own copy is csrc/pse.hip:table_get, the same arithmetic) offered to user code as the reference offers it.
//   TabulatedFunction<real2> table(d_table, N, rmin, rmax, [](double r) { return real2(...); });   // d_table: device memory, N elements
//   TabulatedFunction<real>  table(N, rmin, rmax, foo);                                             // the table allocates (and frees) its own
//   ... in a kernel:  table(r)  /  table.get(r)      (T() beyond rmax, table[0] at or below rmin)
// The interpolation is written with explicit fused multiply-adds (lerp as fma(t, v1, fma(-t, v0, v0)), :33-39): the same value whatever
// the compiler's contraction setting.
#ifndef UAMMD_MI355X_MISC_TABULATEDFUNCTION_CUH
#define UAMMD_MI355X_MISC_TABULATEDFUNCTION_CUH
#include "../uammd.h"
#include <iterator>
#include <vector>

namespace uammd {
template <typename T, typename T2> UAMMD_HD T lerp(T v0, T v1, T2 t) { return ::fmaf(t, v1, ::fmaf(-t, v0, v0)); }
template <typename T2> UAMMD_HD real2 lerp(real2 v0, real2 v1, T2 t) { return make_real2(lerp(v0.x, v1.x, t), lerp(v0.y, v1.y, t)); }
template <typename T2> UAMMD_HD real3 lerp(real3 v0, real3 v1, T2 t) { return make_real3(lerp(v0.x, v1.x, t), lerp(v0.y, v1.y, t), lerp(v0.z, v1.z, t)); }
template <typename T2> UAMMD_HD real4 lerp(real4 v0, real4 v1, T2 t) {
  return make_real4(lerp(v0.x, v1.x, t), lerp(v0.y, v1.y, t), lerp(v0.z, v1.z, t), lerp(v0.w, v1.w, t));
}

struct LinearInterpolation {
  template <class iterator, class T = typename std::iterator_traits<iterator>::value_type>
  UAMMD_HD T operator()(const iterator &table, int Ntable, real dr, real r) const {
    const int i = r * Ntable;
    const real r0 = i * dr;
    const T v0 = table[i], v1 = table[i + 1];
    const real t = (r - r0) * (real)Ntable;
    return lerp(v0, v1, t);
  }
};

template <class T, class Interpolation = LinearInterpolation> struct TabulatedFunction {
  int Ntable = 0;
  real rmin = 0, rmax = 0, interval = 0, dr = 0;
  T *table = nullptr;
  Interpolation interp;
  bool freeTable = false, isCopy = false;

  TabulatedFunction() {}
  // the table in memory of its own (released by the original, not by the copies a kernel launch makes)
  template <class Functor> TabulatedFunction(int N, real rmin, real rmax, Functor foo) : TabulatedFunction(allocate(N), N, rmin, rmax, foo) { freeTable = true; }
  // ... in the caller's device memory, N elements: sample i is foo(rmin + i (rmax - rmin) / (N - 1)), evaluated on the host in double
  template <class Functor>
  TabulatedFunction(T *table, int N, real rmin, real rmax, Functor foo)
      : Ntable(N - 1), rmin(rmin), rmax(rmax), interval(real(1.0 / (rmax - rmin))), dr(real(1.0) / real(N - 1)), table(table) {
    std::vector<T> tableCPU(Ntable + 1);
    for (int i = 0; i <= Ntable; i++) {
      const double x = (i / (double)(Ntable)) * (rmax - rmin) + rmin;
      tableCPU[i] = foo(x);
    }
    detail::hipCheck(hipMemcpy(table, tableCPU.data(), (Ntable + 1) * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy");
  }
  TabulatedFunction(const TabulatedFunction &o)
      : Ntable(o.Ntable), rmin(o.rmin), rmax(o.rmax), interval(o.interval), dr(o.dr), table(o.table), interp(o.interp), freeTable(false), isCopy(true) {}
  void operator=(TabulatedFunction &&o) {
    Ntable = o.Ntable; rmin = o.rmin; rmax = o.rmax; interval = o.interval; dr = o.dr; table = o.table; interp = o.interp;
    freeTable = o.freeTable; isCopy = o.isCopy;
    o.freeTable = false;
    o.isCopy = true;
  }
  ~TabulatedFunction() { if (freeTable && !isCopy) (void)hipFree(table); }

  UAMMD_HD T get(real rs) const { return (*this)(rs); }
  UAMMD_HD T operator()(real rs) const {   // (dereferences device memory: call it from device code)
    const real r = (rs - rmin) * interval;
    if (rs >= rmax) return T();
    if (r <= real(0.0)) return table[0];
    return interp(table, Ntable, dr, r);
  }
private:
  static T *allocate(int N) {
    T *p = nullptr;
    detail::hipCheck(hipMalloc((void **)&p, N * sizeof(T)), "hipMalloc");
    return p;
  }
};
}  // namespace uammd
#endif
