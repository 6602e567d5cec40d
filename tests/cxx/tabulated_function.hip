// misc/TabulatedFunction.cuh in a user kernel: samples, interpolation between them (the lerp written on the host from the same numbers),
// the two boundaries, the self-allocating form, real2 values, copies into a kernel launch.
#include "misc/TabulatedFunction.cuh"
#include <cassert>
#include <cmath>
#include <cstdio>
#include <vector>
using namespace uammd;

template <class Table, class T> __global__ void k_eval(Table t, const real *r, int n, T *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = t(r[i]);
}
int main() {
  const int N = 4096;
  const real rmin = 0.25, rmax = 3.0;
  auto foo = [](double r) { return (real)(std::sin(3.0 * r) / (1.0 + r)); };
  real *d_table;
  (void)hipMalloc(&d_table, N * sizeof(real));
  TabulatedFunction<real> table(d_table, N, rmin, rmax, foo);
  std::vector<real> hr = {rmin, real(0.1), real(-5), rmax, real(rmax + 1), real(1.2345), real(2.9999), real(0.2500001)};
  for (int i = 0; i < 1000; ++i) hr.push_back(rmin + (rmax - rmin) * (i + 0.37f) / 1000.0f);
  const int n = (int)hr.size();
  real *d_r, *d_o;
  (void)hipMalloc(&d_r, n * sizeof(real));
  (void)hipMalloc(&d_o, n * sizeof(real));
  (void)hipMemcpy(d_r, hr.data(), n * sizeof(real), hipMemcpyHostToDevice);
  hipLaunchKernelGGL((k_eval<TabulatedFunction<real>, real>), dim3((n + 127) / 128), dim3(128), 0, 0, table, d_r, n, d_o);
  std::vector<real> ho(n), ht(N);
  (void)hipMemcpy(ho.data(), d_o, n * sizeof(real), hipMemcpyDeviceToHost);
  (void)hipMemcpy(ht.data(), d_table, N * sizeof(real), hipMemcpyDeviceToHost);
  for (int i = 0; i < N; ++i) assert(ht[i] == foo((i / (double)(N - 1)) * (rmax - rmin) + rmin));
  assert(ho[0] == ht[0] && ho[1] == ht[0] && ho[2] == ht[0]);       // at and below rmin: the first sample
  assert(ho[3] == real(0) && ho[4] == real(0));                      // at and beyond rmax: T()
  double worst = 0;
  for (int k = 5; k < n; ++k) {
    const real r = (hr[k] - rmin) * real(1.0 / (rmax - rmin));
    const int i = r * (N - 1);
    const real r0 = i * (real(1.0) / real(N - 1));
    const real t = (r - r0) * (real)(N - 1);
    const real expect = ::fmaf(t, ht[i + 1], ::fmaf(-t, ht[i], ht[i]));
    assert(ho[k] == expect);                                         // the same arithmetic, bit for bit
    worst = std::max(worst, std::fabs((double)ho[k] - std::sin(3.0 * hr[k]) / (1.0 + hr[k])));
  }
  assert(worst < 2e-6);                                              // 4096 points: interpolation error ~ h^2 f'' / 8
  {  // the self-allocating form with real2 values
    TabulatedFunction<real2> t2(1024, real(0), real(2), [](double r) { return make_real2((real)r, (real)(r * r)); });
    real2 *d_o2;
    (void)hipMalloc(&d_o2, n * sizeof(real2));
    hipLaunchKernelGGL((k_eval<TabulatedFunction<real2>, real2>), dim3((n + 127) / 128), dim3(128), 0, 0, t2, d_r, n, d_o2);
    std::vector<real2> h2(n);
    (void)hipMemcpy(h2.data(), d_o2, n * sizeof(real2), hipMemcpyDeviceToHost);
    for (int k = 5; k < n; ++k)
      if (hr[k] < 2) { assert(std::fabs(h2[k].x - hr[k]) < 1e-6 && std::fabs(h2[k].y - hr[k] * hr[k]) < 2e-6); }
      else assert(h2[k].x == 0 && h2[k].y == 0);
    (void)hipFree(d_o2);
  }
  {  // ownership: a copy keeps a self-allocated table's samples alive after the original is gone; assignment releases what it replaces;
     // the pool sees the block come back exactly once, with the last owner
    auto &pool = uammd::detail::DevicePool::instance();
    const size_t live0 = pool.blocksLive();
    TabulatedFunction<real> outer;
    {
      TabulatedFunction<real> inner(512, real(0), real(1), [](double r) { return (real)(2.0 * r); });
      assert(pool.blocksLive() == live0 + 1);
      outer = inner;
      TabulatedFunction<real> third(inner);
      third = TabulatedFunction<real>(64, real(0), real(1), [](double r) { return (real)r; });   // a second block, released with `third`
      assert(pool.blocksLive() == live0 + 2);
    }
    assert(pool.blocksLive() == live0 + 1);   // `inner` and `third` are gone, `outer` still owns the first block
    real rr = real(0.3), *d_one, *d_res, res = 0;
    (void)hipMalloc(&d_one, sizeof(real));
    (void)hipMalloc(&d_res, sizeof(real));
    (void)hipMemcpy(d_one, &rr, sizeof(real), hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k_eval<TabulatedFunction<real>, real>), dim3(1), dim3(64), 0, 0, outer, d_one, 1, d_res);
    (void)hipMemcpy(&res, d_res, sizeof(real), hipMemcpyDeviceToHost);
    assert(std::fabs(res - real(0.6)) < 1e-6);
    assert(pool.blocksLive() == live0 + 1);   // (the launch's copy came and went)
    outer = TabulatedFunction<real>();
    assert(pool.blocksLive() == live0);
    (void)hipFree(d_one);
    (void)hipFree(d_res);
  }
  std::printf("tabulated_function: ok (largest interpolation error %.2e)\n", worst);
  return 0;
}
