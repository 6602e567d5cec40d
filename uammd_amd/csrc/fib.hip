// BDHI::FIB — Fluctuating Immersed Boundary: the fluctuating Stokes equation on a staggered grid instead of a mobility
// kernel.  Another consumer of the spread / FFT / gather engine.
//
// Reference behaviour (Integrator/BDHI/FIB/FIB.cu, as it actually runs: both Scheme values dispatch to forwardMidpoint,
// FIB.cu:1072-1079, and addThermalDrift returns on its first line, :400):
//   g = sqrt(2 eta kT/(dt dV)) D~ W  +  S F        addRandomAdvection :274-391, spreadParticleForces :511-597
//   v = eta^-1 L^-1 g                              3 x R2C, solveStokesFourier :667-724 (face <-> centre phase shifts,
//                                                  k_eff = 2/h sin(k h/2), divergence-free projection), 3 x C2R
//   q^{n+1/2} = q^n + dt/2 J(q^n) v ;  q^{n+1} = q^n + dt J(q^{n+1/2}) v      midPointStep :726-823
// The window is the 3-point Peskin kernel (FIB.cuh:168); every velocity component lives on its own face-centred grid, so a
// particle touches 3 x 27 nodes.  HIP design: three component PLANES transformed in place by one batched rocFFT plan, one
// wave per particle for spreading and interpolation (81 (component, node) pairs over the lanes).
// The reference draws the 6 N_cells fluid random numbers with cuRAND (third party, stream unpinned); here they come from
// Saru(cell + slot * N_cells, seed, step), or from the caller (uammd_fib_set_noise) for the parity tests.
#include "celllist.hpp"
#include "stagger.hpp"

#include <rocfft/rocfft.h>

#include <algorithm>
#include <cmath>

namespace uammd_hip {

int rocfft_setup_once();  // fcm.hip

#define UH_ROCFFT(expr)                                                                      \
  do {                                                                                       \
    rocfft_status s_ = (expr);                                                               \
    if (s_ != rocfft_status_success) {                                                       \
      set_last_error("%s failed with rocfft_status %d (%s:%d)", #expr, (int)s_, __FILE__, __LINE__); \
      return -10 - (int)s_;                                                                  \
    }                                                                                        \
  } while (0)

struct FIB {
  uammd_fib_parameters par{};
  GridT<float> grid{};
  float hKernel = 0.f, rh = 0.f;
  int nxpad = 0;
  size_t planeReal = 0, planeCplx = 0;
  DeviceBuffer gridBuf, random, posOld, work;
  const float *externalNoise = nullptr;
  rocfft_plan fwd = nullptr, inv = nullptr;
  rocfft_execution_info info = nullptr;
  unsigned long long step = 0;
  ~FIB() {
    if (fwd) rocfft_plan_destroy(fwd);
    if (inv) rocfft_plan_destroy(inv);
    if (info) rocfft_execution_info_destroy(info);
  }
};

static int next_fft_wise3(int n) {  // FIB_ns::nextFFTWiseSize3D (FIB.cu:31-84) = utils/Grid.cuh:142-213, one axis
  static const int primes[5] = {2, 3, 5, 7, 11}, maxExp[5] = {64, 64, 5, 4, 3};
  for (int c = std::max(n, 1);; ++c) {
    if (c % 2) continue;
    int m = c;
    bool ok = true;
    for (int p = 0; p < 5; ++p) {
      int e = 0;
      while (m % primes[p] == 0) { m /= primes[p]; ++e; }
      ok = ok && e <= maxExp[p];
    }
    if (ok && m == 1) return c;
  }
}

static int fib_make_plans(FIB *f) {
  if (int e = rocfft_setup_once()) return e;
  const size_t nx = f->grid.cellDim.x, ny = f->grid.cellDim.y, nz = f->grid.cellDim.z, nkx = nx / 2 + 1;
  const size_t lengths[3] = {nx, ny, nz};
  const size_t rstr[3] = {1, (size_t)f->nxpad, (size_t)f->nxpad * ny}, cstr[3] = {1, nkx, nkx * ny};
  rocfft_plan_description d = nullptr;
  UH_ROCFFT(rocfft_plan_description_create(&d));
  UH_ROCFFT(rocfft_plan_description_set_data_layout(d, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, nullptr, nullptr,
                                                     3, rstr, f->planeReal, 3, cstr, f->planeCplx));
  UH_ROCFFT(rocfft_plan_create(&f->fwd, rocfft_placement_inplace, rocfft_transform_type_real_forward, rocfft_precision_single, 3, lengths,
                               3, d));
  UH_ROCFFT(rocfft_plan_description_destroy(d));
  UH_ROCFFT(rocfft_plan_description_create(&d));
  UH_ROCFFT(rocfft_plan_description_set_data_layout(d, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, nullptr, nullptr,
                                                     3, cstr, f->planeCplx, 3, rstr, f->planeReal));
  UH_ROCFFT(rocfft_plan_create(&f->inv, rocfft_placement_inplace, rocfft_transform_type_real_inverse, rocfft_precision_single, 3, lengths,
                               3, d));
  UH_ROCFFT(rocfft_plan_description_destroy(d));
  size_t wf = 0, wi = 0;
  UH_ROCFFT(rocfft_plan_get_work_buffer_size(f->fwd, &wf));
  UH_ROCFFT(rocfft_plan_get_work_buffer_size(f->inv, &wi));
  const size_t w = std::max(wf, wi);
  UH_ROCFFT(rocfft_execution_info_create(&f->info));
  if (w) {
    if (int e = f->work.reserve(w)) return e;
    UH_ROCFFT(rocfft_execution_info_set_work_buffer(f->info, f->work.ptr, w));
  }
  return 0;
}

}  // namespace uammd_hip

using namespace uammd_hip;

extern "C" {

int uammd_fib_create(const uammd_fib_parameters *par, uammd_fib **out, int cells[3], float *hydrodynamicRadius) {
  if (!par || !out) { set_last_error("uammd_fib_create: null argument"); return -1; }
  if (par->hydrodynamicRadius > 0 && par->cells[0] > 0) {
    set_last_error("[BDHI::FIB] Please provide hydrodynamic radius OR cell dimensions, not both.");  // FIB.cu:95-97
    return -2;
  }
  if (par->cells[0] < 0 && par->hydrodynamicRadius < 0) {
    set_last_error("[BHDI::FIB] I need either the hydrodynamic radius or the number of cells!");      // :101-103
    return -2;
  }
  if (!(par->boxSize[0] > 0) || !(par->boxSize[1] > 0) || !(par->boxSize[2] > 0) || !(par->viscosity > 0) || !(par->dt > 0)) {
    set_last_error("uammd_fib_create: box, viscosity and dt must be positive");
    return -1;
  }
  FIB *f = new (std::nothrow) FIB();
  if (!f) { set_last_error("uammd_fib_create: out of host memory"); return -3; }
  f->par = *par;
  int cd[3] = {par->cells[0], par->cells[1], par->cells[2]};
  if (cd[0] < 0) {
    const float hgrid = par->hydrodynamicRadius / 0.91f;  // Peskin::threePoint::adviseGridSize, FIB_kernels.cuh:118-120
    for (int a = 0; a < 3; ++a) cd[a] = next_fft_wise3((int)(par->boxSize[a] / hgrid));
  }
  if (cd[0] < 3) cd[0] = 3;
  if (cd[1] < 3) cd[1] = 3;
  if (cd[2] == 2) cd[2] = 3;
  if (cd[2] < 3) {
    set_last_error("uammd_fib_create: a grid with %d cells along z is not supported (the 2D mode of the reference is untested there)", cd[2]);
    delete f;
    return -2;
  }
  const int per[3] = {1, 1, 1};
  f->grid = make_grid(make_box<float>(par->boxSize, per), make_int3(cd[0], cd[1], cd[2]));
  f->hKernel = std::min(f->grid.cellSize.x, std::min(f->grid.cellSize.y, f->grid.cellSize.z));
  f->rh = f->grid.cellSize.x * 0.91f;  // fixHydrodynamicRadius(h, cellSize.x), FIB.cu:121
  f->nxpad = 2 * (cd[0] / 2 + 1);
  f->planeReal = (size_t)f->nxpad * cd[1] * cd[2];
  f->planeCplx = (size_t)(cd[0] / 2 + 1) * cd[1] * cd[2];
  int e = f->gridBuf.reserve(sizeof(float) * 3 * f->planeReal);
  if (!e && par->temperature != 0.0f) e = f->random.reserve(sizeof(float) * 6 * (size_t)cd[0] * cd[1] * cd[2]);
  if (!e) e = fib_make_plans(f);
  if (e) { delete f; return e; }
  if (cells) for (int a = 0; a < 3; ++a) cells[a] = cd[a];
  if (hydrodynamicRadius) *hydrodynamicRadius = f->rh;
  *out = reinterpret_cast<uammd_fib *>(f);
  return 0;
}

int uammd_fib_destroy(uammd_fib *h) {
  delete reinterpret_cast<FIB *>(h);
  return 0;
}

// Test hook: the 6 * ncells fluid random numbers of the NEXT steps come from this device array (slot-major) instead of Saru.
int uammd_fib_set_noise(uammd_fib *h, const float *d_random) {
  if (!h) { set_last_error("uammd_fib_set_noise: null argument"); return -1; }
  reinterpret_cast<FIB *>(h)->externalNoise = d_random;
  return 0;
}

// forwardMidpoint (:965-1000) after the interactors have run: d_force real4[N] (NULL: no forces), d_pos real4[N] advanced in place.
int uammd_fib_forward(uammd_fib *h, float *d_pos, const float *d_force, int N, void *stream) {
  if (!h || !d_pos) { set_last_error("uammd_fib_forward: null argument"); return -1; }
  FIB *f = reinterpret_cast<FIB *>(h);
  hipStream_t st = (hipStream_t)stream;
  f->step++;
  if (N <= 0) return 0;
  if (int e = f->posOld.reserve(sizeof(float4) * (size_t)N)) return e;
  float *g = (float *)f->gridBuf.ptr;
  const int3 n = f->grid.cellDim;
  const int nc = n.x * n.y * n.z;
  const float T = f->par.temperature;
  UH_CHECK(hipMemsetAsync(g, 0, sizeof(float) * 3 * f->planeReal, st));
  if (T != 0.0f) {
    const float *rnd = f->externalNoise;
    if (!rnd) {
      hipLaunchKernelGGL(k_fib_noise, dim3((3 * nc + 255) / 256), dim3(256), 0, st, (float *)f->random.ptr, nc, f->par.seed, (uint)f->step);
      rnd = (const float *)f->random.ptr;
    }
    const double dV = (double)f->grid.cellSize.x * f->grid.cellSize.y * f->grid.cellSize.z;
    const float pref = (float)sqrt(2 * f->par.viscosity * T / (f->par.dt * dV));  // :973-975
    hipLaunchKernelGGL(k_fib_random_advection, dim3((nc + 255) / 256), dim3(256), 0, st, g, f->planeReal, f->nxpad, f->grid, pref, rnd);
  }
  const float invh = 1.0f / f->hKernel;
  const dim3 gp((N + 3) / 4), bp(256);
  if (d_force)
    hipLaunchKernelGGL(k_fib_spread, gp, bp, 0, st, (const float4 *)d_pos, (const float4 *)d_force, g, f->planeReal, f->nxpad, N, f->grid, invh,
                       1.0f);
  UH_ROCFFT(rocfft_execution_info_set_stream(f->info, (void *)st));
  void *bufs[1] = {g};
  UH_ROCFFT(rocfft_execute(f->fwd, bufs, nullptr, f->info));
  const uint total = (uint)f->planeCplx;
  hipLaunchKernelGGL((k_fib_stokes<false>), dim3((total + 255) / 256), dim3(256), 0, st, (float2 *)g, f->planeCplx, n,
                     real3f{f->par.boxSize[0], f->par.boxSize[1], f->par.boxSize[2]}, f->par.viscosity, make_fastdiv(n.x / 2 + 1),
                     make_fastdiv(n.y), 0.0f, true);
  UH_ROCFFT(rocfft_execute(f->inv, bufs, nullptr, f->info));
  hipLaunchKernelGGL((k_fib_midpoint<0>), gp, bp, 0, st, (float4 *)d_pos, (float4 *)f->posOld.ptr, (const float *)g, f->planeReal, f->nxpad, N,
                     f->grid, invh, f->par.dt);
  hipLaunchKernelGGL((k_fib_midpoint<1>), gp, bp, 0, st, (float4 *)d_pos, (float4 *)f->posOld.ptr, (const float *)g, f->planeReal, f->nxpad, N,
                     f->grid, invh, f->par.dt);
  UH_CHECK(hipGetLastError());
  return 0;
}

float uammd_fib_self_mobility(float hydrodynamicRadius, float viscosity, float L) {  // FIB.cuh:152-163
  const long double rh = hydrodynamicRadius, a = rh / (long double)L, a2 = a * a, a3 = a2 * a;
  const long double c = 2.83729747948061947666591710460773907l, b = 0.19457l;
  const long double a6pref = 16.0l * M_PIl * M_PIl / 45.0l + 630.0L * b * b;
  return (float)(1.0l / (6.0l * M_PIl * viscosity * rh) * (1.0l - c * a + (4.0l / 3.0l) * M_PIl * a3 - a6pref * a3 * a3));
}

}  // extern "C"
