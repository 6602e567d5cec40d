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

// spreadParticleForces (:528-597): one wave per particle, 81 atomics
__global__ void __launch_bounds__(256) k_fib_spread(const float4 *__restrict__ pos, const float4 *__restrict__ force, float *__restrict__ g,
                                                    size_t plane, int nxpad, int N, GridT<float> grid, float invh) {
  const int lane = threadIdx.x & 63;
  const int id = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (id >= N) return;
  const float4 p = pos[id], f = force[id];
  for (int l = lane; l < 81; l += 64) {
    const StagNode s = stag_node(grid, nxpad, invh, real3f{p.x, p.y, p.z}, l);
    const int c = l / 27;
    const float fc = c == 0 ? f.x : (c == 1 ? f.y : f.z);
    unsafeAtomicAdd(&g[c * plane + s.node], s.w * fc);
  }
}

// midPointStep (:726-823).  MODE 0 predictor, 1 corrector, 2 euler
template <int MODE>
__global__ void __launch_bounds__(256) k_fib_midpoint(float4 *__restrict__ pos, float4 *__restrict__ posOld, const float *__restrict__ g,
                                                      size_t plane, int nxpad, int N, GridT<float> grid, float invh, float dt) {
  const int lane = threadIdx.x & 63;
  const int id = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (id >= N) return;
  const float4 p = pos[id];
  const float dV = grid.cellSize.x * grid.cellSize.y * grid.cellSize.z;
  float acc[3] = {0.f, 0.f, 0.f};
  for (int l = lane; l < 81; l += 64) {
    const StagNode s = stag_node(grid, nxpad, invh, real3f{p.x, p.y, p.z}, l);
    const int c = l / 27;
    const float v = s.w * g[c * plane + s.node] * dV;
    acc[0] += c == 0 ? v : 0.0f; acc[1] += c == 1 ? v : 0.0f; acc[2] += c == 2 ? v : 0.0f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    acc[0] += __shfl_xor(acc[0], o, 64); acc[1] += __shfl_xor(acc[1], o, 64); acc[2] += __shfl_xor(acc[2], o, 64);
  }
  if (lane != 0) return;
  if (MODE == 0) {
    posOld[id] = p;
    const float pref = dt * 0.5f;
    pos[id] = make_float4(p.x + pref * acc[0], p.y + pref * acc[1], p.z + pref * acc[2], p.w);
  } else {
    const float4 po = posOld[id];
    pos[id] = make_float4(po.x + dt * acc[0], po.y + dt * acc[1], po.z + dt * acc[2], po.w);
  }
}

// solveStokesFourier (:667-724) on the three complex planes [nz][ny][nkx]
__global__ void __launch_bounds__(256) k_fib_stokes(float2 *__restrict__ g, size_t planeCplx, int3 n, real3f L, float viscosity, FastDiv dkx,
                                                    FastDiv dny) {
  const uint id = blockIdx.x * 256 + threadIdx.x;
  const int nkx = n.x / 2 + 1;
  if (id >= (uint)(nkx * n.y * n.z)) return;
  const uint row = dkx.div(id);
  const int cx = (int)(id - row * (uint)nkx);
  const int cz = (int)dny.div(row);
  const int cy = (int)(row - (uint)cz * (uint)n.y);
  float2 v[3] = {g[id], g[planeCplx + id], g[2 * planeCplx + id]};
  if (id == 0) {
    v[0] = v[1] = v[2] = make_float2(0.f, 0.f);
  } else {
    const float hx = L.x / (float)n.x, hy = L.y / (float)n.y, hz = L.z / (float)n.z;
    const float px = 2.0f * (float)M_PI / L.x, py = 2.0f * (float)M_PI / L.y, pz = 2.0f * (float)M_PI / L.z;
    float kx = (float)cx * px, ky = (float)cy * py, kz = (float)cz * pz;  // cellToWaveNumber with the (n+1)/2 threshold (:603-617)
    if (cx >= (n.x + 1) / 2) kx -= (float)n.x * px;
    if (cy >= (n.y + 1) / 2) ky -= (float)n.y * py;
    if (cz >= (n.z + 1) / 2) kz -= (float)n.z * pz;
    float sn[3], cs[3];
    sincosf(kx * hx * 0.5f, &sn[0], &cs[0]);
    sincosf(ky * hy * 0.5f, &sn[1], &cs[1]);
    sincosf(kz * hz * 0.5f, &sn[2], &cs[2]);
    const real3f keff{2.0f * (1.0f / hx) * sn[0], 2.0f * (1.0f / hy) * sn[1], 2.0f * (1.0f / hz) * sn[2]};
    float re[3], im[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {  // faces -> centres: phase (cos, -sin)
      re[c] = v[c].x * cs[c] - v[c].y * (-sn[c]);
      im[c] = v[c].y * cs[c] + v[c].x * (-sn[c]);
    }
    const float k2 = dot3(keff, keff);
    const float invL = -1.0f / k2;
    const float pref = -1.0f * invL / viscosity;
#pragma unroll
    for (int c = 0; c < 3; ++c) { re[c] *= pref; im[c] *= pref; }
    const float invk2 = 1.0f / k2;
    const float kfr = dot3(keff, real3f{re[0], re[1], re[2]}) * invk2, kfi = dot3(keff, real3f{im[0], im[1], im[2]}) * invk2;
    const float ke[3] = {keff.x, keff.y, keff.z};
    const float norm = 1.0f / (float)(n.x * n.y * n.z);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float tr = re[c] - ke[c] * kfr, ti = im[c] - ke[c] * kfi;
      v[c] = make_float2(norm * (tr * cs[c] - ti * sn[c]), norm * (ti * cs[c] + tr * sn[c]));  // centres -> faces, FFT normalisation
    }
  }
  g[id] = v[0];
  g[planeCplx + id] = v[1];
  g[2 * planeCplx + id] = v[2];
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
    hipLaunchKernelGGL(k_fib_spread, gp, bp, 0, st, (const float4 *)d_pos, (const float4 *)d_force, g, f->planeReal, f->nxpad, N, f->grid, invh);
  UH_ROCFFT(rocfft_execution_info_set_stream(f->info, (void *)st));
  void *bufs[1] = {g};
  UH_ROCFFT(rocfft_execute(f->fwd, bufs, nullptr, f->info));
  const uint total = (uint)f->planeCplx;
  hipLaunchKernelGGL(k_fib_stokes, dim3((total + 255) / 256), dim3(256), 0, st, (float2 *)g, f->planeCplx, n,
                     real3f{f->par.boxSize[0], f->par.boxSize[1], f->par.boxSize[2]}, f->par.viscosity, make_fastdiv(n.x / 2 + 1),
                     make_fastdiv(n.y));
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
