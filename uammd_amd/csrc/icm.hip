// Hydro::ICM — Inertial Coupling Method (zero excess mass): particles advected by an incompressible FLUCTUATING NAVIER-STOKES
// fluid kept on a staggered grid between steps.  Shares the staggered spreading / interpolation, the
// stochastic stress divergence and the shifted Fourier projector with BDHI::FIB (stagger.hpp).
//
// Reference behaviour (Integrator/Hydro/ICM.cu, forwardTime :1191-1224):
//   predictor  q^{n+1/2} = q^n + dt/2 J(q^n) v^n                                            midPointStep :413-500
//   fluid      g = v^n + dt nu/2 L v^n + D~W - dt/rho (3/2 adv^n - 1/2 adv^{n-1})            updateCellVelocityUnperturbed :793-822
//              g += dt/rho S(q^{n+1/2}) F(q^{n+1/2})  [+ RFD thermal drift]                 :86-159, :161-275
//              v^{n+1} = (I - dt nu/2 L)^-1 P g  in Fourier space                           solveStokesFourier :349-411
//   corrector  q^{n+1} = q^n + dt J(q^{n+1/2}) v^{n+1}
// updateCellVelocityUnperturbed updates the field in place while neighbouring threads still read it (a race in the
// reference); here every cell reads the OLD field and writes a second buffer.  The fluid random numbers (cuRAND in the
// reference) and the initial thermal velocities (System::rng gaussians) come from Saru streams of the handle's seed.
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

struct ICMState {
  uammd_icm_parameters par{};
  GridT<float> grid{};
  float rh = 0.f, deltaRFD = 0.f;
  int nxpad = 0;
  size_t planeReal = 0, planeCplx = 0;
  DeviceBuffer velA, velB, advOld, random, posOld, work;
  float *vel = nullptr, *velNext = nullptr;  // vel: v^n (3 padded planes); velNext: scratch the fluid update writes
  const float *externalNoise = nullptr;
  rocfft_plan fwd = nullptr, inv = nullptr;
  rocfft_execution_info info = nullptr;
  unsigned int step = 0;
  ~ICMState() {
    if (fwd) rocfft_plan_destroy(fwd);
    if (inv) rocfft_plan_destroy(inv);
    if (info) rocfft_execution_info_destroy(info);
  }
};

static int next_fft_wise_icm(int n) {  // ICM_ns::nextFFTWiseSize3D (ICM.cu:29-84) = utils/Grid.cuh:142-213, one axis
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

// initFluid (:1001-1023): independent N(0, sqrt(kT/(rho dV))) per face
__global__ void __launch_bounds__(256) k_icm_init(float *__restrict__ v, size_t plane, int nxpad, int3 n, float amp, uint seed) {
  const int ic = blockIdx.x * 256 + threadIdx.x;
  if (ic >= n.x * n.y * n.z) return;
  const int x = ic % n.x, y = (ic / n.x) % n.y, z = ic / (n.x * n.y);
  Saru rng((uint)ic, seed, 0x1c3u);
  const float2 a = rng.gf(0.0f, amp), b = rng.gf(0.0f, amp);
  const size_t node = (size_t)x + (size_t)nxpad * ((size_t)y + (size_t)n.y * (size_t)z);
  v[node] = a.x; v[plane + node] = a.y; v[2 * plane + node] = b.x;
}

// updateCellVelocityUnperturbed (:793-822) with computeVelLaplacian (:592-662) and computeAdvection (:664-779)
__global__ void __launch_bounds__(256) k_icm_update(const float *__restrict__ v, float *__restrict__ vNew, float *__restrict__ advOld,
                                                    size_t plane, int nxpad, GridT<float> grid, float density, float viscosity,
                                                    float noiseAmp, float dt, const float *__restrict__ random) {
  const int ic = blockIdx.x * 256 + threadIdx.x;
  const int3 n = grid.cellDim;
  if (ic >= n.x * n.y * n.z) return;
  const int x = ic % n.x, y = (ic / n.x) % n.y, z = ic / (n.x * n.y);
  auto at = [&](int a, int b, int c, int comp) {
    return v[comp * plane + (size_t)grid.pbc_x(a) + (size_t)nxpad * ((size_t)grid.pbc_y(b) + (size_t)n.y * (size_t)grid.pbc_z(c))];
  };
  real3f dw{0.f, 0.f, 0.f};
  if (noiseAmp != 0.0f) {
    dw = noise_divergence(grid, x, y, z, random);
    dw.x *= noiseAmp; dw.y *= noiseAmp; dw.z *= noiseAmp;
  }
  const real3f ih = grid.invCellSize;
  float lap[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v0 = at(x, y, z, c);
    lap[c] = ih.x * ih.x * (at(x + 1, y, z, c) - 2.0f * v0 + at(x - 1, y, z, c));
    lap[c] += ih.y * ih.y * (at(x, y + 1, z, c) - 2.0f * v0 + at(x, y - 1, z, c));
    lap[c] += ih.z * ih.z * (at(x, y, z + 1, c) - 2.0f * v0 + at(x, y, z - 1, c));
  }
  const float vx = at(x, y, z, 0), vy = at(x, y, z, 1), vz = at(x, y, z, 2);
  const float vx_px = at(x + 1, y, z, 0), vy_px = at(x + 1, y, z, 1), vz_px = at(x + 1, y, z, 2);
  const float vx_mx = at(x - 1, y, z, 0), vy_mx = at(x - 1, y, z, 1), vz_mx = at(x - 1, y, z, 2);
  const float vx_py = at(x, y + 1, z, 0), vy_py = at(x, y + 1, z, 1), vz_py = at(x, y + 1, z, 2);
  const float vx_my = at(x, y - 1, z, 0), vy_my = at(x, y - 1, z, 1), vz_my = at(x, y - 1, z, 2);
  const float vx_pz = at(x, y, z + 1, 0), vy_pz = at(x, y, z + 1, 1), vz_pz = at(x, y, z + 1, 2);
  const float vx_mz = at(x, y, z - 1, 0), vy_mz = at(x, y, z - 1, 1), vz_mz = at(x, y, z - 1, 2);
  const float vy_px_my = at(x + 1, y - 1, z, 1), vz_px_mz = at(x + 1, y, z - 1, 2);
  const float vx_mx_py = at(x - 1, y + 1, z, 0), vz_py_mz = at(x, y + 1, z - 1, 2);
  const float vx_mx_pz = at(x - 1, y, z + 1, 0), vy_my_pz = at(x, y - 1, z + 1, 1);
  float adv[3];
  adv[0] = ih.x * ((vx_px + vx) * (vx_px + vx) - (vx + vx_mx) * (vx + vx_mx));
  adv[0] += ih.y * ((vx_py + vx) * (vy_px + vy) - (vx + vx_my) * (vy_px_my + vy_my));
  adv[0] += ih.z * ((vx_pz + vx) * (vz_px + vz) - (vx + vx_mz) * (vz_px_mz + vz_mz));
  adv[1] = ih.x * ((vy_px + vy) * (vx_py + vx) - (vy + vy_mx) * (vx_mx_py + vx_mx));
  adv[1] += ih.y * ((vy_py + vy) * (vy_py + vy) - (vy + vy_my) * (vy + vy_my));
  adv[1] += ih.z * ((vy_pz + vy) * (vz_py + vz) - (vy + vy_mz) * (vz_py_mz + vz_mz));
  adv[2] = ih.x * ((vz_px + vz) * (vx_pz + vx) - (vz + vz_mx) * (vx_mx_pz + vx_mx));
  adv[2] += ih.y * ((vz_py + vz) * (vy_pz + vy) - (vz + vz_my) * (vy_my_pz + vy_my));
  adv[2] += ih.z * ((vz_pz + vz) * (vz_pz + vz) - (vz + vz_mz) * (vz + vz_mz));
  const float vc[3] = {vx, vy, vz}, dwc[3] = {dw.x, dw.y, dw.z};
  const size_t node = (size_t)x + (size_t)nxpad * ((size_t)y + (size_t)n.y * (size_t)z);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    adv[c] *= 0.25f * density;
    const float old = advOld[c * plane + node];
    vNew[c * plane + node] = vc[c] + ((dt * viscosity * 0.5f / density) * lap[c] + dwc[c] - (dt / density) * (1.5f * adv[c] - 0.5f * old));
    advOld[c * plane + node] = adv[c];
  }
}

// addThermalDrift (:161-275): one wave per particle, W ~ Saru(id, seed, step)
__global__ void __launch_bounds__(256) k_icm_drift(const float4 *__restrict__ pos, float *__restrict__ g, size_t plane, int nxpad, int N,
                                                   GridT<float> grid, float invh, float driftPrefactor, float deltaRFD, uint seed,
                                                   uint step) {
  const int lane = threadIdx.x & 63;
  const int id = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (id >= N) return;
  const float4 p = pos[id];
  Saru rng((uint)id, seed, step);
  const float2 a = rng.gf(0.0f, 1.0f), b = rng.gf(0.0f, 1.0f);
  const float W[3] = {a.x, a.y, b.x};
  const real3f pi{p.x, p.y, p.z};
  const real3f qp{pi.x + 0.5f * deltaRFD * W[0], pi.y + 0.5f * deltaRFD * W[1], pi.z + 0.5f * deltaRFD * W[2]};
  const real3f qm{pi.x - 0.5f * deltaRFD * W[0], pi.y - 0.5f * deltaRFD * W[1], pi.z - 0.5f * deltaRFD * W[2]};
  for (int l = lane; l < 81; l += 64) {
    const int c = l / 27, i = l - 27 * c;
    real3f ps = pi, sp = qp, sm = qm;
    const float hh = c == 0 ? grid.cellSize.x : (c == 1 ? grid.cellSize.y : grid.cellSize.z);
    if (c == 0) { ps.x -= 0.5f * hh; sp.x -= 0.5f * hh; sm.x -= 0.5f * hh; }
    if (c == 1) { ps.y -= 0.5f * hh; sp.y -= 0.5f * hh; sm.y -= 0.5f * hh; }
    if (c == 2) { ps.z -= 0.5f * hh; sp.z -= 0.5f * hh; sm.z -= 0.5f * hh; }
    const int3 cell = grid.getCell(ps);
    const int3 cj = make_int3(grid.pbc_x(cell.x + i % 3 - 1), grid.pbc_y(cell.y + (i / 3) % 3 - 1), grid.pbc_z(cell.z + i / 9 - 1));
    const real3f rp = grid.distanceToCellCenter(sp, cj), rm = grid.distanceToCellCenter(sm, cj);
    float s = peskin3(invh, rp.x) * peskin3(invh, rp.y) * peskin3(invh, rp.z) * W[c];
    s -= peskin3(invh, rm.x) * peskin3(invh, rm.y) * peskin3(invh, rm.z) * W[c];
    unsafeAtomicAdd(&g[c * plane + (size_t)cj.x + (size_t)nxpad * ((size_t)cj.y + (size_t)grid.cellDim.y * (size_t)cj.z)], s * driftPrefactor);
  }
}

// fluid velocity export / import: interleaved real3[nz][ny][nx]; COLLOCATE = interpolateVelocitiesToCellCentersD (ICM.cuh:96-119)
template <bool COLLOCATE>
__global__ void __launch_bounds__(256) k_icm_export(const float *__restrict__ v, size_t plane, int nxpad, GridT<float> grid,
                                                    float *__restrict__ out3) {
  const int ic = blockIdx.x * 256 + threadIdx.x;
  const int3 n = grid.cellDim;
  if (ic >= n.x * n.y * n.z) return;
  const int x = ic % n.x, y = (ic / n.x) % n.y, z = ic / (n.x * n.y);
  auto at = [&](int a, int b, int c, int comp) {
    return v[comp * plane + (size_t)grid.pbc_x(a) + (size_t)nxpad * ((size_t)grid.pbc_y(b) + (size_t)n.y * (size_t)grid.pbc_z(c))];
  };
  float o[3] = {at(x, y, z, 0), at(x, y, z, 1), at(x, y, z, 2)};
  if (COLLOCATE) {
    o[0] = 0.5f * (o[0] + at(x - 1, y, z, 0));
    o[1] = 0.5f * (o[1] + at(x, y - 1, z, 1));
    o[2] = 0.5f * (o[2] + at(x, y, z - 1, 2));
  }
  out3[3 * (size_t)ic] = o[0]; out3[3 * (size_t)ic + 1] = o[1]; out3[3 * (size_t)ic + 2] = o[2];
}
__global__ void __launch_bounds__(256) k_icm_import(float *__restrict__ v, size_t plane, int nxpad, int3 n, const float *__restrict__ in3) {
  const int ic = blockIdx.x * 256 + threadIdx.x;
  if (ic >= n.x * n.y * n.z) return;
  const int x = ic % n.x, y = (ic / n.x) % n.y, z = ic / (n.x * n.y);
  const size_t node = (size_t)x + (size_t)nxpad * ((size_t)y + (size_t)n.y * (size_t)z);
  v[node] = in3[3 * (size_t)ic]; v[plane + node] = in3[3 * (size_t)ic + 1]; v[2 * plane + node] = in3[3 * (size_t)ic + 2];
}

static int icm_make_plans(ICMState *f) {
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

int uammd_icm_create(const uammd_icm_parameters *par, uammd_icm **out, int cells[3], float *hydrodynamicRadius) {
  if (!par || !out) { set_last_error("uammd_icm_create: null argument"); return -1; }
  if (par->density < 0) { set_last_error("[Hydro::ICM] Please provide fluid density"); return -2; }        // ICM.cu:833-834
  if (par->viscosity < 0) { set_last_error("[Hydro::ICM] Please provide fluid viscosity"); return -2; }    // :835-836
  if (par->hydrodynamicRadius > 0 && par->cells[0] > 0) {
    set_last_error("[Hydro::ICM] Please provide hydrodynamic radius OR cell dimensions, not both.");         // :837-839
    return -2;
  }
  if (par->cells[0] < 0 && par->hydrodynamicRadius < 0) {
    set_last_error("[BHDI::ICM] I need either the hydrodynamic radius or the number of cells!");            // :869-872
    return -2;
  }
  if (!(par->boxSize[0] > 0) || !(par->boxSize[1] > 0) || !(par->boxSize[2] > 0) || !(par->dt > 0) || !(par->density > 0)) {
    set_last_error("uammd_icm_create: box, dt and density must be positive");
    return -1;
  }
  ICMState *f = new (std::nothrow) ICMState();
  if (!f) { set_last_error("uammd_icm_create: out of host memory"); return -3; }
  f->par = *par;
  int cd[3] = {par->cells[0], par->cells[1], par->cells[2]};
  if (cd[0] < 0) {
    const float hgrid = (float)(par->hydrodynamicRadius / 0.91);  // :873
    for (int a = 0; a < 3; ++a) cd[a] = next_fft_wise_icm((int)(par->boxSize[a] / hgrid));
  }
  if (cd[0] < 3) cd[0] = 3;
  if (cd[1] < 3) cd[1] = 3;
  if (cd[2] == 2) cd[2] = 3;
  if (cd[2] < 3) {
    set_last_error("uammd_icm_create: a grid with %d cells along z is not supported", cd[2]);
    delete f;
    return -2;
  }
  const int per[3] = {1, 1, 1};
  f->grid = make_grid(make_box<float>(par->boxSize, per), make_int3(cd[0], cd[1], cd[2]));
  f->rh = 0.91f * par->boxSize[0] / (float)cd[0];  // getHydrodynamicRadius, ICM.cuh:169-171
  f->deltaRFD = (float)(1e-4 * (double)f->rh);      // single precision build, ICM.cu:844-848
  f->nxpad = 2 * (cd[0] / 2 + 1);
  f->planeReal = (size_t)f->nxpad * cd[1] * cd[2];
  f->planeCplx = (size_t)(cd[0] / 2 + 1) * cd[1] * cd[2];
  const size_t bytes = sizeof(float) * 3 * f->planeReal;
  int e = f->velA.reserve(bytes);
  if (!e) e = f->velB.reserve(bytes);
  if (!e) e = f->advOld.reserve(bytes);
  if (!e && par->temperature != 0.0f) e = f->random.reserve(sizeof(float) * 6 * (size_t)cd[0] * cd[1] * cd[2]);
  if (!e) e = icm_make_plans(f);
  if (e) { delete f; return e; }
  f->vel = (float *)f->velA.ptr;
  f->velNext = (float *)f->velB.ptr;
  if (hipMemset(f->vel, 0, bytes) != hipSuccess || hipMemset(f->velNext, 0, bytes) != hipSuccess ||
      hipMemset(f->advOld.ptr, 0, bytes) != hipSuccess) {
    set_last_error("uammd_icm_create: hipMemset failed");
    delete f;
    return -4;
  }
  if (par->temperature > 0.0f) {  // initFluid
    const int nc = cd[0] * cd[1] * cd[2];
    const double dV = (double)f->grid.cellSize.x * f->grid.cellSize.y * f->grid.cellSize.z;
    const float amp = (float)sqrt(par->temperature / (par->density * dV));
    hipLaunchKernelGGL(k_icm_init, dim3((nc + 255) / 256), dim3(256), 0, 0, f->vel, f->planeReal, f->nxpad, f->grid.cellDim, amp, par->seed);
    if (hipDeviceSynchronize() != hipSuccess) { set_last_error("uammd_icm_create: initFluid failed"); delete f; return -4; }
  }
  if (cells) for (int a = 0; a < 3; ++a) cells[a] = cd[a];
  if (hydrodynamicRadius) *hydrodynamicRadius = f->rh;
  *out = reinterpret_cast<uammd_icm *>(f);
  return 0;
}

int uammd_icm_destroy(uammd_icm *h) {
  delete reinterpret_cast<ICMState *>(h);
  return 0;
}

int uammd_icm_set_noise(uammd_icm *h, const float *d_random) {
  if (!h) { set_last_error("uammd_icm_set_noise: null argument"); return -1; }
  reinterpret_cast<ICMState *>(h)->externalNoise = d_random;
  return 0;
}

// predictorStep (:1150-1169): q^n is kept, d_pos becomes q^{n+1/2}.  Counts the step.
int uammd_icm_predictor(uammd_icm *h, float *d_pos, int N, void *stream) {
  if (!h || !d_pos) { set_last_error("uammd_icm_predictor: null argument"); return -1; }
  ICMState *f = reinterpret_cast<ICMState *>(h);
  f->step++;
  if (N <= 0) return 0;
  if (int e = f->posOld.reserve(sizeof(float4) * (size_t)N)) return e;
  hipLaunchKernelGGL((k_fib_midpoint<0>), dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, (float4 *)d_pos, (float4 *)f->posOld.ptr,
                     (const float *)f->vel, f->planeReal, f->nxpad, N, f->grid, 1.0f / f->grid.cellSize.x, f->par.dt);
  UH_CHECK(hipGetLastError());
  return 0;
}

// unperturbedFluidForcing + spreadParticleForces + thermalDrift + applyStokesSolutionOperator + correctorStep (:1208-1219).
// d_force real4[N]: the forces at q^{n+1/2} (NULL: no Interactor attached).
int uammd_icm_fluid_and_corrector(uammd_icm *h, float *d_pos, const float *d_force, int N, void *stream) {
  if (!h || !d_pos) { set_last_error("uammd_icm_fluid_and_corrector: null argument"); return -1; }
  ICMState *f = reinterpret_cast<ICMState *>(h);
  hipStream_t st = (hipStream_t)stream;
  const int3 n = f->grid.cellDim;
  const int nc = n.x * n.y * n.z;
  const float T = f->par.temperature, rho = f->par.density, eta = f->par.viscosity, dt = f->par.dt;
  const float *rnd = nullptr;
  float noiseAmp = 0.0f;
  if (T != 0.0f) {
    rnd = f->externalNoise;
    if (!rnd) {
      hipLaunchKernelGGL(k_fib_noise, dim3((3 * nc + 255) / 256), dim3(256), 0, st, (float *)f->random.ptr, nc, f->par.seed, f->step);
      rnd = (const float *)f->random.ptr;
    }
    const float dV = f->grid.cellSize.x * f->grid.cellSize.y * f->grid.cellSize.z;
    noiseAmp = sqrtf(2 * T * eta * dt / dV) / rho;  // :1107
  }
  hipLaunchKernelGGL(k_icm_update, dim3((nc + 255) / 256), dim3(256), 0, st, (const float *)f->vel, f->velNext, (float *)f->advOld.ptr,
                     f->planeReal, f->nxpad, f->grid, rho, eta, noiseAmp, dt, rnd);
  std::swap(f->vel, f->velNext);
  float *g = f->vel;
  const float invh = 1.0f / f->grid.cellSize.x;
  const dim3 gp((N + 3) / 4), bp(256);
  if (d_force && N > 0)
    hipLaunchKernelGGL(k_fib_spread, gp, bp, 0, st, (const float4 *)d_pos, (const float4 *)d_force, g, f->planeReal, f->nxpad, N, f->grid, invh,
                       dt / rho);
  if (f->par.sumThermalDrift && T > 0.0f && N > 0)
    hipLaunchKernelGGL(k_icm_drift, gp, bp, 0, st, (const float4 *)d_pos, g, f->planeReal, f->nxpad, N, f->grid, invh,
                       (dt / rho) * T / f->deltaRFD, f->deltaRFD, f->par.seed, f->step);
  UH_ROCFFT(rocfft_execution_info_set_stream(f->info, (void *)st));
  void *bufs[1] = {g};
  UH_ROCFFT(rocfft_execute(f->fwd, bufs, nullptr, f->info));
  const uint total = (uint)f->planeCplx;
  hipLaunchKernelGGL((k_fib_stokes<true>), dim3((total + 255) / 256), dim3(256), 0, st, (float2 *)g, f->planeCplx, n,
                     real3f{f->par.boxSize[0], f->par.boxSize[1], f->par.boxSize[2]}, eta, make_fastdiv(n.x / 2 + 1), make_fastdiv(n.y),
                     dt / rho, f->par.removeTotalMomentum != 0);
  UH_ROCFFT(rocfft_execute(f->inv, bufs, nullptr, f->info));
  if (N > 0)
    hipLaunchKernelGGL((k_fib_midpoint<1>), gp, bp, 0, st, (float4 *)d_pos, (float4 *)f->posOld.ptr, (const float *)g, f->planeReal, f->nxpad,
                       N, f->grid, invh, dt);
  UH_CHECK(hipGetLastError());
  return 0;
}

// d_out real3[nz][ny][nx]: the face-centred field (collocated = 0) or ICM::getFluidVelocities (collocated = 1, ICM.cuh:176-199)
int uammd_icm_get_fluid_velocity(uammd_icm *h, float *d_out, int collocated, void *stream) {
  if (!h || !d_out) { set_last_error("uammd_icm_get_fluid_velocity: null argument"); return -1; }
  ICMState *f = reinterpret_cast<ICMState *>(h);
  const int nc = f->grid.cellDim.x * f->grid.cellDim.y * f->grid.cellDim.z;
  if (collocated)
    hipLaunchKernelGGL((k_icm_export<true>), dim3((nc + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float *)f->vel, f->planeReal,
                       f->nxpad, f->grid, d_out);
  else
    hipLaunchKernelGGL((k_icm_export<false>), dim3((nc + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float *)f->vel, f->planeReal,
                       f->nxpad, f->grid, d_out);
  UH_CHECK(hipGetLastError());
  return 0;
}

int uammd_icm_set_fluid_velocity(uammd_icm *h, const float *d_in, void *stream) {
  if (!h || !d_in) { set_last_error("uammd_icm_set_fluid_velocity: null argument"); return -1; }
  ICMState *f = reinterpret_cast<ICMState *>(h);
  const int nc = f->grid.cellDim.x * f->grid.cellDim.y * f->grid.cellDim.z;
  hipLaunchKernelGGL(k_icm_import, dim3((nc + 255) / 256), dim3(256), 0, (hipStream_t)stream, f->vel, f->planeReal, f->nxpad, f->grid.cellDim,
                     d_in);
  UH_CHECK(hipGetLastError());
  return 0;
}

}  // extern "C"
