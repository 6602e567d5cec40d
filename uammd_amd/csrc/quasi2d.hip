// BDHI::True2D / BDHI::Quasi2D — Brownian hydrodynamics of particles confined to a plane: a 2D instance of the
// spread / FFT / gather engine of Path B with another Fourier-space operator.
//
// Reference behaviour (Integrator/Hydro/BDHI_quasi2D.cu, .cuh):
//   ctor        grid h = 0.8 a, window support 2(int(3 a n/L)+1)+1, Gaussian variance of the hydrodynamic kernel   .cu:61-88
//   forwardTime spread thermal drift (Quasi2D, T > 0) and forces -> 2D R2C (x, y velocities) -> G_k = g_k k_perp k_perp +
//               f_k k k -> sqrt(G_k) noise with the conjugate symmetry of a real field -> C2R -> gather -> Euler      .cu:179-541
// HIP design: the two velocity components live in two PLANES transformed in place by one batched 2D rocFFT plan; the
// Fourier-space step is ONE kernel in gather form (every node computes its own value: the noise of a node owned by its
// conjugate partner is regenerated from the partner's Saru stream), so no node is written by two threads.  The reference's
// scatter form writes the conjugate of the (nx/2, 0) node one row past the end of its array (indexOfConjugate with
// ik.y = 0, .cu:425-428); that write is not reproduced.
#include "celllist.hpp"
#include "ibm.hpp"
#include "saru.hpp"

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

struct BDHI2D {
  uammd_bdhi2d_parameters par{};
  GridT<float> grid{};
  IBMKernelDev kern{}, kernDriftX{}, kernDriftY{};
  int nxpad = 0;
  size_t planeReal = 0, planeCplx = 0;
  DeviceBuffer gridBuf, work;
  rocfft_plan fwd = nullptr, inv = nullptr;
  rocfft_execution_info info = nullptr;
  unsigned int counter = 0;  // the reference's `static ullint counter` (.cu:455), per handle here
  ~BDHI2D() {
    if (fwd) rocfft_plan_destroy(fwd);
    if (inv) rocfft_plan_destroy(inv);
    if (info) rocfft_execution_info_destroy(info);
  }
};

int next_fft_wise_axis(int n) {  // nextFFTWiseSize3D (utils/Grid.cuh:142-213), one axis
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

// BDHI2D_ns::True2D / Quasi2D::operator() (.cuh:88-92, :100-109)
UH_D float2 hydro_kernel(int mode, float k2, float a) {
  if (mode == 0) return make_float2(0.0f, 1.0f / (k2 * k2));
  const float k = sqrtf(k2);
  const float invk3 = 1.0f / (k2 * k);
  const float inv_sqrtpi = 0.564189583547756f;
  const float kp = k * a * inv_sqrtpi;
  const float fk = 0.5f * invk3 * (erfcf(kp) * (0.5f + kp * kp) * expf(kp * kp) - kp * inv_sqrtpi);
  const float gk = 0.5f * invk3 * erfcf(kp) * expf(kp * kp);
  return make_float2(fk, gk);
}
UH_D float2 wave_number(int ix, int iy, int nx, int ny, float Lx, float Ly) {  // cellToWaveNumber, .cu:314-320
  const float px = (2.0f * (float)M_PI) / Lx, py = (2.0f * (float)M_PI) / Ly;
  return make_float2((float)(ix - nx * (ix >= (nx / 2 + 1))) * px, (float)(iy - ny * (iy >= (ny / 2 + 1))) * py);
}
UH_D float2 project(float2 k, float2 f, float fk, float gk) {  // projectFourier for one real 2-vector, .cu:324-343
  const float dperp = fmaf(f.y, -k.x, f.x * k.y);
  const float dpar = fmaf(f.y, k.y, f.x * k.x);
  return make_float2(fmaf(k.x * fk, dpar, k.y * gk * dperp), fmaf(k.y * fk, dpar, -k.x * gk * dperp));
}
struct Cplx2 { float xr, xi, yr, yi; };
// the noise term of an OWNER node (fourierBrownianNoise, .cu:368-432, up to `gridVelsFourier[id] += factor`)
UH_D Cplx2 noise_factor(int id, int ix, int iy, int nx, int ny, float Lx, float Ly, int mode, float a, float prefactor, uint seed,
                        uint step) {
  const bool isXnyquist = (ix == (nx - ix)) && (nx % 2 == 0);
  const bool isYnyquist = (iy == (ny - iy)) && (ny % 2 == 0);
  const bool isNyquist = (isYnyquist && ix == 0) || (isXnyquist && isYnyquist);
  Saru saru((uint)id, step, seed);
  const float sc = 0.707106781186547f * prefactor;
  float2 n1 = saru.gf(0.0f, sc), n2 = saru.gf(0.0f, sc);
  if (isNyquist) {
    n1.x *= 1.41421356237310f; n2.x *= 1.41421356237310f;
    n1.y = 0.0f; n2.y = 0.0f;
  }
  const float2 k = wave_number(ix, iy, nx, ny, Lx, Ly);
  const float k2 = fmaf(k.y, k.y, k.x * k.x);
  const float2 fg = hydro_kernel(mode, k2, a);
  const float fs = sqrtf(fg.x), gs = sqrtf(fg.y);
  Cplx2 f;
  f.xr = fmaf(fs * n2.x, k.x, gs * n1.x * k.y); f.xi = fmaf(fs * n2.y, k.x, gs * n1.y * k.y);
  f.yr = fmaf(fs * n2.x, k.y, gs * n1.x * (-k.x)); f.yi = fmaf(fs * n2.y, k.y, gs * n1.y * (-k.x));
  return f;
}

// forceFourier2Vel + fourierBrownianNoise in gather form.  gx, gy: the two component planes, complex[ny][nkx].
__global__ void __launch_bounds__(256) k_q2d_kspace(float2 *__restrict__ gx, float2 *__restrict__ gy, int nx, int ny, float Lx, float Ly,
                                                    int mode, float a, float viscosity, bool deterministic, float noisePrefactor,
                                                    uint seed, uint step) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  const int nkx = nx / 2 + 1;
  if (id >= ny * nkx) return;
  const int ix = id % nkx, iy = id / nkx;
  Cplx2 v{0.f, 0.f, 0.f, 0.f};
  if (id != 0) {
    if (deterministic) {
      const float2 k = wave_number(ix, iy, nx, ny, Lx, Ly);
      const float k2 = fmaf(k.y, k.y, k.x * k.x);
      const float2 fg = hydro_kernel(mode, k2, a);
      const float fk = fg.x / (viscosity * (float)(nx * ny)), gk = fg.y / (viscosity * (float)(nx * ny));
      const float2 fx = gx[id], fy = gy[id];
      const float2 vr = project(k, make_float2(fx.x, fy.x), fk, gk), vi = project(k, make_float2(fx.y, fy.y), fk, gk);
      v = Cplx2{vr.x, vi.x, vr.y, vi.y};
    }
    if (noisePrefactor != 0.0f) {
      const bool selfConjColumn = ix == 0 || ix == nx - ix;
      if (selfConjColumn && iy > ny - iy) {
        // owned by the conjugate partner (ix, ny - iy): its factor, conjugated (.cu:422-428)
        const int jy = ny - iy;
        const Cplx2 f = noise_factor(ix + nkx * jy, ix, jy, nx, ny, Lx, Ly, mode, a, noisePrefactor, seed, step);
        v.xr += f.xr; v.xi += -f.xi; v.yr += f.yr; v.yi += -f.yi;
      } else {
        const bool isXnyquist = (ix == (nx - ix)) && (nx % 2 == 0);
        if (isXnyquist && iy == 0) v = Cplx2{0.f, 0.f, 0.f, 0.f};  // .cu:393-395: this node's deterministic part is wiped
        const Cplx2 f = noise_factor(id, ix, iy, nx, ny, Lx, Ly, mode, a, noisePrefactor, seed, step);
        v.xr += f.xr; v.xi += f.xi; v.yr += f.yr; v.yi += f.yi;
      }
    }
  }
  gx[id] = make_float2(v.xr, v.xi);
  gy[id] = make_float2(v.yr, v.yi);
}

// IBM::spread / gather on the two planes, one wave per particle (misc/IBM.cu:83-147, :164-235; 2D branch of IBM.cuh:182-194).
// SPREAD: value = force.xy of the particle, or the constant (cx, cy) when force == nullptr (thermal drift).
// KIND (the window) is a template parameter: with a run-time kind the compiler evaluates EVERY window of phi_axis per weight.
template <bool SPREAD, int KIND>
__global__ void __launch_bounds__(256) k_q2d_ibm(const float4 *__restrict__ pos, const float4 *__restrict__ force, float cx, float cy,
                                                 float *__restrict__ g0, size_t plane, float2 *__restrict__ vel, int N, GridT<float> grid,
                                                 int nxStride, IBMKernelDev kern, FastDiv dsx) {
  kern.kind = KIND;
  const int lane = threadIdx.x & 63;
  const int id = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (id >= N) return;
  const float4 p = pos[id];
  const Stencil s = make_stencil(grid, kern, real3f{p.x, p.y, p.z}, true, lane);
  const int sx = s.support.x, sy = s.support.y;
  const int nn = sx * sy;
  float vx = cx, vy = cy;
  if (SPREAD && force) { const float4 f = force[id]; vx = f.x; vy = f.y; }
  const float wz = __shfl(s.w, sx + sy, 64);  // = 1 for the 2D windows
  const float dV = grid.cellVolume;
  float ax = 0.f, ay = 0.f;
  for (int i0 = 0; i0 < nn; i0 += 64) {
    const int i = i0 + lane;
    const bool in = i < nn;
    const uint iu = in ? (uint)i : 0u;
    const uint jj = dsx.div(iu);
    const uint ii = iu - jj * (uint)sx;
    const float wx = __shfl(s.w, (int)ii, 64), wy = __shfl(s.w, sx + (int)jj, 64);
    if (!in) continue;
    const int gx = grid.pbc_x(s.celli.x + (int)ii - s.P.x), gy = grid.pbc_y(s.celli.y + (int)jj - s.P.y);
    if (gx < 0 || gy < 0 || gx >= grid.cellDim.x || gy >= grid.cellDim.y) continue;
    const size_t node = (size_t)gx + (size_t)nxStride * (size_t)gy;
    if (SPREAD) {
      if (vx != 0.0f) unsafeAtomicAdd(&g0[node], vx * wx * wy * wz);
      if (vy != 0.0f) unsafeAtomicAdd(&g0[plane + node], vy * wx * wy * wz);
    } else {
      ax = fmaf(dV, g0[node] * wx * wy * wz, ax);
      ay = fmaf(dV, g0[plane + node] * wx * wy * wz, ay);
    }
  }
  if (!SPREAD) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ax += __shfl_xor(ax, o, 64); ay += __shfl_xor(ay, o, 64); }
    if (lane == 0) vel[id] = make_float2(ax, ay);
  }
}

// euler_functor (.cu:509-541): pos += make_real4(vel * dt)
__global__ void __launch_bounds__(256) k_q2d_update(float4 *__restrict__ pos, const float2 *__restrict__ vel, int N, float dt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  float4 p = pos[i];
  const float2 v = vel[i];
  p.x = fmaf(v.x, dt, p.x);
  p.y = fmaf(v.y, dt, p.y);
  pos[i] = p;
}

static int q2d_make_plans(BDHI2D *q) {
  if (int e = rocfft_setup_once()) return e;
  const size_t nx = q->grid.cellDim.x, ny = q->grid.cellDim.y, nkx = nx / 2 + 1;
  const size_t len[2] = {nx, ny};
  const size_t rstr[2] = {1, (size_t)q->nxpad}, cstr[2] = {1, nkx};
  rocfft_plan_description d = nullptr;
  UH_ROCFFT(rocfft_plan_description_create(&d));
  UH_ROCFFT(rocfft_plan_description_set_data_layout(d, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, nullptr,
                                                     nullptr, 2, rstr, q->planeReal, 2, cstr, q->planeCplx));
  UH_ROCFFT(rocfft_plan_create(&q->fwd, rocfft_placement_inplace, rocfft_transform_type_real_forward, rocfft_precision_single, 2, len,
                               2, d));
  UH_ROCFFT(rocfft_plan_description_destroy(d));
  UH_ROCFFT(rocfft_plan_description_create(&d));
  UH_ROCFFT(rocfft_plan_description_set_data_layout(d, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, nullptr,
                                                     nullptr, 2, cstr, q->planeCplx, 2, rstr, q->planeReal));
  UH_ROCFFT(rocfft_plan_create(&q->inv, rocfft_placement_inplace, rocfft_transform_type_real_inverse, rocfft_precision_single, 2, len,
                               2, d));
  UH_ROCFFT(rocfft_plan_description_destroy(d));
  size_t wf = 0, wi = 0;
  UH_ROCFFT(rocfft_plan_get_work_buffer_size(q->fwd, &wf));
  UH_ROCFFT(rocfft_plan_get_work_buffer_size(q->inv, &wi));
  const size_t w = std::max(wf, wi);
  UH_ROCFFT(rocfft_execution_info_create(&q->info));
  if (w) {
    if (int e = q->work.reserve(w)) return e;
    UH_ROCFFT(rocfft_execution_info_set_work_buffer(q->info, q->work.ptr, w));
  }
  return 0;
}

}  // namespace uammd_hip

using namespace uammd_hip;

extern "C" {

int uammd_bdhi2d_create(const uammd_bdhi2d_parameters *par, uammd_bdhi2d **out, int cells[2], int *support) {
  if (!par || !out || (par->kernel != UAMMD_BDHI2D_TRUE2D && par->kernel != UAMMD_BDHI2D_QUASI2D)) {
    set_last_error("uammd_bdhi2d_create: bad arguments");
    return -1;
  }
  if (par->boxSize[0] == 0.0f && par->boxSize[1] == 0.0f) { set_last_error("Invalid box"); return -2; }              // .cu:46-52
  if (!(par->hydrodynamicRadius > 0)) { set_last_error("Invalid hydrodynamic radius"); return -2; }                  // .cu:53-57
  if (!(par->viscosity > 0) || !(par->boxSize[0] > 0) || !(par->boxSize[1] > 0)) { set_last_error("uammd_bdhi2d_create: bad arguments"); return -1; }
  BDHI2D *q = new (std::nothrow) BDHI2D();
  if (!q) { set_last_error("uammd_bdhi2d_create: out of host memory"); return -3; }
  q->par = *par;
  const float a = par->hydrodynamicRadius;
  int cd[2] = {par->cells[0], par->cells[1]};
  if (cd[0] <= 0) {  // initializeGrid, .cu:61-73
    const double h = a * 0.8;
    const float hr = (float)h;
    cd[0] = next_fft_wise_axis((int)(par->boxSize[0] / hr));
    cd[1] = next_fft_wise_axis((int)(par->boxSize[1] / hr));
  }
  const float L3[3] = {par->boxSize[0], par->boxSize[1], 0.0f};
  const int per[3] = {1, 1, 0};
  q->grid = make_grid(make_box<float>(L3, per), make_int3(cd[0], cd[1], 1));
  int s = ((int)(3.0 * a * cd[0] / par->boxSize[0]) + 1) * 2 + 1;  // initializeInterpolationKernel, .cu:75-88
  if (s > cd[0]) s = cd[0];
  if (s > kMaxSupport || s > cd[1]) {
    set_last_error("uammd_bdhi2d_create: window support %d is larger than the grid or than the %d nodes per axis one wave evaluates", s,
                   kMaxSupport);
    delete q;
    return -2;
  }
  const double width = par->kernel == UAMMD_BDHI2D_TRUE2D ? (double)(float)pow(a * 0.66556976637237890625, 2)
                                                          : (double)(float)pow(a / sqrt(M_PI), 2);
  const float w = (float)width;
  uammd_ibm_kernel k{};
  k.kind = UAMMD_IBM_KERNEL_GAUSS2D;
  k.support[0] = k.support[1] = s;
  k.support[2] = 1;
  k.prefactor = (float)sqrt(1.0 / (2.0 * M_PI * w));
  k.tau = (float)(-1.0 / (2.0 * w));
  k.rmax = INFINITY;
  q->kern = to_dev(k);
  k.prefactor = (float)(-sqrt(1.0 / (2.0 * M_PI * w * w)));
  k.kind = UAMMD_IBM_KERNEL_GAUSS2D_DRIFT_X;
  q->kernDriftX = to_dev(k);
  k.kind = UAMMD_IBM_KERNEL_GAUSS2D_DRIFT_Y;
  q->kernDriftY = to_dev(k);
  q->nxpad = 2 * (cd[0] / 2 + 1);
  q->planeReal = (size_t)q->nxpad * cd[1];
  q->planeCplx = (size_t)(cd[0] / 2 + 1) * cd[1];
  int e = q->gridBuf.reserve(sizeof(float) * 2 * q->planeReal);
  if (!e) e = q2d_make_plans(q);
  if (e) { delete q; return e; }
  if (cells) { cells[0] = cd[0]; cells[1] = cd[1]; }
  if (support) *support = s;
  *out = reinterpret_cast<uammd_bdhi2d *>(q);
  return 0;
}

int uammd_bdhi2d_destroy(uammd_bdhi2d *h) {
  delete reinterpret_cast<BDHI2D *>(h);
  return 0;
}

int uammd_bdhi2d_velocities(uammd_bdhi2d *h, const float *d_pos, const float *d_force, int N, float *d_vel, void *stream) {
  if (!h) { set_last_error("uammd_bdhi2d_velocities: null argument"); return -1; }
  if (N <= 0) return 0;
  if (!d_pos || !d_vel) { set_last_error("uammd_bdhi2d_velocities: null argument"); return -1; }
  BDHI2D *q = reinterpret_cast<BDHI2D *>(h);
  hipStream_t st = (hipStream_t)stream;
  const float T = q->par.temperature;
  const bool drift = q->par.kernel == UAMMD_BDHI2D_QUASI2D && T > 0;  // hasThermalDrift() and temperature > 0
  const bool deterministic = d_force != nullptr || drift;
  float *g = (float *)q->gridBuf.ptr;
  const dim3 gp((N + 3) / 4), bp(256);
  const FastDiv dsx = make_fastdiv(q->kern.support.x);
  const int nx = q->grid.cellDim.x, ny = q->grid.cellDim.y;
  UH_ROCFFT(rocfft_execution_info_set_stream(q->info, (void *)st));
  void *bufs[1] = {g};
  if (deterministic) {
    UH_CHECK(hipMemsetAsync(g, 0, sizeof(float) * 2 * q->planeReal, st));
    if (drift) {  // spreadThermalDrift, .cu:234-257
      hipLaunchKernelGGL((k_q2d_ibm<true, kKernelGauss2DDriftX>), gp, bp, 0, st, (const float4 *)d_pos, (const float4 *)nullptr, -T, 0.0f, g, q->planeReal,
                         (float2 *)nullptr, N, q->grid, q->nxpad, q->kernDriftX, dsx);
      hipLaunchKernelGGL((k_q2d_ibm<true, kKernelGauss2DDriftY>), gp, bp, 0, st, (const float4 *)d_pos, (const float4 *)nullptr, 0.0f, -T, g, q->planeReal,
                         (float2 *)nullptr, N, q->grid, q->nxpad, q->kernDriftY, dsx);
    }
    if (d_force)  // spreadParticleForces, .cu:268-283
      hipLaunchKernelGGL((k_q2d_ibm<true, kKernelGauss2D>), gp, bp, 0, st, (const float4 *)d_pos, (const float4 *)d_force, 0.0f, 0.0f, g, q->planeReal,
                         (float2 *)nullptr, N, q->grid, q->nxpad, q->kern, dsx);
    UH_ROCFFT(rocfft_execute(q->fwd, bufs, nullptr, q->info));
  }
  if (!deterministic && !(T > 0)) {  // nothing moves the particles
    UH_CHECK(hipMemsetAsync(d_vel, 0, sizeof(float) * 2 * (size_t)N, st));
    return 0;
  }
  float noisePrefactor = 0.0f;
  if (T > 0) {  // addStochastichTermFourier, .cu:450-469
    q->counter++;
    noisePrefactor = (float)sqrt(2.0 * T / (q->par.viscosity * q->par.dt * q->par.boxSize[0] * q->par.boxSize[1]));
  }
  const int total = (int)q->planeCplx;
  hipLaunchKernelGGL(k_q2d_kspace, dim3((total + 255) / 256), dim3(256), 0, st, (float2 *)g, (float2 *)g + q->planeCplx, nx, ny,
                     q->par.boxSize[0], q->par.boxSize[1], q->par.kernel, q->par.hydrodynamicRadius, q->par.viscosity, deterministic,
                     noisePrefactor, q->par.seed, q->counter);
  UH_ROCFFT(rocfft_execute(q->inv, bufs, nullptr, q->info));
  hipLaunchKernelGGL((k_q2d_ibm<false, kKernelGauss2D>), gp, bp, 0, st, (const float4 *)d_pos, (const float4 *)nullptr, 0.0f, 0.0f, g, q->planeReal,
                     (float2 *)d_vel, N, q->grid, q->nxpad, q->kern, dsx);
  UH_CHECK(hipGetLastError());
  return 0;
}

int uammd_bdhi2d_update_positions(float *d_pos, const float *d_vel, int N, float dt, void *stream) {
  if (N <= 0) return 0;
  if (!d_pos || !d_vel) { set_last_error("uammd_bdhi2d_update_positions: null argument"); return -1; }
  hipLaunchKernelGGL(k_q2d_update, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, (float4 *)d_pos, (const float2 *)d_vel, N, dt);
  UH_CHECK(hipGetLastError());
  return 0;
}

int uammd_bdhi2d_get_counter(uammd_bdhi2d *h, unsigned int *counter) {
  if (!h || !counter) { set_last_error("uammd_bdhi2d_get_counter: null argument"); return -1; }
  *counter = reinterpret_cast<BDHI2D *>(h)->counter;
  return 0;
}

}  // extern "C"
