// DOUBLE_PRECISION build of path B — the reference makes `real` a build switch (global/defines.h:9-11,33-44) and compiles every accuracy
// assertion it ships with it (test/CMakeLists.txt:9, test/BDHI/FCM/Makefile:9): Hasimoto self mobility to 1e-8 (test/BDHI/FCM/fcm_test.cu:
// 85-144), Peskin spreading / random-field interpolation to 1e-10 (test/misc/ibm/test_ibm_regular.cu:113-136,240-274), dense SPD Lanczos
// to 1e-7 (test/misc/lanczos/test_lanczos.cu:236-269), PSE self mobility (test/BDHI/PSE/pse_test.cu:64-117).  This file is that build for
// gfx950: the `_f64` entry points of include/uammd_hip.h, so that those known answers run ON THE GPU at the reference's own tolerances
// (tests/test_gpu_f64.py).
//
// What is in it (same algorithms as the single-precision library's layout-generic kernels, `real` = double; MI355X runs f64 vector
// arithmetic at half the f32 rate and has hardware f64 atomics in L2):
//   IBM spread / gather      misc/IBM.cu:10-65 (stencil, support shift, weights), :83-147 (particles2GridD), :164-235 (grid2ParticlesDTPP)
//                            one wave per particle, the 3 x support weights in LDS, global_atomic_add_f64 / shuffle reduction
//   FCM_impl                 Integrator/BDHI/FCM/FCM_impl.cuh:293-304,375-397,399-411,544-581,652-693 (spread, rocFFT R2C in double,
//                            forceFourier2Vel, C2R, gather); deterministic part (every double-precision test of the reference is T = 0 or
//                            statistical)
//   PSE far field            Integrator/BDHI/PSE/FarField.cuh:85-158,605-654 (window, sinc^2 Hasimoto-split greens function, projection)
//   PSE near field           Integrator/BDHI/PSE/NearField.cuh:65-99,120-196 over all pairs (minimum image; rcut <= L/2 is the reference's own
//                            precondition, NearField.cuh:69-78): the double-precision tests hold one particle or a handful
//   lanczos::Solver          lanczos.hip is a template over `real` (uammd_lanczos_*_f64 live there)
// The tuned single-precision hot path (tile-owned MFMA spreading, the in-LDS FFT with the fused operator, the f16-MFMA pair prefilter) is
// single precision by construction and stays that way: `real` = float is UAMMD's default and the benchmarked configuration.
#include "celllist.hpp"
#include "ibm.hpp"
#include "saru.hpp"

#include <rocfft/rocfft.h>

#include <algorithm>
#include <cmath>
#include <mutex>
#include <string>
#include <vector>

namespace uammd_hip {

int rocfft_setup_once();  // fcm.hip

struct Kern64 {
  int kind;
  int3 support;
  double prefactor, tau, rmax, invhx, invhy, invhz;
};

UH_D double phi64_peskin3(double invh, double rr) {  // misc/IBM_kernels.cuh:120-136
  const double r = fabs(rr) * invh;
  if (r < 0.5) return invh * (1 / 3.0) * (1.0 + sqrt(fma(-3.0 * r, r, 1.0)));
  if (r < 1.5) {
    const double omr = 1.0 - r;
    return invh * (1 / 6.0) * (fma(-3.0, r, 5.0) - sqrt(fma(-3.0 * omr, omr, 1.0)));
  }
  return 0.0;
}
UH_D double phi64_peskin4(double invh, double rr) {  // misc/IBM_kernels.cuh:145-158
  const double r = fabs(rr) * invh;
  if (r < 1.0) return invh * 0.125 * (fma(-2.0, r, 3.0) + sqrt(fma(4.0 * r, (1.0 - r), 1.0)));
  if (r < 2.0) return invh * 0.125 * (fma(-2.0, r, 5.0) - sqrt(fma(-(4.0 * r), r, fma(12.0, r, -7.0))));
  return 0.0;
}
UH_D double phi64_axis(const Kern64 &k, int axis, double r) {
  switch (k.kind) {
    case kKernelGaussian: return (r >= k.rmax) ? 0.0 : k.prefactor * exp(k.tau * r * r);  // FCM_kernels.cuh:55-57
    case kKernelPeskin3: return phi64_peskin3(axis == 0 ? k.invhx : (axis == 1 ? k.invhy : k.invhz), r);
    case kKernelPeskin4: return phi64_peskin4(axis == 0 ? k.invhx : (axis == 1 ? k.invhy : k.invhz), r);
    // BDHI2D_ns::Gaussian / GaussianThermalDrift<dir> (Integrator/Hydro/BDHI_quasi2D.cuh:112-153); phiZ = 1 (k_ibm64)
    case kKernelGauss2D: return k.prefactor * exp(k.tau * r * r);
    case kKernelGauss2DDriftX: return k.prefactor * exp(k.tau * r * r) * (axis == 0 ? r : 1.0);
    case kKernelGauss2DDriftY: return k.prefactor * exp(k.tau * r * r) * (axis == 1 ? r : 1.0);
    default: return 1.0;  // the constant window of the reference's test (test_ibm_regular.cu:11-14)
  }
}

UH_D int3 support_shift64(const GridT<double> &g, real3d pos, int3 celli, int3 support) {  // IBM.cu:10-31
  int3 P = make_int3(support.x / 2, support.y / 2, support.z / 2);
  real3d d = g.distanceToCellCenter(pos, make_int3(celli.x - P.x, celli.y - P.y, celli.z - P.z));
  d.x = fabs(d.x); d.y = fabs(d.y); d.z = fabs(d.z);
  if (g.cellSize.x > 0 && d.x > (double)support.x * g.cellSize.x / 2.0) P.x -= 1;
  if (g.cellSize.y > 0 && d.y > (double)support.y * g.cellSize.y / 2.0) P.y -= 1;
  if (g.cellSize.z > 0 && d.z > (double)support.z * g.cellSize.z / 2.0) P.z -= 1;
  return P;
}

// One wave per particle (four per workgroup).  The 3 x support 1-D weights are evaluated by the first lanes and kept in LDS
// (fillSharedWeights, IBM.cu:33-65); lanes then walk the support^3 nodes, x fastest.  Component c of node n lives at
// grid[n * nodeStride + c * compStride]: interleaved user grids (nodeStride = NCOMP, compStride = 1, the reference's real3 grid) or the
// solver's planar grids (nodeStride = 1, compStride = one component grid).  Spread ADDS into the grid, gather ADDS into qout unless
// `overwrite`.
constexpr int kW64 = 3 * kMaxSupport;
template <int NCOMP, bool SPREAD>
__global__ void __launch_bounds__(256) k_ibm64(const double *__restrict__ pos, int posStride, const double *__restrict__ qin, int qStride,
                                                double *__restrict__ qout, double *__restrict__ grid, int N, GridT<double> g, int nxStride,
                                                size_t nodeStride, size_t compStride, Kern64 kern, FastDiv dsx, FastDiv dsxy, bool is2D,
                                                bool overwrite) {
  __shared__ double wsh[4][kW64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int id = blockIdx.x * 4 + wave;
  if (id >= N) return;  // whole wave (no workgroup barrier below)
  double *w = wsh[wave];
  const real3d pi{pos[(size_t)posStride * id], pos[(size_t)posStride * id + 1], pos[(size_t)posStride * id + 2]};
  const int3 celli = g.getCell(pi);
  int3 support = kern.support;
  int3 P = support_shift64(g, pi, celli, support);
  if (is2D) { P.z = 0; support.z = 1; }
  const int sx = support.x, sy = support.y, sz = support.z;
  for (int t = lane; t < sx + sy + sz; t += 64) {
    double v = 0.0;
    if (t < sx) {
      const int cx = g.pbc_x(celli.x + t - P.x);
      if (cx >= 0) v = phi64_axis(kern, 0, g.distanceToCellCenter(pi, make_int3(cx, celli.y, celli.z)).x);
    } else if (t < sx + sy) {
      const int cy = g.pbc_y(celli.y + (t - sx) - P.y);
      if (cy >= 0) v = phi64_axis(kern, 1, g.distanceToCellCenter(pi, make_int3(celli.x, cy, celli.z)).y);
    } else {
      const int cz = g.pbc_z(celli.z + (t - sx - sy) - P.z);
      if (cz >= 0) v = phi64_axis(kern, 2, g.distanceToCellCenter(pi, make_int3(celli.x, celli.y, cz)).z);
      if (is2D && (kern.kind == kKernelPeskin3 || kern.kind == kKernelPeskin4 || kern.kind >= kKernelGauss2D)) v = 1.0;  // test_ibm_regular.cu:83-85
    }
    w[t] = v;
  }
  __builtin_amdgcn_wave_barrier();
  double v[NCOMP], acc[NCOMP];
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) {
    v[c] = SPREAD ? qin[(size_t)qStride * id + c] : 0.0;
    acc[c] = 0.0;
  }
  const double dV = g.cellVolume;
  const int nn = sx * sy * sz;
  for (int i = lane; i < nn; i += 64) {
    const uint kk = dsxy.div((uint)i);
    const uint rem = (uint)i - kk * (uint)(sx * sy);
    const uint jj = dsx.div(rem);
    const uint ii = rem - jj * (uint)sx;
    const int cx = g.pbc_x(celli.x + (int)ii - P.x);
    const int cy = g.pbc_y(celli.y + (int)jj - P.y);
    const int cz = is2D ? 0 : g.pbc_z(celli.z + (int)kk - P.z);
    if (cx < 0 || cy < 0 || cz < 0 || cx >= g.cellDim.x || cy >= g.cellDim.y || cz >= g.cellDim.z) continue;
    const size_t node = (size_t)cx + (size_t)nxStride * ((size_t)cy + (size_t)g.cellDim.y * (size_t)cz);
    if (SPREAD) {
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) unsafeAtomicAdd(&grid[node * nodeStride + (size_t)c * compStride], v[c] * w[ii] * w[sx + jj] * w[sx + sy + kk]);
    } else {
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) acc[c] = fma(dV, grid[node * nodeStride + (size_t)c * compStride] * w[ii] * w[sx + jj] * w[sx + sy + kk], acc[c]);
    }
  }
  if (!SPREAD) {
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      double t = acc[c];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
      if (lane == 0) {
        if (overwrite) qout[(size_t)NCOMP * id + c] = t;
        else qout[(size_t)NCOMP * id + c] += t;
      }
    }
  }
}

static Kern64 to_dev64(const uammd_ibm_kernel_f64 &k) {
  return Kern64{k.kind, make_int3(k.support[0], k.support[1], k.support[2]), k.prefactor, k.tau, k.rmax, k.invh[0], k.invh[1], k.invh[2]};
}
static int check_kernel64(const char *fn, const uammd_ibm_kernel_f64 *k) {
  if (!k) { set_last_error("%s: null kernel", fn); return -1; }
  for (int a = 0; a < 3; ++a)
    if (k->support[a] < 1 || k->support[a] > kMaxSupport) { set_last_error("%s: kernel support %d outside [1, %d]", fn, k->support[a], kMaxSupport); return -1; }
  if (k->kind != kKernelGaussian && k->kind != kKernelPeskin3 && k->kind != kKernelPeskin4 && k->kind != kKernelConstant) {
    set_last_error("%s: the double-precision build has the Gaussian, Peskin 3 / 4 point and constant windows (kind %d asked)", fn, k->kind);
    return -1;
  }
  return 0;
}

// ---- Fourier space, FCM (FCM/utils.cuh:27-74, FCM_impl.cuh:375-397) and PSE far field (FarField.cuh:53-158) ------------------------------
struct Pse64 { double rh, split, eta, shear; bool on; };

// ---- Fourier noise (FCM_impl.cuh:437-512, FarField.cuh:235-308) in GATHER form, as the single-precision operator does it (fcm.hip:
// fcm_kspace_node): a node adds its own draw and, on the kx = 0 / Nyquist planes, the conjugate of its partner's regenerated draw —
// no node is written twice.  The draws are Saru's single-precision Gaussians promoted to double, which is what the reference's
// DOUBLE_PRECISION build does (FCM/utils.cuh:117-131: make_real2(saru.gf(0, sc))). ----
struct C3d { double xr, xi, yr, yi, zr, zi; };
UH_D bool is_nyquist64(int3 c, int3 n) {  // FCM/utils.cuh:133-167
  const bool X = (c.x == n.x - c.x) && (n.x % 2 == 0), Y = (c.y == n.y - c.y) && (n.y % 2 == 0), Z = (c.z == n.z - c.z) && (n.z % 2 == 0);
  return (X && c.y == 0 && c.z == 0) || (X && Y && c.z == 0) || (c.x == 0 && Y && c.z == 0) || (X && c.y == 0 && Z) ||
         (c.x == 0 && c.y == 0 && Z) || (c.x == 0 && Y && Z) || (X && Y && Z);
}
UH_D bool noise_skipped64(int id, int3 c, int3 n) {  // FCM_impl.cuh:456-463
  return id == 0 || (c.x == 0 && c.y == 0 && 2 * c.z >= n.z + 1) || (c.x == 0 && 2 * c.y >= n.y + 1);
}
UH_D C3d draw_noise64(double prefactor, uint id, uint seed1, uint seed2, bool nyquist) {
  Saru rng(id, seed1, seed2);
  const float sc = (float)(0.707106781186547 * prefactor);
  const float2 a = rng.gf(0.0f, sc), b = rng.gf(0.0f, sc), c = rng.gf(0.0f, sc);
  C3d n{a.x, a.y, b.x, b.y, c.x, c.y};
  if (nyquist) {
    const double q = 1.41421356237310;
    n.xr *= q; n.xi = 0.0; n.yr *= q; n.yi = 0.0; n.zr *= q; n.zi = 0.0;
  }
  return n;
}
UH_D C3d project64(const real3d &dk, double invk2, const C3d &f) {   // (I - dk dk / k^2) f, FCM/utils.cuh:70-74 and FarField.cuh:53-73
  const double sr = dot3(real3d{f.xr, f.yr, f.zr}, dk) * invk2, si = dot3(real3d{f.xi, f.yi, f.zi}, dk) * invk2;
  return C3d{fma(-dk.x, sr, f.xr), fma(-dk.x, si, f.xi), fma(-dk.y, sr, f.yr), fma(-dk.y, si, f.yi), fma(-dk.z, sr, f.zr), fma(-dk.z, si, f.zi)};
}

// One thread per Fourier node, in place: forceFourier2Vel (FCM_impl.cuh:375-397 / FarField.cuh:137-158) when the grid holds transformed
// forces, plus the noise when noisePrefactor != 0.
__global__ void __launch_bounds__(256) k_kspace64(double2 *__restrict__ g0, size_t planeC, int3 nk, real3d L, double viscosity, Pse64 pse,
                                                  bool haveForce, double noisePrefactor, uint seed1, uint seed2) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int nkx = nk.x / 2 + 1;
  const size_t total = (size_t)nk.z * nk.y * nkx;
  if (t >= total) return;
  const int3 cell = make_int3((int)(t % nkx), (int)((t / nkx) % nk.y), (int)(t / ((size_t)nkx * nk.y)));
  int3 ik = cell;  // indexToWaveNumber
  ik.x -= nk.x * (ik.x >= nkx);
  ik.y -= nk.y * (ik.y >= (nk.y / 2 + 1));
  ik.z -= nk.z * (ik.z >= (nk.z / 2 + 1));
  double2 *g1 = g0 + planeC, *g2 = g0 + 2 * planeC;
  const double2 zero = make_double2(0.0, 0.0);
  if (t == 0) { g0[0] = zero; g1[0] = zero; g2[0] = zero; return; }
  const double twopi = 2.0 * 3.14159265358979323846;
  const real3d k{(twopi / L.x) * (double)ik.x, (twopi / L.y) * (double)ik.y, (twopi / L.z) * (double)ik.z};
  const double k2 = dot3(k, k);
  C3d v{0, 0, 0, 0, 0, 0};
  double Bsq;       // sqrt of the operator's scalar, for the noise
  real3d pk;        // the vector the projector is built from
  double invp2;
  bool noiseAfterProjection;   // PSE: sqrt(B) multiplies the PROJECTED draw (FarField.cuh:286-290); FCM: the draw, then the projection
  if (pse.on) {
    real3d kE = k;
    kE.y = fma(-pse.shear, k.x, k.y);
    const double kE2 = dot3(kE, kE);
    const double kmod = sqrt(kE2), invk2 = 1.0 / kE2;
    const double sink = sin(kmod * pse.rh);
    const double kEw = kE2 / (4.0 * pse.split * pse.split), kNU = k2 / (4.0 * pse.split * pse.split);
    const double tau = fma(pse.eta, kNU, -kEw);
    const double hashimoto = (1.0 + kEw) * exp(tau) / kE2;
    double B = sink * sink * invk2 * hashimoto / (viscosity * pse.rh * pse.rh);
    B /= (double)nk.x * (double)nk.y * (double)nk.z;
    if (haveForce) {
      const double2 fx = g0[t], fy = g1[t], fz = g2[t];
      v = project64(kE, invk2, C3d{fx.x * B, fx.y * B, fy.x * B, fy.y * B, fz.x * B, fz.y * B});
    }
    Bsq = sqrt(B); pk = kE; invp2 = invk2; noiseAfterProjection = true;
  } else {
    // getGradientFourier: unpaired (Nyquist) components are zero in the projector (FCM/utils.cuh:41-51)
    const real3d dk{ik.x == (nk.x - ik.x) ? 0.0 : k.x, ik.y == (nk.y - ik.y) ? 0.0 : k.y, ik.z == (nk.z - ik.z) ? 0.0 : k.z};
    const double invk2 = 1.0 / k2;
    if (haveForce) {
      const double2 fx = g0[t], fy = g1[t], fz = g2[t];
      const double B = 1.0 / (viscosity * k2);
      const double sc = B / ((double)nk.x * (double)nk.y * (double)nk.z);  // the FFT normalisation lives here (FCM_impl.cuh:392)
      const C3d pr = project64(dk, invk2, C3d{fx.x, fx.y, fy.x, fy.y, fz.x, fz.y});
      v = C3d{pr.xr * sc, pr.xi * sc, pr.yr * sc, pr.yi * sc, pr.zr * sc, pr.zi * sc};
    }
    Bsq = sqrt(1.0 / (k2 * viscosity)); pk = dk; invp2 = invk2; noiseAfterProjection = false;
  }
  if (noisePrefactor != 0.0) {
    const int id = (int)t;
    const bool own = !noise_skipped64(id, cell, nk);
    int idp = -1;   // conjugate partner: only stored (and only written by the reference) on the kx == 0 / kx == nx - kx planes
    if (cell.x == 0 || cell.x == nk.x - cell.x) {
      const int3 pc = make_int3(cell.x, (cell.y > 0) * (nk.y - cell.y), (cell.z > 0) * (nk.z - cell.z));
      const int cand = pc.x + nkx * (pc.y + pc.z * nk.y);
      if (cand != id && !noise_skipped64(cand, pc, nk) && !is_nyquist64(pc, nk)) idp = cand;
    }
    auto shaped = [&](C3d f) -> C3d {
      if (noiseAfterProjection) {
        const C3d z = project64(pk, invp2, f);
        return C3d{z.xr * Bsq, z.xi * Bsq, z.yr * Bsq, z.yi * Bsq, z.zr * Bsq, z.zi * Bsq};
      }
      return project64(pk, invp2, C3d{f.xr * Bsq, f.xi * Bsq, f.yr * Bsq, f.yi * Bsq, f.zr * Bsq, f.zi * Bsq});
    };
    C3d mine{0, 0, 0, 0, 0, 0}, theirs = mine;
    if (own) mine = shaped(draw_noise64(noisePrefactor, (uint)id, seed1, seed2, is_nyquist64(cell, nk)));
    if (idp >= 0) {
      C3d f = draw_noise64(noisePrefactor, (uint)idp, seed1, seed2, false);
      f.xi = -f.xi; f.yi = -f.yi; f.zi = -f.zi;
      theirs = shaped(f);
    }
    const C3d first = (idp >= 0 && idp < id) ? theirs : mine, second = (idp >= 0 && idp < id) ? mine : theirs;   // (a sequential sweep's order)
    v.xr += first.xr; v.xi += first.xi; v.yr += first.yr; v.yi += first.yi; v.zr += first.zr; v.zi += first.zi;
    v.xr += second.xr; v.xi += second.xi; v.yr += second.yr; v.yi += second.yi; v.zr += second.zr; v.zi += second.zi;
  }
  g0[t] = make_double2(v.xr, v.xi);
  g1[t] = make_double2(v.yr, v.yi);
  g2[t] = make_double2(v.zr, v.zi);
}

struct FCM64 {
  GridT<double> grid;
  Kern64 kern;
  double L[3], viscosity;
  int nxpad = 0;
  size_t planeReal = 0, planeCplx = 0;
  DeviceBuffer gridBuf, work;
  Pse64 pse{0, 0, 0, 0, false};
  bool accumulate = false;
  rocfft_plan fwd = nullptr, inv = nullptr;
  rocfft_execution_info info = nullptr;
  ~FCM64() {
    if (fwd) rocfft_plan_destroy(fwd);
    if (inv) rocfft_plan_destroy(inv);
    if (info) rocfft_execution_info_destroy(info);
  }
};

#define UH_ROCFFT64(expr)                                                                      \
  do {                                                                                         \
    const rocfft_status s_ = (expr);                                                           \
    if (s_ != rocfft_status_success) {                                                         \
      set_last_error("%s failed with rocfft status %d (%s:%d)", #expr, (int)s_, __FILE__, __LINE__); \
      return -10;                                                                              \
    }                                                                                          \
  } while (0)

static int fcm64_plans(FCM64 *f) {
  rocfft_setup_once();
  const size_t nx = (size_t)f->grid.cellDim.x, ny = (size_t)f->grid.cellDim.y, nz = (size_t)f->grid.cellDim.z, nkx = nx / 2 + 1;
  const size_t lengths[3] = {nx, ny, nz};
  const size_t rstr[3] = {1, (size_t)f->nxpad, (size_t)f->nxpad * ny}, cstr[3] = {1, nkx, nkx * ny};
  rocfft_plan_description d = nullptr;
  UH_ROCFFT64(rocfft_plan_description_create(&d));
  UH_ROCFFT64(rocfft_plan_description_set_data_layout(d, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, nullptr, nullptr, 3,
                                                      rstr, f->planeReal, 3, cstr, f->planeCplx));
  UH_ROCFFT64(rocfft_plan_create(&f->fwd, rocfft_placement_inplace, rocfft_transform_type_real_forward, rocfft_precision_double, 3, lengths, 3, d));
  UH_ROCFFT64(rocfft_plan_description_destroy(d));
  UH_ROCFFT64(rocfft_plan_description_create(&d));
  UH_ROCFFT64(rocfft_plan_description_set_data_layout(d, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, nullptr, nullptr, 3,
                                                      cstr, f->planeCplx, 3, rstr, f->planeReal));
  UH_ROCFFT64(rocfft_plan_create(&f->inv, rocfft_placement_inplace, rocfft_transform_type_real_inverse, rocfft_precision_double, 3, lengths, 3, d));
  UH_ROCFFT64(rocfft_plan_description_destroy(d));
  size_t wf = 0, wi = 0;
  UH_ROCFFT64(rocfft_plan_get_work_buffer_size(f->fwd, &wf));
  UH_ROCFFT64(rocfft_plan_get_work_buffer_size(f->inv, &wi));
  const size_t wb = std::max(wf, wi);
  UH_ROCFFT64(rocfft_execution_info_create(&f->info));
  if (wb) {
    if (int e = f->work.reserve(wb)) return e;
    UH_ROCFFT64(rocfft_execution_info_set_work_buffer(f->info, f->work.ptr, wb));
  }
  return 0;
}

// RPYPSE_near::FandG (pse.hip, host, double)
void rpy_near_FandG(double r, double rh, double psi, double rcut, double *F, double *G);

// NearField Mdot over all pairs with the minimum image (NearField.cuh:134-185): Mv[i] (+)= sum_j F(r) v_j + (G - F) (r.v_j) r / r^2,
// F and G by linear interpolation of the tabulated closed form (TabulatedFunction.cuh:63-75,148-157)
__global__ void __launch_bounds__(128) k_pse_near64(const double *__restrict__ pos, const double *__restrict__ v, int vstride, int N, real3d L,
                                                     double rcut, const double2 *__restrict__ table, int Ntable, double *__restrict__ Mv,
                                                     bool overwrite) {
  const int i = blockIdx.x * 128 + threadIdx.x;
  if (i >= N) return;
  const real3d pi{pos[4 * (size_t)i], pos[4 * (size_t)i + 1], pos[4 * (size_t)i + 2]};
  const double interval = 1.0 / rcut, dr = 1.0 / (double)Ntable, rcut2 = rcut * rcut;
  double tx = 0, ty = 0, tz = 0;
  for (int j = 0; j < N; ++j) {
    real3d rij{pos[4 * (size_t)j] - pi.x, pos[4 * (size_t)j + 1] - pi.y, pos[4 * (size_t)j + 2] - pi.z};
    rij.y = fma(-L.y, round(rij.y / L.y), rij.y);
    rij.z = fma(-L.z, round(rij.z / L.z), rij.z);
    rij.x = fma(-L.x, round(rij.x / L.x), rij.x);
    const double r2 = dot3(rij, rij);
    if (r2 >= rcut2) continue;
    const double rs = sqrt(r2);
    double f = 0, g = 0;
    if (!(rs >= rcut)) {
      const double r = rs * interval;
      if (r <= 0.0) { f = table[0].x; g = table[0].y; }
      else {
        const int t = (int)(r * (double)Ntable);
        const double r0 = (double)t * dr;
        const double2 v0 = table[t], v1 = table[t + 1];
        const double w = (r - r0) * (double)Ntable;
        f = fma(w, v1.x, fma(-w, v0.x, v0.x));
        g = fma(w, v1.y, fma(-w, v0.y, v0.y));
      }
    }
    const double vx = v[(size_t)vstride * j], vy = v[(size_t)vstride * j + 1], vz = v[(size_t)vstride * j + 2];
    if (r2 == 0.0) { tx += f * vx; ty += f * vy; tz += f * vz; }
    else {
      const double gmfv = (g - f) * dot3(rij, real3d{vx, vy, vz}) * (1.0 / r2);
      tx += fma(gmfv, rij.x, f * vx);
      ty += fma(gmfv, rij.y, f * vy);
      tz += fma(gmfv, rij.z, f * vz);
    }
  }
  double *o = Mv + 3 * (size_t)i;
  if (overwrite) { o[0] = tx; o[1] = ty; o[2] = tz; } else { o[0] += tx; o[1] += ty; o[2] += tz; }
}

struct PSENear64 {
  DeviceBuffer table, noise;
  int nPointsTable = 0;
  double rcut = 0, L[3] = {0, 0, 0};
};


// ---- BDHI::True2D / BDHI::Quasi2D with real = double (Integrator/Hydro/BDHI_quasi2D.cu:61-88, :179-541; the single-precision build and
// the account of the gather-form noise: quasi2d.hip).  Spread and gather are k_ibm64<2, .> on the two component planes. ----
UH_D double2 hydro_kernel64(int mode, double k2, double a) {   // BDHI2D_ns::True2D / Quasi2D::operator(), .cuh:88-92, :100-109
  if (mode == UAMMD_BDHI2D_TRUE2D) return make_double2(0.0, 1.0 / (k2 * k2));
  const double k = sqrt(k2);
  const double invk3 = 1.0 / (k2 * k);
  const double inv_sqrtpi = 0.564189583547756;
  const double kp = k * a * inv_sqrtpi;
  const double fk = 0.5 * invk3 * (erfc(kp) * (0.5 + kp * kp) * exp(kp * kp) - kp * inv_sqrtpi);
  const double gk = 0.5 * invk3 * erfc(kp) * exp(kp * kp);
  return make_double2(fk, gk);
}
UH_D double2 wave_number64(int ix, int iy, int nx, int ny, double Lx, double Ly) {  // cellToWaveNumber, .cu:314-320
  const double px = (2.0 * M_PI) / Lx, py = (2.0 * M_PI) / Ly;
  return make_double2((double)(ix - nx * (ix >= (nx / 2 + 1))) * px, (double)(iy - ny * (iy >= (ny / 2 + 1))) * py);
}
UH_D double2 project2d64(double2 k, double2 f, double fk, double gk) {  // projectFourier for one real 2-vector, .cu:324-343
  const double dperp = fma(f.y, -k.x, f.x * k.y);
  const double dpar = fma(f.y, k.y, f.x * k.x);
  return make_double2(fma(k.x * fk, dpar, k.y * gk * dperp), fma(k.y * fk, dpar, -k.x * gk * dperp));
}
struct Cplx2d { double xr, xi, yr, yi; };
// the noise term of an OWNER node (fourierBrownianNoise, .cu:368-432); the draws are Saru's single-precision Gaussians promoted to double
UH_D Cplx2d noise_factor2d64(int id, int ix, int iy, int nx, int ny, double Lx, double Ly, int mode, double a, double prefactor, uint seed,
                             uint step) {
  const bool isXnyquist = (ix == (nx - ix)) && (nx % 2 == 0);
  const bool isYnyquist = (iy == (ny - iy)) && (ny % 2 == 0);
  const bool isNyquist = (isYnyquist && ix == 0) || (isXnyquist && isYnyquist);
  Saru saru((uint)id, step, seed);
  const float sc = (float)(0.707106781186547 * prefactor);
  const float2 a1 = saru.gf(0.0f, sc), a2 = saru.gf(0.0f, sc);
  double2 n1 = make_double2(a1.x, a1.y), n2 = make_double2(a2.x, a2.y);
  if (isNyquist) {
    n1.x *= 1.41421356237310; n2.x *= 1.41421356237310;
    n1.y = 0.0; n2.y = 0.0;
  }
  const double2 k = wave_number64(ix, iy, nx, ny, Lx, Ly);
  const double k2 = fma(k.y, k.y, k.x * k.x);
  const double2 fg = hydro_kernel64(mode, k2, a);
  const double fs = sqrt(fg.x), gs = sqrt(fg.y);
  Cplx2d f;
  f.xr = fma(fs * n2.x, k.x, gs * n1.x * k.y); f.xi = fma(fs * n2.y, k.x, gs * n1.y * k.y);
  f.yr = fma(fs * n2.x, k.y, gs * n1.x * (-k.x)); f.yi = fma(fs * n2.y, k.y, gs * n1.y * (-k.x));
  return f;
}
// forceFourier2Vel + fourierBrownianNoise in gather form on the two component planes, complex[ny][nkx] each
__global__ void __launch_bounds__(256) k_q2d_kspace64(double2 *__restrict__ gx, double2 *__restrict__ gy, int nx, int ny, double Lx, double Ly,
                                                      int mode, double a, double viscosity, bool deterministic, double noisePrefactor,
                                                      uint seed, uint step) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  const int nkx = nx / 2 + 1;
  if (id >= ny * nkx) return;
  const int ix = id % nkx, iy = id / nkx;
  Cplx2d v{0, 0, 0, 0};
  if (id != 0) {
    if (deterministic) {
      const double2 k = wave_number64(ix, iy, nx, ny, Lx, Ly);
      const double k2 = fma(k.y, k.y, k.x * k.x);
      const double2 fg = hydro_kernel64(mode, k2, a);
      const double fk = fg.x / (viscosity * (double)(nx * ny)), gk = fg.y / (viscosity * (double)(nx * ny));
      const double2 fx = gx[id], fy = gy[id];
      const double2 vr = project2d64(k, make_double2(fx.x, fy.x), fk, gk), vi = project2d64(k, make_double2(fx.y, fy.y), fk, gk);
      v = Cplx2d{vr.x, vi.x, vr.y, vi.y};
    }
    if (noisePrefactor != 0.0) {
      const bool selfConjColumn = ix == 0 || ix == nx - ix;
      if (selfConjColumn && iy > ny - iy) {   // owned by the conjugate partner (ix, ny - iy): its factor, conjugated (.cu:422-428)
        const int jy = ny - iy;
        const Cplx2d f = noise_factor2d64(ix + nkx * jy, ix, jy, nx, ny, Lx, Ly, mode, a, noisePrefactor, seed, step);
        v.xr += f.xr; v.xi += -f.xi; v.yr += f.yr; v.yi += -f.yi;
      } else {
        const bool isXnyquist = (ix == (nx - ix)) && (nx % 2 == 0);
        if (isXnyquist && iy == 0) v = Cplx2d{0, 0, 0, 0};  // .cu:393-395: this node's deterministic part is wiped
        const Cplx2d f = noise_factor2d64(id, ix, iy, nx, ny, Lx, Ly, mode, a, noisePrefactor, seed, step);
        v.xr += f.xr; v.xi += f.xi; v.yr += f.yr; v.yi += f.yi;
      }
    }
  }
  gx[id] = make_double2(v.xr, v.xi);
  gy[id] = make_double2(v.yr, v.yi);
}
// euler_functor (.cu:509-541): pos += make_real4(vel * dt)
__global__ void __launch_bounds__(256) k_q2d_update64(double *__restrict__ pos, const double2 *__restrict__ vel, int N, double dt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const double2 v = vel[i];
  pos[4 * (size_t)i] = fma(v.x, dt, pos[4 * (size_t)i]);
  pos[4 * (size_t)i + 1] = fma(v.y, dt, pos[4 * (size_t)i + 1]);
}
struct BDHI2D64 {
  uammd_bdhi2d_parameters_f64 par{};
  GridT<double> grid{};
  Kern64 kern{}, kernDriftX{}, kernDriftY{};
  int nxpad = 0;
  size_t planeReal = 0, planeCplx = 0;
  DeviceBuffer gridBuf, work, drift;   // drift: the thermal drift's constant "quantities" {-T, 0} and {0, -T}
  rocfft_plan fwd = nullptr, inv = nullptr;
  rocfft_execution_info info = nullptr;
  unsigned int counter = 0;
  ~BDHI2D64() {
    if (fwd) rocfft_plan_destroy(fwd);
    if (inv) rocfft_plan_destroy(inv);
    if (info) rocfft_execution_info_destroy(info);
  }
};
int next_fft_wise_axis(int n);   // quasi2d.hip
static int q2d64_plans(BDHI2D64 *q) {
  if (int e = rocfft_setup_once()) return e;
  const size_t nx = q->grid.cellDim.x, ny = q->grid.cellDim.y, nkx = nx / 2 + 1;
  const size_t len[2] = {nx, ny};
  const size_t rstr[2] = {1, (size_t)q->nxpad}, cstr[2] = {1, nkx};
  rocfft_plan_description d = nullptr;
  UH_ROCFFT64(rocfft_plan_description_create(&d));
  UH_ROCFFT64(rocfft_plan_description_set_data_layout(d, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, nullptr, nullptr, 2, rstr,
                                                      q->planeReal, 2, cstr, q->planeCplx));
  UH_ROCFFT64(rocfft_plan_create(&q->fwd, rocfft_placement_inplace, rocfft_transform_type_real_forward, rocfft_precision_double, 2, len, 2, d));
  UH_ROCFFT64(rocfft_plan_description_destroy(d));
  UH_ROCFFT64(rocfft_plan_description_create(&d));
  UH_ROCFFT64(rocfft_plan_description_set_data_layout(d, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, nullptr, nullptr, 2, cstr,
                                                      q->planeCplx, 2, rstr, q->planeReal));
  UH_ROCFFT64(rocfft_plan_create(&q->inv, rocfft_placement_inplace, rocfft_transform_type_real_inverse, rocfft_precision_double, 2, len, 2, d));
  UH_ROCFFT64(rocfft_plan_description_destroy(d));
  size_t wf = 0, wi = 0;
  UH_ROCFFT64(rocfft_plan_get_work_buffer_size(q->fwd, &wf));
  UH_ROCFFT64(rocfft_plan_get_work_buffer_size(q->inv, &wi));
  const size_t w = std::max(wf, wi);
  UH_ROCFFT64(rocfft_execution_info_create(&q->info));
  if (w) {
    if (int e = q->work.reserve(w)) return e;
    UH_ROCFFT64(rocfft_execution_info_set_work_buffer(q->info, q->work.ptr, w));
  }
  return 0;
}


// ---- Poisson with real = double (Interactor/SpectralEwaldPoisson.cu:15-62 closed forms, :71-160 set-up, :222-329 near field, :332-360 and
// :410-559 far field; the single-precision build with its tile-owned spread and column gather: poisson.hip).  Spread and gather are
// k_ibm64 on planar grids; the near field runs over ALL pairs with the minimum image (the reference's own precondition: cut-off <= L/2,
// .cu:111-116) — the double-precision tests hold three charges or a thousand. ----
static double greens64(double r2, double gw, double split, double epsilon) {  // .cu:15-38
  double G = 0;
  if (r2 > gw * gw * gw * gw) {
    const double r = std::sqrt(r2);
    const double farw = std::sqrt(4 * gw * gw + 1 / (split * split));
    G = (1.0 / (4.0 * M_PI * epsilon * r) * (std::erf(r / (2 * gw)) - std::erf(r / farw)));
  } else {
    const double pi32 = std::pow(M_PI, 1.5);
    const double gw2 = gw * gw;
    const double invsp2 = 1.0 / (split * split);
    const double selfterm = 1.0 / (4 * pi32 * gw) - 1.0 / (2 * pi32 * std::sqrt(4 * gw2 + invsp2));
    const double r2term = 1.0 / (6.0 * pi32 * std::pow(4.0 * gw2 + invsp2, 1.5)) - 1.0 / (48.0 * pi32 * gw2 * gw);
    const double r4term = 1.0 / (640.0 * pi32 * gw2 * gw2 * gw) - 1.0 / (20.0 * pi32 * std::pow(4 * gw2 + invsp2, 2.5));
    G = 1.0 / epsilon * (selfterm + r2 * r2term + r2 * r2 * r4term);
  }
  return G;
}
static double greens_field64(double r, double gw, double split, double epsilon) {  // .cu:40-62
  const double r2 = r * r;
  const double gw2 = gw * gw;
  const double newgw = std::sqrt(gw2 + 1 / (4.0 * split * split));
  const double newgw2 = newgw * newgw;
  double fmod = 0;
  if (r2 > gw * gw * gw * gw) {
    const double invrterm = std::exp(-0.25 * r2 / newgw2) / std::sqrt(M_PI * newgw2) - std::exp(-0.25 * r2 / gw2) / std::sqrt(M_PI * gw2);
    const double invr2term = std::erf(0.5 * r / newgw) - std::erf(0.5 * r / gw);
    fmod += 1 / (4 * M_PI) * (invrterm / r - invr2term / r2);
  } else if (r2 > 0) {
    const double pi32 = std::pow(M_PI, 1.5);
    const double rterm = 1 / (24 * pi32) * (1.0 / (gw2 * gw) - 1 / (newgw2 * newgw));
    const double r3term = 1 / (160 * pi32) * (1.0 / (newgw2 * newgw2 * newgw) - 1.0 / (gw2 * gw2 * gw));
    fmod += r * rterm + r2 * r * r3term;
  }
  return fmod / epsilon;
}
struct Table64 {
  const double *table;
  int Nm1;
  double rmax, interval, dr;
};
// TabulatedFunction::operator() with LinearInterpolation (misc/TabulatedFunction.cuh:63-75, :148-157), rmin = 0
UH_D double table_get64(const Table64 &t, double rs) {
  const double r = rs * t.interval;
  if (rs >= t.rmax) return 0.0;
  if (r <= 0.0) return t.table[0];
  const int i = (int)(r * (double)t.Nm1);
  const double r0 = (double)i * t.dr;
  const double v0 = t.table[i], v1 = t.table[i + 1];
  const double w = (r - r0) * (double)t.Nm1;
  return fma(w, v1, fma(-w, v0, v0));
}
// chargeFourier2FieldAndPotential (.cu:433-476); planes: Ex, Ey, Ez, phi, each complex[nz][ny][nkx]
__global__ void __launch_bounds__(256) k_poisson_convolve64(const double2 *__restrict__ qk, double2 *__restrict__ planes, size_t planeCplx, int3 n,
                                                            real3d L, double epsilon) {
  const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int nkx = n.x / 2 + 1;
  if (id >= (size_t)nkx * n.y * n.z) return;
  const int cx = (int)(id % nkx), cy = (int)((id / nkx) % n.y), cz = (int)(id / ((size_t)nkx * n.y));
  const double2 zero = make_double2(0.0, 0.0);
  double2 ex = zero, ey = zero, ez = zero, ph = zero;
  const bool xn = (cx == n.x - cx) && (n.x % 2 == 0), yn = (cy == n.y - cy) && (n.y % 2 == 0), zn = (cz == n.z - cz) && (n.z % 2 == 0);
  const bool nyquist = (xn && cy == 0 && cz == 0) || (xn && yn && cz == 0) || (cx == 0 && yn && cz == 0) || (xn && cy == 0 && zn) ||
                       (cx == 0 && cy == 0 && zn) || (cx == 0 && yn && zn) || (xn && yn && zn);
  if (!(cx == 0 && cy == 0 && cz == 0) && !nyquist) {
    const double px = (2.0 * M_PI) / L.x, py = (2.0 * M_PI) / L.y, pz = (2.0 * M_PI) / L.z;
    double kx = (double)cx * px, ky = (double)cy * py, kz = (double)cz * pz;
    if (cx >= n.x / 2 + 1) kx -= (double)n.x * px;
    if (cy >= n.y / 2 + 1) ky -= (double)n.y * py;
    if (cz >= n.z / 2 + 1) kz -= (double)n.z * pz;
    const double k2 = fma(kz, kz, fma(ky, ky, kx * kx));
    const double2 fk = qk[id];
    const double B = 1.0 / (k2 * epsilon * ((double)n.x * (double)n.y * (double)n.z));
    ex = make_double2(kx * fk.y * B, -kx * fk.x * B);
    ey = make_double2(ky * fk.y * B, -ky * fk.x * B);
    ez = make_double2(kz * fk.y * B, -kz * fk.x * B);
    ph = make_double2(fk.x * B, fk.y * B);
  }
  planes[id] = ex;
  planes[planeCplx + id] = ey;
  planes[2 * planeCplx + id] = ez;
  planes[3 * planeCplx + id] = ph;
}
// UnZip2Real4 (.cu:529-559) after the gather: force += q E, energy += q phi, fieldPotential += (E, phi)
__global__ void __launch_bounds__(256) k_poisson_unzip64(const double *__restrict__ gathered, const double *__restrict__ charge, int N,
                                                         double *__restrict__ force, double *__restrict__ energy, double *__restrict__ fieldPotential) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const double ex = gathered[4 * (size_t)i], ey = gathered[4 * (size_t)i + 1], ez = gathered[4 * (size_t)i + 2], ph = gathered[4 * (size_t)i + 3];
  const double q = charge[i];
  if (force) { force[4 * (size_t)i] += q * ex; force[4 * (size_t)i + 1] += q * ey; force[4 * (size_t)i + 2] += q * ez; }
  if (energy) energy[i] += q * ph;
  if (fieldPotential) {
    fieldPotential[4 * (size_t)i] += ex; fieldPotential[4 * (size_t)i + 1] += ey; fieldPotential[4 * (size_t)i + 2] += ez; fieldPotential[4 * (size_t)i + 3] += ph;
  }
}
// NearField{Force,Energy,FieldPotential}Transverser (.cu:222-329) over all pairs, the self pair included.  MODE 0: force4 += (total, 0);
// 1: energy += total; 2: fieldPotential4 += (E, phi)
template <int MODE>
__global__ void __launch_bounds__(128) k_poisson_near64(const double *__restrict__ pos, const double *__restrict__ charge, int N, real3d L,
                                                        Table64 tabF, Table64 tabP, double *__restrict__ out) {
  const int i = blockIdx.x * 128 + threadIdx.x;
  if (i >= N) return;
  const real3d pi{pos[4 * (size_t)i], pos[4 * (size_t)i + 1], pos[4 * (size_t)i + 2]};
  const double qi = charge[i];
  double tx = 0, ty = 0, tz = 0, tw = 0;
  for (int j = 0; j < N; ++j) {
    const double qj = charge[j];
    real3d rij{pos[4 * (size_t)j] - pi.x, pos[4 * (size_t)j + 1] - pi.y, pos[4 * (size_t)j + 2] - pi.z};
    rij.x -= floor(rij.x / L.x + 0.5) * L.x;   // Box::apply_pbc (utils/Box.cuh:51-58)
    rij.y -= floor(rij.y / L.y + 0.5) * L.y;
    rij.z -= floor(rij.z / L.z + 0.5) * L.z;
    const double r2 = dot3(rij, rij);
    if (MODE == 1) tw += qi * qj * table_get64(tabP, r2);
    else if (MODE == 0) {
      const double r = sqrt(r2);
      const double fmod = -qi * qj * table_get64(tabF, r);
      if (r2 > 0.0) { const double invr = 1.0 / r; tx += invr * (fmod * rij.x); ty += invr * (fmod * rij.y); tz += invr * (fmod * rij.z); }
    } else {
      tw += qj * table_get64(tabP, r2);
      if (r2 > 0.0) {
        const double r = sqrt(r2);
        const double fmod = -qj * table_get64(tabF, r);
        const double invr = 1.0 / r;
        tx += invr * (fmod * rij.x); ty += invr * (fmod * rij.y); tz += invr * (fmod * rij.z);
      }
    }
  }
  if (MODE == 1) out[i] += tw;
  else {
    out[4 * (size_t)i] += tx; out[4 * (size_t)i + 1] += ty; out[4 * (size_t)i + 2] += tz;
    if (MODE == 2) out[4 * (size_t)i + 3] += tw;
  }
}
struct Poisson64 {
  uammd_poisson_parameters_f64 par{};
  double L[3] = {0, 0, 0};
  int cells[3] = {0, 0, 0};
  GridT<double> grid{};
  Kern64 kern{};
  int nxpad = 0;
  size_t planeReal = 0, planeCplx = 0;
  double cutoff = 0;
  int ntable = 0;
  DeviceBuffer tableField, tablePotential, gridQ, planes, gathered, work;
  rocfft_plan fwd = nullptr, inv = nullptr;
  rocfft_execution_info info = nullptr;
  ~Poisson64() {
    if (fwd) rocfft_plan_destroy(fwd);
    if (inv) rocfft_plan_destroy(inv);
    if (info) rocfft_execution_info_destroy(info);
  }
};
static int poisson64_plans(Poisson64 *p) {
  if (int e = rocfft_setup_once()) return e;
  const size_t nx = p->cells[0], ny = p->cells[1], nz = p->cells[2], nkx = nx / 2 + 1;
  const size_t lengths[3] = {nx, ny, nz};
  const size_t rstr[3] = {1, (size_t)p->nxpad, (size_t)p->nxpad * ny}, cstr[3] = {1, nkx, nkx * ny};
  rocfft_plan_description d = nullptr;
  UH_ROCFFT64(rocfft_plan_description_create(&d));
  UH_ROCFFT64(rocfft_plan_description_set_data_layout(d, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, nullptr, nullptr, 3, rstr,
                                                      p->planeReal, 3, cstr, p->planeCplx));
  UH_ROCFFT64(rocfft_plan_create(&p->fwd, rocfft_placement_inplace, rocfft_transform_type_real_forward, rocfft_precision_double, 3, lengths, 1, d));
  UH_ROCFFT64(rocfft_plan_description_destroy(d));
  UH_ROCFFT64(rocfft_plan_description_create(&d));
  UH_ROCFFT64(rocfft_plan_description_set_data_layout(d, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, nullptr, nullptr, 3, cstr,
                                                      p->planeCplx, 3, rstr, p->planeReal));
  UH_ROCFFT64(rocfft_plan_create(&p->inv, rocfft_placement_inplace, rocfft_transform_type_real_inverse, rocfft_precision_double, 3, lengths, 4, d));
  UH_ROCFFT64(rocfft_plan_description_destroy(d));
  size_t wf = 0, wi = 0;
  UH_ROCFFT64(rocfft_plan_get_work_buffer_size(p->fwd, &wf));
  UH_ROCFFT64(rocfft_plan_get_work_buffer_size(p->inv, &wi));
  const size_t w = std::max(wf, wi);
  UH_ROCFFT64(rocfft_execution_info_create(&p->info));
  if (w) {
    if (int e = p->work.reserve(w)) return e;
    UH_ROCFFT64(rocfft_execution_info_set_work_buffer(p->info, p->work.ptr, w));
  }
  return 0;
}
static Table64 view64(const DeviceBuffer &b, int ntable, double rmax) {
  return Table64{(const double *)b.ptr, ntable - 1, rmax, 1.0 / rmax, 1.0 / (double)(ntable - 1)};
}
// farField (.cu:332-360): any of d_force / d_energy / d_fieldPotential may be null
static int poisson64_far(Poisson64 *p, const double *d_pos, const double *d_charge, int N, double *d_force, double *d_energy, double *d_fieldPotential,
                         hipStream_t st) {
  double *gq = (double *)p->gridQ.ptr;
  const FastDiv dsx = make_fastdiv(p->kern.support.x), dsxy = make_fastdiv(p->kern.support.x * p->kern.support.y);
  const dim3 gp((N + 3) / 4), bp(256);
  UH_CHECK(hipMemsetAsync(gq, 0, sizeof(double) * p->planeReal, st));
  hipLaunchKernelGGL((k_ibm64<1, true>), gp, bp, 0, st, d_pos, 4, d_charge, 1, (double *)nullptr, gq, N, p->grid, p->nxpad, (size_t)1, (size_t)1, p->kern,
                     dsx, dsxy, false, false);
  UH_ROCFFT64(rocfft_execution_info_set_stream(p->info, (void *)st));
  void *bq[1] = {gq};
  UH_ROCFFT64(rocfft_execute(p->fwd, bq, nullptr, p->info));
  const int3 n = p->grid.cellDim;
  const size_t total = p->planeCplx;
  hipLaunchKernelGGL(k_poisson_convolve64, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const double2 *)gq, (double2 *)p->planes.ptr,
                     p->planeCplx, n, real3d{p->L[0], p->L[1], p->L[2]}, p->par.epsilon);
  void *bpl[1] = {p->planes.ptr};
  UH_ROCFFT64(rocfft_execute(p->inv, bpl, nullptr, p->info));
  if (int e = p->gathered.reserve(sizeof(double) * 4 * (size_t)N)) return e;
  hipLaunchKernelGGL((k_ibm64<4, false>), gp, bp, 0, st, d_pos, 4, (const double *)nullptr, 4, (double *)p->gathered.ptr, (double *)p->planes.ptr, N, p->grid,
                     p->nxpad, (size_t)1, p->planeReal, p->kern, dsx, dsxy, false, true);
  hipLaunchKernelGGL(k_poisson_unzip64, dim3((N + 255) / 256), dim3(256), 0, st, (const double *)p->gathered.ptr, d_charge, N, d_force, d_energy,
                     d_fieldPotential);
  UH_CHECK(hipGetLastError());
  return 0;
}
template <int MODE> static int poisson64_near(Poisson64 *p, const double *d_pos, const double *d_charge, int N, double *d_out, hipStream_t st) {
  hipLaunchKernelGGL((k_poisson_near64<MODE>), dim3((N + 127) / 128), dim3(128), 0, st, d_pos, d_charge, N, real3d{p->L[0], p->L[1], p->L[2]},
                     view64(p->tableField, p->ntable, p->cutoff), view64(p->tablePotential, p->ntable, p->cutoff * p->cutoff), d_out);
  UH_CHECK(hipGetLastError());
  return 0;
}
int next_fft_wise_axis(int n);   // quasi2d.hip

static double fcm_upsampling64(double tolerance) {  // FCM_kernels.cuh:24-30
  const double amin = 0.55, amax = 1.65;
  const double x = -std::log10(3 * tolerance) / 10.0;
  const double factor = amin + x * (amax - amin);
  return factor < amax ? factor : amax;
}

}  // namespace uammd_hip

using namespace uammd_hip;

extern "C" {

// FCM_ns::Kernels::Gaussian(h, tolerance), BDHI/FCM/FCM_kernels.cuh:22-58 over IBM_kernels::Gaussian (misc/IBM_kernels.cuh:28-40)
int uammd_fcm_gaussian_kernel_f64(double h, double tolerance, uammd_ibm_kernel_f64 *out, double *a_eff) {
  if (!out || !(h > 0) || !(tolerance > 0)) { set_last_error("uammd_fcm_gaussian_kernel_f64: bad arguments"); return -1; }
  const double ups = fcm_upsampling64(tolerance), width = h * ups;
  const double prefactor = std::pow(2.0 * M_PI * width * width, -0.5), tau = -0.5 / (width * width);
  const double dr = 0.5 * h;
  double r = dr;
  while (prefactor * std::exp(tau * r * r) > tolerance) r += dr;
  int support = (int)(2 * r / h + 0.5);
  if (support < 3) support = 3;
  out->kind = UAMMD_IBM_KERNEL_GAUSSIAN;
  out->support[0] = out->support[1] = out->support[2] = support;
  out->prefactor = prefactor;
  out->tau = tau;
  out->rmax = (double)support * h;
  out->invh[0] = out->invh[1] = out->invh[2] = 0.0;
  if (a_eff) *a_eff = (h * ups) * std::sqrt(M_PI);
  return 0;
}
double uammd_fcm_advise_grid_size_f64(double hydrodynamicRadius, double tolerance) {  // FCM_kernels.cuh:47-50
  return hydrodynamicRadius / (std::sqrt(M_PI) * fcm_upsampling64(tolerance));
}

// IBM::spread / IBM::gather (misc/IBM.cuh:99-203) on a user-owned grid with interleaved components, LinearIndex3D(nxStride, ny, nz)
int uammd_ibm_spread_f64(const double *d_pos, int posStride, const double *d_quantity, int ncomp, int N, const double L[3],
                         const int periodic[3], const int cellDim[3], int nxStride, const uammd_ibm_kernel_f64 *kernel, double *d_grid,
                         void *stream) {
  if (int e = check_kernel64("uammd_ibm_spread_f64", kernel)) return e;
  if (posStride < 3 || (ncomp != 1 && ncomp != 3) || nxStride < cellDim[0]) { set_last_error("uammd_ibm_spread_f64: bad arguments"); return -1; }
  if (N <= 0) return 0;
  const GridT<double> g = make_grid<double>(make_box<double>(L, periodic), make_int3(cellDim[0], cellDim[1], cellDim[2]));
  const bool is2D = g.cellDim.z == 1;  // IBM.cuh:189-194
  const Kern64 k = to_dev64(*kernel);
  const FastDiv dsx = make_fastdiv(k.support.x), dsxy = make_fastdiv(k.support.x * k.support.y);
  const dim3 gr((N + 3) / 4), b(256);
  if (ncomp == 1)
    hipLaunchKernelGGL((k_ibm64<1, true>), gr, b, 0, (hipStream_t)stream, d_pos, posStride, d_quantity, 1, (double *)nullptr, d_grid, N, g,
                       nxStride, (size_t)1, (size_t)1, k, dsx, dsxy, is2D, false);
  else
    hipLaunchKernelGGL((k_ibm64<3, true>), gr, b, 0, (hipStream_t)stream, d_pos, posStride, d_quantity, 3, (double *)nullptr, d_grid, N, g,
                       nxStride, (size_t)3, (size_t)1, k, dsx, dsxy, is2D, false);
  UH_CHECK(hipGetLastError());
  return 0;
}
int uammd_ibm_gather_f64(const double *d_pos, int posStride, double *d_out, int ncomp, int N, const double L[3], const int periodic[3],
                         const int cellDim[3], int nxStride, const uammd_ibm_kernel_f64 *kernel, const double *d_grid, void *stream) {
  if (int e = check_kernel64("uammd_ibm_gather_f64", kernel)) return e;
  if (posStride < 3 || (ncomp != 1 && ncomp != 3) || nxStride < cellDim[0]) { set_last_error("uammd_ibm_gather_f64: bad arguments"); return -1; }
  if (N <= 0) return 0;
  const GridT<double> g = make_grid<double>(make_box<double>(L, periodic), make_int3(cellDim[0], cellDim[1], cellDim[2]));
  const bool is2D = g.cellDim.z == 1;
  const Kern64 k = to_dev64(*kernel);
  const FastDiv dsx = make_fastdiv(k.support.x), dsxy = make_fastdiv(k.support.x * k.support.y);
  const dim3 gr((N + 3) / 4), b(256);
  if (ncomp == 1)
    hipLaunchKernelGGL((k_ibm64<1, false>), gr, b, 0, (hipStream_t)stream, d_pos, posStride, (const double *)nullptr, 1, d_out,
                       const_cast<double *>(d_grid), N, g, nxStride, (size_t)1, (size_t)1, k, dsx, dsxy, is2D, false);
  else
    hipLaunchKernelGGL((k_ibm64<3, false>), gr, b, 0, (hipStream_t)stream, d_pos, posStride, (const double *)nullptr, 3, d_out,
                       const_cast<double *>(d_grid), N, g, nxStride, (size_t)3, (size_t)1, k, dsx, dsxy, is2D, false);
  UH_CHECK(hipGetLastError());
  return 0;
}

// FCM_impl (Integrator/BDHI/FCM/FCM_impl.cuh:56-119): triply periodic, three planar padded grids transformed in place by rocFFT (double)
int uammd_fcm_create_f64(const uammd_fcm_parameters_f64 *par, uammd_fcm_f64 **out) {
  if (!par || !out) { set_last_error("uammd_fcm_create_f64: null argument"); return -1; }
  if (int e = check_kernel64("uammd_fcm_create_f64", &par->kernel)) return e;
  for (int a = 0; a < 3; ++a) {
    // A support that is not smaller than the grid: the reference logs an ERROR and goes on (BDHI_FCM.cuh:58-64) — its acceptance script
    // walks through such boxes at its tolerance of 1e-14 (test/BDHI/FCM/test.bash:33, FCM.cu selfMobilityCubicBox) — and the stencil
    // wraps as far as the reference's own Grid::pbc_cell goes (one +- n): the grid must hold half a support (uammd_fcm_create's rule).
    if (par->cells[a] >= 2 && par->boxSize[a] > 0 && 2 * par->cells[a] < par->kernel.support[a] + 1) {
      set_last_error("[BDHI::FCM] Kernel support is too big, try lowering the tolerance or increasing the box size!.");
      return -2;
    }
    if (par->cells[a] < 2 || !(par->boxSize[a] > 0)) {
      set_last_error("uammd_fcm_create_f64: bad grid (cells %d %d %d, support %d)", par->cells[0], par->cells[1], par->cells[2], par->kernel.support[0]);
      return -2;
    }
  }
  FCM64 *f = new FCM64();
  const int periodic[3] = {1, 1, 1};
  f->grid = make_grid<double>(make_box<double>(par->boxSize, periodic), make_int3(par->cells[0], par->cells[1], par->cells[2]));
  f->kern = to_dev64(par->kernel);
  for (int a = 0; a < 3; ++a) f->L[a] = par->boxSize[a];
  f->viscosity = par->viscosity;
  f->nxpad = 2 * (par->cells[0] / 2 + 1);
  f->planeReal = (size_t)f->nxpad * par->cells[1] * par->cells[2];
  f->planeCplx = f->planeReal / 2;
  if (int e = f->gridBuf.reserve(sizeof(double) * 3 * f->planeReal)) { delete f; return e; }
  if (int e = fcm64_plans(f)) { delete f; return e; }
  *out = reinterpret_cast<uammd_fcm_f64 *>(f);
  return 0;
}
int uammd_fcm_destroy_f64(uammd_fcm_f64 *h) {
  delete reinterpret_cast<FCM64 *>(h);
  return 0;
}
// FCM_impl::computeHydrodynamicDisplacements (FCM_impl.cuh:652-693) / FarField::computeHydrodynamicDisplacements (FarField.cuh:569-589):
// d_velocity real3[N] = M F + prefactor sqrt(2 T M) dW (overwritten; a PSE far-field handle ADDS, as FarField does).  d_pos / d_force
// real4[N]; d_force may be NULL (noise only: deterministicPart of FarField.cuh:451-467).  seed1 / seed2 key the Fourier noise as
// Saru(node, seed1, seed2).
int uammd_fcm_displacements_thermal_f64(uammd_fcm_f64 *h, const double *d_pos, const double *d_force, int N, double temperature, double prefactor,
                                        unsigned int seed1, unsigned int seed2, double *d_velocity, void *stream) {
  if (!h || (N > 0 && (!d_pos || !d_velocity))) { set_last_error("uammd_fcm_displacements_thermal_f64: null argument"); return -1; }
  if (N <= 0) return 0;
  FCM64 *f = reinterpret_cast<FCM64 *>(h);
  hipStream_t st = (hipStream_t)stream;
  double *g = (double *)f->gridBuf.ptr;
  const FastDiv dsx = make_fastdiv(f->kern.support.x), dsxy = make_fastdiv(f->kern.support.x * f->kern.support.y);
  const dim3 gp((N + 3) / 4), bp(256);
  const bool thermal = temperature > 0.0 && prefactor != 0.0;
  if (!d_force && !thermal) {   // nothing to add: M 0 = 0
    if (!f->accumulate) UH_CHECK(hipMemsetAsync(d_velocity, 0, sizeof(double) * 3 * (size_t)N, st));
    return 0;
  }
  UH_ROCFFT64(rocfft_execution_info_set_stream(f->info, st));
  void *bufs[1] = {g};
  if (d_force) {
    UH_CHECK(hipMemsetAsync(g, 0, sizeof(double) * 3 * f->planeReal, st));
    hipLaunchKernelGGL((k_ibm64<3, true>), gp, bp, 0, st, d_pos, 4, d_force, 4, (double *)nullptr, g, N, f->grid, f->nxpad, (size_t)1, f->planeReal,
                       f->kern, dsx, dsxy, false, false);
    UH_ROCFFT64(rocfft_execute(f->fwd, bufs, nullptr, f->info));
  }
  double noisePrefactor = 0.0;
  if (thermal) {
    const double dV = f->grid.cellVolume;
    const double fourierNormalization = 1.0 / ((double)f->grid.cellDim.x * f->grid.cellDim.y * f->grid.cellDim.z);
    noisePrefactor = f->pse.on ? prefactor * std::sqrt(2 * temperature / dV)                             // FarField.cuh:503: the 1 / N lives in B
                               : prefactor * std::sqrt(fourierNormalization * 2 * temperature / dV);    // FCM_impl.cuh:527-530
  }
  const size_t total = f->planeCplx;
  hipLaunchKernelGGL(k_kspace64, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (double2 *)g, f->planeCplx, f->grid.cellDim,
                     real3d{f->L[0], f->L[1], f->L[2]}, f->viscosity, f->pse, d_force != nullptr, noisePrefactor, seed1, seed2);
  UH_ROCFFT64(rocfft_execute(f->inv, bufs, nullptr, f->info));
  hipLaunchKernelGGL((k_ibm64<3, false>), gp, bp, 0, st, d_pos, 4, (const double *)nullptr, 3, d_velocity, g, N, f->grid, f->nxpad, (size_t)1,
                     f->planeReal, f->kern, dsx, dsxy, false, !f->accumulate);
  UH_CHECK(hipGetLastError());
  return 0;
}
// the deterministic part alone (T = 0)
int uammd_fcm_displacements_f64(uammd_fcm_f64 *h, const double *d_pos, const double *d_force, int N, double *d_velocity, void *stream) {
  if (N > 0 && !d_force) { set_last_error("uammd_fcm_displacements_f64: null argument"); return -1; }
  return uammd_fcm_displacements_thermal_f64(h, d_pos, d_force, N, 0.0, 0.0, 0u, 0u, d_velocity, stream);
}

// ---- PSE (Integrator/BDHI/PSE) ------------------------------------------------------------------------------------------------------------
int uammd_pse_far_raw_cells_f64(const double boxSize[3], double psi, double tolerance, int cells_out[3]) {  // FarField.cuh:646-654
  if (!boxSize || !cells_out) { set_last_error("uammd_pse_far_raw_cells_f64: null argument"); return -1; }
  const double kcut = 2 * psi * std::sqrt(-std::log(tolerance));
  const double hgrid = 2 * M_PI / kcut;
  for (int k = 0; k < 3; ++k) cells_out[k] = (int)(2 * boxSize[k] / hgrid) + 1;
  return 0;
}
// FarField ctor + initializeKernel (FarField.cuh:318-342,605-644)
int uammd_pse_far_create_f64(const double boxSize[3], const int cells[3], double viscosity, double hydrodynamicRadius, double tolerance, double psi,
                             double shearStrain, uammd_fcm_f64 **out, int *support_out, double *eta_out) {
  if (!boxSize || !cells || !out) { set_last_error("uammd_pse_far_create_f64: null argument"); return -1; }
  const double C = 0.976;
  double m = 1;
  while (std::erfc(m / std::sqrt(2.0)) > 0.1 * tolerance) m += 0.01;
  int support;
  while ((support = int(std::pow(m / C, 2) / M_PI + 0.5) + 1) % 2 == 0) m += tolerance;
  int P = support / 2;
  const int minCellDim = std::min(cells[0], std::min(cells[1], cells[2]));
  if (support > minCellDim) {
    support = minCellDim;
    if (support % 2 == 0) support--;
    P = support / 2;
    m = C * std::sqrt(M_PI * support);
  }
  const double pw = 2 * P + 1;
  const double cs[3] = {boxSize[0] / cells[0], boxSize[1] / cells[1], boxSize[2] / cells[2]};
  const double hh = std::min(cs[0], std::min(cs[1], cs[2]));
  const double w = pw * hh / 2.0;
  const double eta = std::pow(2.0 * psi * w / m, 2);
  const double width = std::sqrt(eta) / (2.0 * psi);  // pse_ns::Kernel(P, width), FarField.cuh:25-41
  if (2 * P + 1 > kMaxSupport) { set_last_error("uammd_pse_far_create_f64: support %d above %d", 2 * P + 1, kMaxSupport); return -2; }
  uammd_fcm_parameters_f64 p{};
  for (int k = 0; k < 3; ++k) { p.boxSize[k] = boxSize[k]; p.cells[k] = cells[k]; }
  p.viscosity = viscosity;
  p.kernel.kind = UAMMD_IBM_KERNEL_GAUSSIAN;
  p.kernel.support[0] = p.kernel.support[1] = p.kernel.support[2] = 2 * P + 1;
  p.kernel.prefactor = std::cbrt(1.0 / (width * width * width * std::pow(2.0 * M_PI, 1.5)));
  p.kernel.tau = -0.5 / (width * width);
  p.kernel.rmax = INFINITY;  // this window is not cut (FarField.cuh:37-39)
  if (int e = uammd_fcm_create_f64(&p, out)) return e;
  FCM64 *f = reinterpret_cast<FCM64 *>(*out);
  f->pse = Pse64{hydrodynamicRadius, psi, eta, shearStrain, true};
  f->accumulate = true;  // ibm.gather adds into MF (FarField.cuh:563-566)
  if (support_out) *support_out = 2 * P + 1;
  if (eta_out) *eta_out = eta;
  return 0;
}

// NearField::initializeDeterministicPart (NearField.cuh:65-99) + TabulatedFunction<real2>
int uammd_pse_near_create_f64(const double boxSize[3], double viscosity, double hydrodynamicRadius, double tolerance, double psi,
                              uammd_pse_near_f64 **out, double *rcut_out, int *nPointsTable_out) {
  if (!boxSize || !out) { set_last_error("uammd_pse_near_create_f64: null argument"); return -1; }
  const double rcut = std::sqrt(-std::log(tolerance)) / psi;
  if (0.5 * boxSize[0] < rcut) { set_last_error("[BDHI::PSE] Cut off is too large, try increasing psi"); return -2; }
  const double textureTolerance = hydrodynamicRadius * tolerance;
  double np = rcut / textureTolerance + 0.5;
  if (np > 2e30) np = 2e30;
  unsigned nPointsTable = np >= 4294967295.0 ? 4294967295u : (unsigned)np;
  nPointsTable = std::min(1u << 22, std::max(1u << 14, nPointsTable));
  const int Ntable = (int)nPointsTable - 1;
  const double normalization = 6 * M_PI * hydrodynamicRadius * viscosity;
  std::vector<double2> host((size_t)Ntable + 1);
  for (int i = 0; i <= Ntable; ++i) {
    const double x = (i / (double)Ntable) * rcut;
    double F, G;
    rpy_near_FandG(x, hydrodynamicRadius, psi, rcut, &F, &G);
    host[i] = make_double2(F / normalization, G / normalization);
  }
  PSENear64 *p = new PSENear64();
  for (int k = 0; k < 3; ++k) p->L[k] = boxSize[k];
  p->rcut = rcut;
  p->nPointsTable = (int)nPointsTable;
  if (p->table.reserve(sizeof(double2) * host.size()) ||
      hipMemcpy(p->table.ptr, host.data(), sizeof(double2) * host.size(), hipMemcpyHostToDevice) != hipSuccess) {
    delete p;
    set_last_error("uammd_pse_near_create_f64: could not upload the RPY table");
    return -3;
  }
  *out = reinterpret_cast<uammd_pse_near_f64 *>(p);
  if (rcut_out) *rcut_out = rcut;
  if (nPointsTable_out) *nPointsTable_out = (int)nPointsTable;
  return 0;
}
int uammd_pse_near_destroy_f64(uammd_pse_near_f64 *h) {
  delete reinterpret_cast<PSENear64 *>(h);
  return 0;
}
// NearField::Mdot (NearField.cuh:239-250): d_MF real3[N] += M_near F with d_force real4[N] (vstride 4), or a real3 vector (vstride 3)
int uammd_pse_near_mdot_f64(uammd_pse_near_f64 *h, const double *d_pos, const double *d_v, int vstride, int N, double *d_MF, void *stream) {
  if (!h || (N > 0 && (!d_pos || !d_v || !d_MF)) || (vstride != 3 && vstride != 4)) { set_last_error("uammd_pse_near_mdot_f64: bad arguments"); return -1; }
  if (N <= 0) return 0;
  PSENear64 *p = reinterpret_cast<PSENear64 *>(h);
  hipLaunchKernelGGL(k_pse_near64, dim3((N + 127) / 128), dim3(128), 0, (hipStream_t)stream, d_pos, d_v, vstride, N,
                     real3d{p->L[0], p->L[1], p->L[2]}, p->rcut, (const double2 *)p->table.ptr, p->nPointsTable - 1, d_MF, false);
  UH_CHECK(hipGetLastError());
  return 0;
}

// NearField::computeStochasticDisplacements (NearField.cuh:255-284): d_BdW real3[N] = sqrt(M_near) dW * prefactor sqrt(2 T) — the Lanczos
// result OVERWRITES d_BdW, as lanczos->run does there — noise = SaruTransform (:218-228) keyed (particle, seed1, seed2), the solver's
// tolerance the near field's.  `solver` is the caller's lanczos handle (kept between steps for its adaptive check schedule).
static int pse_near64_dot(void *ctx, const double *d_v, double *d_Mv, int n, void *stream) {
  struct Ctx { PSENear64 *p; const double *pos; };
  const Ctx *c = static_cast<const Ctx *>(ctx);
  const int N = n / 3;
  hipLaunchKernelGGL(k_pse_near64, dim3((N + 127) / 128), dim3(128), 0, (hipStream_t)stream, c->pos, d_v, 3, N,
                     real3d{c->p->L[0], c->p->L[1], c->p->L[2]}, c->p->rcut, (const double2 *)c->p->table.ptr, c->p->nPointsTable - 1, d_Mv, true);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
__global__ void __launch_bounds__(256) k_pse_noise64(double *__restrict__ out3, int N, double variance, uint seed1, uint seed2) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  Saru rng((uint)i, seed1, seed2);
  const float2 a = rng.gf(0.0f, 1.0f);
  const float2 b = rng.gf(0.0f, 1.0f);
  out3[3 * (size_t)i] = (double)a.x * variance;
  out3[3 * (size_t)i + 1] = (double)a.y * variance;
  out3[3 * (size_t)i + 2] = (double)b.x * variance;
}
int uammd_pse_near_stochastic_f64(uammd_pse_near_f64 *h, uammd_lanczos_f64 *solver, const double *d_pos, int N, double temperature, double prefactor,
                                  unsigned int seed1, unsigned int seed2, double tolerance, double *d_BdW, void *stream, int *iterations) {
  if (iterations) *iterations = 0;
  if (!h || !solver || (N > 0 && (!d_pos || !d_BdW))) { set_last_error("uammd_pse_near_stochastic_f64: null argument"); return -1; }
  if (temperature == 0.0 || N <= 0) return 0;
  PSENear64 *p = reinterpret_cast<PSENear64 *>(h);
  if (int e = p->noise.reserve(sizeof(double) * 3 * (size_t)N)) return e;
  hipLaunchKernelGGL(k_pse_noise64, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, (double *)p->noise.ptr, N,
                     prefactor * std::sqrt(2 * temperature), seed1, seed2);
  UH_CHECK(hipGetLastError());
  struct Ctx { PSENear64 *p; const double *pos; } ctx{p, d_pos};
  return uammd_lanczos_run_f64(solver, &pse_near64_dot, &ctx, d_BdW, (const double *)p->noise.ptr, tolerance, 3 * N, stream, iterations);
}

// BDHI::EulerMaruyama_ns::integrateGPUD (Integrator/BDHI/BDHI_EulerMaruyama.cu:82-113) with real = double: pos += dt (K pos + MF) + sqrt2Tdt BdW
struct Shear9d { double k[9]; };
__global__ void __launch_bounds__(256) k_bdhi_euler_maruyama64(double *__restrict__ pos, const int *__restrict__ index, const double *__restrict__ MF,
                                                               const double *__restrict__ BdW, Shear9d K, bool haveK, int N, double sqrt2Tdt,
                                                               double dt, bool is2D) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= N) return;
  const size_t i = (size_t)(index ? index[id] : id);
  double x = pos[4 * i], y = pos[4 * i + 1], z = pos[4 * i + 2];
  if (haveK) {
    const double krx = fma(K.k[2], z, fma(K.k[1], y, K.k[0] * x));
    const double kry = fma(K.k[5], z, fma(K.k[4], y, K.k[3] * x));
    const double krz = is2D ? 0.0 : fma(K.k[8], z, fma(K.k[7], y, K.k[6] * x));
    x = fma(krx, dt, x); y = fma(kry, dt, y); z = fma(krz, dt, z);
  }
  x = fma(MF[3 * (size_t)id], dt, x);
  y = fma(MF[3 * (size_t)id + 1], dt, y);
  z = fma(MF[3 * (size_t)id + 2], dt, z);
  if (BdW) {
    x = fma(sqrt2Tdt, BdW[3 * (size_t)id], x);
    y = fma(sqrt2Tdt, BdW[3 * (size_t)id + 1], y);
    z = fma(sqrt2Tdt, is2D ? 0.0 : BdW[3 * (size_t)id + 2], z);
  }
  pos[4 * i] = x; pos[4 * i + 1] = y; pos[4 * i + 2] = z;
}
int uammd_bdhi_euler_maruyama_f64(double *d_pos, const int *d_index, const double *d_MF, const double *d_BdW, const double K[9], int N,
                                  double sqrt2Tdt, double dt, int is2D, void *stream) {
  if (N <= 0) return 0;
  if (!d_pos || !d_MF) { set_last_error("uammd_bdhi_euler_maruyama_f64: null argument"); return -1; }
  Shear9d k{};
  if (K) for (int t = 0; t < 9; ++t) k.k[t] = K[t];
  hipLaunchKernelGGL(k_bdhi_euler_maruyama64, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_pos, d_index, d_MF, d_BdW, k, K != nullptr, N,
                     sqrt2Tdt, dt, is2D != 0);
  UH_CHECK(hipGetLastError());
  return 0;
}

// ---- BD::EulerMaruyama / MidPoint / AdamsBashforth / Leimkuhler with real = double (Integrator/BrownianDynamics.cu:119-144, :178-214,
// :262-289, :313-345; single precision: integrators.hip).  One streaming kernel, the scheme a template parameter; the same expressions
// as the single-precision kernels in double.  The draws stay FLOAT Gaussians, as in the reference's double build: Saru::gf takes and
// returns floats (third_party/saruprng.cuh:291-301), the amplitude is rounded to float on the way in.
//   SCHEME 0             EulerMaruyama: sqrt(2 T M dt), the generator keyed by the particle index i
//   SCHEME 1, SUB 0 / 1  MidPoint: half a step from the current forces (the starting point kept in aux[id]), then a whole step from the
//                        kept point with the midpoint's forces; sqrt(T M dt) per draw, the second sub-step repeats the first draw and
//                        adds another; keyed by the GROUP index id
//   SCHEME 2             AdamsBashforth: forces 3/2 F_n - 1/2 F_(n-1) (aux[id] = F_(n-1) in group order), sqrt(2 T M dt), keyed by id
//   SCHEME 3             Leimkuhler: Euler drift, noise sqrt(T M dt / 2) (dW_n + dW_(n-1)), keyed by originalIndex[i]
}  // extern "C"
struct Shear9c { double k[9]; };
template <int SCHEME, int SUB>
__global__ void __launch_bounds__(256) k_bd_scheme64(double *__restrict__ pos, double *__restrict__ aux, const int *__restrict__ index,
                                                     const int *__restrict__ originalIndex, const double *__restrict__ force, Shear9c K,
                                                     double selfMobility, const double *__restrict__ radius, double dt, int is2D,
                                                     double temperature, int N, uint stepNum, uint seed) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= N) return;
  const size_t i = (size_t)(index ? index[id] : id);
  double px = pos[4 * i], py = pos[4 * i + 1], pz = pos[4 * i + 2];
  if (SCHEME == 1) {
    if (SUB == 0) { aux[4 * (size_t)id] = px; aux[4 * (size_t)id + 1] = py; aux[4 * (size_t)id + 2] = pz; aux[4 * (size_t)id + 3] = pos[4 * i + 3]; }
    else { px = aux[4 * (size_t)id]; py = aux[4 * (size_t)id + 1]; pz = aux[4 * (size_t)id + 2]; }
  }
  double fx = force[4 * i], fy = force[4 * i + 1], fz = force[4 * i + 2];
  double KRx = fma(K.k[2], pz, fma(K.k[1], py, K.k[0] * px)), KRy = fma(K.k[5], pz, fma(K.k[4], py, K.k[3] * px)),
         KRz = fma(K.k[8], pz, fma(K.k[7], py, K.k[6] * px));
  const double M = selfMobility * (radius ? (1.0 / radius[i]) : 1.0);
  if (SCHEME == 1 && SUB == 0) { fx *= 0.5; fy *= 0.5; fz *= 0.5; KRx *= 0.5; KRy *= 0.5; KRz *= 0.5; }
  if (SCHEME == 2) {
    const double qx = aux[4 * (size_t)id], qy = aux[4 * (size_t)id + 1], qz = aux[4 * (size_t)id + 2];
    fx = fma(-0.5, qx, 1.5 * fx); fy = fma(-0.5, qy, 1.5 * fy); fz = fma(-0.5, qz, 1.5 * fz);
  }
  double rx = fma(dt, fma(M, fx, KRx), px), ry = fma(dt, fma(M, fy, KRy), py), rz = fma(dt, fma(M, fz, KRz), pz);
  if (temperature > 0.0) {
    if (SCHEME == 3) {
      const uint ori = (uint)(originalIndex ? originalIndex[i] : (int)i);
      const double B = sqrt(0.5 * temperature * M * dt);
      Saru a(ori, stepNum, seed), b(ori, stepNum - 1u, seed);
      const float2 a01 = a.gf(0.0f, 1.0f);
      const float a2 = a.gf(0.0f, 1.0f).x;
      const float2 b01 = b.gf(0.0f, 1.0f);
      const float b2 = b.gf(0.0f, 1.0f).x;
      rx = fma(B, (double)a01.x + (double)b01.x, rx); ry = fma(B, (double)a01.y + (double)b01.y, ry); rz = fma(B, (double)a2 + (double)b2, rz);
    } else {
      const float B = (float)(SCHEME == 1 ? sqrt(temperature * M * dt) : sqrt(2.0 * temperature * M * dt));
      Saru rng(SCHEME == 0 ? (uint)i : (uint)id, stepNum, seed);
      const float2 d01 = rng.gf(0.0f, B);
      const float d2 = rng.gf(0.0f, B).x;
      rx += (double)d01.x; ry += (double)d01.y; rz += (double)d2;
      if (SCHEME == 1 && SUB == 1) {
        const float2 e01 = rng.gf(0.0f, B);
        const float e2 = rng.gf(0.0f, B).x;
        rx += (double)e01.x; ry += (double)e01.y; rz += (double)e2;
      }
    }
  }
  pos[4 * i] = rx;
  pos[4 * i + 1] = ry;
  if (!is2D) pos[4 * i + 2] = rz;   // (pos[i].w is the particle's own: never written)
}
extern "C" {
int uammd_bd_scheme_step_f64(int scheme, int substep, double *d_pos, double *d_aux, const int *d_index, const int *d_originalIndex,
                             const double *d_force, const double K[9], double selfMobility, const double *d_radius, double dt, int is2D,
                             double temperature, int N, unsigned int stepNum, unsigned int seed, void *stream) {
  if (N <= 0) return 0;
  if (!d_pos || !d_force || ((scheme == UAMMD_BD_MIDPOINT || scheme == UAMMD_BD_ADAMS_BASHFORTH) && !d_aux)) {
    set_last_error("uammd_bd_scheme_step_f64: null argument");
    return -1;
  }
  Shear9c S{};
  if (K) for (int t = 0; t < 9; ++t) S.k[t] = K[t];
  const dim3 g((N + 255) / 256), b(256);
  hipStream_t st = (hipStream_t)stream;
#define UH_BD64(SC, SU) hipLaunchKernelGGL((k_bd_scheme64<SC, SU>), g, b, 0, st, d_pos, d_aux, d_index, d_originalIndex, d_force, S, selfMobility, \
                                           d_radius, dt, is2D, temperature, N, stepNum, seed)
  if (scheme == UAMMD_BD_EULER_MARUYAMA) UH_BD64(0, 0);
  else if (scheme == UAMMD_BD_MIDPOINT && substep == 0) UH_BD64(1, 0);
  else if (scheme == UAMMD_BD_MIDPOINT && substep == 1) UH_BD64(1, 1);
  else if (scheme == UAMMD_BD_ADAMS_BASHFORTH) UH_BD64(2, 0);
  else if (scheme == UAMMD_BD_LEIMKUHLER) UH_BD64(3, 0);
  else { set_last_error("uammd_bd_scheme_step_f64: unknown scheme %d / sub-step %d", scheme, substep); return -1; }
#undef UH_BD64
  UH_CHECK(hipGetLastError());
  return 0;
}

// ---- BDHI::True2D / Quasi2D, real = double (the single-precision entry points: quasi2d.hip) ---------------------------------------------
int uammd_bdhi2d_create_f64(const uammd_bdhi2d_parameters_f64 *par, uammd_bdhi2d_f64 **out, int cells[2], int *support) {
  if (!par || !out || (par->kernel != UAMMD_BDHI2D_TRUE2D && par->kernel != UAMMD_BDHI2D_QUASI2D)) {
    set_last_error("uammd_bdhi2d_create_f64: bad arguments");
    return -1;
  }
  if (par->boxSize[0] == 0.0 && par->boxSize[1] == 0.0) { set_last_error("Invalid box"); return -2; }              // .cu:46-52
  if (!(par->hydrodynamicRadius > 0)) { set_last_error("Invalid hydrodynamic radius"); return -2; }                // .cu:53-57
  if (!(par->viscosity > 0) || !(par->boxSize[0] > 0) || !(par->boxSize[1] > 0)) { set_last_error("uammd_bdhi2d_create_f64: bad arguments"); return -1; }
  BDHI2D64 *q = new (std::nothrow) BDHI2D64();
  if (!q) { set_last_error("uammd_bdhi2d_create_f64: out of host memory"); return -3; }
  q->par = *par;
  const double a = par->hydrodynamicRadius;
  int cd[2] = {par->cells[0], par->cells[1]};
  if (cd[0] <= 0) {  // initializeGrid, .cu:61-73
    const double h = a * 0.8;
    cd[0] = next_fft_wise_axis((int)(par->boxSize[0] / h));
    cd[1] = next_fft_wise_axis((int)(par->boxSize[1] / h));
  }
  const double L3[3] = {par->boxSize[0], par->boxSize[1], 0.0};
  const int per[3] = {1, 1, 0};
  q->grid = make_grid<double>(make_box<double>(L3, per), make_int3(cd[0], cd[1], 1));
  int s = ((int)(3.0 * a * cd[0] / par->boxSize[0]) + 1) * 2 + 1;  // initializeInterpolationKernel, .cu:75-88
  if (s > cd[0]) s = cd[0];
  if (s > kMaxSupport || s > cd[1]) {
    set_last_error("uammd_bdhi2d_create_f64: window support %d is larger than the grid or than the %d nodes per axis one wave evaluates", s, kMaxSupport);
    delete q;
    return -2;
  }
  const double w = par->kernel == UAMMD_BDHI2D_TRUE2D ? std::pow(a * 0.66556976637237890625, 2) : std::pow(a / std::sqrt(M_PI), 2);
  q->kern = Kern64{kKernelGauss2D, make_int3(s, s, 1), std::sqrt(1.0 / (2.0 * M_PI * w)), -1.0 / (2.0 * w), INFINITY, 0.0, 0.0, 0.0};
  q->kernDriftX = q->kern;
  q->kernDriftX.kind = kKernelGauss2DDriftX;
  q->kernDriftX.prefactor = -std::sqrt(1.0 / (2.0 * M_PI * w * w));
  q->kernDriftY = q->kernDriftX;
  q->kernDriftY.kind = kKernelGauss2DDriftY;
  q->nxpad = 2 * (cd[0] / 2 + 1);
  q->planeReal = (size_t)q->nxpad * cd[1];
  q->planeCplx = (size_t)(cd[0] / 2 + 1) * cd[1];
  int e = q->gridBuf.reserve(sizeof(double) * 2 * q->planeReal);
  if (!e) e = q->drift.reserve(sizeof(double) * 4);
  if (!e) {
    const double T = par->temperature, c[4] = {-T, 0.0, 0.0, -T};
    if (hipMemcpy(q->drift.ptr, c, sizeof(c), hipMemcpyHostToDevice) != hipSuccess) { set_last_error("uammd_bdhi2d_create_f64: hipMemcpy failed"); e = -4; }
  }
  if (!e) e = q2d64_plans(q);
  if (e) { delete q; return e; }
  if (cells) { cells[0] = cd[0]; cells[1] = cd[1]; }
  if (support) *support = s;
  *out = reinterpret_cast<uammd_bdhi2d_f64 *>(q);
  return 0;
}
int uammd_bdhi2d_destroy_f64(uammd_bdhi2d_f64 *h) {
  delete reinterpret_cast<BDHI2D64 *>(h);
  return 0;
}
int uammd_bdhi2d_velocities_f64(uammd_bdhi2d_f64 *h, const double *d_pos, const double *d_force, int N, double *d_vel, void *stream) {
  if (!h) { set_last_error("uammd_bdhi2d_velocities_f64: null argument"); return -1; }
  if (N <= 0) return 0;
  if (!d_pos || !d_vel) { set_last_error("uammd_bdhi2d_velocities_f64: null argument"); return -1; }
  BDHI2D64 *q = reinterpret_cast<BDHI2D64 *>(h);
  hipStream_t st = (hipStream_t)stream;
  const double T = q->par.temperature;
  const bool drift = q->par.kernel == UAMMD_BDHI2D_QUASI2D && T > 0;  // hasThermalDrift() and temperature > 0
  const bool deterministic = d_force != nullptr || drift;
  double *g = (double *)q->gridBuf.ptr;
  const dim3 gp((N + 3) / 4), bp(256);
  const FastDiv dsx = make_fastdiv(q->kern.support.x), dsxy = make_fastdiv(q->kern.support.x * q->kern.support.y);
  const int nx = q->grid.cellDim.x, ny = q->grid.cellDim.y;
  UH_ROCFFT64(rocfft_execution_info_set_stream(q->info, (void *)st));
  void *bufs[1] = {g};
  const double *c = (const double *)q->drift.ptr;
#define UH_Q2D_SPREAD(quantity, qstride, kernel)                                                                                                 \
  hipLaunchKernelGGL((k_ibm64<2, true>), gp, bp, 0, st, d_pos, 4, quantity, qstride, (double *)nullptr, g, N, q->grid, q->nxpad, (size_t)1, q->planeReal, \
                     kernel, dsx, dsxy, true, false)
  if (deterministic) {
    UH_CHECK(hipMemsetAsync(g, 0, sizeof(double) * 2 * q->planeReal, st));
    if (drift) {  // spreadThermalDrift, .cu:234-257: every particle spreads the constant (-T, 0) with the x window, (0, -T) with the y window
      UH_Q2D_SPREAD(c, 0, q->kernDriftX);
      UH_Q2D_SPREAD(c + 2, 0, q->kernDriftY);
    }
    if (d_force) UH_Q2D_SPREAD(d_force, 4, q->kern);   // spreadParticleForces, .cu:268-283
    UH_ROCFFT64(rocfft_execute(q->fwd, bufs, nullptr, q->info));
  }
#undef UH_Q2D_SPREAD
  if (!deterministic && !(T > 0)) {  // nothing moves the particles
    UH_CHECK(hipMemsetAsync(d_vel, 0, sizeof(double) * 2 * (size_t)N, st));
    return 0;
  }
  double noisePrefactor = 0.0;
  if (T > 0) {  // addStochastichTermFourier, .cu:450-469
    q->counter++;
    noisePrefactor = std::sqrt(2.0 * T / (q->par.viscosity * q->par.dt * q->par.boxSize[0] * q->par.boxSize[1]));
  }
  const int total = (int)q->planeCplx;
  hipLaunchKernelGGL(k_q2d_kspace64, dim3((total + 255) / 256), dim3(256), 0, st, (double2 *)g, (double2 *)g + q->planeCplx, nx, ny, q->par.boxSize[0],
                     q->par.boxSize[1], q->par.kernel, q->par.hydrodynamicRadius, q->par.viscosity, deterministic, noisePrefactor, q->par.seed,
                     q->counter);
  UH_ROCFFT64(rocfft_execute(q->inv, bufs, nullptr, q->info));
  hipLaunchKernelGGL((k_ibm64<2, false>), gp, bp, 0, st, d_pos, 4, (const double *)nullptr, 2, d_vel, g, N, q->grid, q->nxpad, (size_t)1, q->planeReal,
                     q->kern, dsx, dsxy, true, true);
  UH_CHECK(hipGetLastError());
  return 0;
}
int uammd_bdhi2d_update_positions_f64(double *d_pos, const double *d_vel, int N, double dt, void *stream) {
  if (N <= 0) return 0;
  if (!d_pos || !d_vel) { set_last_error("uammd_bdhi2d_update_positions_f64: null argument"); return -1; }
  hipLaunchKernelGGL(k_q2d_update64, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_pos, (const double2 *)d_vel, N, dt);
  UH_CHECK(hipGetLastError());
  return 0;
}

// ---- Poisson, real = double (the single-precision entry points: poisson.hip) ------------------------------------------------------------
int uammd_poisson_create_f64(const uammd_poisson_parameters_f64 *par, uammd_poisson_f64 **out, uammd_poisson_info_f64 *info) {
  if (!par || !out) { set_last_error("uammd_poisson_create_f64: null argument"); return -1; }
  if (!(par->boxSize[0] > 0) || !(par->boxSize[1] > 0) || !(par->boxSize[2] > 0) || !(par->epsilon > 0) || !(par->gw > 0) || !(par->tolerance > 0)) {
    set_last_error("uammd_poisson_create_f64: box, epsilon, gw and tolerance must be positive");
    return -2;
  }
  Poisson64 *p = new (std::nothrow) Poisson64();
  if (!p) { set_last_error("uammd_poisson_create_f64: out of host memory"); return -3; }
  p->par = *par;
  const double gw = par->gw, split = par->split, epsilon = par->epsilon, tolerance = par->tolerance;
  for (int a = 0; a < 3; ++a) p->L[a] = par->boxSize[a];
  // grid (.cu:75-90)
  const double fw = split > 0 ? std::sqrt(gw * gw + 1.0 / (4.0 * split * split)) : gw;
  double h;
  if (par->upsampling > 0) h = 1.0 / par->upsampling;
  else h = (1.3 - std::min((-std::log10(tolerance)) / 10.0, 0.9)) * fw;
  h = std::min(h, p->L[0] / 32.0);
  for (int a = 0; a < 3; ++a) p->cells[a] = next_fft_wise_axis((int)(p->L[a] / h));
  const int per[3] = {1, 1, 1};
  p->grid = make_grid<double>(make_box<double>(p->L, per), make_int3(p->cells[0], p->cells[1], p->cells[2]));
  const double hx = p->grid.cellSize.x;
  // window (SpectralEwaldPoisson.cuh:65-70, .cu:93-104)
  const double prefactor = std::cbrt(std::pow(2 * M_PI * fw * fw, -1.5));
  const double tau = -1.0 / (2.0 * fw * fw);
  const double rmax = std::sqrt(std::log(tolerance * std::sqrt(2 * M_PI * fw * fw)) / tau);
  int support = std::max(3, (int)(2 * rmax / hx + 0.5));
  if (support > p->cells[0] / 2 - 1) {
    set_last_error("[Poisson] Kernel support (%d) is too large for this configuration (max is %d), try increasing splitting "
                   "parameter or decrasing tolerance", support, p->cells[0] / 2 - 1);
    delete p;
    return -2;
  }
  support = std::min(support, p->cells[0] / 2 - 2);
  if (support > kMaxSupport) {
    set_last_error("uammd_poisson_create_f64: window support %d exceeds the %d nodes per axis one wave evaluates", support, kMaxSupport);
    delete p;
    return -2;
  }
  p->kern = Kern64{kKernelGaussian, make_int3(support, support, support), prefactor, tau, INFINITY, 0.0, 0.0, 0.0};   // Poisson_ns::Gaussian::phi has no cut
  // near field cut-off and tables (.cu:105-118, :140-160).  The reference marches r from the far-field width in steps of gw / 1000 until
  // |G(r)| <= tolerance (tens of millions of erf pairs at gw = 1e-3); G decreases monotonically from there on, so the same first step is found by
  // marching a thousand steps at a time and then the last thousand one by one.
  if (split > 0) {
    const long double step = 0.001l * gw;
    auto above = [&](long double r) { return std::fabs(greens64((double)(r * r), gw, split, epsilon)) > tolerance; };
    long long nSteps = 0;
    {
      long long coarse = 0;
      while (above((long double)fw + (long double)((coarse + 1) * 1000) * step)) ++coarse;
      nSteps = coarse * 1000;
      do { ++nSteps; } while (above((long double)fw + (long double)nSteps * step));
    }
    p->cutoff = (double)((long double)fw + (long double)nSteps * step);
    if (p->cutoff > p->L[0] / 2.0) {
      set_last_error("[Poisson] Near field cut off is too large, increase splitting parameter.");
      delete p;
      return -2;
    }
    p->ntable = std::max(4096, (int)std::min((double)(1 << 16), p->cutoff / (gw * tolerance * 1e3)));
    const int Nm1 = p->ntable - 1;
    std::vector<double> tf(p->ntable), tp(p->ntable);
    const double rmaxF = p->cutoff, rmaxP = p->cutoff * p->cutoff;
    for (int i = 0; i <= Nm1; ++i) {  // TabulatedFunction ctor, misc/TabulatedFunction.cuh:103-117
      tf[i] = greens_field64((i / (double)Nm1) * rmaxF, gw, split, epsilon);
      tp[i] = greens64((i / (double)Nm1) * rmaxP, gw, split, epsilon);
    }
    int e = p->tableField.reserve(sizeof(double) * tf.size());
    if (!e) e = p->tablePotential.reserve(sizeof(double) * tp.size());
    if (e) { delete p; return e; }
    if (hipMemcpy(p->tableField.ptr, tf.data(), sizeof(double) * tf.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p->tablePotential.ptr, tp.data(), sizeof(double) * tp.size(), hipMemcpyHostToDevice) != hipSuccess) {
      set_last_error("uammd_poisson_create_f64: table upload failed");
      delete p;
      return -4;
    }
  }
  p->nxpad = 2 * (p->cells[0] / 2 + 1);
  p->planeReal = (size_t)p->nxpad * p->cells[1] * p->cells[2];
  p->planeCplx = (size_t)(p->cells[0] / 2 + 1) * p->cells[1] * p->cells[2];
  int e = p->gridQ.reserve(sizeof(double) * p->planeReal);
  if (!e) e = p->planes.reserve(sizeof(double) * 4 * p->planeReal);
  if (!e) e = poisson64_plans(p);
  if (e) { delete p; return e; }
  if (info) {
    for (int a = 0; a < 3; ++a) info->cells[a] = p->cells[a];
    info->support = support;
    info->nearFieldCutOff = p->cutoff;
    info->nTable = p->ntable;
    info->h = hx;
  }
  *out = reinterpret_cast<uammd_poisson_f64 *>(p);
  return 0;
}
int uammd_poisson_destroy_f64(uammd_poisson_f64 *h) {
  delete reinterpret_cast<Poisson64 *>(h);
  return 0;
}
int uammd_poisson_sum_f64(uammd_poisson_f64 *h, const double *d_pos, const double *d_charge, int N, double *d_force, double *d_energy, int nearForce,
                          int nearEnergy, void *stream) {
  if (!h) { set_last_error("uammd_poisson_sum_f64: null argument"); return -1; }
  if (N <= 0) return 0;
  if (!d_pos || !d_charge) { set_last_error("uammd_poisson_sum_f64: null argument"); return -1; }
  if ((nearForce && !d_force) || (nearEnergy && !d_energy)) { set_last_error("uammd_poisson_sum_f64: missing output array"); return -1; }
  Poisson64 *p = reinterpret_cast<Poisson64 *>(h);
  hipStream_t st = (hipStream_t)stream;
  if (int e = poisson64_far(p, d_pos, d_charge, N, d_force, d_energy, nullptr, st)) return e;
  if (p->par.split > 0) {
    if (nearForce) if (int e = poisson64_near<0>(p, d_pos, d_charge, N, d_force, st)) return e;
    if (nearEnergy) if (int e = poisson64_near<1>(p, d_pos, d_charge, N, d_energy, st)) return e;
  }
  return 0;
}
int uammd_poisson_field_potential_f64(uammd_poisson_f64 *h, const double *d_pos, const double *d_charge, int N, double *d_fieldPotential, double *d_force,
                                      double *d_energy, void *stream) {
  if (!h) { set_last_error("uammd_poisson_field_potential_f64: null argument"); return -1; }
  if (N <= 0) return 0;
  if (!d_pos || !d_charge || !d_fieldPotential) { set_last_error("uammd_poisson_field_potential_f64: null argument"); return -1; }
  Poisson64 *p = reinterpret_cast<Poisson64 *>(h);
  hipStream_t st = (hipStream_t)stream;
  if (int e = poisson64_far(p, d_pos, d_charge, N, d_force, d_energy, d_fieldPotential, st)) return e;
  if (p->par.split > 0) if (int e = poisson64_near<2>(p, d_pos, d_charge, N, d_fieldPotential, st)) return e;
  return 0;
}

}  // extern "C"
