/* ORACLE — TEST INFRASTRUCTURE ONLY (see common.h).
 *
 * Path B, Fourier-space part of the Force Coupling Method: CPU restatement of
 *   fcm_detail::indexToWaveNumber / waveNumberToWaveVector / getGradientFourier / projectFourier
 *                                                    Integrator/BDHI/FCM/utils.cuh:27-103
 *   fcm_detail::generateNoise / isNyquistWaveNumber   Integrator/BDHI/FCM/utils.cuh:117-167
 *   fcm_detail::forceFourier2Vel        (K16)         Integrator/BDHI/FCM/FCM_impl.cuh:375-397
 *   fcm_detail::fourierBrownianNoise    (K17)         Integrator/BDHI/FCM/FCM_impl.cuh:437-512
 *   addBrownianNoise prefactor                        Integrator/BDHI/FCM/FCM_impl.cuh:514-542
 *   FCM_impl::getSelfMobility (Hasimoto)              Integrator/BDHI/FCM/FCM_impl.cuh:102-119
 * The FFTs themselves (cuFFT batched R2C/C2R, FCM_impl.cuh:179-234) are a third-party dependency that
 * is not in /root/reference (CUDA toolkit, unpinned); a DFT has one answer up to rounding and the
 * oracle takes it from numpy/scipy pocketfft (oracle/fcm.py).
 *
 * Data layout: complex3 = {x.re, x.im, y.re, y.im, z.re, z.im} per Fourier node, node index
 * id = ikx + (nx/2+1) * (iky + ny * ikz)   (utils/cufftComplex3.cuh, FCM_impl.cuh:186-198).
 *
 * NOTE (reference behaviour, reproduced): on the kx = nx/2 plane (nx even) both a node and its
 * conjugate partner are "owners", so each receives its own draw plus the conjugate of its
 * partner's (FCM_impl.cuh:486-511); the device version does these two += from different threads
 * without atomics.  The oracle applies them sequentially in id order (no lost update).
 */
#include "common.h"
#include "saru.h"

typedef struct { real xr, xi, yr, yi, zr, zi; } complex3;

static inline int3 indexToWaveNumber(int i, int3 nk) { /* utils.cuh:27-35 */
  int ikx = i % (nk.x / 2 + 1);
  int iky = (i / (nk.x / 2 + 1)) % nk.y;
  int ikz = i / ((nk.x / 2 + 1) * nk.y);
  ikx -= nk.x * (ikx >= (nk.x / 2 + 1));
  iky -= nk.y * (iky >= (nk.y / 2 + 1));
  ikz -= nk.z * (ikz >= (nk.z / 2 + 1));
  return mki3(ikx, iky, ikz);
}
static inline real3 waveNumberToWaveVector(int3 ik, real3 L) { /* utils.cuh:37-39 */
  const real twopi = (real)2.0 * (real)M_PI;
  return mk3((twopi / L.x) * (real)ik.x, (twopi / L.y) * (real)ik.y, (twopi / L.z) * (real)ik.z);
}
static inline real3 getGradientFourier(int3 ik, int3 nk, real3 L) { /* utils.cuh:41-51 */
  const int ux = ik.x == (nk.x - ik.x), uy = ik.y == (nk.y - ik.y), uz = ik.z == (nk.z - ik.z);
  const real3 k = waveNumberToWaveVector(ik, L);
  return mk3(ux ? 0 : k.x, uy ? 0 : k.y, uz ? 0 : k.z);
}
static inline real3 projectFourierR(real k2, real3 dk, real3 fr) { /* utils.cuh:70-74 */
  const real invk2 = (real)1.0 / k2;
  const real s = dot3(fr, mk3(dk.x * invk2, dk.y * invk2, dk.z * invk2));
  return mk3(FMA(-dk.x, s, fr.x), FMA(-dk.y, s, fr.y), FMA(-dk.z, s, fr.z));
}
static inline complex3 projectFourierC(real k2, real3 dk, complex3 f) { /* utils.cuh:94-102 */
  real3 re = projectFourierR(k2, dk, mk3(f.xr, f.yr, f.zr));
  real3 im = projectFourierR(k2, dk, mk3(f.xi, f.yi, f.zi));
  complex3 r = {re.x, im.x, re.y, im.y, re.z, im.z};
  return r;
}
static inline complex3 c3scale(complex3 a, real s) {
  complex3 r = {a.xr * s, a.xi * s, a.yr * s, a.yi * s, a.zr * s, a.zi * s};
  return r;
}
static inline void c3add(complex3 *a, complex3 b) {
  a->xr += b.xr; a->xi += b.xi; a->yr += b.yr; a->yi += b.yi; a->zr += b.zr; a->zi += b.zi;
}

/* K16: FCM_impl.cuh:375-397, in place */
ORACLE_API void oracle_fcm_force_fourier_to_vel(real *grid6, real vis, const real *L3, const int *cellDim) {
  complex3 *g = (complex3 *)grid6;
  const int3 n = mki3(cellDim[0], cellDim[1], cellDim[2]);
  const real3 L = mk3(L3[0], L3[1], L3[2]);
  const int nk = n.z * n.y * (n.x / 2 + 1);
  memset(&g[0], 0, sizeof(complex3));
  const real norm = (real)(n.x * n.y * n.z);
#pragma omp parallel for schedule(static) if (oracle_get_parallel())
  for (int id = 1; id < nk; id++) {
    const int3 wn = indexToWaveNumber(id, n);
    const real3 k = waveNumberToWaveVector(wn, L);
    const real k2 = dot3(k, k);
    const real B = (real)1.0 / (vis * k2);
    const real3 dk = getGradientFourier(wn, n, L);
    g[id] = c3scale(projectFourierC(k2, dk, g[id]), B / norm);
  }
}

static inline complex3 generateNoise(real prefactor, uint id, uint seed1, uint seed2) { /* utils.cuh:117-131 */
  Saru saru = saru3(id, seed1, seed2);
  const real sc = (real)0.707106781186547 * prefactor;
  float a, b;
  complex3 n;
  saru_gf(&saru, 0, (float)sc, &a, &b); n.xr = a; n.xi = b;
  saru_gf(&saru, 0, (float)sc, &a, &b); n.yr = a; n.yi = b;
  saru_gf(&saru, 0, (float)sc, &a, &b); n.zr = a; n.zi = b;
  return n;
}
static inline int isNyquistWaveNumber(int3 cell, int3 nc) { /* utils.cuh:133-167 */
  const int X = (cell.x == nc.x - cell.x) && (nc.x % 2 == 0);
  const int Y = (cell.y == nc.y - cell.y) && (nc.y % 2 == 0);
  const int Z = (cell.z == nc.z - cell.z) && (nc.z % 2 == 0);
  return (X && cell.y == 0 && cell.z == 0) || (X && Y && cell.z == 0) || (cell.x == 0 && Y && cell.z == 0) ||
         (X && cell.y == 0 && Z) || (cell.x == 0 && cell.y == 0 && Z) || (cell.x == 0 && Y && Z) || (X && Y && Z);
}

/* K17: FCM_impl.cuh:437-512.  `prefactor` is the noisePrefactor of addBrownianNoise. */
static void fcm_noise_node(complex3 *g, int id, int3 nk, real3 L, real prefactor, real viscosity, uint seed1, uint seed2) {
  const int3 cell = mki3(id % (nk.x / 2 + 1), (id / (nk.x / 2 + 1)) % nk.y, id / ((nk.x / 2 + 1) * nk.y));
  if (id == 0 || (cell.x == 0 && cell.y == 0 && 2 * cell.z >= nk.z + 1) || (cell.x == 0 && 2 * cell.y >= nk.y + 1)) return;
  complex3 noise = generateNoise(prefactor, (uint)id, seed1, seed2);
  const int nyquist = isNyquistWaveNumber(cell, nk);
  if (nyquist) {
    const real nqsc = (real)1.41421356237310;
    noise.xr *= nqsc; noise.xi = 0;
    noise.yr *= nqsc; noise.yi = 0;
    noise.zr *= nqsc; noise.zi = 0;
  }
  {
    const int3 ik = indexToWaveNumber(id, nk);
    const real3 k = waveNumberToWaveVector(ik, L);
    const real k2 = dot3(k, k);
    const real B = (real)1.0 / (k2 * viscosity);
    complex3 factor = c3scale(noise, SQRT(B));
    const real3 dk = getGradientFourier(ik, nk, L);
    c3add(&g[id], projectFourierC(k2, dk, factor));
  }
  if (nyquist) return;
  if (cell.x == nk.x - cell.x || cell.x == 0) {
    const int xc = cell.x;
    const int yc = (cell.y > 0) * (nk.y - cell.y);
    const int zc = (cell.z > 0) * (nk.z - cell.z);
    const int id_conj = xc + (nk.x / 2 + 1) * (yc + zc * nk.y);
    const int3 ik = indexToWaveNumber(id_conj, nk);
    const real3 k = waveNumberToWaveVector(ik, L);
    const real k2 = dot3(k, k);
    const real B = (real)1.0 / (k2 * viscosity);
    const real Bsq = SQRT(B);
    complex3 factor = c3scale(noise, Bsq);
    factor.xi *= (real)(-1.0); factor.yi *= (real)(-1.0); factor.zi *= (real)(-1.0);
    const real3 dk = getGradientFourier(ik, nk, L);
    c3add(&g[id_conj], projectFourierC(k2, dk, factor));
  }
}

ORACLE_API void oracle_fcm_fourier_brownian_noise(real *grid6, const real *L3, const int *cellDim, real prefactor,
                                                  real viscosity, uint seed1, uint seed2) {
  complex3 *g = (complex3 *)grid6;
  const int3 nk = mki3(cellDim[0], cellDim[1], cellDim[2]);
  const real3 L = mk3(L3[0], L3[1], L3[2]);
  const int nxh = nk.x / 2 + 1;
  const int N = nk.z * nk.y * nxh;
  if (!oracle_get_parallel()) {  /* the reference's order: one loop over the nodes (what every parity test runs) */
    for (int id = 0; id < N; id++) fcm_noise_node(g, id, nk, L, prefactor, viscosity, seed1, seed2);
    return;
  }
  /* Parallel mode (bench.py's cpu_baseline leg only).  A node on the kx = 0 or kx = nx/2 plane also writes its conjugate partner.
   * Where the partner is one of the SKIPPED nodes no two iterations touch the same element; where it is not — the whole kx = nx/2 plane
   * (even nx) and the line kx = 0, ky = ny/2 (even ny) — a node and its partner both run and each adds to the other's element: a race
   * under `omp parallel for` (round-3 verdict).  The threads therefore leave out every node whose partner is active, and those nodes
   * are walked afterwards by one thread in the reference's order: the same sums, bit for bit, as the serial loop. */
#define FCM_NOISE_SKIPPED(i, c) ((i) == 0 || ((c).x == 0 && (c).y == 0 && 2 * (c).z >= nk.z + 1) || ((c).x == 0 && 2 * (c).y >= nk.y + 1))
#define FCM_NOISE_DEFERRED(id, out)                                                                        \
  do {                                                                                                     \
    const int3 c_ = mki3((id) % nxh, ((id) / nxh) % nk.y, (id) / (nxh * nk.y));                           \
    (out) = 0;                                                                                             \
    if (c_.x == 0 || c_.x == nk.x - c_.x) {                                                                \
      const int3 p_ = mki3(c_.x, (c_.y > 0) * (nk.y - c_.y), (c_.z > 0) * (nk.z - c_.z));                 \
      const int ip_ = p_.x + nxh * (p_.y + p_.z * nk.y);                                                   \
      (out) = ip_ != (id) && !FCM_NOISE_SKIPPED(ip_, p_);                                                  \
    }                                                                                                      \
  } while (0)
#pragma omp parallel for schedule(static)
  for (int id = 0; id < N; id++) {
    int deferred;
    FCM_NOISE_DEFERRED(id, deferred);
    if (!deferred) fcm_noise_node(g, id, nk, L, prefactor, viscosity, seed1, seed2);
  }
  for (int id = 0; id < N; id++) {
    if (id % nxh != 0 && id % nxh != nk.x - id % nxh) continue;
    int deferred;
    FCM_NOISE_DEFERRED(id, deferred);
    if (deferred) fcm_noise_node(g, id, nk, L, prefactor, viscosity, seed1, seed2);
  }
#undef FCM_NOISE_DEFERRED
#undef FCM_NOISE_SKIPPED
}

/* addBrownianNoise: noisePrefactor = prefactor * sqrt(fourierNormalization*2*T/dV), FCM_impl.cuh:526-532 */
ORACLE_API real oracle_fcm_noise_prefactor(real prefactor, real temperature, const real *L3, const int *cellDim) {
  Box box = box_make(mk3(L3[0], L3[1], L3[2]));
  Grid g = grid_make(box, mki3(cellDim[0], cellDim[1], cellDim[2]));
  const real dV = g.cellVolume;
  const real fourierNormalization = (real)(1.0 / ((double)cellDim[0] * cellDim[1] * cellDim[2]));
  /* host code, all in `real`: std::sqrt(real) */
  return prefactor * SQRT(fourierNormalization * 2 * temperature / dV);
}

/* FCM_impl::getSelfMobility / test selfMobility(): FCM_impl.cuh:102-119, test/BDHI/FCM/fcm_test.cu:64-81 */
ORACLE_API double oracle_fcm_self_mobility(double hydrodynamicRadius, double viscosity, double Lx) {
  long double rh = hydrodynamicRadius;
  long double L = Lx;
  long double a = rh / L;
  long double a2 = a * a;
  long double a3 = a2 * a;
  long double c = 2.83729747948061947666591710460773907l;
  long double b = 0.19457l;
  long double pi = 3.141592653589793238462643383279502884L;
  long double a6pref = 16.0l * pi * pi / 45.0l + 630.0L * b * b;
  return (double)(1.0l / (6.0l * pi * viscosity * rh) * (1.0l - c * a + (4.0l / 3.0l) * pi * a3 - a6pref * a3 * a3));
}

/* ---- torques / rotation ------------------------------------------------------------------------------------
 *   FCM_ns::Kernels::GaussianTorque(width, h, tolerance)     Integrator/BDHI/FCM/FCM_kernels.cuh:60-80
 *   detail::initializeKernelTorque (width = a/(6 sqrt(pi))^(1/3))  Integrator/BDHI/BDHI_FCM.cuh:69-80
 *   addTorqueCurl / computeVelocityCurlFourier               Integrator/BDHI/FCM/FCM_impl.cuh:306-327, :590-617
 *   FCM_ns::integrateEulerMaruyamaD with orientations        Integrator/BDHI/BDHI_FCM.cu:67-92, utils/quaternion.cuh:179-196
 */
ORACLE_API int oracle_fcm_torque_gaussian_init(real hydrodynamicRadius, real h, real tolerance, real *out3) {
  const real width = (real)(hydrodynamicRadius / (pow(6 * sqrt(M_PI), 1 / 3.)));
  const real prefactor = (real)pow(2.0 * M_PI * (double)width * (double)width, -0.5);
  const real tau = (real)(-0.5 / ((double)width * (double)width));
  const real dr = (real)(0.5 * h);
  real r = dr;
  while (prefactor * EXP(tau * r * r) > tolerance) r += dr;
  int support = (int)(2 * r / h + 0.5);
  if (support < 3) support = 3;
  out3[0] = prefactor; out3[1] = tau; out3[2] = (real)support * h;
  return support;
}

/* half * i dk x g : the operator shared by addTorqueCurl (accumulate = 1: out += ...) and computeVelocityCurlFourier
 * (accumulate = 0: out = ...).  Component expressions exactly as written in the reference. */
ORACLE_API void oracle_fcm_half_curl_fourier(const real *in6, real *out6, const real *L3, const int *cellDim, int accumulate) {
  const complex3 *in = (const complex3 *)in6;
  complex3 *out = (complex3 *)out6;
  const int3 n = mki3(cellDim[0], cellDim[1], cellDim[2]);
  const real3 L = mk3(L3[0], L3[1], L3[2]);
  const int nk = n.z * n.y * (n.x / 2 + 1);
  const real half = (real)0.5;
  for (int id = 0; id < nk; id++) {
    const int3 ik = indexToWaveNumber(id, n);
    const real3 dk = getGradientFourier(ik, n, L);
    const complex3 g = in[id];
    complex3 c;
    c.xr = half * FMA(-dk.y, g.zi, dk.z * g.yi); c.xi = half * FMA(dk.y, g.zr, -(dk.z * g.yr));
    c.yr = half * FMA(-dk.z, g.xi, dk.x * g.zi); c.yi = half * FMA(dk.z, g.xr, -(dk.x * g.zr));
    c.zr = half * FMA(-dk.x, g.yi, dk.y * g.xi); c.zi = half * FMA(dk.x, g.yr, -(dk.y * g.xr));
    if (accumulate) c3add(&out[id], c);
    else out[id] = c;
  }
}
