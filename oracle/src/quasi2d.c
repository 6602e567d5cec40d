/* ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the product.
 *
 * CPU restatement of the 2D / quasi-2D hydrodynamics integrator, Integrator/Hydro/BDHI_quasi2D.cu(h):
 *   BDHI2D_ns::True2D / Quasi2D (f_k, g_k, Gaussian variance)     BDHI_quasi2D.cuh:83-110
 *   initializeGrid / initializeInterpolationKernel                BDHI_quasi2D.cu:61-88
 *   cellToWaveNumber, projectFourier, forceFourier2Vel            :314-366
 *   fourierBrownianNoise                                          :368-432
 * mode 0 = True2D, 1 = Quasi2D.
 */
#include "common.h"
#include "saru.h"

#ifdef DOUBLE_PRECISION
#define ERFC(a) erfc(a)
#else
#define ERFC(a) erfcf(a)
#endif

ORACLE_API real oracle_q2d_gaussian_variance(int mode, real a) { /* .cuh:84-86, :97-99 */
  if (mode == 0) return (real)pow(a * 0.66556976637237890625, 2);
  return (real)pow(a / sqrt(M_PI), 2);
}

static inline void hydro_kernel(int mode, real k2, real a, real *fk, real *gk) { /* .cuh:88-92, :100-109 */
  if (mode == 0) {
    *fk = 0;
    *gk = (real)1.0 / (k2 * k2);
  } else {
    const real k = SQRT(k2);
    const real invk3 = (real)1.0 / (k2 * k);
    const real inv_sqrtpi = (real)0.564189583547756;
    const real kp = k * a * inv_sqrtpi;
    *fk = (real)0.5 * invk3 * (ERFC(kp) * ((real)0.5 + kp * kp) * EXP(kp * kp) - kp * inv_sqrtpi);
    *gk = (real)0.5 * invk3 * ERFC(kp) * EXP(kp * kp);
  }
}
ORACLE_API void oracle_q2d_hydro_kernel(int mode, real k2, real a, real *out2) { hydro_kernel(mode, k2, a, out2, out2 + 1); }

/* grid and window: raw cell counts (before nextFFTWiseSize3D), then support and the three prefactors / tau */
ORACLE_API void oracle_q2d_raw_cells(const real *boxL, real a, int *cells2) {
  const double h = a * 0.8;
  const real hr = (real)h;
  cells2[0] = (int)(boxL[0] / hr);
  cells2[1] = (int)(boxL[1] / hr);
}
ORACLE_API void oracle_q2d_window(int mode, const real *boxL, const int *cells2, real a, int *support, real *prefactor,
                                  real *prefactorDrift, real *tau) {
  int s = ((int)(3.0 * a * cells2[0] / boxL[0]) + 1) * 2 + 1;
  if (s > cells2[0]) s = cells2[0];
  *support = s;
  const double width = oracle_q2d_gaussian_variance(mode, a);
  const real w = (real)width; /* the window constructors take a real */
  *prefactor = (real)sqrt(1.0 / (2.0 * M_PI * w));
  *tau = (real)(-1.0 / (2.0 * w));
  *prefactorDrift = (real)(-sqrt(1.0 / (2.0 * M_PI * w * w)));
}

static inline real2 wave_number(int ix, int iy, int nx, int ny, const real *L) {
  const real px = ((real)2.0 * (real)M_PI) / L[0], py = ((real)2.0 * (real)M_PI) / L[1];
  real2 k = {(real)(ix - nx * (ix >= (nx / 2 + 1))) * px, (real)(iy - ny * (iy >= (ny / 2 + 1))) * py};
  return k;
}
/* G_k * factor (Algorithm 1.3 of the paper) for one real 2-vector */
static inline real2 project(real2 k, real2 f, real fk, real gk) {
  const real dperp = FMA(f.y, -k.x, f.x * k.y); /* dot(f, {k.y, -k.x}) */
  const real dpar = FMA(f.y, k.y, f.x * k.x);   /* dot(f, k) */
  real2 v = {FMA(k.x * fk, dpar, k.y * gk * dperp), FMA(k.y * fk, dpar, -k.x * gk * dperp)};
  return v;
}

/* forceFourier2Vel in place: grid4 = complex2[ny][nx/2+1] as (x.re, x.im, y.re, y.im) */
ORACLE_API void oracle_q2d_force_fourier_to_vel(real *grid4, int mode, real viscosity, const real *L, const int *cells2, real a) {
  const int nx = cells2[0], ny = cells2[1], nkx = nx / 2 + 1;
  for (int id = 0; id < ny * nkx; id++) {
    real *g = grid4 + 4 * (size_t)id;
    if (id == 0) { g[0] = g[1] = g[2] = g[3] = 0; continue; }
    const int ix = id % nkx, iy = id / nkx;
    const real2 k = wave_number(ix, iy, nx, ny, L);
    const real k2 = FMA(k.y, k.y, k.x * k.x);
    real fk, gk;
    hydro_kernel(mode, k2, a, &fk, &gk);
    fk = fk / (viscosity * (real)(nx * ny));
    gk = gk / (viscosity * (real)(nx * ny));
    const real2 re = {g[0], g[2]}, im = {g[1], g[3]};
    const real2 vr = project(k, re, fk, gk), vi = project(k, im, fk, gk);
    g[0] = vr.x; g[2] = vr.y; g[1] = vi.x; g[3] = vi.y;
  }
}

/* fourierBrownianNoise: scatter form as the reference, except that the conjugate of the (nx/2, 0) node — which the
 * reference writes one row past the end of the array (indexOfConjugate with ik.y = 0) — is dropped. */
ORACLE_API void oracle_q2d_fourier_brownian_noise(real *grid4, int mode, const real *L, const int *cells2, real prefactor, real a,
                                                  uint seed, uint step) {
  const int nx = cells2[0], ny = cells2[1], nkx = nx / 2 + 1;
  for (int id = 0; id < ny * nkx; id++) {
    real *g = grid4 + 4 * (size_t)id;
    const int ix = id % nkx, iy = id / nkx;
    if (id == 0) { g[0] = g[1] = g[2] = g[3] = 0; continue; }
    if (ix == 0 && iy > (ny - iy)) continue;
    if (ix == nx - ix && iy > (ny - iy)) continue;
    const int isXnyquist = (ix == (nx - ix)) && (nx % 2 == 0);
    const int isYnyquist = (iy == (ny - iy)) && (ny % 2 == 0);
    if (isXnyquist && iy == 0) { g[0] = g[1] = g[2] = g[3] = 0; }
    const int isNyquist = (isYnyquist && ix == 0) || (isXnyquist && isYnyquist);
    Saru rng = saru3((uint)id, step, seed);
    const real sc = (real)0.707106781186547 * prefactor;
    real2 nx_, ny_; /* complex noise for the perpendicular / parallel parts (gf draws floats in either build) */
    {
      float a0, a1;
      saru_gf(&rng, 0, (float)sc, &a0, &a1); nx_.x = a0; nx_.y = a1;
      saru_gf(&rng, 0, (float)sc, &a0, &a1); ny_.x = a0; ny_.y = a1;
    }
    if (isNyquist) {
      nx_.x *= (real)1.41421356237310; ny_.x *= (real)1.41421356237310;
      nx_.y = 0; ny_.y = 0;
    }
    const real2 k = wave_number(ix, iy, nx, ny, L);
    const real k2 = FMA(k.y, k.y, k.x * k.x);
    real fk, gk;
    hydro_kernel(mode, k2, a, &fk, &gk);
    const real fs = SQRT(fk), gs = SQRT(gk);
    /* factor.x = gs*noise.x*k.y + fs*noise.y*k.x ; factor.y = gs*noise.x*(-k.x) + fs*noise.y*k.y  (complex each) */
    const real fxr = FMA(fs * ny_.x, k.x, gs * nx_.x * k.y), fxi = FMA(fs * ny_.y, k.x, gs * nx_.y * k.y);
    const real fyr = FMA(fs * ny_.x, k.y, gs * nx_.x * (-k.x)), fyi = FMA(fs * ny_.y, k.y, gs * nx_.y * (-k.x));
    g[0] += fxr; g[1] += fxi; g[2] += fyr; g[3] += fyi;
    if (isNyquist) continue;
    if (ix == (nx - ix) || ix == 0) {
      if (iy == 0) continue; /* reference: out-of-bounds write */
      real *c = grid4 + 4 * (size_t)(ix + nkx * (ny - iy));
      c[0] += fxr; c[1] += -fxi; c[2] += fyr; c[3] += -fyi;
    }
  }
}
