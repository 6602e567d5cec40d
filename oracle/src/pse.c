/* ORACLE — TEST INFRASTRUCTURE ONLY (see common.h).
 *
 * Positively Split Ewald RPY: CPU restatement of
 *   RPYPSE_near::FandG / params2FG                       Integrator/BDHI/PSE/RPY_PSE.cuh:45-128   (host double, erfc/exp)
 *   TabulatedFunction ctor / operator() / LinearInterpolation / lerp
 *                                                        misc/TabulatedFunction.cuh:36-45, :63-75, :104-115, :148-157
 *   NearField::initializeDeterministicPart (rcut, table size)   Integrator/BDHI/PSE/NearField.cuh:65-99
 *   RPYNearTransverser::{computeShearedDistancePBC, compute, set}   NearField.cuh:134-185
 *   SaruTransform                                        NearField.cuh:218-228
 *   pse_ns::Kernel (Gaussian window)                     Integrator/BDHI/PSE/FarField.cuh:25-41
 *   projectFourier / greensFunction / forceFourier2Vel / fourierBrownianNoise   FarField.cuh:44-119, :137-158, :235-308
 *   FarField::initializeKernel / initializeGrid          FarField.cuh:605-654
 *   shearWaveVector, waveNumberToWaveVector              Integrator/BDHI/PSE/utils.cuh:26-43
 * The FFTs (cuFFT, third party) come from numpy/scipy in oracle/pse.py; the noise kernel is applied sequentially in id
 * order (the reference races on the kx = nx/2 plane, as FCM does).
 * FMA contract for the near-field pair: f*vj + gmfv*rij  ->  FMA(gmfv, rij.k, f*vj.k); accumulate = plain add.
 */
#include "common.h"
#include "saru.h"

#include <math.h>

/* ---- near field ------------------------------------------------------------------------------------------------ */
static double params2FG(double rh, double psi, double r, const double f[8]) { /* RPY_PSE.cuh:117-128 */
  const double psisq = psi * psi;
  const double a2mr = 2 * rh - r;
  const double a2pr = 2 * rh + r;
  const double rsq = r * r;
  return f[0] + f[1] * exp(-psisq * a2pr * a2pr) + f[2] * exp(-a2mr * a2mr * psisq) + f[3] * exp(-psisq * rsq) +
         f[4] * erfc(a2mr * psi) + f[5] * erfc(-a2mr * psi) + f[6] * erfc(a2pr * psi) + f[7] * erfc(r * psi);
}

ORACLE_API void oracle_pse_rpy_near_fandg(double r, double rh, double psi, double rcut, double *out2) { /* RPY_PSE.cuh:45-115 */
  const double r2 = r * r;
  if (r >= rcut) { out2[0] = out2[1] = 0.0; return; }
  if (r <= 0.0) {
    const double pi = M_PI;
    out2[0] = (1.0 / (4 * sqrt(pi) * psi * rh)) * (1 - exp(-4 * rh * rh * psi * psi) + 4 * sqrt(pi) * rh * psi * erfc(2 * rh * psi));
    out2[1] = 0;
    return;
  }
  const double a2mr = 2 * rh - r, a2pr = 2 * rh + r;
  const double rh2 = rh * rh, rh4 = rh2 * rh2;
  const double psi2 = psi * psi, psi3 = psi2 * psi, psi4 = psi2 * psi2;
  const double r3 = r2 * r, r4 = r3 * r;
  double f[8], g[8];
  if (r > 2 * rh) {
    f[0] = (64.0 * rh4 * psi4 + 96.0 * rh2 * r2 * psi4 - 128.0 * rh * r3 * psi4 + 36.0 * r4 * psi4 - 3.0) / (128.0 * rh * r3 * psi4);
    f[4] = (3.0 - 4.0 * psi4 * a2mr * a2mr * (4.0 * rh2 + 4.0 * rh * r + 9.0 * r2)) / (256.0 * rh * r3 * psi4);
    f[5] = 0;
    g[0] = (-64.0 * rh4 * psi4 + 96.0 * rh2 * r2 * psi4 - 64.0 * rh * r3 * psi4 + 12.0 * r4 * psi4 + 3.0) / (64.0 * rh * r3 * psi4);
    g[4] = (4.0 * psi4 * a2mr * a2mr * a2mr * (2.0 * rh + 3.0 * r) - 3.0) / (128.0 * rh * r3 * psi4);
    g[5] = 0;
  } else {
    f[0] = (-16.0 * rh4 - 24.0 * rh2 * r2 + 32.0 * rh * r3 - 9.0 * r4) / (32.0 * rh * r3);
    f[4] = 0;
    f[5] = (4.0 * psi4 * a2mr * a2mr * (4.0 * rh2 + 4.0 * rh * r + 9.0 * r2) - 3.0) / (256.0 * rh * r3 * psi4);
    g[0] = a2mr * a2mr * a2mr * (2.0 * rh + 3.0 * r) / (16.0 * rh * r3);
    g[4] = 0;
    g[5] = (3.0 - 4.0 * psi4 * a2mr * a2mr * a2mr * (2.0 * rh + 3.0 * r)) / (128.0 * rh * r3 * psi4);
  }
  f[1] = (-2.0 * psi2 * a2pr * (4.0 * rh2 - 4.0 * rh * r + 9.0 * r2) + 2.0 * rh - 3.0 * r) / (128.0 * rh * r3 * psi3 * sqrt(M_PI));
  f[2] = (2.0 * psi2 * a2mr * (4.0 * rh2 + 4.0 * rh * r + 9.0 * r2) - 2.0 * rh - 3.0 * r) / (128.0 * rh * r3 * psi3 * sqrt(M_PI));
  f[3] = 3.0 * (6.0 * r2 * psi2 + 1.0) / (64.0 * sqrt(M_PI) * rh * r2 * psi3);
  f[6] = (4.0 * psi4 * a2pr * a2pr * (4.0 * rh2 - 4.0 * rh * r + 9.0 * r2) - 3.0) / (256.0 * rh * r3 * psi4);
  f[7] = 3.0 * (1.0 - 12.0 * r4 * psi4) / (128.0 * rh * r3 * psi4);
  g[1] = (2.0 * psi2 * a2pr * a2pr * (2.0 * rh - 3.0 * r) - 2.0 * rh + 3.0 * r) / (64.0 * sqrt(M_PI) * rh * r3 * psi3);
  g[2] = (-2.0 * psi2 * a2mr * a2mr * (2.0 * rh + 3.0 * r) + 2.0 * rh + 3.0 * r) / (64.0 * sqrt(M_PI) * rh * r3 * psi3);
  g[3] = (3.0 * (2.0 * r2 * psi2 - 1.0)) / (32.0 * sqrt(M_PI) * rh * r2 * psi3);
  g[6] = (3.0 - 4.0 * psi4 * (2.0 * rh - 3.0 * r) * a2pr * a2pr * a2pr) / (128.0 * rh * r3 * psi4);
  g[7] = -3.0 * (4.0 * r4 * psi4 + 1.0) / (64.0 * rh * r3 * psi4);
  out2[0] = params2FG(rh, psi, r, f);
  out2[1] = params2FG(rh, psi, r, g);
}

/* NearField.cuh:65-99.  The reference stores these in `real` members / arguments: rcut is real, RPYPSE_near takes
 * (real rh, real psi, real normalization, real rcut).  Returns nPointsTable; *rcut_out = the real cut-off. */
ORACLE_API int oracle_pse_near_setup(real hydrodynamicRadius, real psi, real tolerance, real *rcut_out) {
  const double split = psi;
  const real rcut = (real)(sqrt(-log(tolerance)) / split);
  *rcut_out = rcut;
  const double a = hydrodynamicRadius;
  const real textureTolerance = (real)(a * tolerance);
  const unsigned maximumTextureElements = 1u << 22;
  double np = rcut / textureTolerance + 0.5;
  if (np > 2e30) np = 2e30;
  unsigned nPointsTable = (np >= 4294967295.0) ? 4294967295u : (unsigned)np;
  if (nPointsTable < (1u << 14)) nPointsTable = 1u << 14;
  if (nPointsTable > maximumTextureElements) nPointsTable = maximumTextureElements;
  return (int)nPointsTable;
}

/* TabulatedFunction<real2>(table, N = nPointsTable, 0, rcut, rpy): Ntable = N-1, entries 0..Ntable (TabulatedFunction.cuh:104-115) */
ORACLE_API void oracle_pse_near_table(real hydrodynamicRadius, real psi, real viscosity, real rcut, int nPointsTable, real *table2) {
  const int Ntable = nPointsTable - 1;
  const real rmin = 0, rmax = rcut;
  const double a = hydrodynamicRadius;
  const real normalization = (real)(6 * M_PI * a * viscosity);
  for (int i = 0; i <= Ntable; i++) {
    const double x = (i / (double)(Ntable)) * (rmax - rmin) + rmin;
    double fg[2];
    oracle_pse_rpy_near_fandg(x, (double)hydrodynamicRadius, (double)psi, (double)rcut, fg);
    table2[2 * i] = (real)(fg[0] / (double)normalization);
    table2[2 * i + 1] = (real)(fg[1] / (double)normalization);
  }
}

typedef struct { real x, y; } real2_t;
static inline real lerp1(real v0, real v1, real t) { return FMA(t, v1, FMA(-t, v0, v0)); } /* TabulatedFunction.cuh:36-45 */
/* TabulatedFunction::operator() + LinearInterpolation (:63-75, :148-157) */
static inline real2_t table_get(const real2_t *table, int Ntable, real rmin, real rmax, real rs) {
  const real interval = (real)(1.0 / (rmax - rmin));
  const real dr = (real)(1.0 / (real)Ntable);
  real2_t zero = {0, 0};
  const real r = (rs - rmin) * interval;
  if (rs >= rmax) return zero;
  if (r <= (real)0.0) return table[0];
  const int i = (int)(r * Ntable);
  const real r0 = i * dr;
  const real2_t v0 = table[i], v1 = table[i + 1];
  const real t = (r - r0) * (real)Ntable;
  real2_t out = {lerp1(v0.x, v1.x, t), lerp1(v0.y, v1.y, t)};
  return out;
}

ORACLE_API void oracle_pse_table_get(const real *table2, int nPointsTable, real rcut, const real *rs, int n, real *out2) {
  for (int i = 0; i < n; i++) {
    real2_t v = table_get((const real2_t *)table2, nPointsTable - 1, 0, rcut, rs[i]);
    out2[2 * i] = v.x; out2[2 * i + 1] = v.y;
  }
}

static inline real3 sheared_distance_pbc(real3 pi, real3 pj, real3 L, real shearStrain) { /* NearField.cuh:134-152 */
  real3 rij = mk3(pj.x - pi.x, pj.y - pi.y, pj.z - pi.z);
  rij.x = FMA(shearStrain, rij.y, rij.x);
  const real s1 = ROUND(rij.y / L.y);
  rij.x = FMA(-(shearStrain * L.y), s1, rij.x);
  rij.y = FMA(-L.y, s1, rij.y);
  rij.z = FMA(-L.z, ROUND(rij.z / L.z), rij.z);
  rij.x = FMA(-L.x, ROUND(rij.x / L.x), rij.x);
  return rij;
}

/* transverseWithNeighbourContainer<RPYNearTransverser<vtype>> over a built cell list (common.cuh:10-34).  v is indexed with
 * the GROUP index of the neighbour (getInfo), Mv[ori] += total (set).  vstride = 3 (real3) or 4 (real4 forces). */
ORACLE_API void oracle_pse_near_mdot(const real4 *sortPos, const int *groupIndex, int N, const uint *cellStart, const int *cellEnd,
                                     uint validCell, const real *gridL, const int *gridPeriodic, const int *cellDim,
                                     const real *boxL, real shearStrain, real rcut, const real *table2, int nPointsTable,
                                     const real *v, int vstride, real *Mv3) {
  Box gbox = box_from(gridL, gridPeriodic);
  Grid grid = grid_make(gbox, mki3(cellDim[0], cellDim[1], cellDim[2]));
  const real3 L = mk3(boxL[0], boxL[1], boxL[2]);
  const real rcut2 = rcut * rcut;
  const int Ntable = nPointsTable - 1;
  const real2_t *table = (const real2_t *)table2;
  const int3 n = grid.cellDim;
  const int3 nperdim = mki3((n.x > 1 ? 3 : 1), (n.y > 1 ? 3 : 1), (n.z > 1 ? 3 : 1));
  const int numberNeighbourCells = nperdim.x * nperdim.y * nperdim.z;
#pragma omp parallel for schedule(static)
  for (int id = 0; id < N; id++) {
    const int ori = groupIndex[id];
    const real4 pi = sortPos[id];
    real3 total = mk3(0, 0, 0);
    const int3 celli = grid_get_cell(&grid, mk3(pi.x, pi.y, pi.z));
    for (int currentCell = 0; currentCell < numberNeighbourCells; currentCell++) {
      int3 cellj = celli;
      if (nperdim.x > 1) cellj.x += currentCell % 3 - 1;
      if (nperdim.y > 1) cellj.y += (currentCell / nperdim.x) % 3 - 1;
      if (nperdim.z > 1) cellj.z += currentCell / (nperdim.x * nperdim.y) - 1;
      cellj = grid_pbc_cell(&grid, cellj);
      if (cellj.x < 0 || cellj.x >= n.x || cellj.y < 0 || cellj.y >= n.y || cellj.z < 0 || cellj.z >= n.z) continue;
      const int icellj = grid_cell_index(&grid, cellj);
      const uint cs = cellStart[icellj];
      if (cs < validCell) continue;
      const int first = (int)(cs - validCell), last = cellEnd[icellj];
      for (int j = first; j < last; j++) {
        const real4 pj = sortPos[j];
        const real *vp = v + (size_t)vstride * groupIndex[j];
        const real3 vj = mk3(vp[0], vp[1], vp[2]);
        const real3 rij = sheared_distance_pbc(mk3(pi.x, pi.y, pi.z), mk3(pj.x, pj.y, pj.z), L, shearStrain);
        const real r2 = dot3(rij, rij);
        if (r2 >= rcut2) continue; /* returns real3(): adds zero */
        const real2_t fg = table_get(table, Ntable, 0, rcut, SQRT(r2));
        const real f = fg.x, g = fg.y;
        real3 res;
        if (r2 == (real)0.0) {
          res = mk3(f * vj.x, f * vj.y, f * vj.z);
        } else {
          const real invr2 = (real)1.0 / r2;
          const real gmfv = (g - f) * dot3(rij, vj) * invr2;
          res = mk3(FMA(gmfv, rij.x, f * vj.x), FMA(gmfv, rij.y, f * vj.y), FMA(gmfv, rij.z, f * vj.z));
        }
        total.x += res.x; total.y += res.y; total.z += res.z;
      }
    }
    Mv3[3 * ori] += total.x; Mv3[3 * ori + 1] += total.y; Mv3[3 * ori + 2] += total.z;
  }
}

/* SaruTransform (NearField.cuh:218-228): noise_i = make_real3(gf(0,1), gf(0,1).x) * variance, left-to-right evaluation */
ORACLE_API void oracle_pse_near_noise(int N, real variance, uint seed1, uint seed2, real *out3) {
  for (int i = 0; i < N; i++) {
    Saru rng = saru3((uint)i, seed1, seed2);
    float a0, a1, b0, b1;
    saru_gf(&rng, 0, 1, &a0, &a1);
    saru_gf(&rng, 0, 1, &b0, &b1);
    out3[3 * i] = (real)a0 * variance; out3[3 * i + 1] = (real)a1 * variance; out3[3 * i + 2] = (real)b0 * variance;
  }
}

/* ---- far field ---------------------------------------------------------------------------------------------------- */
/* FarField::initializeGrid (:646-654) + initializeKernel (:605-644) + Kernel ctor (:25-41).
 * out: cells[3] BEFORE nextFFTWiseSize3D (the caller applies it), then with the final cells: support, eta, prefactor, tau. */
ORACLE_API void oracle_pse_far_raw_cells(const real *boxL, real psi, real tolerance, int *cells3) {
  const real kcut = (real)(2 * psi * sqrt(-log(tolerance)));
  const double hgrid = 2 * M_PI / kcut;
  for (int k = 0; k < 3; k++) cells3[k] = (int)(2 * boxL[k] / hgrid) + 1; /* make_int3(real3) truncates, then + 1 */
}
ORACLE_API void oracle_pse_far_kernel(const real *boxL, const int *cells3, real psi, real tolerance, int *support_out,
                                      real *eta_out, real *prefactor_out, real *tau_out) {
  const double C = 0.976;
  double m = 1;
  while (erfc(m / sqrt(2)) > 0.1 * tolerance) m += 0.01;
  int support;
  while ((support = (int)(pow(m / C, 2) / M_PI + 0.5) + 1) % 2 == 0) m += tolerance;
  int P = support / 2;
  int minCellDim = cells3[0] < cells3[1] ? cells3[0] : cells3[1];
  if (cells3[2] < minCellDim) minCellDim = cells3[2];
  if (support > minCellDim) {
    support = minCellDim;
    if (support % 2 == 0) support--;
    P = support / 2;
    m = C * sqrt(M_PI * support);
  }
  const double pw = 2 * P + 1;
  real cs[3];
  for (int k = 0; k < 3; k++) cs[k] = boxL[k] / (real)cells3[k]; /* Grid::cellSize */
  double h = cs[0] < cs[1] ? cs[0] : cs[1];
  if (cs[2] < h) h = cs[2];
  const double gaussM = m;
  const double w = pw * h / 2.0;
  const real eta = (real)pow(2.0 * psi * w / gaussM, 2);
  const real width = (real)(sqrt(eta) / (2.0 * psi)); /* Kernel(int P, real width) */
  *support_out = 2 * P + 1;
  *eta_out = eta;
  *prefactor_out = (real)cbrt(1.0 / (width * width * width * pow(2.0 * M_PI, 1.5)));
  *tau_out = (real)(-0.5 / (width * width));
}

typedef struct { real xr, xi, yr, yi, zr, zi; } pcomplex3;

static inline int3 pse_indexToWaveNumber(int i, int3 nk) {
  int ikx = i % (nk.x / 2 + 1);
  int iky = (i / (nk.x / 2 + 1)) % nk.y;
  int ikz = i / ((nk.x / 2 + 1) * nk.y);
  ikx -= nk.x * (ikx >= (nk.x / 2 + 1));
  iky -= nk.y * (iky >= (nk.y / 2 + 1));
  ikz -= nk.z * (ikz >= (nk.z / 2 + 1));
  return mki3(ikx, iky, ikz);
}
static inline real3 pse_waveVector(int3 ik, real3 L) { /* utils.cuh:39-42: (2 pi / L) * ik */
  const real twopi = (real)2.0 * (real)M_PI;
  return mk3((twopi / L.x) * (real)ik.x, (twopi / L.y) * (real)ik.y, (twopi / L.z) * (real)ik.z);
}
static inline real3 pse_shear(real3 k, real shearStrain) { k.y = FMA(-shearStrain, k.x, k.y); return k; }

static inline pcomplex3 pse_project(real3 k, pcomplex3 f) { /* FarField.cuh:53-73 */
  const real invk2 = (real)1.0 / dot3(k, k);
  pcomplex3 r;
  {
    const real3 fr = mk3(f.xr, f.yr, f.zr);
    const real kfr = dot3(k, fr) * invk2;
    r.xr = FMA(-k.x, kfr, fr.x); r.yr = FMA(-k.y, kfr, fr.y); r.zr = FMA(-k.z, kfr, fr.z);
  }
  {
    const real3 fi = mk3(f.xi, f.yi, f.zi);
    const real kfi = dot3(k, fi) * invk2;
    r.xi = FMA(-k.x, kfi, fi.x); r.yi = FMA(-k.y, kfi, fi.y); r.zi = FMA(-k.z, kfi, fi.z);
  }
  return r;
}
static inline pcomplex3 pscale(pcomplex3 a, real s) {
  pcomplex3 r = {a.xr * s, a.xi * s, a.yr * s, a.yi * s, a.zr * s, a.zi * s};
  return r;
}

static inline real pse_greens(real3 waveVector, real shearStrain, real rh, real viscosity, real split, real eta, int3 n) { /* :85-119 */
  const real k2 = dot3(waveVector, waveVector);
  if (k2 == 0) return 0;
  const real3 K_NUFFT = waveVector;
  const real3 K_Ewald = pse_shear(waveVector, shearStrain);
  const real K_Ewald2 = dot3(K_Ewald, K_Ewald);
  const real K_NUFFT2 = dot3(K_NUFFT, K_NUFFT);
  const real kmod = SQRT(K_Ewald2);
  const real invk2 = (real)1.0 / K_Ewald2;
  const real sink = SIN(kmod * rh);
  const real kEw2_invsplit2_4 = K_Ewald2 / ((real)4.0 * split * split);
  const real kNU2_invsplit2_4 = K_NUFFT2 / ((real)4.0 * split * split);
  const real tau = FMA(eta, kNU2_invsplit2_4, -kEw2_invsplit2_4);
  const real hashimoto = ((real)1.0 + kEw2_invsplit2_4) * EXP(tau) / K_Ewald2;
  real B = sink * sink * invk2 * hashimoto / (viscosity * rh * rh);
  B /= (real)(n.x * n.y * n.z);
  return B;
}

/* forceFourier2Vel (FarField.cuh:137-158), in place, complex3 interleaved */
ORACLE_API void oracle_pse_force_fourier_to_vel(real *grid6, real shearStrain, real rh, real viscosity, real split, real eta,
                                                const real *L3, const int *cellDim) {
  pcomplex3 *g = (pcomplex3 *)grid6;
  const int3 n = mki3(cellDim[0], cellDim[1], cellDim[2]);
  const real3 L = mk3(L3[0], L3[1], L3[2]);
  const int nk = n.z * n.y * (n.x / 2 + 1);
  memset(&g[0], 0, sizeof(pcomplex3));
  for (int id = 1; id < nk; id++) {
    const int3 wn = pse_indexToWaveNumber(id, n);
    const real3 k = pse_waveVector(wn, L);
    const real B = pse_greens(k, shearStrain, rh, viscosity, split, eta, n);
    g[id] = pse_project(pse_shear(k, shearStrain), pscale(g[id], B));
  }
}

static inline int pse_is_nyquist(int3 c, int3 n) {
  const int X = (c.x == n.x - c.x) && (n.x % 2 == 0), Y = (c.y == n.y - c.y) && (n.y % 2 == 0), Z = (c.z == n.z - c.z) && (n.z % 2 == 0);
  return (X && c.y == 0 && c.z == 0) || (X && Y && c.z == 0) || (c.x == 0 && Y && c.z == 0) || (X && c.y == 0 && Z) ||
         (c.x == 0 && c.y == 0 && Z) || (c.x == 0 && Y && Z) || (X && Y && Z);
}

/* fourierBrownianNoise (FarField.cuh:235-308), sequential in id */
ORACLE_API void oracle_pse_fourier_brownian_noise(real *grid6, const real *L3, const int *cellDim, real prefactor, real shearStrain,
                                                  real rh, real viscosity, real split, real eta, uint seed1, uint seed2) {
  pcomplex3 *g = (pcomplex3 *)grid6;
  const int3 nk = mki3(cellDim[0], cellDim[1], cellDim[2]);
  const real3 L = mk3(L3[0], L3[1], L3[2]);
  const int nkx = nk.x / 2 + 1;
  const int total = nk.z * nk.y * nkx;
  for (int id = 0; id < total; id++) {
    const int3 cell = mki3(id % nkx, (id / nkx) % nk.y, id / (nkx * nk.y));
    if (id == 0 || (cell.x == 0 && cell.y == 0 && 2 * cell.z >= nk.z + 1) || (cell.x == 0 && 2 * cell.y >= nk.y + 1)) continue;
    Saru rng = saru3((uint)id, seed1, seed2);
    const real sc = (real)0.707106781186547 * prefactor;
    float a0, a1, b0, b1, c0, c1;
    saru_gf(&rng, 0, (float)sc, &a0, &a1);
    saru_gf(&rng, 0, (float)sc, &b0, &b1);
    saru_gf(&rng, 0, (float)sc, &c0, &c1);
    pcomplex3 noise = {a0, a1, b0, b1, c0, c1};
    const int nyquist = pse_is_nyquist(cell, nk);
    if (nyquist) {
      const real nqsc = (real)1.41421356237310;
      noise.xr *= nqsc; noise.xi = 0; noise.yr *= nqsc; noise.yi = 0; noise.zr *= nqsc; noise.zi = 0;
    }
    {
      const int3 ik = pse_indexToWaveNumber(id, nk);
      const real3 k = pse_waveVector(ik, L);
      const real B = pse_greens(k, shearStrain, rh, viscosity, split, eta, nk);
      const pcomplex3 z = pscale(pse_project(pse_shear(k, shearStrain), noise), SQRT(B));
      g[id].xr += z.xr; g[id].xi += z.xi; g[id].yr += z.yr; g[id].yi += z.yi; g[id].zr += z.zr; g[id].zi += z.zi;
    }
    if (nyquist) continue;
    if (cell.x == nk.x - cell.x || cell.x == 0) {
      const int xc = cell.x, yc = (cell.y > 0) * (nk.y - cell.y), zc = (cell.z > 0) * (nk.z - cell.z);
      const int id_conj = xc + nkx * (yc + zc * nk.y);
      const int3 ik = pse_indexToWaveNumber(id_conj, nk);
      const real3 k = pse_waveVector(ik, L);
      pcomplex3 factor = noise;
      factor.xi *= (real)-1.0; factor.yi *= (real)-1.0; factor.zi *= (real)-1.0;
      const real B = pse_greens(k, shearStrain, rh, viscosity, split, eta, nk);
      const pcomplex3 z = pscale(pse_project(pse_shear(k, shearStrain), factor), SQRT(B));
      g[id_conj].xr += z.xr; g[id_conj].xi += z.xi; g[id_conj].yr += z.yr; g[id_conj].yi += z.yi; g[id_conj].zr += z.zr; g[id_conj].zi += z.zi;
    }
  }
}

/* ---- BDHI::Lanczos (open boundary, dense RPY mobility) -------------------------------------------------------------------
 *   RotnePragerYamakawa::{RPY_differentSizes, operator()}       Integrator/BDHI/BDHI.cuh:27-96
 *   Lanczos_ns::NbodyMatrixFreeMobilityDot::{compute,accumulate,set}   Integrator/BDHI/BDHI_Lanczos.cu:56-118
 *   NBody::transverse order: j ascending over the group         Interactor/NBodyBase.cuh:79-110
 * Contract: f*vj + gv*rij -> FMA(gv, rij.k, f*vj.k); total += cur plain add; set OVERWRITES Mv[id]. */
static inline void rpy_different_sizes(real M0, real r, real ai, real aj, real *c1, real *c2) {
  const real asum = ai + aj;
  const real asub = FABS(ai - aj);
  if (r > asum) {
    const real invr = (real)1.0 / r;
    const real pref = M0 * (real)3.0 * (real)0.25 * invr;
    const real denom = FMA(ai, ai, aj * aj) / ((real)3.0 * r * r);
    *c1 = pref * ((real)1.0 + denom);
    *c2 = pref * FMA((real)-3.0, denom, (real)1.0) * invr * invr;
  } else if (r > asub) {
    const real pref = M0 / (ai * aj * (real)32.0 * r * r * r);
    real num = FMA((real)3.0 * r, r, asub * asub);
    *c1 = pref * FMA((real)16.0 * r * r * r, asum, -(num * num));
    num = FMA(-r, r, asub * asub);
    *c2 = pref * ((real)3.0 * num * num) / (r * r);
  } else {
    *c1 = M0 / (ai > aj ? ai : aj);
    *c2 = 0;
  }
}

ORACLE_API void oracle_rpy_nbody_mdot(const real4 *pos, const real *v, int vstride, const real *radius, real rh, real viscosity,
                                      int N, real *Mv3) {
  const real M0 = (real)(1 / (6 * M_PI * viscosity)); /* BDHI.cuh:29 */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; i++) {
    const real4 pi = pos[i];
    const real ai = radius ? radius[i] : rh;
    real3 total = mk3(0, 0, 0);
    for (int j = 0; j < N; j++) {
      const real4 pj = pos[j];
      const real aj = radius ? radius[j] : rh;
      const real3 rij = mk3(pi.x - pj.x, pi.y - pj.y, pi.z - pj.z);
      const real r = SQRT(dot3(rij, rij));
      const real3 vj = mk3(v[(size_t)vstride * j], v[(size_t)vstride * j + 1], v[(size_t)vstride * j + 2]);
      real f, gdivr2;
      rpy_different_sizes(M0, r, ai, aj, &f, &gdivr2);
      real3 cur;
      if (r == (real)0.0) {
        cur = mk3(f * vj.x, f * vj.y, f * vj.z);
      } else {
        const real gv = gdivr2 * dot3(rij, vj);
        cur = mk3(FMA(gv, rij.x, f * vj.x), FMA(gv, rij.y, f * vj.y), FMA(gv, rij.z, f * vj.z));
      }
      total.x += cur.x; total.y += cur.y; total.z += cur.z;
    }
    Mv3[3 * i] = total.x; Mv3[3 * i + 1] = total.y; Mv3[3 * i + 2] = total.z;
  }
}

/* Cholesky_ns::fillMobilityRPYD (Integrator/BDHI/BDHI_Cholesky.cu:34-80) completed to the full symmetric matrix: column
 * major 3N x 3N, M[3i+k + n(3j+l)] = c2 rij_k rij_l + c1 delta_kl with rij = pos_j - pos_i, self blocks (M0/a_i) I. */
ORACLE_API void oracle_rpy_dense(const real4 *pos, const real *radius, real rh, real viscosity, int N, real *M) {
  const real M0 = (real)(1 / (6 * M_PI * viscosity));
  const size_t n = 3 * (size_t)N;
  for (int i = 0; i < N; i++) {
    const real ai = radius ? radius[i] : rh;
    for (int j = i; j < N; j++) {
      const real aj = radius ? radius[j] : rh;
      real b[3][3];
      real c1, c2;
      if (i == j) {
        rpy_different_sizes(M0, 0, ai, ai, &c1, &c2);
        for (int k = 0; k < 3; k++) for (int l = 0; l < 3; l++) b[k][l] = k == l ? c1 : 0;
      } else {
        const real rij[3] = {pos[j].x - pos[i].x, pos[j].y - pos[i].y, pos[j].z - pos[i].z};
        const real r = SQRT(FMA(rij[2], rij[2], FMA(rij[1], rij[1], rij[0] * rij[0])));
        rpy_different_sizes(M0, r, ai, aj, &c1, &c2);
        for (int k = 0; k < 3; k++) for (int l = 0; l < 3; l++) b[k][l] = c2 * rij[k] * rij[l];
        for (int k = 0; k < 3; k++) b[k][k] += c1;
      }
      for (int k = 0; k < 3; k++)
        for (int l = 0; l < 3; l++) {
          M[3 * (size_t)i + k + n * (3 * (size_t)j + l)] = b[k][l];
          M[3 * (size_t)j + l + n * (3 * (size_t)i + k)] = b[k][l];
        }
    }
  }
}
