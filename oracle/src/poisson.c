/* ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the product (uammd_amd/, include/).
 *
 * CPU restatement of the triply periodic spectral Ewald Poisson solver, Interactor/SpectralEwaldPoisson.cu(h):
 *   Poisson_ns::Gaussian (window, support)                     SpectralEwaldPoisson.cuh:63-81
 *   Poisson::Poisson (grid, support clamp, near cut-off, table) SpectralEwaldPoisson.cu:71-160
 *   greensFunction / greensFunctionField                        SpectralEwaldPoisson.cu:15-62
 *   chargeFourier2FieldAndPotential (+ cellToWaveNumber, isNyquist)  :410-476
 *   NearField{Force,Energy,FieldPotential}Transverser           :222-329 over a CellList (NeighbourList/common.cuh:10-34)
 *   UnZip2Real4 (force += q E, energy += q phi)                 :529-559
 * The reference mixes `real` and double on the host (the closed forms are evaluated in double from real arguments and
 * rounded to real); the same mix is kept here, so the f32 build restates a SINGLE_PRECISION reference and the f64
 * build a DOUBLE_PRECISION one.  sqrt/log10 of a real argument are the real overloads, as nvcc resolves them.
 */
#include "common.h"

#ifdef DOUBLE_PRECISION
#define LOG10R(a) log10(a)
#else
#define LOG10R(a) log10f(a)
#endif

/* :15-38.  r2 is the squared distance. */
static real poisson_greens(real r2, real gw, real split, real epsilon) {
  double G = 0;
  if (r2 > gw * gw * gw * gw) {
    const double r = SQRT(r2);
    const real farw = SQRT((real)4 * gw * gw + (real)1 / (split * split));
    G = (1.0 / (4.0 * M_PI * epsilon * r) * (erf(r / ((real)2 * gw)) - erf(r / farw)));
  } else {
    const double pi32 = pow(M_PI, 1.5);
    const double gw2 = gw * gw;
    const double invsp2 = 1.0 / (split * split);
    const double selfterm = 1.0 / (4 * pi32 * gw) - 1.0 / (2 * pi32 * sqrt(4 * gw2 + invsp2));
    const double r2term = 1.0 / (6.0 * pi32 * pow(4.0 * gw2 + invsp2, 1.5)) - 1.0 / (48.0 * pi32 * gw2 * gw);
    const double r4term = 1.0 / (640.0 * pi32 * gw2 * gw2 * gw) - 1.0 / (20.0 * pi32 * pow(4 * gw2 + invsp2, 2.5));
    G = 1.0 / epsilon * (selfterm + r2 * r2term + r2 * r2 * r4term);
  }
  return (real)G;
}

/* :40-62.  r is the distance; returns -dG/dr / ... the modulus used by the force transverser. */
static real poisson_greens_field(real r, real gw, real split, real epsilon) {
  const double r2 = r * r;
  const double gw2 = gw * gw;
  const double newgw = sqrt(gw2 + 1 / (4.0 * split * split));
  const double newgw2 = newgw * newgw;
  double fmod = 0;
  if (r2 > gw * gw * gw * gw) {
    const double invrterm = exp(-0.25 * r2 / newgw2) / sqrt(M_PI * newgw2) - exp(-0.25 * r2 / gw2) / sqrt(M_PI * gw2);
    const double invr2term = erf(0.5 * r / newgw) - erf(0.5 * r / gw);
    fmod += 1 / (4 * M_PI) * (invrterm / r - invr2term / r2);
  } else if (r2 > 0) {
    const double pi32 = pow(M_PI, 1.5);
    const double rterm = 1 / (24 * pi32) * (1.0 / (gw2 * gw) - 1 / (newgw2 * newgw));
    const double r3term = 1 / (160 * pi32) * (1.0 / (newgw2 * newgw2 * newgw) - 1.0 / (gw2 * gw2 * gw));
    fmod += r * rterm + r2 * r * r3term;
  }
  return (real)(fmod / epsilon);
}

ORACLE_API real oracle_poisson_greens(real r2, real gw, real split, real epsilon) { return poisson_greens(r2, gw, split, epsilon); }
ORACLE_API real oracle_poisson_greens_field(real r, real gw, real split, real epsilon) {
  return poisson_greens_field(r, gw, split, epsilon);
}

static double far_field_width(real gw, real split) { /* :76-79 */
  double w = gw;
  if (split > 0) w = sqrt(gw * gw + 1.0 / (4.0 * split * split));
  return w;
}

/* :75-90, first half: the cell counts before nextFFTWiseSize3D */
ORACLE_API void oracle_poisson_raw_cells(const real *boxL, real gw, real split, real tolerance, real upsampling, int *cells3) {
  const double w = far_field_width(gw, split);
  double h;
  if (upsampling > 0) h = 1.0 / upsampling;
  else {
    const double t = (-LOG10R(tolerance)) / 10.0;
    h = (1.3 - (t < 0.9 ? t : 0.9)) * w;
  }
  const double hmax = boxL[0] / 32.0;
  if (hmax < h) h = hmax;
  const real hr = (real)h; /* real3 / double goes through the real3 / real operator */
  for (int a = 0; a < 3; a++) cells3[a] = (int)(boxL[a] / hr);
}

/* :87-118 after the grid is known, and :140-160.  Returns 0, -1 (support too large) or -2 (near cut-off too large).
 * out: window prefactor, tau; support; nearFieldCutOff; Ntable. */
ORACLE_API int oracle_poisson_setup(const real *boxL, const int *cells3, real gw, real split, real epsilon, real tolerance,
                                    real *prefactor, real *tau_out, int *support_out, real *cutoff_out, int *ntable_out) {
  const real h = boxL[0] / (real)cells3[0]; /* grid.cellSize.x */
  const real width = (real)far_field_width(gw, split);
  /* Poisson_ns::Gaussian(tolerance, width, h), SpectralEwaldPoisson.cuh:65-70 */
  *prefactor = (real)cbrt(pow(2 * M_PI * width * width, -1.5));
  const real tau = (real)(-1.0 / (2.0 * width * width));
  *tau_out = tau;
  const real rmax = (real)sqrt(log(tolerance * sqrt(2 * M_PI * width * width)) / tau);
  int support = (int)((real)2 * rmax / h + 0.5);
  if (support < 3) support = 3;
  if (support > cells3[0] / 2 - 1) return -1;
  if (support > cells3[0] / 2 - 2) support = cells3[0] / 2 - 2;
  *support_out = support;
  *cutoff_out = 0;
  *ntable_out = 0;
  if (split > 0) {
    long double E = 1;
    long double r = far_field_width(gw, split);
    while (fabsl(E) > tolerance) {
      r += 0.001l * gw;
      E = poisson_greens((real)(r * r), gw, split, epsilon);
    }
    const real rc = (real)r;
    *cutoff_out = rc;
    if (rc > boxL[0] / 2.0) return -2;
    int n = (int)(rc / (gw * tolerance * 1e3));
    if (n > (1 << 16)) n = 1 << 16;
    if (n < 4096) n = 4096;
    *ntable_out = n;
  }
  return 0;
}

/* TabulatedFunction ctor (misc/TabulatedFunction.cuh:103-117): N samples on [0, rmax], x evaluated in double */
ORACLE_API void oracle_poisson_tables(real gw, real split, real epsilon, real cutoff, int ntable, real *tableField,
                                      real *tablePotential) {
  const int Nm1 = ntable - 1;
  const real rmaxF = cutoff, rmaxP = cutoff * cutoff;
  for (int i = 0; i <= Nm1; i++) {
    const double xf = (i / (double)Nm1) * (rmaxF - (real)0) + (real)0;
    tableField[i] = poisson_greens_field((real)xf, gw, split, epsilon);
    const double xp = (i / (double)Nm1) * (rmaxP - (real)0) + (real)0;
    tablePotential[i] = poisson_greens((real)xp, gw, split, epsilon);
  }
}

/* TabulatedFunction::operator() with LinearInterpolation (TabulatedFunction.cuh:63-75, :148-157), rmin = 0 */
static inline real table1(const real *table, int Nm1, real rmax, real rs) {
  const real interval = (real)(1.0 / (rmax - (real)0));
  const real dr = (real)(1.0 / (real)Nm1);
  const real r = (rs - (real)0) * interval;
  if (rs >= rmax) return 0;
  if (r <= (real)0.0) return table[0];
  const int i = (int)(r * Nm1);
  const real r0 = i * dr;
  const real v0 = table[i], v1 = table[i + 1];
  const real t = (r - r0) * (real)Nm1;
  return FMA(t, v1, FMA(-t, v0, v0));
}
ORACLE_API void oracle_poisson_table_get(const real *table, int ntable, real rmax, const real *rs, int n, real *out) {
  for (int i = 0; i < n; i++) out[i] = table1(table, ntable - 1, rmax, rs[i]);
}

/* chargeFourier2FieldAndPotential (:433-476).  charges: complex[nz][ny][nx/2+1]; out: complex4 (Ex,Ey,Ez,phi) same order */
ORACLE_API void oracle_poisson_convolve(const real *chargesFourier, real *fieldPotential8, const real *boxL, const int *cells3,
                                        real epsilon) {
  const int nx = cells3[0], ny = cells3[1], nz = cells3[2], nkx = nx / 2 + 1;
  const real3 pi2invL = mk3(((real)2.0 * (real)M_PI) / boxL[0], ((real)2.0 * (real)M_PI) / boxL[1], ((real)2.0 * (real)M_PI) / boxL[2]);
  const int ncells = nx * ny * nz;
  for (int cz = 0; cz < nz; cz++)
    for (int cy = 0; cy < ny; cy++)
      for (int cx = 0; cx < nkx; cx++) {
        const size_t ic = (size_t)cx + ((size_t)cy + (size_t)cz * ny) * nkx;
        real *o = fieldPotential8 + 8 * ic;
        for (int t = 0; t < 8; t++) o[t] = 0;
        if (cx == 0 && cy == 0 && cz == 0) continue;
        real3 k = mk3(cx * pi2invL.x, cy * pi2invL.y, cz * pi2invL.z);
        if (cx >= nx / 2 + 1) k.x -= (real)nx * pi2invL.x;
        if (cy >= ny / 2 + 1) k.y -= (real)ny * pi2invL.y;
        if (cz >= nz / 2 + 1) k.z -= (real)nz * pi2invL.z;
        const real k2 = FMA(k.z, k.z, FMA(k.y, k.y, k.x * k.x));
        const int xn = (cx == nx - cx) && (nx % 2 == 0), yn = (cy == ny - cy) && (ny % 2 == 0), zn = (cz == nz - cz) && (nz % 2 == 0);
        const int nyquist = (xn && cy == 0 && cz == 0) || (xn && yn && cz == 0) || (cx == 0 && yn && cz == 0) ||
                            (xn && cy == 0 && zn) || (cx == 0 && cy == 0 && zn) || (cx == 0 && yn && zn) || (xn && yn && zn);
        if (nyquist) continue;
        const real fx = chargesFourier[2 * ic], fy = chargesFourier[2 * ic + 1];
        const real B = (real)1.0 / (k2 * epsilon * ncells);
        o[0] = k.x * fy * B; o[1] = -k.x * fx * B;
        o[2] = k.y * fy * B; o[3] = -k.y * fx * B;
        o[4] = k.z * fy * B; o[5] = -k.z * fx * B;
        o[6] = fx * B;       o[7] = fy * B;
      }
}

/* UnZip2Real4::operator+= (:548-553) after IBM::gather of the real4 grid */
ORACLE_API void oracle_poisson_apply_charges(const real *fieldPotential4, const real *charge, int N, real *force4, real *energy) {
  for (int i = 0; i < N; i++) {
    const real q = charge[i];
    if (force4) {
      force4[4 * i] += q * fieldPotential4[4 * i];
      force4[4 * i + 1] += q * fieldPotential4[4 * i + 1];
      force4[4 * i + 2] += q * fieldPotential4[4 * i + 2];
      force4[4 * i + 3] += q * (real)0;
    }
    if (energy) energy[i] += q * fieldPotential4[4 * i + 3];
  }
}

/* transverseList with the three near-field Transversers (:222-329) over a built cell list.  The charge is indexed with the
 * group index (getInfo).  mode 0: force4[ori] += (total, 0); 1: energy[ori] += total; 2: fieldPotential4[ori] += total. */
ORACLE_API void oracle_poisson_near(const real4 *sortPos, const int *groupIndex, int N, const uint *cellStart, const int *cellEnd,
                                    uint validCell, const real *gridL, const int *gridPeriodic, const int *cellDim,
                                    const real *boxL, const real *charge, const real *tableField, const real *tablePotential,
                                    int ntable, real cutoff, int mode, real *out) {
  Box gbox = box_from(gridL, gridPeriodic);
  Grid grid = grid_make(gbox, mki3(cellDim[0], cellDim[1], cellDim[2]));
  const int per1[3] = {1, 1, 1};
  Box box = box_from(boxL, per1);
  const int Nm1 = ntable - 1;
  const real rc = cutoff, rc2 = cutoff * cutoff;
  const int3 n = grid.cellDim;
  const int3 nperdim = mki3((n.x > 1 ? 3 : 1), (n.y > 1 ? 3 : 1), (n.z > 1 ? 3 : 1));
  const int numberNeighbourCells = nperdim.x * nperdim.y * nperdim.z;
#pragma omp parallel for schedule(static)
  for (int id = 0; id < N; id++) {
    const int ori = groupIndex[id];
    const real4 pi = sortPos[id];
    const real qi = charge[ori];
    real tx = 0, ty = 0, tz = 0, tw = 0;
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
        const real qj = charge[groupIndex[j]];
        const real3 rij = box_apply_pbc(&box, mk3(pj.x - pi.x, pj.y - pi.y, pj.z - pi.z));
        const real r2 = dot3(rij, rij);
        if (mode == 1) {
          tw += qi * qj * table1(tablePotential, Nm1, rc2, r2);
        } else if (mode == 0) {
          const real r = SQRT(r2);
          const real fmod = -qi * qj * table1(tableField, Nm1, rc, r);
          if (r2 > 0) {
            const real invr = (real)1.0 / r; /* real3 / real multiplies by the reciprocal (utils/vector.cuh:191-193) */
            tx += invr * (fmod * rij.x); ty += invr * (fmod * rij.y); tz += invr * (fmod * rij.z);
          }
        } else {
          const real phi = qj * table1(tablePotential, Nm1, rc2, r2);
          real ex = 0, ey = 0, ez = 0;
          if (r2 > 0) {
            const real r = SQRT(r2);
            const real fmod = -qj * table1(tableField, Nm1, rc, r);
            const real invr = (real)1.0 / r;
            ex = invr * (fmod * rij.x); ey = invr * (fmod * rij.y); ez = invr * (fmod * rij.z);
          }
          tx += ex; ty += ey; tz += ez; tw += phi;
        }
      }
    }
    if (mode == 1) out[ori] += tw;
    else if (mode == 0) { out[4 * ori] += tx; out[4 * ori + 1] += ty; out[4 * ori + 2] += tz; out[4 * ori + 3] += 0; }
    else { out[4 * ori] += tx; out[4 * ori + 1] += ty; out[4 * ori + 2] += tz; out[4 * ori + 3] += tw; }
  }
}
