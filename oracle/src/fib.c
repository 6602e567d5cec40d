/* ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the product.
 *
 * CPU restatement of the Fluctuating Immersed Boundary integrator, Integrator/BDHI/FIB/FIB.cu(h), as it actually runs:
 * forwardTime dispatches BOTH schemes to forwardMidpoint (FIB.cu:1072-1079) and addThermalDrift returns on its first line
 * (:400), so a step is  g = noise + S F  ->  v = eta^-1 L^-1 g  ->  predictor  ->  corrector.
 *   FIB::FIB (grid from Kernel::adviseGridSize, hydrodynamic radius 0.91 h)    FIB.cu:87-135, FIB_kernels.cuh:108-130
 *   addRandomAdvection     :274-391     spreadParticleForces  :511-597
 *   cellToWaveNumber, projectFourier, shiftVelocity, solveStokesFourier       :601-724
 *   midPointStep           :726-823
 * The fluid noise comes from cuRAND in the reference (third party, stream unpinned): here it is an input array.
 */
#include "common.h"

real oracle_phi_peskin3(real invh, real r); /* ibm.c: IBM_kernels::Peskin::threePoint::phi */

ORACLE_API void oracle_fib_raw_cells(const real *boxL, real hydrodynamicRadius, int *cells3) { /* :104-108 */
  const real hgrid = hydrodynamicRadius / (real)0.91;
  for (int a = 0; a < 3; a++) cells3[a] = (int)(boxL[a] / hgrid);
}

typedef struct { int3 cell, P; real3 shifted; } Stag;
/* staggered cell of one velocity component: getCell(pi - h/2 e_c), P = support/2 = 1 (odd support: no shift, :511-526) */
static inline Stag stag(const Grid *g, real3 pi, int c) {
  Stag s;
  s.shifted = pi;
  if (c == 0) s.shifted.x = pi.x - (real)0.5 * g->cellSize.x;
  if (c == 1) s.shifted.y = pi.y - (real)0.5 * g->cellSize.y;
  if (c == 2) s.shifted.z = pi.z - (real)0.5 * g->cellSize.z;
  s.cell = grid_get_cell(g, s.shifted);
  s.P = mki3(1, 1, 1);
  return s;
}
static inline real delta3(const Grid *g, real3 r) {
  return oracle_phi_peskin3((real)1.0 / g->cellSize.x, r.x) * oracle_phi_peskin3((real)1.0 / g->cellSize.x, r.y) *
         oracle_phi_peskin3((real)1.0 / g->cellSize.x, r.z); /* the kernel is built with ONE h = min cell size (:119-120) */
}

/* spreadParticleForces: gridVels real3[nz][ny][nx] += S F */
ORACLE_API void oracle_fib_spread(const real4 *pos, const real4 *force, int N, const real *boxL, const int *cells3, real hKernel,
                                  real *gridVels3) {
  const int per[3] = {1, 1, 1};
  Box box = box_from(boxL, per);
  Grid g = grid_make(box, mki3(cells3[0], cells3[1], cells3[2]));
  const real invh = (real)1.0 / hKernel;
  for (int id = 0; id < N; id++) {
    const real3 pi = mk3(pos[id].x, pos[id].y, pos[id].z);
    const real f[3] = {force[id].x, force[id].y, force[id].z};
    for (int c = 0; c < 3; c++) {
      const Stag s = stag(&g, pi, c);
      for (int i = 0; i < 27; i++) {
        int3 cj = mki3(s.cell.x + i % 3 - s.P.x, s.cell.y + (i / 3) % 3 - s.P.y, s.cell.z + i / 9 - s.P.z);
        cj = grid_pbc_cell(&g, cj);
        const int jc = grid_cell_index(&g, cj);
        const real3 r = grid_distance_to_cell_center(&g, s.shifted, cj);
        gridVels3[3 * (size_t)jc + c] += oracle_phi_peskin3(invh, r.x) * oracle_phi_peskin3(invh, r.y) * oracle_phi_peskin3(invh, r.z) * f[c];
      }
    }
  }
}

/* addRandomAdvection: random = real[6][ncells] in the order XX, YY, ZZ, XY, XZ, YZ */
ORACLE_API void oracle_fib_random_advection(real *gridVels3, const real *boxL, const int *cells3, real noisePrefactor, const real *random) {
  const int per[3] = {1, 1, 1};
  Box box = box_from(boxL, per);
  Grid g = grid_make(box, mki3(cells3[0], cells3[1], cells3[2]));
  const int nx = cells3[0], ny = cells3[1], nz = cells3[2], nc = nx * ny * nz;
  const real sqrt2 = (real)1.41421356237310;
  const real3 ih = mk3((real)1.0 / g.cellSize.x, (real)1.0 / g.cellSize.y, (real)1.0 / g.cellSize.z);
#define IDX(x, y, z) grid_cell_index(&g, grid_pbc_cell(&g, mki3((x), (y), (z))))
  for (int z = 0; z < nz; z++)
    for (int y = 0; y < ny; y++)
      for (int x = 0; x < nx; x++) {
        const int ic = x + nx * (y + ny * z);
        real dx = 0, dy = 0, dz = 0;
        dx += sqrt2 * ih.x * (random[IDX(x + 1, y, z) + nc * 0] - random[ic + nc * 0]);
        dy += sqrt2 * ih.y * (random[IDX(x, y + 1, z) + nc * 1] - random[ic + nc * 1]);
        dz += sqrt2 * ih.z * (random[IDX(x, y, z + 1) + nc * 2] - random[ic + nc * 2]);
        const real wxy = random[ic + nc * 3], wxz = random[ic + nc * 4], wyz = random[ic + nc * 5];
        dx += ih.y * (wxy - random[IDX(x, y - 1, z) + nc * 3]);
        dy += ih.x * (wxy - random[IDX(x - 1, y, z) + nc * 3]);
        dx += ih.z * (wxz - random[IDX(x, y, z - 1) + nc * 4]);
        dz += ih.x * (wxz - random[IDX(x - 1, y, z) + nc * 4]);
        dy += ih.z * (wyz - random[IDX(x, y, z - 1) + nc * 5]);
        dz += ih.y * (wyz - random[IDX(x, y - 1, z) + nc * 5]);
        gridVels3[3 * (size_t)ic] += dx * noisePrefactor;
        gridVels3[3 * (size_t)ic + 1] += dy * noisePrefactor;
        gridVels3[3 * (size_t)ic + 2] += dz * noisePrefactor;
      }
#undef IDX
}

/* solveStokesFourier in place on the compact half-complex layout: complex3[nz][ny][nx/2+1] as (x.re,x.im,y.re,y.im,z.re,z.im) */
ORACLE_API void oracle_fib_solve_stokes(real *grid6, real viscosity, const real *boxL, const int *cells3) {
  const int nx = cells3[0], ny = cells3[1], nz = cells3[2], nkx = nx / 2 + 1, nc = nx * ny * nz;
  const real3 h = mk3(boxL[0] / (real)nx, boxL[1] / (real)ny, boxL[2] / (real)nz);
  const real3 ih = mk3((real)1.0 / h.x, (real)1.0 / h.y, (real)1.0 / h.z);
  const real3 p2 = mk3((real)2.0 * (real)M_PI / boxL[0], (real)2.0 * (real)M_PI / boxL[1], (real)2.0 * (real)M_PI / boxL[2]);
  for (int cz = 0; cz < nz; cz++)
    for (int cy = 0; cy < ny; cy++)
      for (int cx = 0; cx < nkx; cx++) {
        real *v = grid6 + 6 * ((size_t)cx + (size_t)nkx * ((size_t)cy + (size_t)ny * cz));
        if (cx == 0 && cy == 0 && cz == 0) { for (int t = 0; t < 6; t++) v[t] = 0; continue; }
        real3 k = mk3(cx * p2.x, cy * p2.y, cz * p2.z); /* cellToWaveNumber with the (n+1)/2 threshold, :603-617 */
        if (cx >= (nx + 1) / 2) k.x -= (real)nx * p2.x;
        if (cy >= (ny + 1) / 2) k.y -= (real)ny * p2.y;
        if (cz >= (nz + 1) / 2) k.z -= (real)nz * p2.z;
        const real ax = k.x * h.x * (real)0.5, ay = k.y * h.y * (real)0.5, az = k.z * h.z * (real)0.5;
        const real3 sink = mk3(SIN(ax), SIN(ay), SIN(az)), cosk = mk3(COS(ax), COS(ay), COS(az));
        const real3 keff = mk3((real)2.0 * ih.x * sink.x, (real)2.0 * ih.y * sink.y, (real)2.0 * ih.z * sink.z);
        /* shift to the cell centres: phase (cosk, -sink) */
        real re[3], im[3];
        const real cs[3] = {cosk.x, cosk.y, cosk.z}, sn[3] = {sink.x, sink.y, sink.z};
        for (int c = 0; c < 3; c++) {
          const real tr = v[2 * c], ti = v[2 * c + 1];
          re[c] = tr * cs[c] - ti * (-sn[c]);
          im[c] = ti * cs[c] + tr * (-sn[c]);
        }
        const real invL = (real)-1.0 / dot3(keff, keff);
        const real pref = (real)-1.0 * invL / viscosity;
        for (int c = 0; c < 3; c++) { re[c] *= pref; im[c] *= pref; }
        const real invk2 = (real)1.0 / dot3(keff, keff);
        const real kfr = dot3(keff, mk3(re[0], re[1], re[2])) * invk2, kfi = dot3(keff, mk3(im[0], im[1], im[2])) * invk2;
        const real ke[3] = {keff.x, keff.y, keff.z};
        for (int c = 0; c < 3; c++) { re[c] = re[c] - ke[c] * kfr; im[c] = im[c] - ke[c] * kfi; }
        const real norm = (real)1.0 / (real)nc;
        for (int c = 0; c < 3; c++) { /* back to the faces: phase (cosk, sink), FFT normalisation */
          const real tr = re[c], ti = im[c];
          v[2 * c] = norm * (tr * cs[c] - ti * sn[c]);
          v[2 * c + 1] = norm * (ti * cs[c] + tr * sn[c]);
        }
      }
}

/* midPointStep: mode 0 predictor (posOld = pos; pos += dt/2 J v), 1 corrector (pos = posOld + dt J(pos) v), 2 euler */
ORACLE_API void oracle_fib_midpoint_step(int mode, real4 *pos, real4 *posOld, const real *gridVels3, int N, const real *boxL,
                                         const int *cells3, real hKernel, real dt) {
  const int per[3] = {1, 1, 1};
  Box box = box_from(boxL, per);
  Grid g = grid_make(box, mki3(cells3[0], cells3[1], cells3[2]));
  const real invh = (real)1.0 / hKernel;
  const real dV = g.cellSize.x * g.cellSize.y * g.cellSize.z;
  real prefactor = dt;
  if (mode == 0) prefactor *= (real)0.5;
  for (int id = 0; id < N; id++) {
    const real3 pc = mk3(pos[id].x, pos[id].y, pos[id].z);
    if (mode == 0) posOld[id] = pos[id];
    real pn[3] = {0, 0, 0};
    for (int i = 0; i < 27; i++)
      for (int c = 0; c < 3; c++) {
        const Stag s = stag(&g, pc, c);
        int3 cj = mki3(s.cell.x + i % 3 - s.P.x, s.cell.y + (i / 3) % 3 - s.P.y, s.cell.z + i / 9 - s.P.z);
        cj = grid_pbc_cell(&g, cj);
        const int jc = grid_cell_index(&g, cj);
        const real3 r = grid_distance_to_cell_center(&g, s.shifted, cj);
        pn[c] += oracle_phi_peskin3(invh, r.x) * oracle_phi_peskin3(invh, r.y) * oracle_phi_peskin3(invh, r.z) * gridVels3[3 * (size_t)jc + c] * dV;
      }
    if (mode == 0) {
      pos[id].x = pc.x + prefactor * pn[0]; pos[id].y = pc.y + prefactor * pn[1]; pos[id].z = pc.z + prefactor * pn[2];
    } else {
      const real4 po = posOld[id];
      pos[id].x = po.x + prefactor * pn[0]; pos[id].y = po.y + prefactor * pn[1]; pos[id].z = po.z + prefactor * pn[2];
      pos[id].w = po.w;
    }
  }
}
