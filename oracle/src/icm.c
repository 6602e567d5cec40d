/* ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the product.
 *
 * CPU restatement of the Inertial Coupling Method integrator (zero excess mass), Integrator/Hydro/ICM.cu(h):
 *   updateCellVelocityUnperturbed = Laplacian + stochastic stress divergence + Adams-Bashforth advection   ICM.cu:503-822
 *   spreadParticleForces (prefactor dt/rho) :86-159        addThermalDrift (random finite differences) :161-275
 *   solveStokesFourier ((I - dt eta/(2 rho) L)^-1 P with staggered shifts) :349-411
 *   midPointStep :413-500 (the same interpolation as FIB: oracle_fib_midpoint_step)
 *   interpolateVelocitiesToCellCentersD  ICM.cuh:96-119
 * updateCellVelocityUnperturbed updates the velocity field IN PLACE while neighbouring threads still read it (a data race in
 * the reference); here — and in the HIP kernel — every cell reads the OLD field.  The fluid noise (cuRAND, unpinned) and the
 * initial thermal velocities (System::rng gaussians) are inputs.
 */
#include "common.h"
#include "saru.h"

real oracle_phi_peskin3(real invh, real r); /* ibm.c */

#define AT(v, x, y, z, c) v[3 * (size_t)(grid_cell_index(&g, grid_pbc_cell(&g, mki3((x), (y), (z))))) + (c)]

/* vNew = v + (dt eta/(2 rho)) L v + DivNoise - (dt/rho)(1.5 adv - 0.5 advOld);  advOld <- adv.   random: real[6][ncells] or NULL */
ORACLE_API void oracle_icm_update_unperturbed(const real *v, real *vNew, real *advOld, const real *boxL, const int *cells3, real density,
                                              real viscosity, real noiseAmp, real dt, const real *random) {
  const int per[3] = {1, 1, 1};
  Box box = box_from(boxL, per);
  Grid g = grid_make(box, mki3(cells3[0], cells3[1], cells3[2]));
  const int nx = cells3[0], ny = cells3[1], nz = cells3[2], nc = nx * ny * nz;
  const real3 ih = mk3((real)1.0 / g.cellSize.x, (real)1.0 / g.cellSize.y, (real)1.0 / g.cellSize.z);
  const real sqrt2 = (real)1.41421356237310;
#define RI(x, y, z) grid_cell_index(&g, grid_pbc_cell(&g, mki3((x), (y), (z))))
  for (int z = 0; z < nz; z++)
    for (int y = 0; y < ny; y++)
      for (int x = 0; x < nx; x++) {
        const int ic = x + nx * (y + ny * z);
        real dw[3] = {0, 0, 0};
        if (noiseAmp != (real)0.0) { /* computeNoiseDivergence, :503-590 */
          dw[0] += sqrt2 * ih.x * (random[RI(x + 1, y, z)] - random[ic]);
          dw[1] += sqrt2 * ih.y * (random[RI(x, y + 1, z) + nc] - random[ic + nc]);
          dw[2] += sqrt2 * ih.z * (random[RI(x, y, z + 1) + 2 * nc] - random[ic + 2 * nc]);
          const real wxy = random[ic + 3 * nc], wxz = random[ic + 4 * nc], wyz = random[ic + 5 * nc];
          dw[0] += ih.y * (wxy - random[RI(x, y - 1, z) + 3 * nc]);
          dw[1] += ih.x * (wxy - random[RI(x - 1, y, z) + 3 * nc]);
          dw[0] += ih.z * (wxz - random[RI(x, y, z - 1) + 4 * nc]);
          dw[2] += ih.x * (wxz - random[RI(x - 1, y, z) + 4 * nc]);
          dw[1] += ih.z * (wyz - random[RI(x, y, z - 1) + 5 * nc]);
          dw[2] += ih.y * (wyz - random[RI(x, y - 1, z) + 5 * nc]);
          for (int c = 0; c < 3; c++) dw[c] *= noiseAmp;
        }
        real lap[3];
        for (int c = 0; c < 3; c++) { /* computeVelLaplacian, :592-662 */
          const real v0 = AT(v, x, y, z, c);
          lap[c] = ih.x * ih.x * (AT(v, x + 1, y, z, c) - (real)2.0 * v0 + AT(v, x - 1, y, z, c));
          lap[c] += ih.y * ih.y * (AT(v, x, y + 1, z, c) - (real)2.0 * v0 + AT(v, x, y - 1, z, c));
          lap[c] += ih.z * ih.z * (AT(v, x, y, z + 1, c) - (real)2.0 * v0 + AT(v, x, y, z - 1, c));
        }
        /* computeAdvection, :664-779: D(rho v v^T) on the staggered grid */
        const real vx = AT(v, x, y, z, 0), vy = AT(v, x, y, z, 1), vz = AT(v, x, y, z, 2);
        const real vx_px = AT(v, x + 1, y, z, 0), vy_px = AT(v, x + 1, y, z, 1), vz_px = AT(v, x + 1, y, z, 2);
        const real vx_mx = AT(v, x - 1, y, z, 0), vy_mx = AT(v, x - 1, y, z, 1), vz_mx = AT(v, x - 1, y, z, 2);
        const real vx_py = AT(v, x, y + 1, z, 0), vy_py = AT(v, x, y + 1, z, 1), vz_py = AT(v, x, y + 1, z, 2);
        const real vx_my = AT(v, x, y - 1, z, 0), vy_my = AT(v, x, y - 1, z, 1), vz_my = AT(v, x, y - 1, z, 2);
        const real vx_pz = AT(v, x, y, z + 1, 0), vy_pz = AT(v, x, y, z + 1, 1), vz_pz = AT(v, x, y, z + 1, 2);
        const real vx_mz = AT(v, x, y, z - 1, 0), vy_mz = AT(v, x, y, z - 1, 1), vz_mz = AT(v, x, y, z - 1, 2);
        const real vy_px_my = AT(v, x + 1, y - 1, z, 1), vz_px_mz = AT(v, x + 1, y, z - 1, 2);
        const real vx_mx_py = AT(v, x - 1, y + 1, z, 0), vz_py_mz = AT(v, x, y + 1, z - 1, 2);
        const real vx_mx_pz = AT(v, x - 1, y, z + 1, 0), vy_my_pz = AT(v, x, y - 1, z + 1, 1);
        real adv[3];
        adv[0] = ih.x * ((vx_px + vx) * (vx_px + vx) - (vx + vx_mx) * (vx + vx_mx));
        adv[0] += ih.y * ((vx_py + vx) * (vy_px + vy) - (vx + vx_my) * (vy_px_my + vy_my));
        adv[0] += ih.z * ((vx_pz + vx) * (vz_px + vz) - (vx + vx_mz) * (vz_px_mz + vz_mz));
        adv[1] = ih.x * ((vy_px + vy) * (vx_py + vx) - (vy + vy_mx) * (vx_mx_py + vx_mx));
        adv[1] += ih.y * ((vy_py + vy) * (vy_py + vy) - (vy + vy_my) * (vy + vy_my));
        adv[1] += ih.z * ((vy_pz + vy) * (vz_py + vz) - (vy + vy_mz) * (vz_py_mz + vz_mz));
        adv[2] = ih.x * ((vz_px + vz) * (vx_pz + vx) - (vz + vz_mx) * (vx_mx_pz + vx_mx));
        adv[2] += ih.y * ((vz_py + vz) * (vy_pz + vy) - (vz + vz_my) * (vy_my_pz + vy_my));
        adv[2] += ih.z * ((vz_pz + vz) * (vz_pz + vz) - (vz + vz_mz) * (vz + vz_mz));
        for (int c = 0; c < 3; c++) {
          adv[c] *= (real)0.25 * density;
          vNew[3 * (size_t)ic + c] = AT(v, x, y, z, c) + ((dt * viscosity * (real)0.5 / density) * lap[c] + dw[c] -
                                                         (dt / density) * ((real)1.5 * adv[c] - (real)0.5 * advOld[3 * (size_t)ic + c]));
          advOld[3 * (size_t)ic + c] = adv[c];
        }
      }
#undef RI
}

/* solveStokesFourier in place on complex3[nz][ny][nx/2+1]: (I - dt eta/(2 rho) L)^-1 P, :349-411 */
ORACLE_API void oracle_icm_solve_stokes(real *grid6, real viscosity, real density, real dt, const real *boxL, const int *cells3,
                                        int removeTotalMomentum) {
  const int nx = cells3[0], ny = cells3[1], nz = cells3[2], nkx = nx / 2 + 1, nc = nx * ny * nz;
  const real3 h = mk3(boxL[0] / (real)nx, boxL[1] / (real)ny, boxL[2] / (real)nz);
  const real3 ih = mk3((real)1.0 / h.x, (real)1.0 / h.y, (real)1.0 / h.z);
  const real3 p2 = mk3((real)2.0 * (real)M_PI / boxL[0], (real)2.0 * (real)M_PI / boxL[1], (real)2.0 * (real)M_PI / boxL[2]);
  for (int cz = 0; cz < nz; cz++)
    for (int cy = 0; cy < ny; cy++)
      for (int cx = 0; cx < nkx; cx++) {
        real *v = grid6 + 6 * ((size_t)cx + (size_t)nkx * ((size_t)cy + (size_t)ny * cz));
        if (cx == 0 && cy == 0 && cz == 0) {
          for (int t = 0; t < 6; t++) v[t] = removeTotalMomentum ? 0 : v[t] / (real)nc;
          continue;
        }
        real3 k = mk3(cx * p2.x, cy * p2.y, cz * p2.z);
        if (cx >= (nx + 1) / 2) k.x -= (real)nx * p2.x;
        if (cy >= (ny + 1) / 2) k.y -= (real)ny * p2.y;
        if (cz >= (nz + 1) / 2) k.z -= (real)nz * p2.z;
        const real ax = k.x * h.x * (real)0.5, ay = k.y * h.y * (real)0.5, az = k.z * h.z * (real)0.5;
        const real sn[3] = {SIN(ax), SIN(ay), SIN(az)}, cs[3] = {COS(ax), COS(ay), COS(az)};
        const real3 keff = mk3((real)2.0 * ih.x * sn[0], (real)2.0 * ih.y * sn[1], (real)2.0 * ih.z * sn[2]);
        real re[3], im[3];
        for (int c = 0; c < 3; c++) {
          const real tr = v[2 * c], ti = v[2 * c + 1];
          re[c] = tr * cs[c] - ti * (-sn[c]);
          im[c] = ti * cs[c] + tr * (-sn[c]);
        }
        const real Lk = -dot3(keff, keff);
        const real pref = (real)1.0 / ((real)1.0 - (dt / density) * (real)0.5 * viscosity * Lk);
        for (int c = 0; c < 3; c++) { re[c] *= pref; im[c] *= pref; }
        const real invk2 = (real)1.0 / dot3(keff, keff);
        const real kfr = dot3(keff, mk3(re[0], re[1], re[2])) * invk2, kfi = dot3(keff, mk3(im[0], im[1], im[2])) * invk2;
        const real ke[3] = {keff.x, keff.y, keff.z};
        const real norm = (real)1.0 / (real)nc;
        for (int c = 0; c < 3; c++) {
          const real tr = re[c] - ke[c] * kfr, ti = im[c] - ke[c] * kfi;
          v[2 * c] = norm * (tr * cs[c] - ti * sn[c]);
          v[2 * c + 1] = norm * (ti * cs[c] + tr * sn[c]);
        }
      }
}

/* addThermalDrift (:161-275): gridVels += driftPrefactor [S(q + d/2 W) - S(q - d/2 W)] W, W ~ Saru(id, seed, step) */
ORACLE_API void oracle_icm_thermal_drift(const real4 *pos, int N, real *gridVels3, const real *boxL, const int *cells3,
                                         real driftPrefactor, real deltaRFD, uint seed, uint step) {
  const int per[3] = {1, 1, 1};
  Box box = box_from(boxL, per);
  Grid g = grid_make(box, mki3(cells3[0], cells3[1], cells3[2]));
  const real invh = (real)1.0 / g.cellSize.x; /* Kernel(grid.cellSize.x) */
  for (int id = 0; id < N; id++) {
    const real3 pi = mk3(pos[id].x, pos[id].y, pos[id].z);
    Saru rng = saru3((uint)id, seed, step);
    float a0, a1, b0, b1;
    saru_gf(&rng, 0, 1, &a0, &a1);
    saru_gf(&rng, 0, 1, &b0, &b1);
    const real W[3] = {a0, a1, b0};
    const real3 qp = mk3(pi.x + (real)0.5 * deltaRFD * W[0], pi.y + (real)0.5 * deltaRFD * W[1], pi.z + (real)0.5 * deltaRFD * W[2]);
    const real3 qm = mk3(pi.x - (real)0.5 * deltaRFD * W[0], pi.y - (real)0.5 * deltaRFD * W[1], pi.z - (real)0.5 * deltaRFD * W[2]);
    for (int c = 0; c < 3; c++) {
      real3 ps = pi, sp = qp, sm = qm; /* positions seen from the grid of component c */
      const real hh[3] = {g.cellSize.x, g.cellSize.y, g.cellSize.z};
      if (c == 0) { ps.x -= (real)0.5 * hh[0]; sp.x -= (real)0.5 * hh[0]; sm.x -= (real)0.5 * hh[0]; }
      if (c == 1) { ps.y -= (real)0.5 * hh[1]; sp.y -= (real)0.5 * hh[1]; sm.y -= (real)0.5 * hh[1]; }
      if (c == 2) { ps.z -= (real)0.5 * hh[2]; sp.z -= (real)0.5 * hh[2]; sm.z -= (real)0.5 * hh[2]; }
      const int3 cell = grid_get_cell(&g, ps);
      for (int i = 0; i < 27; i++) {
        int3 cj = grid_pbc_cell(&g, mki3(cell.x + i % 3 - 1, cell.y + (i / 3) % 3 - 1, cell.z + i / 9 - 1));
        const int jc = grid_cell_index(&g, cj);
        const real3 rp = grid_distance_to_cell_center(&g, sp, cj), rm = grid_distance_to_cell_center(&g, sm, cj);
        real s = oracle_phi_peskin3(invh, rp.x) * oracle_phi_peskin3(invh, rp.y) * oracle_phi_peskin3(invh, rp.z) * W[c];
        s -= oracle_phi_peskin3(invh, rm.x) * oracle_phi_peskin3(invh, rm.y) * oracle_phi_peskin3(invh, rm.z) * W[c];
        gridVels3[3 * (size_t)jc + c] += s * driftPrefactor;
      }
    }
  }
}

/* interpolateVelocitiesToCellCentersD (ICM.cuh:96-119) */
ORACLE_API void oracle_icm_collocate(const real *v, real *out, const real *boxL, const int *cells3) {
  const int per[3] = {1, 1, 1};
  Box box = box_from(boxL, per);
  Grid g = grid_make(box, mki3(cells3[0], cells3[1], cells3[2]));
  for (int z = 0; z < cells3[2]; z++)
    for (int y = 0; y < cells3[1]; y++)
      for (int x = 0; x < cells3[0]; x++) {
        const size_t ic = (size_t)x + (size_t)cells3[0] * ((size_t)y + (size_t)cells3[1] * z);
        out[3 * ic] = (real)0.5 * (AT(v, x, y, z, 0) + AT(v, x - 1, y, z, 0));
        out[3 * ic + 1] = (real)0.5 * (AT(v, x, y, z, 1) + AT(v, x, y - 1, z, 1));
        out[3 * ic + 2] = (real)0.5 * (AT(v, x, y, z, 2) + AT(v, x, y, z - 1, 2));
      }
}
