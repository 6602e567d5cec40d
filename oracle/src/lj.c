/* ORACLE — TEST INFRASTRUCTURE ONLY (see common.h).
 *
 * Path A, traversal + Lennard-Jones: CPU restatement of
 *   NeighbourIterator::nextcell / increment          Interactor/NeighbourList/CellList/NeighbourContainer.cuh:95-138
 *   transverseWithNeighbourContainer (K6)            Interactor/NeighbourList/common.cuh:10-34
 *   nbody_ns::transverseGPU (K9)                     Interactor/NBodyBase.cuh:46-116   (j ascending over the group)
 *   Radial<LJFunctor>::Transverser::compute / set    Interactor/Potential/RadialPotential.cuh:107-127
 *   LJFunctor::force / energy / processPairParameters Interactor/Potential/Potential.cuh:37-82
 *   BasicParameterHandler::Iterator::operator()      Interactor/Potential/ParameterHandler.cuh:41-60
 *   default accumulate = total + current             utils/TransverserUtils.cuh:240-244, utils/ForceEnergyVirial.cuh:14-17
 *
 * Deviation (documented, DESIGN.md "non-periodic neighbours"): in a NON-periodic dimension the
 * reference's pbc_cell leaves cell -1 / cellDim unwrapped and then indexes cellStart with it
 * (NeighbourContainer.cuh:112-118 never fires because |cellj-celli| is 1) — an out-of-range read
 * whose pairs are all beyond the cut-off.  The oracle (and the HIP path) skip such cells.
 */
#include "common.h"

typedef struct { real cutOff2, sigma2, epsilonDivSigma2, shift; } LJPairParameters; /* Potential.cuh:31-35 */

/* Potential.cuh:66-82 */
ORACLE_API void oracle_lj_process_pair_parameters(real cutOff, real sigma, real epsilon, int shift, real *out4) {
  LJPairParameters p;
  p.cutOff2 = cutOff * cutOff;
  p.sigma2 = sigma * sigma;
  p.epsilonDivSigma2 = epsilon / p.sigma2;
  if (shift) {
    real invCutOff2 = p.sigma2 / p.cutOff2;
    real invrc6 = invCutOff2 * invCutOff2 * invCutOff2;
    p.shift = epsilon * (real)4.0 * invrc6 * (invrc6 - (real)1.0);
  } else
    p.shift = (real)0.0;
  out4[0] = p.cutOff2; out4[1] = p.sigma2; out4[2] = p.epsilonDivSigma2; out4[3] = p.shift;
}

/* Potential.cuh:37-46 */
static inline real lj_force(real r2, LJPairParameters p) {
  if (r2 >= p.cutOff2) return 0;
  const real invr2 = p.sigma2 / r2;
  const real invr6 = invr2 * invr2 * invr2;
  real fmod = p.epsilonDivSigma2 * FMA((real)(-48.0), invr6, (real)24.0) * invr6 * invr2;
  return fmod;
}
/* Potential.cuh:48-65 */
static inline real lj_energy(real r2, LJPairParameters p) {
  if (r2 >= p.cutOff2) return 0;
  real invr2 = p.sigma2 / r2;
  real invr6 = invr2 * invr2 * invr2;
  real E = FMA(p.epsilonDivSigma2 * p.sigma2 * (real)4.0 * invr6, (invr6 - (real)1.0), -p.shift);
  return (real)0.5 * E;
}
/* ParameterHandler.cuh:41-60 */
static inline LJPairParameters lj_params(const LJPairParameters *tbl, int ntypes, int ti, int tj) {
  if (ntypes == 1) return tbl[0];
  if (ti > tj) { int t = ti; ti = tj; tj = t; }
  int typeIndex = ti + ntypes * tj;
  if (ti >= ntypes || tj >= ntypes) typeIndex = 0;
  return tbl[typeIndex];
}

typedef struct { real3 force; real energy, virial; } FEV;

/* One pair: RadialPotential.cuh:107-118 (compute) followed by the default accumulate
 * (TransverserUtils.cuh:240-244: total = total + current).
 * Contract for the fused sum: with F = fmod*r12 inlined into total.force + F the a*b+c pattern
 * appears, so the force is accumulated as FMA(fmod, r12.k, total.k).  The virial needs the rounded
 * product F itself: V = dot(F, r12) with F.k = fmod*r12.k. */
static inline void lj_pair_acc(FEV *t, const Box *box, const LJPairParameters *tbl, int ntypes, real4 ri, real4 rj,
                               int wantF, int wantE, int wantV) {
  real3 r12 = box_apply_pbc(box, mk3(rj.x - ri.x, rj.y - ri.y, rj.z - ri.z));
  LJPairParameters params = lj_params(tbl, ntypes, (int)ri.w, (int)rj.w);
  const real r2 = dot3(r12, r12);
  if (r2 == (real)0.0) return; /* returns {} : adds zeros */
  if (wantE) t->energy += lj_energy(r2, params);
  if (wantF || wantV) {
    const real fm = lj_force(r2, params);
    t->force.x = FMA(fm, r12.x, t->force.x);
    t->force.y = FMA(fm, r12.y, t->force.y);
    t->force.z = FMA(fm, r12.z, t->force.z);
    if (wantV) t->virial += dot3(mk3(fm * r12.x, fm * r12.y, fm * r12.z), r12);
  }
}

/* K6 driven by the cell list.  `groupIndex` = sorted->group index (ParticleSorter index array),
 * `globalIndex` = group->global (nullable = identity, the "All" group).  force/energy/virial are
 * nullable and are ACCUMULATED into (Transverser::set does +=). */
ORACLE_API void oracle_lj_transverse_celllist(const real4 *sortPos, const int *groupIndex, const int *globalIndex, int N,
                                              const uint *cellStart, const int *cellEnd, uint validCell,
                                              const real *gridL, const int *gridPeriodic, const int *cellDim,
                                              const real *boxL, const int *boxPeriodic, const real *paramTable,
                                              int ntypes, real4 *force, real *energy, real *virial) {
  Box gbox = box_from(gridL, gridPeriodic);
  Grid grid = grid_make(gbox, mki3(cellDim[0], cellDim[1], cellDim[2]));
  Box box = box_from(boxL, boxPeriodic); /* the Potential's box (PairForces::Parameters::box) */
  const LJPairParameters *tbl = (const LJPairParameters *)paramTable;
  const int3 n = grid.cellDim;
  const int3 nperdim = mki3((n.x > 1 ? 3 : 1), (n.y > 1 ? 3 : 1), (n.z > 1 ? 3 : 1));
  const int numberNeighbourCells = nperdim.x * nperdim.y * nperdim.z;
#pragma omp parallel for schedule(static)
  for (int id = 0; id < N; id++) {
    const int gi = groupIndex[id];
    const int ori = globalIndex ? globalIndex[gi] : gi;
    const real4 pi = sortPos[id];
    FEV quantity = {{0, 0, 0}, 0, 0};
    const int3 celli = grid_get_cell(&grid, mk3(pi.x, pi.y, pi.z));
    for (int currentCell = 0; currentCell < numberNeighbourCells; currentCell++) { /* nextcell(), :95-130 */
      int3 cellj = celli;
      if (nperdim.x > 1) cellj.x += currentCell % 3 - 1;
      if (nperdim.y > 1) cellj.y += (currentCell / nperdim.x) % 3 - 1;
      if (nperdim.z > 1) cellj.z += currentCell / (nperdim.x * nperdim.y) - 1;
      cellj = grid_pbc_cell(&grid, cellj);
      /* documented deviation: out-of-range cell in a non periodic dimension -> skipped */
      if (cellj.x < 0 || cellj.x >= n.x || cellj.y < 0 || cellj.y >= n.y || cellj.z < 0 || cellj.z >= n.z) continue;
      const int icellj = grid_cell_index(&grid, cellj);
      const uint cs = cellStart[icellj];
      if (cs < validCell) continue; /* empty */
      const int first = (int)(cs - validCell), last = cellEnd[icellj];
      for (int j = first; j < last; j++) {
        lj_pair_acc(&quantity, &box, tbl, ntypes, pi, sortPos[j], force || virial, energy != NULL, virial != NULL);
      }
    }
    if (force) { force[ori].x += quantity.force.x; force[ori].y += quantity.force.y; force[ori].z += quantity.force.z; force[ori].w += 0; }
    if (energy) energy[ori] += quantity.energy;
    if (virial) virial[ori] += quantity.virial;
  }
}

/* K9: all pairs, j ascending over the group (NBodyBase.cuh:79-110); the small-box fallback of
 * PairForces (PairForces.cu:49-53) and an independent cross-check of the cell-list path. */
ORACLE_API void oracle_lj_transverse_nbody(const real4 *pos, const int *globalIndex, int N, const real *boxL,
                                           const int *boxPeriodic, const real *paramTable, int ntypes, real4 *force,
                                           real *energy, real *virial) {
  Box box = box_from(boxL, boxPeriodic);
  const LJPairParameters *tbl = (const LJPairParameters *)paramTable;
#pragma omp parallel for schedule(static)
  for (int t = 0; t < N; t++) {
    const int id = globalIndex ? globalIndex[t] : t;
    const real4 pi = pos[id];
    FEV quantity = {{0, 0, 0}, 0, 0};
    for (int c = 0; c < N; c++) {
      const int j = globalIndex ? globalIndex[c] : c;
      lj_pair_acc(&quantity, &box, tbl, ntypes, pi, pos[j], force || virial, energy != NULL, virial != NULL);
    }
    if (force) { force[id].x += quantity.force.x; force[id].y += quantity.force.y; force[id].z += quantity.force.z; }
    if (energy) energy[id] += quantity.energy;
    if (virial) virial[id] += quantity.virial;
  }
}

/* Double-precision all-pairs LJ force irrespective of the build's `real` (accuracy yardstick for
 * the "vs double oracle" tolerance of SURVEY §8d).  Minimum image in double. */
ORACLE_API void oracle_lj_nbody_f64(const double *pos4, int N, const double *L, const int *periodic, double cutOff,
                                    double sigma, double epsilon, double *force3) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; i++) {
    double fx = 0, fy = 0, fz = 0;
    for (int j = 0; j < N; j++) {
      if (j == i) continue;
      double d[3];
      for (int k = 0; k < 3; k++) {
        d[k] = pos4[4 * j + k] - pos4[4 * i + k];
        if (periodic[k]) d[k] -= floor(d[k] / L[k] + 0.5) * L[k];
      }
      double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
      if (r2 == 0 || r2 >= cutOff * cutOff) continue;
      double s2 = sigma * sigma / r2, s6 = s2 * s2 * s2;
      double fm = epsilon / (sigma * sigma) * (-48.0 * s6 + 24.0) * s6 * s2;
      fx += fm * d[0]; fy += fm * d[1]; fz += fm * d[2];
    }
    force3[3 * i] = fx; force3[3 * i + 1] = fy; force3[3 * i + 2] = fz;
  }
}

/* ---- VerletList -----------------------------------------------------------------------------------------
 *   BasicNeighbourList_ns::fillBasicNeighbourList (K7)   Interactor/NeighbourList/BasicList/BasicListBase.cuh:41-75
 *   BasicNeighbourList_ns::NeighbourIterator              Interactor/NeighbourList/BasicList/NeighbourContainer.cuh:54-104
 *   VerletListBase_ns::checkMaximumDrift (K8)             Interactor/NeighbourList/VerletList/VerletListBase.cuh:55-69
 * Details that are easy to "fix" by accident: the particle itself IS in its own list (r2 = 0 <= cutOff2); the test is
 * `<=`; a particle that reaches maxNeighboursPerParticle raises the flag with atomicMax(nneigh) and returns WITHOUT writing
 * numberNeighbours; entry k of particle i lives at neighbourList[k*N + i].  Same non-periodic deviation as above. */
ORACLE_API int oracle_verletlist_fill(const real4 *sortPos, int N, const uint *cellStart, const int *cellEnd, uint validCell,
                                      const real *gridL, const int *gridPeriodic, const int *cellDim, const real *boxL,
                                      const int *boxPeriodic, real cutOff2, int maxNeighboursPerParticle, int *neighbourList,
                                      int *numberNeighbours) {
  Box gbox = box_from(gridL, gridPeriodic);
  Grid grid = grid_make(gbox, mki3(cellDim[0], cellDim[1], cellDim[2]));
  Box box = box_from(boxL, boxPeriodic);
  const int3 n = grid.cellDim;
  const int3 nperdim = mki3((n.x > 1 ? 3 : 1), (n.y > 1 ? 3 : 1), (n.z > 1 ? 3 : 1));
  const int numberNeighbourCells = nperdim.x * nperdim.y * nperdim.z;
  int tooMany = 0;
  for (int id = 0; id < N; id++) {
    int nneigh = 0, aborted = 0;
    const real4 pi = sortPos[id];
    const int3 celli = grid_get_cell(&grid, mk3(pi.x, pi.y, pi.z));
    for (int currentCell = 0; currentCell < numberNeighbourCells && !aborted; currentCell++) {
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
        const real3 rij = box_apply_pbc(&box, mk3(pj.x - pi.x, pj.y - pi.y, pj.z - pi.z));
        if (dot3(rij, rij) <= cutOff2) {
          nneigh++;
          if (nneigh >= maxNeighboursPerParticle) {
            if (nneigh > tooMany) tooMany = nneigh;
            aborted = 1;
            break;
          }
          neighbourList[(size_t)(nneigh - 1) * N + id] = j;
        }
      }
    }
    if (!aborted) numberNeighbours[id] = nneigh;
  }
  return tooMany;
}

/* number of particles with |pbc(current - stored)|^2 >= maxDistAllowed^2 */
ORACLE_API unsigned oracle_verletlist_check_drift(const real4 *currentPos, const real4 *storedPos, int N, real maxDistAllowed,
                                                  const real *boxL, const int *boxPeriodic) {
  Box box = box_from(boxL, boxPeriodic);
  unsigned flag = 0;
  for (int id = 0; id < N; id++) {
    const real3 rij = box_apply_pbc(&box, mk3(currentPos[id].x - storedPos[id].x, currentPos[id].y - storedPos[id].y,
                                              currentPos[id].z - storedPos[id].z));
    if (dot3(rij, rij) >= maxDistAllowed * maxDistAllowed) flag++;
  }
  return flag;
}

/* transverseWithNeighbourContainer (common.cuh:10-34) driven by the list: neighbours k = 0..numberNeighbours[i]-1 in list
 * order, positions from the CURRENT sortPos (VerletListBase::updateSortedPositions, VerletListBase.cuh:140-151). */
ORACLE_API void oracle_lj_transverse_verletlist(const real4 *sortPos, const int *groupIndex, const int *globalIndex, int N,
                                                const int *neighbourList, const int *numberNeighbours, const real *boxL,
                                                const int *boxPeriodic, const real *paramTable, int ntypes, real4 *force,
                                                real *energy, real *virial) {
  Box box = box_from(boxL, boxPeriodic);
  const LJPairParameters *tbl = (const LJPairParameters *)paramTable;
#pragma omp parallel for schedule(static)
  for (int id = 0; id < N; id++) {
    const int gi = groupIndex[id];
    const int ori = globalIndex ? globalIndex[gi] : gi;
    const real4 pi = sortPos[id];
    FEV quantity = {{0, 0, 0}, 0, 0};
    const int nn = numberNeighbours[id];
    for (int k = 0; k < nn; k++) {
      const int j = neighbourList[(size_t)k * N + id];
      lj_pair_acc(&quantity, &box, tbl, ntypes, pi, sortPos[j], force || virial, energy != NULL, virial != NULL);
    }
    if (force) { force[ori].x += quantity.force.x; force[ori].y += quantity.force.y; force[ori].z += quantity.force.z; force[ori].w += 0; }
    if (energy) energy[ori] += quantity.energy;
    if (virial) virial[ori] += quantity.virial;
  }
}
