// Path A — neighbour traversal with the Lennard-Jones Transverser, for gfx950.
//
// Reference behaviour (what must come out, in which order it is summed):
//   transverseWithNeighbourContainer   Interactor/NeighbourList/common.cuh:10-34
//   27-cell walk, x fastest then y, z  Interactor/NeighbourList/CellList/NeighbourContainer.cuh:95-130
//   Radial<LJ>::Transverser            Interactor/Potential/RadialPotential.cuh:107-127
//   LJFunctor::force/energy            Interactor/Potential/Potential.cuh:37-65
//   NBody tile loop                    Interactor/NBodyBase.cuh:46-116
//
// Kernels here keep the reference's SUMMATION ORDER (bit-identical forces, UAMMD_LJ_ALGO_EXACT); the fast path that does not is in
// lj_tile.hip (UAMMD_LJ_ALGO_TILE, what AUTO selects):
//   k_lj_general  thread per sorted particle, walks the cells through global memory.  Any grid (collapsed dimensions, non periodic,
//                 2D), any number of types, F/E/V; two-phase evaluation (scan -> per-lane FIFO -> drain).
//   k_lj_ring     the same with a ring FIFO and the per-cell range table (partial drains).
//   k_lj_ringh    + half-precision superset prefilter, two candidates per load (the fastest bit-exact kernel: 0.27 ms at C3).
//   k_lj_nbody    all pairs with LDS tiles (small boxes; PairForces.cu:49-53).
// (Round 1 also carried an LDS brick kernel, a scalar-streamed "quad" kernel, a wave-staged walk and a cell-per-wave kernel; all were
// measured slower than k_lj_ringh — DESIGN.md 5.2 keeps the numbers — and were removed when the tile kernels replaced them.)
#include "celllist.hpp"
#include "lj_common.hpp"
#include "gj_step.hpp"
#include "ring_scan.hpp"

#include <string>

namespace uammd_hip {

// ---- two-phase pair evaluation ---------------------------------------------------------------------
// Only ~15 % of the candidate pairs of a 27-cell walk are inside the cut-off (4.19 rc^3 of 27 rc^3),
// and on gfx950 the IEEE division + force polynomial cost ~3x the distance test.  So each lane first
// SCANS its candidates (distance test only) and appends the index of every candidate inside the
// cut-off to a small per-lane FIFO in LDS; when any lane's FIFO is nearly full (wave-uniform test)
// the wave DRAINS: every lane evaluates its queued pairs, in FIFO order.  The order of the float
// accumulation is therefore still the reference's j order and the results stay bit-identical.
template <class QT, int QCAP, int QSTRIDE> struct PairQueue {
  QT *slot;  // this lane's entry 0; entry t lives at slot[t * QSTRIDE]
  int n;
};

template <bool PBC, bool NT1, bool WE, bool WV, class QT, int QCAP, int QSTRIDE>
UH_D void lj_drain(Acc &acc, PairQueue<QT, QCAP, QSTRIDE> &Q, const float4 *__restrict__ P, const float4 &pi,
                   const BoxT<float> &box, const LJParams &p1, const LJParams *tbl, int ntypes) {
  const int n = Q.n;
  for (int t = 0; t < n; t += 4) {
    int jj[4];
    float4 c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) jj[u] = (int)Q.slot[min(t + u, n - 1) * QSTRIDE];
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = P[jj[u]];
    real3f r[4];
    float f[4], e[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (NT1) lj_eval<PBC, WE>(box, p1, pi, c[u], r[u], f[u], e[u]);
      else lj_eval<PBC, WE>(box, lj_lookup(tbl, ntypes, (int)pi.w, (int)c[u].w), pi, c[u], r[u], f[u], e[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // added in FIFO order; the clamped tail repeats the last pair with weight 0
      const bool live = t + u < n;
      lj_acc<WE, WV>(acc, r[u], live ? f[u] : 0.0f, live ? e[u] : 0.0f);
    }
  }
  Q.n = 0;
}

// Scans j in [jb, je) in order; rc2 = largest squared cut-off of the type table (a superset filter,
// the drain applies the exact per-pair cut-off).  !(r2 >= rc2) keeps NaN pairs, like the reference.
template <bool PBC, bool NT1, bool WE, bool WV, class QT, int QCAP, int QSTRIDE>
UH_D void lj_scan(Acc &acc, PairQueue<QT, QCAP, QSTRIDE> &Q, bool drainPBC, const float4 *__restrict__ P, int jb,
                  int je, const float4 &pi, const BoxT<float> &box, float rc2, const LJParams &p1,
                  const LJParams *tbl, int ntypes) {
  // The FIFO state of this loop is ONE 32-bit LDS address per lane (the next free entry): an append is a masked
  // ds_write + one add, with no per-candidate branch.
  using LdsQT = __attribute__((address_space(3))) QT;
  constexpr uint kStep = QSTRIDE * sizeof(QT);
  const uint q0 = (uint)(uintptr_t)(LdsQT *)Q.slot;
  uint qa = q0 + (uint)Q.n * kStep;
  for (int j = jb; j < je; j += 8) {
    if (__any(qa > q0 + (QCAP - 8) * kStep)) {  // wave-uniform; the full minimum image is exact for every queued pair
      Q.n = (int)((qa - q0) / kStep);
      if (drainPBC) lj_drain<true, NT1, WE, WV>(acc, Q, P, pi, box, p1, tbl, ntypes);
      else lj_drain<false, NT1, WE, WV>(acc, Q, P, pi, box, p1, tbl, ntypes);
      qa = q0;
    }
    // eight candidates from ONE address (immediate offsets); entries past the end of the cell are other particles or the
    // padding of the array (CellList::update allocates N + 8) and are masked by u < rem below
    const float4 *__restrict__ pj = P + j;
    float4 c[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = pj[u];
    float d[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) d[u] = lj_dist2<PBC>(box, pi, c[u]);
    const int rem = je - j;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool hit = !(d[u] >= rc2) & (u < rem);
      if (hit) { *(LdsQT *)(uintptr_t)qa = (QT)(j + u); qa += kStep; }
    }
  }
  Q.n = (int)((qa - q0) / kStep);
}

constexpr int kQCapGeneral = 24;  // per-lane FIFO depth of the global-memory kernels (uint entries)
constexpr int kQCapBrick = 32;    // per-lane FIFO depth of the all-pairs kernel (ushort LDS tile indices)

// ---- general walk ----------------------------------------------------------------------------------
template <bool NT1, bool WE, bool WV, class QT, int QCAP, int QSTRIDE>
UH_D void walk_global(Acc &acc, PairQueue<QT, QCAP, QSTRIDE> &Q, const ListView &cl, const GridT<float> &grid,
                      const BoxT<float> &box, const LJParams *tbl, int ntypes, const LJParams &p1, float rc2,
                      const float4 &pi) {
  const int3 n = grid.cellDim;
  const int npx = n.x > 1 ? 3 : 1, npy = n.y > 1 ? 3 : 1, npz = n.z > 1 ? 3 : 1;
  const int numberNeighbourCells = npx * npy * npz;
  const int3 celli = grid.getCell(real3f{pi.x, pi.y, pi.z});
  // The minimum image is the identity (offset 0, r + 0*L == r) for a pair whose two particles are stored
  // inside the primary box and whose cells are direct (unwrapped) neighbours on a grid with >= 5 cells
  // per dimension: then |d| <= 2 cells <= 0.4 L and floor(d*(-1/L)+0.5) == 0 exactly.  Skipping it is
  // decided per neighbour cell for the whole wave (ballot), so the code stays convergent.
  const bool sameBox = box.boxSize.x == grid.box.boxSize.x && box.boxSize.y == grid.box.boxSize.y &&
                       box.boxSize.z == grid.box.boxSize.z && box.px() == grid.box.px() &&
                       box.py() == grid.box.py() && box.pz() == grid.box.pz();
  const bool smallGrid = n.x < 5 || n.y < 5 || n.z < 5 || !cl.cellOutside || !sameBox;
  const float hx = 0.5f * box.boxSize.x, hy = 0.5f * box.boxSize.y, hz = 0.5f * box.boxSize.z;
  const bool iOut = !(pi.x >= -hx && pi.x < hx && pi.y >= -hy && pi.y < hy && pi.z >= -hz && pi.z < hz);
  bool drainPBC = false;
  if (cl.cellRange) {
    // One 8-byte load per neighbour cell instead of the dependent cellStart -> cellEnd -> cellOutside chain, issued one cell
    // ahead of the scan that needs it.  A neighbour outside a non periodic box reads the {0, 0} entry past the last cell.
    const int ncells = n.x * n.y * n.z;
    auto fetch = [&](int cc, uint2 &rg, bool &wrapped) {
      int3 cellj = celli;
      if (npx > 1) cellj.x += cc % 3 - 1;
      if (npy > 1) cellj.y += (cc / npx) % 3 - 1;
      if (npz > 1) cellj.z += cc / (npx * npy) - 1;
      const int3 raw = cellj;
      cellj.x = grid.pbc_x(cellj.x);
      cellj.y = grid.pbc_y(cellj.y);
      cellj.z = grid.pbc_z(cellj.z);
      const bool exists = !(cellj.x < 0 || cellj.x >= n.x || cellj.y < 0 || cellj.y >= n.y || cellj.z < 0 || cellj.z >= n.z);
      wrapped = raw.x != cellj.x || raw.y != cellj.y || raw.z != cellj.z;
      rg = cl.cellRange[exists ? grid.getCellIndex(cellj) : ncells];
    };
    uint2 rg;
    bool wrapped;
    fetch(0, rg, wrapped);
    for (int cc = 0; cc < numberNeighbourCells; ++cc) {
      uint2 rgNext = make_uint2(0u, 0u);
      bool wrappedNext = false;
      if (cc + 1 < numberNeighbourCells) fetch(cc + 1, rgNext, wrappedNext);
      const int first = (int)rg.x, last = (int)(rg.y & 0x7fffffffu);
      const bool needPBC = first < last && (smallGrid || iOut || wrapped || (rg.y >> 31) != 0u);
      if (__any(needPBC)) {
        drainPBC = true;
        lj_scan<true, NT1, WE, WV>(acc, Q, drainPBC, cl.sortPos, first, last, pi, box, rc2, p1, tbl, ntypes);
      } else {
        lj_scan<false, NT1, WE, WV>(acc, Q, drainPBC, cl.sortPos, first, last, pi, box, rc2, p1, tbl, ntypes);
      }
      rg = rgNext;
      wrapped = wrappedNext;
    }
    if (drainPBC) lj_drain<true, NT1, WE, WV>(acc, Q, cl.sortPos, pi, box, p1, tbl, ntypes);
    else lj_drain<false, NT1, WE, WV>(acc, Q, cl.sortPos, pi, box, p1, tbl, ntypes);
    return;
  }
  for (int cc = 0; cc < numberNeighbourCells; ++cc) {
    int3 cellj = celli;
    if (npx > 1) cellj.x += cc % 3 - 1;
    if (npy > 1) cellj.y += (cc / npx) % 3 - 1;
    if (npz > 1) cellj.z += cc / (npx * npy) - 1;
    const int3 raw = cellj;
    cellj.x = grid.pbc_x(cellj.x);
    cellj.y = grid.pbc_y(cellj.y);
    cellj.z = grid.pbc_z(cellj.z);
    // outside a non periodic box: no such cell (see DESIGN.md "non-periodic neighbours")
    const bool exists = !(cellj.x < 0 || cellj.x >= n.x || cellj.y < 0 || cellj.y >= n.y || cellj.z < 0 || cellj.z >= n.z);
    int first = 0, last = 0;
    bool needPBC = false;
    if (exists) {
      const int icellj = grid.getCellIndex(cellj);
      const uint cs = cl.cellStart[icellj];
      if (cs >= cl.validCell) {
        first = (int)(cs - cl.validCell);
        last = cl.cellEnd[icellj];
        const bool wrapped = raw.x != cellj.x || raw.y != cellj.y || raw.z != cellj.z;
        needPBC = smallGrid || iOut || wrapped || cl.cellOutside[icellj] != 0;
      }
    }
    if (__any(needPBC)) {
      drainPBC = true;
      lj_scan<true, NT1, WE, WV>(acc, Q, drainPBC, cl.sortPos, first, last, pi, box, rc2, p1, tbl, ntypes);
    } else {
      lj_scan<false, NT1, WE, WV>(acc, Q, drainPBC, cl.sortPos, first, last, pi, box, rc2, p1, tbl, ntypes);
    }
  }
  if (drainPBC) lj_drain<true, NT1, WE, WV>(acc, Q, cl.sortPos, pi, box, p1, tbl, ntypes);
  else lj_drain<false, NT1, WE, WV>(acc, Q, cl.sortPos, pi, box, p1, tbl, ntypes);
}

template <bool NT1, bool WE, bool WV>
__global__ void __launch_bounds__(128) k_lj_general(ListView cl, GridT<float> grid, BoxT<float> box,
                                                     const LJParams *__restrict__ tbl, int ntypes, Outputs out) {
  __shared__ uint gq[kQCapGeneral * 128];
  const int id = (int)xcd_contiguous_block(blockIdx.x, gridDim.x) * 128 + threadIdx.x;
  if (id >= cl.N) return;
  const int gi = cl.groupIndex[id];
  if (gi >= cl.numOwned) return;  // a ghost: ghost cells are whole waves at the slab faces, so this skips their walks
  const int ori = out.globalIndex ? out.globalIndex[gi] : gi;
  const float4 pi = cl.sortPos[id];
  LJParams p1 = tbl[0];
  const float rc2 = NT1 ? p1.cutOff2 : lj_max_cutoff2(tbl, ntypes);
  PairQueue<uint, kQCapGeneral, 128> Q{gq + threadIdx.x, 0};
  Acc acc;
  walk_global<NT1, WE, WV>(acc, Q, cl, grid, box, tbl, ntypes, p1, rc2, pi);
  write_out(out, ori, acc);
}

// ---- ring FIFO: drain a few pairs from EVERY lane instead of everything from the fullest -------------------------------
// With the linear FIFO above a drain empties every lane as soon as ONE lane is nearly full.  The hits arrive in bursts (a
// lane's own cell is all hits, a corner cell none), so at that moment the typical lane holds ~6 pairs while the loop runs
// to the fullest lane's ~20: measured at C3, a wave spends 200 pair evaluations per lane-slot for 52 useful ones and the
// drain is 44 % of the kernel's VALU instructions.  Here the FIFO is a ring of kRingCap entries per lane and a drain takes
// at most kRingTake pairs from each lane: lanes that fill slowly keep their pairs until they have a full batch, and the
// number of drain iterations follows the busiest lane's TOTAL instead of the sum of the per-drain maxima.  The order in
// which a lane evaluates its pairs is unchanged, so the results are bit-identical to the linear FIFO.
template <bool PBC, bool NT1, bool WE, bool WV>
UH_D void lj_drain_ring(Acc &acc, RingQ &Q, int take, const float4 *__restrict__ P, const float4 &pi,
                        const BoxT<float> &box, const LJParams &p1, const LJParams *tbl, int ntypes) {
  const int n = min((int)(Q.bytes() / kRingStep), take);
  const bool fastDivOK = NT1 && p1.sigma2 >= kDivLo && p1.sigma2 <= kDivHi && p1.cutOff2 <= kDivHi;  // uniform
  __builtin_amdgcn_s_setprio(0);  // the drain yields to waves that are scanning (their loads start earlier): 0.281 -> 0.274 ms at C3
  for (int t = 0; t < n; t += 4) {
    // slots past the lane's n-th entry read whatever the ring holds there — a neighbour queued earlier or the lane's own
    // particle (the kernels initialise the ring with it): a valid address and no foreign NaN — and are masked below (weight 0)
    uint jj[4];
    float4 c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) jj[u] = *(const LdsU32 *)(uintptr_t)Q.wrap(Q.head + (uint)(t + u) * kRingStep);
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = P[jj[u]];
    real3f r[4];
    float f[4], e[4];
    if (NT1) {
      // one type: sigma2 / r2 without the scaling and fix-up instructions of the general division when the wave's operands
      // allow it (div_in_range); pairs beyond the cut-off are masked whatever their quotient
      float r2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        r[u] = real3f{c[u].x - pi.x, c[u].y - pi.y, c[u].z - pi.z};
        if (PBC) r[u] = box.apply_pbc(r[u]);
        r2[u] = dot3(r[u], r[u]);
      }
      const float r2min = fminf(fminf(r2[0], r2[1]), fminf(r2[2], r2[3]));
      const bool plain = fastDivOK && !__any(r2min < kDivLo);
      if (plain) {
#pragma unroll
        for (int u = 0; u < 4; ++u) lj_eval_r2_fastdiv<WE>(p1, r2[u], f[u], e[u]);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) lj_eval<PBC, WE>(box, p1, pi, c[u], r[u], f[u], e[u]);
      }
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        lj_eval<PBC, WE>(box, lj_lookup(tbl, ntypes, (int)pi.w, (int)c[u].w), pi, c[u], r[u], f[u], e[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // added in FIFO order; the clamped tail repeats the last pair with weight 0
      const bool live = t + u < n;
      lj_acc<WE, WV>(acc, r[u], live ? f[u] : 0.0f, live ? e[u] : 0.0f);
    }
  }
  __builtin_amdgcn_s_setprio(1);
  Q.head = Q.wrap(Q.head + (uint)n * kRingStep);
}

template <bool PBC, bool NT1, bool WE, bool WV>
UH_D void lj_scan_ring(Acc &acc, RingQ &Q, bool drainPBC, const float4 *__restrict__ P, int jb, int je, const float4 &pi,
                       const BoxT<float> &box, float rc2, const LJParams &p1, const LJParams *tbl, int ntypes) {
  for (int j = jb; j < je; j += 8) {
    while (__any(Q.bytes() > (kRingCap - 9) * kRingStep)) {  // wave-uniform: some lane could not take 8 more
      if (drainPBC) lj_drain_ring<true, NT1, WE, WV>(acc, Q, kRingTake, P, pi, box, p1, tbl, ntypes);
      else lj_drain_ring<false, NT1, WE, WV>(acc, Q, kRingTake, P, pi, box, p1, tbl, ntypes);
    }
    const float4 *__restrict__ pj = P + j;
    float4 c[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = pj[u];
    float d[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) d[u] = lj_dist2<PBC>(box, pi, c[u]);
    const int rem = je - j;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool hit = !(d[u] >= rc2) & (u < rem);
      if (hit) { *(LdsU32 *)(uintptr_t)Q.tail = (uint)(j + u); Q.tail = Q.wrap(Q.tail + kRingStep); }
    }
  }
}

template <bool NT1, bool WE, bool WV>
__global__ void __launch_bounds__(kRingLanes) k_lj_ring(ListView cl, GridT<float> grid, BoxT<float> box,
                                                  const LJParams *__restrict__ tbl, int ntypes, Outputs out) {
  __shared__ __attribute__((aligned(kRingCap * kRingLanes * 4))) uint ring[kRingCap * kRingLanes];
  const int id = (int)xcd_contiguous_block(blockIdx.x, gridDim.x) * kRingLanes + threadIdx.x;
  if (id >= cl.N) return;
  const int gi = cl.groupIndex[id];
  if (gi >= cl.numOwned) return;
  const int ori = out.globalIndex ? out.globalIndex[gi] : gi;
  const float4 pi = cl.sortPos[id];
  LJParams p1 = tbl[0];
  const float rc2 = NT1 ? p1.cutOff2 : lj_max_cutoff2(tbl, ntypes);
  RingQ Q;
  Q.base = (uint)(uintptr_t)(LdsU32 *)ring;
  Q.head = Q.tail = Q.base + threadIdx.x * 4u;
#pragma unroll
  for (int t = 0; t < kRingCap; ++t) ring[t * kRingLanes + threadIdx.x] = (uint)id;  // every entry is a valid index from the start: the particle itself
  Acc acc;

  const int3 n = grid.cellDim;
  const int npx = n.x > 1 ? 3 : 1, npy = n.y > 1 ? 3 : 1, npz = n.z > 1 ? 3 : 1;
  const int numberNeighbourCells = npx * npy * npz;
  const int ncells = n.x * n.y * n.z;
  const int3 celli = grid.getCell(real3f{pi.x, pi.y, pi.z});
  const bool sameBox = box.boxSize.x == grid.box.boxSize.x && box.boxSize.y == grid.box.boxSize.y &&
                       box.boxSize.z == grid.box.boxSize.z && box.px() == grid.box.px() &&
                       box.py() == grid.box.py() && box.pz() == grid.box.pz();
  const bool smallGrid = n.x < 5 || n.y < 5 || n.z < 5 || !sameBox;
  const float hx = 0.5f * box.boxSize.x, hy = 0.5f * box.boxSize.y, hz = 0.5f * box.boxSize.z;
  const bool iOut = !(pi.x >= -hx && pi.x < hx && pi.y >= -hy && pi.y < hy && pi.z >= -hz && pi.z < hz);
  auto fetch = [&](int cc, uint2 &rg, bool &wrapped) {
    int3 cellj = celli;
    if (npx > 1) cellj.x += cc % 3 - 1;
    if (npy > 1) cellj.y += (cc / npx) % 3 - 1;
    if (npz > 1) cellj.z += cc / (npx * npy) - 1;
    const int3 raw = cellj;
    cellj.x = grid.pbc_x(cellj.x);
    cellj.y = grid.pbc_y(cellj.y);
    cellj.z = grid.pbc_z(cellj.z);
    const bool exists = !(cellj.x < 0 || cellj.x >= n.x || cellj.y < 0 || cellj.y >= n.y || cellj.z < 0 || cellj.z >= n.z);
    wrapped = raw.x != cellj.x || raw.y != cellj.y || raw.z != cellj.z;
    rg = cl.cellRange[exists ? grid.getCellIndex(cellj) : ncells];
  };
  uint2 rg;
  bool wrapped;
  fetch(0, rg, wrapped);
  bool drainPBC = false;
  for (int cc = 0; cc < numberNeighbourCells; ++cc) {
    uint2 rgNext = make_uint2(0u, 0u);
    bool wrappedNext = false;
    if (cc + 1 < numberNeighbourCells) fetch(cc + 1, rgNext, wrappedNext);
    const int first = (int)rg.x, last = (int)(rg.y & 0x7fffffffu);
    const bool needPBC = first < last && (smallGrid || iOut || wrapped || (rg.y >> 31) != 0u);
    if (__any(needPBC)) {
      drainPBC = true;
      lj_scan_ring<true, NT1, WE, WV>(acc, Q, drainPBC, cl.sortPos, first, last, pi, box, rc2, p1, tbl, ntypes);
    } else {
      lj_scan_ring<false, NT1, WE, WV>(acc, Q, drainPBC, cl.sortPos, first, last, pi, box, rc2, p1, tbl, ntypes);
    }
    rg = rgNext;
    wrapped = wrappedNext;
  }
  if (drainPBC) lj_drain_ring<true, NT1, WE, WV>(acc, Q, kRingCap, cl.sortPos, pi, box, p1, tbl, ntypes);
  else lj_drain_ring<false, NT1, WE, WV>(acc, Q, kRingCap, cl.sortPos, pi, box, p1, tbl, ntypes);
  write_out(out, ori, acc);
}

// ---- half-precision prefilter: two candidates per load --------------------------------------------------------------
// After the ring FIFO the walk is bound by the texture addresser again: a 64-lane global load of 8..16 bytes costs ~16.5
// clocks of the CU's single addresser whatever its width and whatever the lanes' addresses (tools/vmem_ubench.hip), and the
// scan issues one per candidate.  Here the scan reads the packed copy of the cell list (two candidates per 12-byte load,
// half precision, relative to the candidates' cell centre) and tests |r|^2 < rc^2 + margin with three packed-half
// instructions per candidate.  The test is a strict SUPERSET of the exact one (margin derived below), the ring entries are
// the same particle indices in the same order, and the drain re-tests every pair in full precision: a false hit adds
// fma(0, r, acc) == acc, so the forces stay bit-identical to k_lj_general.
//   error budget, in units of the largest cell edge (rc <= 1, |candidate| <= 0.5, |particle relative to the neighbour
//   centre| <= 1.5): candidate rounding 2^-13, particle rounding 2^-11, subtraction rounding 2^-11 -> |d_err| <= 1.1e-3 per
//   component; near the threshold (r <= 1) that is 2 sqrt(3) r d_err = 3.8e-3 in r^2, plus three half-precision roundings of
//   the squares and sums (<= 1.2e-3): 5e-3.  kHalfMargin is more than twice that.
// The displacement uses the RAW neighbour offset (no minimum image): on a grid with >= 3 cells along every periodic
// direction the image of j within the cut-off of i, if any, is the one in the raw-adjacent cell (cells are >= rc wide).
template <bool NT1, bool WE, bool WV>
UH_D void lj_scan_ringh(Acc &acc, RingQ &Q, bool drainPBC, const float4 *__restrict__ P, const uint3 *__restrict__ PK, int jb,
                        int je, half2_t px, half2_t py, half2_t pz, _Float16 thr, const float4 &pi, const BoxT<float> &box,
                        const LJParams &p1, const LJParams *tbl, int ntypes) {
  half_scan(Q, PK, jb, je, px, py, pz, thr, [&]() {
    while (__any(Q.bytes() > (kRingCap - 9) * kRingStep)) {  // wave-uniform: some lane could not take 8 more
      if (drainPBC) lj_drain_ring<true, NT1, WE, WV>(acc, Q, kRingTake, P, pi, box, p1, tbl, ntypes);
      else lj_drain_ring<false, NT1, WE, WV>(acc, Q, kRingTake, P, pi, box, p1, tbl, ntypes);
    }
  });
}

template <bool NT1, bool WE, bool WV>
// 7 waves per SIMD (<= 72 VGPRs): measured 0.321 / 0.296 / 0.281 / 0.274 / 0.280 ms at 4 / 5 / 6 (the compiler's choice) / 7 / 8
__attribute__((amdgpu_waves_per_eu(7, 7)))
__global__ void __launch_bounds__(kRingLanes) k_lj_ringh(ListView cl, GridT<float> grid, BoxT<float> box,
                                                   const LJParams *__restrict__ tbl, int ntypes, Outputs out) {
  __shared__ __attribute__((aligned(kRingCap * kRingLanes * 4))) uint ring[kRingCap * kRingLanes];
  const int id = (int)xcd_contiguous_block(blockIdx.x, gridDim.x) * kRingLanes + threadIdx.x;
  if (id >= cl.N) return;
  const int gi = cl.groupIndex[id];
  if (gi >= cl.numOwned) return;
  const int ori = out.globalIndex ? out.globalIndex[gi] : gi;
  const float4 pi = cl.sortPos[id];
  LJParams p1 = tbl[0];
  const float rc2 = NT1 ? p1.cutOff2 : lj_max_cutoff2(tbl, ntypes);
  RingQ Q;
  Q.base = (uint)(uintptr_t)(LdsU32 *)ring;
  Q.head = Q.tail = Q.base + threadIdx.x * 4u;
#pragma unroll
  for (int t = 0; t < kRingCap; ++t) ring[t * kRingLanes + threadIdx.x] = (uint)id;  // every entry is a valid index from the start: the particle itself
  Acc acc;

  const int3 n = grid.cellDim;
  const int npx = n.x > 1 ? 3 : 1, npy = n.y > 1 ? 3 : 1, npz = n.z > 1 ? 3 : 1;
  const int numberRows = npy * npz;
  const int ncells = n.x * n.y * n.z;
  const int3 celli = grid.getCell(real3f{pi.x, pi.y, pi.z});
  const bool smallGrid = n.x < 5 || n.y < 5 || n.z < 5;  // the launcher guarantees that box is the grid's box
  const float hx = 0.5f * box.boxSize.x, hy = 0.5f * box.boxSize.y, hz = 0.5f * box.boxSize.z;
  const bool iOut = !(pi.x >= -hx && pi.x < hx && pi.y >= -hy && pi.y < hy && pi.z >= -hz && pi.z < hz);
  const bool alwaysPBC = smallGrid || iOut;
  // the particle relative to its own cell centre, and the threshold, in units of the largest cell edge
  const float s = cl.packScale;
  const real3f own = grid.distanceToCellCenter(real3f{pi.x, pi.y, pi.z}, celli);
  const float ox = own.x * s, oy = own.y * s, oz = own.z * s;
  const float sx = grid.cellSize.x * s, sy = grid.cellSize.y * s, sz = grid.cellSize.z * s;
  // a cut-off larger than a cell edge breaks the one-image argument above: accept everything, the drain decides
  const float hmin = fminf(grid.cellSize.x, fminf(grid.cellSize.y, grid.cellSize.z));
  const _Float16 thr = rc2 <= hmin * hmin * 1.0001f ? (_Float16)((rc2 * s * s + kHalfMargin) * 1.002f) : (_Float16)__builtin_inff();

  // The walk is the reference's (x fastest, then y, then z) as 9 rows of 3 cells.  What depends on the x offset only is
  // computed once per particle: the wrapped x index (or "no such cell"), whether it wrapped, and the particle's packed x
  // relative to that column.  A row adds its base index (wrapped y, z), its flags and the packed y, z.  The three ranges of a
  // row are loaded while the previous row is scanned.
  constexpr int kNoCell = -(1 << 30);
  int xIdx[3];
  bool xWrap[3];
  half2_t qx[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int raw = celli.x + (k - 1), w = grid.pbc_x(raw);
    xIdx[k] = (w < 0 || w >= n.x) ? kNoCell : w;
    xWrap[k] = w != raw;
    const _Float16 q = (_Float16)fmaf(-(float)(k - 1), sx, ox);
    qx[k] = half2_t{q, q};
  }
  auto row_base = [&](int row, int &base, bool &wrapped, int &offy, int &offz) {
    offy = npy > 1 ? row % 3 - 1 : 0;
    offz = npz > 1 ? row / npy - 1 : 0;
    const int rawy = celli.y + offy, rawz = celli.z + offz;
    const int wy = grid.pbc_y(rawy), wz = grid.pbc_z(rawz);
    const bool exists = !(wy < 0 || wy >= n.y || wz < 0 || wz >= n.z);
    base = exists ? (wz * n.y + wy) * n.x : kNoCell;
    wrapped = wy != rawy || wz != rawz;
  };
  auto fetch_row = [&](int base, uint2 (&rg)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int idx = base + xIdx[k];  // negative when the row or the column does not exist
      rg[k] = (npx > 1 || k == 1) ? cl.cellRange[idx < 0 ? ncells : idx] : make_uint2(0u, 0u);
    }
  };
  int base, offy, offz;
  bool rowWrapped;
  uint2 rg[3];
  row_base(0, base, rowWrapped, offy, offz);
  fetch_row(base, rg);
  bool drainPBC = false;
  for (int row = 0; row < numberRows; ++row) {
    int baseN, offyN, offzN;
    bool rowWrappedN;
    uint2 rgN[3];
    row_base(row + 1 < numberRows ? row + 1 : row, baseN, rowWrappedN, offyN, offzN);
    fetch_row(baseN, rgN);
    const _Float16 qyh = (_Float16)fmaf(-(float)offy, sy, oy), qzh = (_Float16)fmaf(-(float)offz, sz, oz);
    const half2_t qy = {qyh, qyh}, qz = {qzh, qzh};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (npx == 1 && k != 1) continue;
      const int first = (int)rg[k].x, last = (int)(rg[k].y & 0x7fffffffu);
      const bool needPBC = first < last && (alwaysPBC || rowWrapped || xWrap[k] || (rg[k].y >> 31) != 0u);
      if (__any(needPBC)) drainPBC = true;
      lj_scan_ringh<NT1, WE, WV>(acc, Q, drainPBC, cl.sortPos, cl.packHalf, first, last, qx[k], qy, qz, thr, pi, box, p1, tbl,
                                 ntypes);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) rg[k] = rgN[k];
    rowWrapped = rowWrappedN;
    offy = offyN;
    offz = offzN;
  }
  if (drainPBC) lj_drain_ring<true, NT1, WE, WV>(acc, Q, kRingCap, cl.sortPos, pi, box, p1, tbl, ntypes);
  else lj_drain_ring<false, NT1, WE, WV>(acc, Q, kRingCap, cl.sortPos, pi, box, p1, tbl, ntypes);
  write_out(out, ori, acc);
}

// ---- all pairs -------------------------------------------------------------------------------------
template <bool NT1, bool WE, bool WV>
__global__ void __launch_bounds__(128) k_lj_nbody(const float4 *__restrict__ pos, int N, BoxT<float> box,
                                                   const LJParams *__restrict__ tbl, int ntypes, Outputs out) {
  __shared__ float4 tile[128];
  __shared__ unsigned short nq[kQCapBrick * 128];
  const int t = blockIdx.x * 128 + threadIdx.x;
  const bool active = t < N;
  const int id = active ? (out.globalIndex ? out.globalIndex[t] : t) : 0;
  const float4 pi = active ? pos[id] : make_float4(0.f, 0.f, 0.f, 0.f);
  const LJParams p1 = tbl[0];
  const float rc2 = NT1 ? p1.cutOff2 : lj_max_cutoff2(tbl, ntypes);
  PairQueue<unsigned short, kQCapBrick, 128> Q{nq + threadIdx.x, 0};
  Acc acc;
  const int numTiles = (N + 127) / 128;
  for (int tileIdx = 0; tileIdx < numTiles; ++tileIdx) {
    const int iload = tileIdx * 128 + threadIdx.x;
    if (iload < N) tile[threadIdx.x] = pos[out.globalIndex ? out.globalIndex[iload] : iload];
    __syncthreads();
    if (active) {
      const int cnt = min(128, N - tileIdx * 128);
      lj_scan<true, NT1, WE, WV>(acc, Q, true, tile, 0, cnt, pi, box, rc2, p1, tbl, ntypes);
      lj_drain<true, NT1, WE, WV>(acc, Q, tile, pi, box, p1, tbl, ntypes);  // queue indices are tile-local
    }
    __syncthreads();
  }
  if (active) write_out(out, id, acc);
}

// ---- dispatch --------------------------------------------------------------------------------------
// lj_tile.hip: cell-pair tiles, distance test on the matrix pipe (same pairs, another summation order)
bool lj_tile_supported(const CellList *h, const BoxT<float> &box, float maxCutOff2);
float lj_tile_max_cutoff2(const GridT<float> &g);
template <bool NT1, bool WE, bool WV>
int launch_lj_tile(CellList *h, const ListView &cl, const BoxT<float> &box, const LJParams *tbl, int ntypes, const Outputs &out,
                   int shape, hipStream_t st);

template <bool NT1, bool WE, bool WV>
static int dispatch_celllist(CellList *h, int algo, const BoxT<float> &box, const LJParams *tbl, int ntypes, const Outputs &out,
                             hipStream_t st) {
  ListView cl;
  cl.cellStart = (const uint *)h->cellStart.ptr;
  cl.cellEnd = (const int *)h->cellEnd.ptr;
  cl.sortPos = (const float4 *)h->sortPos.ptr;
  cl.groupIndex = (const int *)h->index.ptr;
  cl.sortHash = (const uint *)h->sortHash.ptr;
  cl.keyStart = (const uint *)h->keyStart.ptr;
  cl.cellOutside = h->haveCellOutside ? (const unsigned char *)h->cellOutside.ptr : nullptr;
  cl.cellRange = h->haveCellOutside ? (const uint2 *)h->cellRange.ptr : nullptr;
  cl.packHalf = nullptr;
  cl.packScale = 0.f;
  cl.validCell = h->validCell;
  cl.N = h->numberParticlesBuilt;
  cl.numOwned = h->numOwned;
  cl.maxCut2Allowed = lj_tile_max_cutoff2(h->grid);
  cl.errFlag = h->devErr;
  cl.tileStats = h->tileStatsOn ? (uint *)h->tileStats.ptr : nullptr;
  switch (algo) {
    case UAMMD_LJ_ALGO_AUTO: case UAMMD_LJ_ALGO_GENERAL: case UAMMD_LJ_ALGO_RING: case UAMMD_LJ_ALGO_RING_HALF: case UAMMD_LJ_ALGO_TILE:
    case UAMMD_LJ_ALGO_TILE1: case UAMMD_LJ_ALGO_EXACT: break;
    default:
      set_last_error("uammd_lj_transverse_celllist: unknown algorithm %d (the brick / quad / staged / cell-per-wave kernels of round 1 "
                     "were removed: TILE replaces them)", algo);
      return -3;
  }
  bool tileOK = false;
  if (algo == UAMMD_LJ_ALGO_TILE || algo == UAMMD_LJ_ALGO_TILE1 || algo == UAMMD_LJ_ALGO_AUTO) {
    // the table lives in device memory: its largest cut-off is read back once per (pointer, size) and remembered with the list
    float maxCut2 = 0.f;
    if (int e = h->lj_max_cutoff2(tbl, ntypes, st, &maxCut2)) return e;
    tileOK = lj_tile_supported(h, box, maxCut2);
  }
  if ((algo == UAMMD_LJ_ALGO_TILE || algo == UAMMD_LJ_ALGO_TILE1) && !tileOK) {
    set_last_error("uammd_lj_transverse_celllist: the tile kernel needs a tabulated list built on the potential's box with 1 (non "
                   "periodic) or >= 3 cells per dimension (>= 4 along a periodic x) and no cell edge shorter than the largest cut-off");
    return -3;
  }
  if (algo == UAMMD_LJ_ALGO_TILE || algo == UAMMD_LJ_ALGO_TILE1 || (algo == UAMMD_LJ_ALGO_AUTO && tileOK))
    return launch_lj_tile<NT1, WE, WV>(h, cl, box, tbl, ntypes, out, algo == UAMMD_LJ_ALGO_TILE1 ? 1 : 4, st);
  if (algo == UAMMD_LJ_ALGO_EXACT) algo = UAMMD_LJ_ALGO_AUTO;  // from here on AUTO = the fastest bit-exact kernel for the grid
  const GridT<float> &g = h->grid;
  if (cl.cellRange && (algo == UAMMD_LJ_ALGO_AUTO || algo == UAMMD_LJ_ALGO_RING || algo == UAMMD_LJ_ALGO_RING_HALF)) {
    // the half-precision prefilter needs the grid's own box, >= 3 cells along every periodic direction with more than one
    // cell, and a 3D grid (ensure_pack); otherwise the full-precision ring walk
    const bool sameBoxAny = box.boxSize.x == g.box.boxSize.x && box.boxSize.y == g.box.boxSize.y &&
                            box.boxSize.z == g.box.boxSize.z && box.px() == g.box.px() && box.py() == g.box.py() &&
                            box.pz() == g.box.pz();
    const bool dimsOK = (g.cellDim.x >= 3 || !g.box.px()) && (g.cellDim.y >= 3 || !g.box.py()) && (g.cellDim.z >= 3 || !g.box.pz());
    bool half = algo != UAMMD_LJ_ALGO_RING && sameBoxAny && dimsOK;
    if (half) {
      const int e = h->ensure_pack(st);
      if (e < 0) return e;
      half = e == 0;
    }
    if (algo == UAMMD_LJ_ALGO_RING_HALF && !half) {
      set_last_error("uammd_lj_transverse_celllist: the half-precision prefilter needs a 3D grid built on the potential's box "
                     "with >= 3 cells along every periodic direction");
      return -3;
    }
    cl.packHalf = half ? (const uint3 *)h->packHalf.ptr : nullptr;
    cl.packScale = h->packScale;
    if (half)
      hipLaunchKernelGGL((k_lj_ringh<NT1, WE, WV>), dim3((cl.N + kRingLanes - 1) / kRingLanes), dim3(kRingLanes), 0, st, cl, g, box, tbl, ntypes, out);
    else
      hipLaunchKernelGGL((k_lj_ring<NT1, WE, WV>), dim3((cl.N + kRingLanes - 1) / kRingLanes), dim3(kRingLanes), 0, st, cl, g, box, tbl, ntypes, out);
    return 0;
  }
  if ((algo == UAMMD_LJ_ALGO_RING || algo == UAMMD_LJ_ALGO_RING_HALF) && !cl.cellRange) {
    set_last_error("uammd_lj_transverse_celllist: the ring kernels need the per-cell range table (a tabulated key space)");
    return -3;
  }
  hipLaunchKernelGGL((k_lj_general<NT1, WE, WV>), dim3((cl.N + 127) / 128), dim3(128), 0, st, cl, g, box, tbl, ntypes, out);
  return 0;
}

template <bool NT1, bool WE, bool WV>
static int dispatch_nbody(const float4 *pos, int N, const BoxT<float> &box, const LJParams *tbl, int ntypes,
                          const Outputs &out, hipStream_t st) {
  hipLaunchKernelGGL((k_lj_nbody<NT1, WE, WV>), dim3((N + 127) / 128), dim3(128), 0, st, pos, N, box, tbl, ntypes, out);
  return 0;
}

}  // namespace uammd_hip

using namespace uammd_hip;

extern "C" {

int uammd_lj_process_pair_parameters(float cutOff, float sigma, float epsilon, int shift,
                                     uammd_lj_pair_parameters *out) {
  // LJFunctor::processPairParameters, Interactor/Potential/Potential.cuh:66-82
  out->cutOff2 = cutOff * cutOff;
  out->sigma2 = sigma * sigma;
  out->epsilonDivSigma2 = epsilon / out->sigma2;
  if (shift) {
    const float invCutOff2 = out->sigma2 / out->cutOff2;
    const float invrc6 = invCutOff2 * invCutOff2 * invCutOff2;
    out->shift = epsilon * 4.0f * invrc6 * (invrc6 - 1.0f);
  } else
    out->shift = 0.0f;
  return 0;
}

int uammd_hip_set_tunable(const char *name, int value) {
  (void)value;
  set_last_error("uammd_hip_set_tunable: unknown tunable or bad value");
  return -1;
}

#define UH_DISPATCH_FEV(FN, ...)                                               \
  do {                                                                         \
    const bool nt1 = ntypes == 1, we = d_energy != nullptr, wv = d_virial != nullptr; \
    if (nt1 && !we && !wv) rc = FN<true, false, false>(__VA_ARGS__);           \
    else if (nt1) rc = FN<true, true, true>(__VA_ARGS__);                      \
    else if (!we && !wv) rc = FN<false, false, false>(__VA_ARGS__);            \
    else rc = FN<false, true, true>(__VA_ARGS__);                              \
  } while (0)

int uammd_lj_transverse_celllist(uammd_celllist *hh, const uammd_lj_pair_parameters *d_paramTable, int ntypes,
                                 const float boxL[3], const int boxPeriodic[3], float *d_force, float *d_energy,
                                 float *d_virial, const int *d_globalIndex, int algo, void *stream) {
  if (!hh || !d_paramTable || ntypes < 1) { set_last_error("uammd_lj_transverse_celllist: bad arguments"); return -1; }
  CellList *h = reinterpret_cast<CellList *>(hh);
  if (h->numberParticlesBuilt == 0) return 0;
  const BoxT<float> box = make_box<float>(boxL, boxPeriodic);
  Outputs out{reinterpret_cast<float4 *>(d_force), d_energy, d_virial, d_globalIndex};
  const LJParams *tbl = reinterpret_cast<const LJParams *>(d_paramTable);
  int rc = 0;
  // E/V kernels also need valid pointers for the quantities not requested: the templates test them.
  UH_DISPATCH_FEV(dispatch_celllist, h, algo, box, tbl, ntypes, out, (hipStream_t)stream);
  if (rc) return rc;
  UH_CHECK(hipGetLastError());
  return 0;
}

// One VerletNVT::GronbechJensen::forwardTime (GronbechJensen.cu:88-115) whose only interactor is PairForces<Potential::LJ, CellList> on the
// whole system (PairForces.cu:43-78), in five launches instead of seven: the first half step rides in the cell list's hash kernel (the new
// position is stored and hashed in one pass), the second in the traversal's store (the force is still in registers).  Same arithmetic as
// uammd_verletnvt_gj(1) -> uammd_celllist_update -> uammd_lj_transverse_celllist -> uammd_verletnvt_gj(2), bit for bit; where the list
// does not take the aggregated counting build or the tile kernel, exactly that sequence runs instead.  d_force must hold f(t) on entry
// (the integrator's first step computes it the plain way) and holds f(t + dt) on return.
int uammd_verletnvt_gj_lj_step(uammd_celllist *hh, float *d_pos, float *d_vel, float *d_force, const float *d_mass, float defaultMass,
                               int N, const float boxL[3], const int boxPeriodic[3], const float updateL[3], const int updatePeriodic[3],
                               const int cellDim[3], const uammd_lj_pair_parameters *d_paramTable, int ntypes, float dt, float friction,
                               int is2D, float noiseAmplitude, unsigned int stepNum, unsigned int seed, int algo, void *stream) {
  if (!hh || !d_pos || !d_vel || !d_force || !d_paramTable || ntypes < 1 || N < 0) { set_last_error("uammd_verletnvt_gj_lj_step: bad arguments"); return -1; }
  if (!d_mass && !(defaultMass > 0)) { set_last_error("uammd_verletnvt_gj_lj_step: no mass array and defaultMass <= 0"); return -1; }
  if (N == 0) return 0;
  CellList *h = reinterpret_cast<CellList *>(hh);
  hipStream_t st = (hipStream_t)stream;
  // keepForce: when the previous fused step on this list ended in the tile traversal, this one will too unless the grid changes — the
  // traversal's store overwrites every force, so the half step does not have to zero them (fixed up below if the guess was wrong)
  const GJFuse gj{d_vel, reinterpret_cast<float4 *>(d_force), d_mass, defaultMass, dt, friction, noiseAmplitude, is2D, stepNum, seed,
                  h->lastFusedTile ? 1 : 0};
  if (int e = h->update(reinterpret_cast<const float4 *>(d_pos), N, updateL, updatePeriodic, cellDim, st, &gj)) {
    if (!h->gjDone) {
      // (the build refused — a flag raised by an earlier build, a bad grid — before the half step: nothing was changed)
    }
    return e;
  }
  const BoxT<float> box = make_box<float>(boxL, boxPeriodic);
  const LJParams *tbl = reinterpret_cast<const LJParams *>(d_paramTable);
  float maxCut2 = 0.f;
  if (int e = h->lj_max_cutoff2(tbl, ntypes, st, &maxCut2)) return e;
  const bool tile = (algo == UAMMD_LJ_ALGO_AUTO || algo == UAMMD_LJ_ALGO_TILE) && h->numOwned == 0x7fffffff && lj_tile_supported(h, box, maxCut2);
  Outputs out{reinterpret_cast<float4 *>(d_force), nullptr, nullptr, nullptr};
  if (tile) {
    out.vel = d_vel; out.mass = d_mass; out.defaultMass = defaultMass; out.dt = dt; out.is2D = is2D;
    out.invDefaultMass = defaultMass > 0 ? 1.0f / defaultMass : 0.f;
  }
  if (!tile && gj.keepForce && h->gjInHash) UH_CHECK(hipMemsetAsync(d_force, 0, sizeof(float4) * (size_t)N, st));
  h->lastFusedTile = tile;
  int rc = 0;
  if (ntypes == 1) rc = dispatch_celllist<true, false, false>(h, algo, box, tbl, ntypes, out, st);
  else rc = dispatch_celllist<false, false, false>(h, algo, box, tbl, ntypes, out, st);
  if (rc) return rc;
  UH_CHECK(hipGetLastError());
  if (!tile)
    return uammd_verletnvt_gj(2, d_pos, d_vel, d_force, d_mass, defaultMass, nullptr, N, dt, friction, is2D, noiseAmplitude, stepNum, seed, stream);
  return 0;
}

// The second half of that fusion on its own, for the domain-decomposed drivers: between the first half step and the list build they
// exchange the halo, so the build cannot carry the first half step — but the traversal's store can still carry the second one.  The
// list may hold ghosts (option num_owned): they act as neighbours only.  d_force must be zero on the owned rows (GronbechJensen's
// first half step leaves it so) and holds f(t + dt) on return.  Same bits as uammd_lj_transverse_celllist followed by
// uammd_verletnvt_gj(2) on the owned rows, which is what runs where the tile kernel does not take the list.
int uammd_lj_transverse_celllist_gj2(uammd_celllist *hh, const uammd_lj_pair_parameters *d_paramTable, int ntypes, const float boxL[3],
                                     const int boxPeriodic[3], float *d_force, float *d_vel, const float *d_mass, float defaultMass, float dt,
                                     int is2D, int algo, void *stream) {
  if (!hh || !d_paramTable || ntypes < 1 || !d_force || !d_vel) { set_last_error("uammd_lj_transverse_celllist_gj2: bad arguments"); return -1; }
  if (!d_mass && !(defaultMass > 0)) { set_last_error("uammd_lj_transverse_celllist_gj2: no mass array and defaultMass <= 0"); return -1; }
  CellList *h = reinterpret_cast<CellList *>(hh);
  if (h->numberParticlesBuilt == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const BoxT<float> box = make_box<float>(boxL, boxPeriodic);
  const LJParams *tbl = reinterpret_cast<const LJParams *>(d_paramTable);
  float maxCut2 = 0.f;
  if (int e = h->lj_max_cutoff2(tbl, ntypes, st, &maxCut2)) return e;
  const bool tile = (algo == UAMMD_LJ_ALGO_AUTO || algo == UAMMD_LJ_ALGO_TILE) && lj_tile_supported(h, box, maxCut2);
  Outputs out{reinterpret_cast<float4 *>(d_force), nullptr, nullptr, nullptr};
  if (tile) {
    out.vel = d_vel; out.mass = d_mass; out.defaultMass = defaultMass; out.dt = dt; out.is2D = is2D;
    out.invDefaultMass = defaultMass > 0 ? 1.0f / defaultMass : 0.f;
  }
  int rc = 0;
  if (ntypes == 1) rc = dispatch_celllist<true, false, false>(h, algo, box, tbl, ntypes, out, st);
  else rc = dispatch_celllist<false, false, false>(h, algo, box, tbl, ntypes, out, st);
  if (rc) return rc;
  UH_CHECK(hipGetLastError());
  if (!tile) {
    const int nOwned = h->numOwned < h->numberParticlesBuilt ? h->numOwned : h->numberParticlesBuilt;
    return uammd_verletnvt_gj(2, nullptr, d_vel, d_force, d_mass, defaultMass, nullptr, nOwned, dt, 0.f, is2D, 0.f, 0u, 0u, stream);
  }
  return 0;
}

int uammd_lj_transverse_nbody(const float *d_pos, int numberParticles, const uammd_lj_pair_parameters *d_paramTable,
                              int ntypes, const float boxL[3], const int boxPeriodic[3], float *d_force,
                              float *d_energy, float *d_virial, const int *d_globalIndex, void *stream) {
  if (!d_pos || !d_paramTable || ntypes < 1) { set_last_error("uammd_lj_transverse_nbody: bad arguments"); return -1; }
  if (numberParticles <= 0) return 0;
  const BoxT<float> box = make_box<float>(boxL, boxPeriodic);
  Outputs out{reinterpret_cast<float4 *>(d_force), d_energy, d_virial, d_globalIndex};
  const LJParams *tbl = reinterpret_cast<const LJParams *>(d_paramTable);
  int rc = 0;
  UH_DISPATCH_FEV(dispatch_nbody, reinterpret_cast<const float4 *>(d_pos), numberParticles, box, tbl, ntypes, out,
                  (hipStream_t)stream);
  if (rc) return rc;
  UH_CHECK(hipGetLastError());
  return 0;
}

}  // extern "C"
