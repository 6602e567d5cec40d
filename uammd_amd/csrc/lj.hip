// Path A — neighbour traversal with the Lennard-Jones Transverser, for gfx950.
//
// Reference behaviour (what must come out, in which order it is summed):
//   transverseWithNeighbourContainer   Interactor/NeighbourList/common.cuh:10-34
//   27-cell walk, x fastest then y, z  Interactor/NeighbourList/CellList/NeighbourContainer.cuh:95-130
//   Radial<LJ>::Transverser            Interactor/Potential/RadialPotential.cuh:107-127
//   LJFunctor::force/energy            Interactor/Potential/Potential.cuh:37-65
//   NBody tile loop                    Interactor/NBodyBase.cuh:46-116
//
// Three kernels:
//   k_lj_general  thread per sorted particle, walks the cells through global memory.  Any grid
//                 (collapsed dimensions, non periodic, 2D), any number of types, F/E/V.
//   k_lj_brick    the MI355X fast path.  A workgroup owns a Morton-aligned brick of 2^K cells
//                 (K=3: 2x2x2 ... K=6: 4x4x4); because the particles are sorted by Morton key its
//                 i-particles are ONE contiguous range of sortPos.  The brick's halo of cells is
//                 staged once into LDS in x-fastest order, so that the three x-neighbouring cells
//                 of a (dy,dz) row are one contiguous LDS range: 9 ranges per particle instead of
//                 27 scattered global ranges, read with ds_read_b128.  The j order is exactly the
//                 reference's (cells x-fastest, particles ascending), so the float sums are
//                 bit-identical to the thread-per-particle walk.
//                 The minimum-image arithmetic is skipped — exactly, not approximately — for waves
//                 whose cells do not touch the box faces when every staged position lies in the
//                 primary box (then floor(d*(-1/L)+0.5) == 0 for every pair).
//   k_lj_nbody    all pairs with LDS tiles (small boxes; PairForces.cu:49-53).
#include "celllist.hpp"
#include "lj_common.hpp"
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
#ifdef UAMMD_EXP_LOADONLY
    for (int u = 0; u < 8; ++u) d[u] = fabsf(c[u].x) + fabsf(c[u].y) + fabsf(c[u].z) + 7.0f;
#else
    for (int u = 0; u < 8; ++u) d[u] = lj_dist2<PBC>(box, pi, c[u]);
#endif
    const int rem = je - j;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool hit = !(d[u] >= rc2) & (u < rem);
      if (hit) { *(LdsQT *)(uintptr_t)qa = (QT)(j + u); qa += kStep; }
    }
  }
  Q.n = (int)((qa - q0) / kStep);
}

// 16-byte LDS accesses through 32-bit LDS addresses (HIP's float4 is a class and cannot live behind an address-space pointer)
typedef float f4v __attribute__((ext_vector_type(4)));
using LdsV4 = __attribute__((address_space(3))) f4v;
UH_D float4 lds_load4(uint addr) {
  const f4v v = *(const LdsV4 *)(uintptr_t)addr;
  return make_float4(v.x, v.y, v.z, v.w);
}
UH_D void lds_store4(uint addr, const float4 &v) {
  const f4v t = {v.x, v.y, v.z, v.w};
  *(LdsV4 *)(uintptr_t)addr = t;
}

// The same scan over candidates a wave has staged in LDS (k_lj_staged): the lane's candidates are `cnt` consecutive float4
// starting at LDS byte address `sa`; candidate t is particle `first + t` of the sorted array (what the FIFO records and the
// drain re-reads from global memory).  A 64-lane ds_read_b128 costs 7.4 clocks of the CU's LDS against 16.5 clocks of its
// texture addresser for ANY global load wider than a dword, whatever the lanes' addresses (tools/vmem_ubench.hip).
template <bool PBC, bool NT1, bool WE, bool WV, class QT, int QCAP, int QSTRIDE>
UH_D void lj_scan_lds(Acc &acc, PairQueue<QT, QCAP, QSTRIDE> &Q, bool drainPBC, const float4 *__restrict__ P, uint sa,
                      int first, int cnt, const float4 &pi, const BoxT<float> &box, float rc2, const LJParams &p1,
                      const LJParams *tbl, int ntypes) {
  using LdsQT = __attribute__((address_space(3))) QT;
  constexpr uint kStep = QSTRIDE * sizeof(QT);
  const uint q0 = (uint)(uintptr_t)(LdsQT *)Q.slot;
  uint qa = q0 + (uint)Q.n * kStep;
  for (int j = 0; j < cnt; j += 8) {
    if (__any(qa > q0 + (QCAP - 8) * kStep)) {
      Q.n = (int)((qa - q0) / kStep);
      if (drainPBC) lj_drain<true, NT1, WE, WV>(acc, Q, P, pi, box, p1, tbl, ntypes);
      else lj_drain<false, NT1, WE, WV>(acc, Q, P, pi, box, p1, tbl, ntypes);
      qa = q0;
    }
    const uint pj = sa + (uint)j * 16u;
    float4 c[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = lds_load4(pj + 16u * u);
    float d[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) d[u] = lj_dist2<PBC>(box, pi, c[u]);
    const int rem = cnt - j;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool hit = !(d[u] >= rc2) & (u < rem);
      if (hit) { *(LdsQT *)(uintptr_t)qa = (QT)(first + j + u); qa += kStep; }
    }
  }
  Q.n = (int)((qa - q0) / kStep);
}


constexpr int kQCapGeneral = 24;  // per-lane FIFO depth of the global-memory kernels (uint entries)
constexpr int kQCapBrick = 32;    // per-lane FIFO depth of the brick kernel (ushort LDS indices)

// ---- general walk (also the in-kernel fallback of the brick kernel) ------------------------------
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

// ---- wave-staged walk: the candidates of a neighbour offset go through LDS ONCE per wave -------------------------------
// k_lj_general is bound by the texture addresser, not by the VALU: every lane issues its own global load per candidate and
// a 64-lane load of >= 8 bytes costs ~16.5 clocks of the CU's one addresser even when the lanes of a cell all read the
// same address (0.27 ms of the 0.365 ms at C3; with the distance arithmetic removed the kernel still takes 0.29 ms).
// Here the 64 sorted particles of a wave are runs of lanes that share a cell ("groups", ~5 per wave); for one neighbour
// offset each group needs ONE range of the sorted array, and the wave copies the concatenation of its groups' ranges
// (~75 particles) into LDS with at most two loads per lane (element k and k + 64 of the concatenation), issued one
// neighbour offset ahead of the scan that reads them.  Every lane then scans ITS range from LDS (broadcast reads) in the
// same order as k_lj_general, so the FIFO contents, the drain and the accumulated floats are bit-identical to it.
// A neighbour offset whose ranges exceed the staging buffer (very crowded cells) takes the global-memory scan instead.
constexpr int kStageCap = 128;  // staged candidates per wave and neighbour offset (two per lane)

template <bool NT1, bool WE, bool WV>
__global__ void __launch_bounds__(128) k_lj_staged(ListView cl, GridT<float> grid, BoxT<float> box,
                                                    const LJParams *__restrict__ tbl, int ntypes, Outputs out) {
  __shared__ uint gq[kQCapGeneral * 128];
  __shared__ float4 stageBuf[2][kStageCap + 8];
  const int lane = threadIdx.x & 63;
  const int id = (int)xcd_contiguous_block(blockIdx.x, gridDim.x) * 128 + threadIdx.x;
  const bool inRange = id < cl.N;
  const int gi = inRange ? cl.groupIndex[id] : 0;
  const bool owned = inRange && gi < cl.numOwned;
  if (!__any(owned)) return;  // ghost cells are whole waves at the slab faces
  const float4 pi = inRange ? cl.sortPos[id] : make_float4(0.f, 0.f, 0.f, 0.f);
  LJParams p1 = tbl[0];
  const float rc2 = NT1 ? p1.cutOff2 : lj_max_cutoff2(tbl, ntypes);
  PairQueue<uint, kQCapGeneral, 128> Q{gq + threadIdx.x, 0};
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

  // groups: runs of lanes with the same cell (the array is sorted by cell); the same for all neighbour offsets
  const int myCell = inRange ? grid.getCellIndex(celli) : -1 - lane;
  const int prevCell = __shfl_up(myCell, 1, 64);
  const unsigned long long leaderMask = __ballot(lane == 0 || myCell != prevCell);

  auto fetch = [&](int cc, uint2 &rg, bool &wrapped) {
    int3 cellj = celli;
    if (npx > 1) cellj.x += cc % 3 - 1;
    if (npy > 1) cellj.y += (cc / npx) % 3 - 1;
    if (npz > 1) cellj.z += cc / (npx * npy) - 1;
    const int3 raw = cellj;
    cellj.x = grid.pbc_x(cellj.x);
    cellj.y = grid.pbc_y(cellj.y);
    cellj.z = grid.pbc_z(cellj.z);
    const bool exists = inRange && !(cellj.x < 0 || cellj.x >= n.x || cellj.y < 0 || cellj.y >= n.y || cellj.z < 0 || cellj.z >= n.z);
    wrapped = raw.x != cellj.x || raw.y != cellj.y || raw.z != cellj.z;
    rg = cl.cellRange[exists ? grid.getCellIndex(cellj) : ncells];
  };
  // element k of the concatenated ranges -> index in the sorted array (src0: k = lane, src1: k = lane + 64; -1: none), and
  // the lane's own offset into the concatenation; returns the total
  auto plan = [&](int first, int cnt, int &src0, int &src1, int &myOff) {
    unsigned long long m = leaderMask & __ballot(cnt > 0);
    int sOff = 0;
    src0 = -1; src1 = -1; myOff = 0;
    while (m) {
      const int L = __builtin_ctzll(m);
      m &= m - 1;
      const int sFirst = __builtin_amdgcn_readlane(first, L);
      const int sCnt = __builtin_amdgcn_readlane(cnt, L);
      const uint d0 = (uint)(lane - sOff), d1 = d0 + 64u;
      if (d0 < (uint)sCnt) src0 = sFirst + (int)d0;
      if (d1 < (uint)sCnt) src1 = sFirst + (int)d1;
      if (lane >= L) myOff = sOff;
      sOff += sCnt;
    }
    return sOff;
  };
  const uint stage0 = (uint)(uintptr_t)(__attribute__((address_space(3))) char *)(char *)&stageBuf[threadIdx.x >> 6][0];
  const float4 *__restrict__ P = cl.sortPos;

  uint2 rg, rgNext = make_uint2(0u, 0u);
  bool wrapped, wrappedNext = false;
  fetch(0, rg, wrapped);
  if (numberNeighbourCells > 1) fetch(1, rgNext, wrappedNext);
  int first = (int)rg.x, cnt = (int)(rg.y & 0x7fffffffu) - first;
  int src0, src1, myOff;
  int total = plan(first, cnt, src0, src1, myOff);
  if (total <= kStageCap) {
    if (src0 >= 0) lds_store4(stage0 + (uint)lane * 16u, P[src0]);
    if (src1 >= 0) lds_store4(stage0 + (uint)(lane + 64) * 16u, P[src1]);
  }
  bool drainPBC = false;
  for (int cc = 0; cc < numberNeighbourCells; ++cc) {
    // plan and load the next offset's candidates, fetch the range of the one after
    const bool more = cc + 1 < numberNeighbourCells;
    const int firstN = (int)rgNext.x, cntN = (int)(rgNext.y & 0x7fffffffu) - firstN;
    const bool outsideN = (rgNext.y >> 31) != 0u, wrapN = wrappedNext;
    int src0N = -1, src1N = -1, myOffN = 0, totalN = 0;
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (more) {
      totalN = plan(firstN, cntN, src0N, src1N, myOffN);
      if (totalN <= kStageCap) {
        if (src0N >= 0) v0 = P[src0N];
        if (src1N >= 0) v1 = P[src1N];
      }
      rgNext = make_uint2(0u, 0u);
      wrappedNext = false;
      if (cc + 2 < numberNeighbourCells) fetch(cc + 2, rgNext, wrappedNext);
    }
    // scan the current offset
    const int myCnt = owned ? cnt : 0;
    const bool needPBC = myCnt > 0 && (smallGrid || iOut || wrapped || (rg.y >> 31) != 0u);
    const bool anyPBC = __any(needPBC);
    if (anyPBC) drainPBC = true;
    if (total <= kStageCap) {
      const uint sa = stage0 + (uint)myOff * 16u;
      if (anyPBC) lj_scan_lds<true, NT1, WE, WV>(acc, Q, drainPBC, P, sa, first, myCnt, pi, box, rc2, p1, tbl, ntypes);
      else lj_scan_lds<false, NT1, WE, WV>(acc, Q, drainPBC, P, sa, first, myCnt, pi, box, rc2, p1, tbl, ntypes);
    } else {
      if (anyPBC) lj_scan<true, NT1, WE, WV>(acc, Q, drainPBC, P, first, first + myCnt, pi, box, rc2, p1, tbl, ntypes);
      else lj_scan<false, NT1, WE, WV>(acc, Q, drainPBC, P, first, first + myCnt, pi, box, rc2, p1, tbl, ntypes);
    }
    // the scan above has finished reading the buffer: stage the next offset
    __builtin_amdgcn_wave_barrier();
    if (more && totalN <= kStageCap) {
      if (src0N >= 0) lds_store4(stage0 + (uint)lane * 16u, v0);
      if (src1N >= 0) lds_store4(stage0 + (uint)(lane + 64) * 16u, v1);
    }
    __builtin_amdgcn_wave_barrier();
    rg.y = outsideN ? 0x80000000u : 0u;  // only the flag of rg is read below
    wrapped = wrapN;
    first = firstN; cnt = cntN; myOff = myOffN; total = totalN;
  }
  if (drainPBC) lj_drain<true, NT1, WE, WV>(acc, Q, P, pi, box, p1, tbl, ntypes);
  else lj_drain<false, NT1, WE, WV>(acc, Q, P, pi, box, p1, tbl, ntypes);
  if (owned) write_out(out, out.globalIndex ? out.globalIndex[gi] : gi, acc);
}

// ---- cell-per-wave kernel (tolerance-level: same pairs, different summation order) -------------------------------
// The thread-per-particle walk keeps the reference's summation order but pays for it: the lanes of a wave belong to ~5
// different cells whose neighbour cells have different populations, so 40 % of the lanes idle (measured), and every lane
// tests its own 340 candidates.  Here a WAVE owns one cell A: the ~340 particles of the 27 neighbour cells are loaded
// once (5-6 per lane, kept in registers and in LDS) and reused for all ~13 particles i of A:
//   phase 1  each lane tests its candidates against i (broadcast through SGPRs); hits are compacted across the wave
//            (ballot + mbcnt) into a list in LDS;
//   phase 2  the ~52 hits of i are evaluated 64 at a time, one per lane (81 % lane use instead of 15 %), then one wave
//            reduction gives F_i.
// The sum over j is in a different order than the reference's -> forces agree to rounding (tests: <= 1e-5 max|F|, the
// bar of SURVEY 8d), not bit for bit; the bit-exact kernels above stay selectable (UAMMD_LJ_ALGO_GENERAL).
constexpr int kCWSlots = 8;
constexpr int kCWChunk = 64 * kCWSlots;

// sum over the 64 lanes with DPP adds (no LDS crossbar): quad swaps, row rotations, then the row_bcast steps of the GCN/CDNA
// reduction idiom; the total lands in lane 63 and is returned wave-uniform
template <int CTRL, int ROWMASK> UH_D float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWMASK, 0xf, false);
  return v + __int_as_float(moved);
}
UH_D float wave_sum(float v) {
  v = dpp_add<0xb1, 0xf>(v);   // quad_perm:[1,0,3,2]
  v = dpp_add<0x4e, 0xf>(v);   // quad_perm:[2,3,0,1]
  v = dpp_add<0x124, 0xf>(v);  // row_ror:4
  v = dpp_add<0x128, 0xf>(v);  // row_ror:8  -> every lane of a row holds the row total
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
UH_D float lane_value(float v, int k) {  // k wave-uniform
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k));
}
UH_D void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <bool PBC, bool NT1, bool WE, bool WV>
UH_D void cellwave_rows(Acc &mine, const float4 (&pj)[kCWSlots], int nC, const float4 &pA, int nI, int lane,
                        const float4 *__restrict__ cand, unsigned short *__restrict__ lst, const BoxT<float> &box, float rc2,
                        const LJParams &p1, const LJParams *__restrict__ tbl, int ntypes) {
  const int nSlots = (nC + 63) >> 6;
  for (int k = 0; k < nI; ++k) {
    const float4 pi = make_float4(lane_value(pA.x, k), lane_value(pA.y, k), lane_value(pA.z, k), lane_value(pA.w, k));
    int cnt = 0;
#pragma unroll
    for (int s = 0; s < kCWSlots; ++s) {
      if (s < nSlots) {
        const int c = s * 64 + lane;
        const float r2 = lj_dist2<PBC>(box, pi, pj[s]);
        const bool in = c < nC && !(r2 >= rc2) && r2 != 0.0f;
        const unsigned long long m = __ballot(in);
        if (in) lst[cnt + (int)__builtin_amdgcn_mbcnt_hi((uint)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint)m, 0u))] = (unsigned short)c;
        cnt += (int)__popcll(m);
      }
    }
    wave_lds_fence();
    Acc a;
    for (int b = 0; b < cnt; b += 64) {
      const int q = b + lane;
      if (q < cnt) {
        const float4 cj = cand[lst[q]];
        real3f r12;
        float fm, e;
        if (NT1) lj_eval<PBC, WE>(box, p1, pi, cj, r12, fm, e);
        else lj_eval<PBC, WE>(box, lj_lookup(tbl, ntypes, (int)pi.w, (int)cj.w), pi, cj, r12, fm, e);
        lj_acc<WE, WV>(a, r12, fm, e);
      }
    }
    wave_lds_fence();  // the list is rewritten for the next i
    const float fx = wave_sum(a.fx), fy = wave_sum(a.fy), fz = wave_sum(a.fz);
    float e = 0.f, v = 0.f;
    if (WE) e = wave_sum(a.e);
    if (WV) v = wave_sum(a.v);
    if (lane == k) { mine.fx += fx; mine.fy += fy; mine.fz += fz; mine.e += e; mine.v += v; }
  }
}

template <bool NT1, bool WE, bool WV>
__global__ void __launch_bounds__(256) k_lj_cellwave(ListView cl, GridT<float> grid, BoxT<float> box,
                                                      const LJParams *__restrict__ tbl, int ntypes, Outputs out) {
  __shared__ float4 candAll[4 * kCWChunk];
  __shared__ unsigned short lstAll[4 * kCWChunk];
  __shared__ int rngFirst[4 * 32], rngPre[4 * 32];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int3 n = grid.cellDim;
  const int ncells = n.x * n.y * n.z;
  const int cell = (int)xcd_contiguous_block(blockIdx.x, gridDim.x) * 4 + wave;
  if (cell >= ncells) return;  // whole waves leave; there is no block-level barrier below
  const uint csA = cl.cellStart[cell];
  if (csA < cl.validCell) return;
  const int aStart = (int)(csA - cl.validCell), aLen = cl.cellEnd[cell] - aStart;
  float4 *cand = candAll + wave * kCWChunk;
  unsigned short *lst = lstAll + wave * kCWChunk;
  int *rFirst = rngFirst + wave * 32, *rPre = rngPre + wave * 32;
  const int3 celli = make_int3(cell % n.x, (cell / n.x) % n.y, cell / (n.x * n.y));
  const int npx = n.x > 1 ? 3 : 1, npy = n.y > 1 ? 3 : 1, npz = n.z > 1 ? 3 : 1;
  const int nnc = npx * npy * npz;
  // lanes 0..nnc-1: the range of one neighbour cell each
  int myFirst = 0, myLen = 0;
  bool myPBC = false;
  if (lane < nnc) {
    int3 cellj = celli;
    if (npx > 1) cellj.x += lane % 3 - 1;
    if (npy > 1) cellj.y += (lane / npx) % 3 - 1;
    if (npz > 1) cellj.z += lane / (npx * npy) - 1;
    const int3 raw = cellj;
    cellj.x = grid.pbc_x(cellj.x);
    cellj.y = grid.pbc_y(cellj.y);
    cellj.z = grid.pbc_z(cellj.z);
    const bool exists = !(cellj.x < 0 || cellj.x >= n.x || cellj.y < 0 || cellj.y >= n.y || cellj.z < 0 || cellj.z >= n.z);
    if (exists) {
      const int icellj = grid.getCellIndex(cellj);
      const uint cs = cl.cellStart[icellj];
      if (cs >= cl.validCell) {
        myFirst = (int)(cs - cl.validCell);
        myLen = cl.cellEnd[icellj] - myFirst;
        myPBC = raw.x != cellj.x || raw.y != cellj.y || raw.z != cellj.z || !cl.cellOutside || cl.cellOutside[icellj] != 0;
      }
    }
  }
  int incl = myLen;
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  const int total = __shfl(incl, nnc - 1, 64);
  if (lane < 32) { rFirst[lane] = myFirst; rPre[lane] = lane < nnc ? incl - myLen : total; }
  wave_lds_fence();
  // minimum image needed? (see walk_global: direct neighbours on a >= 5-cell grid with particles stored inside the box)
  const bool sameBox = box.boxSize.x == grid.box.boxSize.x && box.boxSize.y == grid.box.boxSize.y &&
                       box.boxSize.z == grid.box.boxSize.z && box.px() == grid.box.px() &&
                       box.py() == grid.box.py() && box.pz() == grid.box.pz();
  const bool needPBC = __any(myPBC) || n.x < 5 || n.y < 5 || n.z < 5 || !sameBox;
  const LJParams p1 = tbl[0];
  const float rc2 = NT1 ? p1.cutOff2 : lj_max_cutoff2(tbl, ntypes);
  for (int ib = 0; ib < aLen; ib += 64) {
    const int nI = min(64, aLen - ib);
    const float4 pA = lane < nI ? cl.sortPos[aStart + ib + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    Acc mine;
    for (int chunk = 0; chunk < total; chunk += kCWChunk) {
      const int nC = min(kCWChunk, total - chunk);
      float4 pj[kCWSlots];
#pragma unroll
      for (int s = 0; s < kCWSlots; ++s) {
        const int c = chunk + s * 64 + lane;
        pj[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < total) {
          int lo = 0, hi = nnc;  // rPre[lo] <= c < rPre[hi]
#pragma unroll
          for (int it = 0; it < 5; ++it) {
            const int mid = (lo + hi) >> 1;
            if (rPre[mid] <= c) lo = mid; else hi = mid;
          }
          pj[s] = cl.sortPos[rFirst[lo] + (c - rPre[lo])];
          cand[s * 64 + lane] = pj[s];
        }
      }
      wave_lds_fence();
      if (needPBC) cellwave_rows<true, NT1, WE, WV>(mine, pj, nC, pA, nI, lane, cand, lst, box, rc2, p1, tbl, ntypes);
      else cellwave_rows<false, NT1, WE, WV>(mine, pj, nC, pA, nI, lane, cand, lst, box, rc2, p1, tbl, ntypes);
      wave_lds_fence();
    }
    if (lane < nI) {
      const int gi = cl.groupIndex[aStart + ib + lane];
      write_out(out, out.globalIndex ? out.globalIndex[gi] : gi, mine);
    }
  }
}

// ---- LDS-tiled brick kernel ------------------------------------------------------------------------
template <int K> struct Brick {
  static constexpr int BX = (K >= 4) ? 4 : 2;
  static constexpr int BY = (K >= 5) ? 4 : 2;
  static constexpr int BZ = (K >= 6) ? 4 : 2;
  static constexpr int HX = BX + 2, HY = BY + 2, HZ = BZ + 2;
  static constexpr int NH = HX * HY * HZ;
  static constexpr int NCELL = 1 << K;
  // ~12.6 particles per cell at liquid density: 2^K cells -> 101/201/403/805 i-particles per brick
  static constexpr int THREADS = (K == 3) ? 128 : (K == 4) ? 256 : (K == 5) ? 512 : 1024;
};

// Flat walk of the 9 rows of one i-particle from the LDS tile.  Every lane keeps its own (row, j)
// cursor, so lanes whose rows have different lengths do not wait for each other at row ends; the wave
// only reconverges for the queue drains.  Rows are multiples of 4 slots (cells are padded with +inf
// dummies), so the body needs no tail handling.  `d < rc2` (not !(d >= rc2)): dummies and NaN fail.
template <class Bk, bool PBC, bool NT1, bool WE, bool WV, class QT, int QCAP, int QSTRIDE>
UH_D void brick_walk(Acc &acc, PairQueue<QT, QCAP, QSTRIDE> &Q, bool active, const float4 *__restrict__ spos,
                     const int *__restrict__ off, int cbase, const float4 &pi, const BoxT<float> &box, float rc2,
                     const LJParams &p1, const LJParams *stbl, int ntypes) {
  int r = -1, j = 0, je = 0;
  bool more = active;
  while (__any(more)) {
    if (more) {
      while (j == je) {  // next non-empty row
        if (++r == 9) { more = false; break; }
        const int c0 = cbase + Bk::HX * ((r % 3) + Bk::HY * (r / 3));
        j = off[c0];
        je = off[c0 + 3];
      }
    }
    if (__any(Q.n > QCAP - 4)) lj_drain<PBC, NT1, WE, WV>(acc, Q, spos, pi, box, p1, stbl, ntypes);
    if (more) {
      const float4 c0 = spos[j], c1 = spos[j + 1], c2 = spos[j + 2], c3 = spos[j + 3];
      const float d0 = lj_dist2<PBC>(box, pi, c0), d1 = lj_dist2<PBC>(box, pi, c1);
      const float d2 = lj_dist2<PBC>(box, pi, c2), d3 = lj_dist2<PBC>(box, pi, c3);
      if (d0 < rc2) { Q.slot[Q.n * QSTRIDE] = (QT)j; ++Q.n; }
      if (d1 < rc2) { Q.slot[Q.n * QSTRIDE] = (QT)(j + 1); ++Q.n; }
      if (d2 < rc2) { Q.slot[Q.n * QSTRIDE] = (QT)(j + 2); ++Q.n; }
      if (d3 < rc2) { Q.slot[Q.n * QSTRIDE] = (QT)(j + 3); ++Q.n; }
      j += 4;
    }
  }
  lj_drain<PBC, NT1, WE, WV>(acc, Q, spos, pi, box, p1, stbl, ntypes);
}

constexpr int kMaxTypesLds = 8;  // type tables up to 8x8 are cached in LDS by the brick kernel

template <int K, bool NT1, bool WE, bool WV>
__global__ void __launch_bounds__(Brick<K>::THREADS)
k_lj_brick(ListView cl, GridT<float> grid, BoxT<float> box, const LJParams *__restrict__ tbl, int ntypes, Outputs out,
           int capacity, const int *__restrict__ brickList) {
  using Bk = Brick<K>;
  constexpr int T = Bk::THREADS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4 *spos = reinterpret_cast<float4 *>(smem);                                   // [capacity]
  int *off = reinterpret_cast<int *>(smem + sizeof(float4) * (size_t)capacity);       // [NH+1]
  int *gstart = off + (Bk::NH + 1);                                                    // [NH]
  int *cnt = gstart + Bk::NH;                                                          // [NH]
  int *misc = cnt + Bk::NH;                                                            // [4]: total, allInBox
  LJParams *stbl = reinterpret_cast<LJParams *>(misc + 4);                             // [kMaxTypesLds^2]
  unsigned short *queue = reinterpret_cast<unsigned short *>(stbl + kMaxTypesLds * kMaxTypesLds);  // [T*kQCapBrick]

  const int tid = threadIdx.x;
  const uint b = brickList ? (uint)brickList[blockIdx.x] : blockIdx.x;
  const uint key0 = b << K;
  const int pStart = (int)cl.keyStart[key0];
  const int pEnd = (int)cl.keyStart[key0 + Bk::NCELL];
  if (pStart == pEnd) return;  // block-uniform
  const int bx = (int)compact10(key0), by = (int)compact10(key0 >> 1), bz = (int)compact10(key0 >> 2);
  const int3 n = grid.cellDim;

  // 1. halo cell table: global range and count of every halo cell, x fastest.
  if (tid == 0) { misc[0] = 0; misc[1] = 1; }
  for (int t = tid; t < Bk::NH; t += T) {
    const int hx = t % Bk::HX, hy = (t / Bk::HX) % Bk::HY, hz = t / (Bk::HX * Bk::HY);
    int gx = bx + hx - 1, gy = by + hy - 1, gz = bz + hz - 1;
    // a halo cell is needed only if it neighbours an existing cell of the brick (bricks at the
    // upper faces of a grid whose size is not a multiple of the brick are partial)
    const int vx = min(Bk::BX, n.x - bx), vy = min(Bk::BY, n.y - by), vz = min(Bk::BZ, n.z - bz);
    const bool needed = hx <= vx + 1 && hy <= vy + 1 && hz <= vz + 1;
    gx = gx < 0 ? gx + n.x : (gx >= n.x ? gx - n.x : gx);
    gy = gy < 0 ? gy + n.y : (gy >= n.y ? gy - n.y : gy);
    gz = gz < 0 ? gz + n.z : (gz >= n.z ? gz - n.z : gz);
    int s = 0, c = 0;
    if (needed) {
      const uint hh = morton_hash(make_int3(gx, gy, gz));
      s = (int)cl.keyStart[hh];
      c = (int)cl.keyStart[hh + 1] - s;
    }
    gstart[t] = s;
    cnt[t] = c;
    off[t + 1] = (c + 3) & ~3;  // padded counts (every cell occupies a multiple of 4 slots), scanned below
  }
  if (tid == 0) off[0] = 0;
  __syncthreads();
  // 2. exclusive scan of NH (<= 216) padded counts: a single wave does it with shuffles.
  if (tid < 64) {
    constexpr int PER = (Bk::NH + 63) / 64;
    int v[PER];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int idx = tid * PER + k;
      v[k] = (idx < Bk::NH) ? off[idx + 1] : 0;
      sum += v[k];
    }
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d, 64);
      if (tid >= d) incl += o;
    }
    int run = incl - sum;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int idx = tid * PER + k;
      run += v[k];
      if (idx < Bk::NH) off[idx + 1] = run;
    }
    if (tid == 63) misc[0] = incl;
  }
  if (!NT1) {
    for (int t = tid; t < ntypes * ntypes; t += T) stbl[t] = tbl[t];
  }
  __syncthreads();
  const int total = misc[0];
  const LJParams p1 = tbl[0];
  const float rc2 = NT1 ? p1.cutOff2 : lj_max_cutoff2(stbl, ntypes);

  if (total > capacity) {
    // Too dense for the LDS tile (block-uniform): walk global memory like k_lj_general.
    for (int i = pStart + tid; i < pEnd; i += T) {
      const int gi = cl.groupIndex[i];
      const int ori = out.globalIndex ? out.globalIndex[gi] : gi;
      const float4 pi = cl.sortPos[i];
      Acc acc;
      PairQueue<uint, kQCapGeneral, T> Qg{reinterpret_cast<uint *>(smem) + tid, 0};  // the unused tile region
      walk_global<NT1, WE, WV>(acc, Qg, cl, grid, box, tbl, ntypes, p1, rc2, pi);
      write_out(out, ori, acc);
    }
    return;
  }

  // 3. stage the halo: 16-lane groups copy one cell at a time (a cell is ~13 contiguous float4) and
  //    fill the cell's padding slots with +inf positions (r2 = inf or NaN: never inside the cut-off).
  {
    const int g = tid >> 4, l = tid & 15;
    const float hxL = 0.5f * box.boxSize.x, hyL = 0.5f * box.boxSize.y, hzL = 0.5f * box.boxSize.z;
    const float inf = __builtin_inff();
    bool inBox = true;
    for (int t = g; t < Bk::NH; t += T / 16) {
      const int s = gstart[t], o = off[t], c = cnt[t], cpad = off[t + 1] - o;
      for (int k = l; k < cpad; k += 16) {
        float4 p = make_float4(inf, inf, inf, 0.0f);
        if (k < c) {
          p = cl.sortPos[s + k];
          inBox = inBox && (p.x >= -hxL && p.x < hxL && p.y >= -hyL && p.y < hyL && p.z >= -hzL && p.z < hzL);
        }
        spos[o + k] = p;
      }
    }
    if (!inBox) misc[1] = 0;  // benign race: every writer stores 0
  }
  __syncthreads();
  const bool allInBox = misc[1] != 0;
  const bool smallGrid = (n.x < 5) || (n.y < 5) || (n.z < 5);

  // 4. traversal: one i-particle per lane; its 9 (dy,dz) rows are contiguous, 4-aligned LDS ranges that
  //    the lane walks as ONE flat loop (no per-row reconvergence), 4 candidates per trip.
  for (int i0 = pStart + (tid & ~63); i0 < pEnd; i0 += T) {  // wave-uniform trip count
    const int i = i0 + (tid & 63);
    const bool active = i < pEnd;
    float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
    int lx = 0, ly = 0, lz = 0;
    bool atFace = false;
    if (active) {
      pi = cl.sortPos[i];
      const uint h = cl.sortHash[i] - key0;  // local Morton key inside the brick, < 2^K
      lx = (int)compact10(h);
      ly = (int)compact10(h >> 1);
      lz = (int)compact10(h >> 2);
      const int gx = bx + lx, gy = by + ly, gz = bz + lz;
      atFace = gx == 0 || gx == n.x - 1 || gy == 0 || gy == n.y - 1 || gz == 0 || gz == n.z - 1;
    }
    const bool needPBC = !allInBox || smallGrid || (__ballot(atFace) != 0ull);  // wave-uniform
    Acc acc;
    PairQueue<unsigned short, kQCapBrick, T> Q{queue + tid, 0};
    const int cbase = lx + Bk::HX * (ly + Bk::HY * lz);
    if (needPBC) brick_walk<Bk, true, NT1, WE, WV>(acc, Q, active, spos, off, cbase, pi, box, rc2, p1, stbl, ntypes);
    else brick_walk<Bk, false, NT1, WE, WV>(acc, Q, active, spos, off, cbase, pi, box, rc2, p1, stbl, ntypes);
    if (active) {
      const int gi = cl.groupIndex[i];
      const int ori = out.globalIndex ? out.globalIndex[gi] : gi;
      write_out(out, ori, acc);
    }
  }
}

// ---- uniform-j "quad" kernel ------------------------------------------------------------------------
// One wave owns a Morton-aligned 2x2x1 quad of cells (keys 4q..4q+3): its i-particles are one
// contiguous range of sortPos (~50 of 64 lanes at liquid density).  The wave then streams the 4x4x3
// block of cells around the quad in the reference's order (z, y, x ascending; particles ascending).
// The candidate j is WAVE-UNIFORM: its position is fetched by the scalar unit (s_load) and used as
// an SGPR operand, the loop control is scalar, and each lane pays only 3 subs + 3 FMAs + 1 compare
// per candidate; lanes whose own cell is not a neighbour of the j-cell are masked.  Candidates
// inside the cut-off are appended to the lane's FIFO and evaluated later in FIFO order (lj_drain),
// so every lane still accumulates exactly the reference's sequence of pairs.
// Minimum image: when every particle of the i-lanes and of the j-cell lies in the primary box and
// the grid has >= 5 cells per dimension, floor(d*(-1/L)+0.5) is the same for every pair of the two
// cells (0, or -/+1 across a face), so r = d + off*L is applied as a per-cell scalar shift — the
// same float operations the reference performs, minus the floor.  Otherwise the full arithmetic runs.
constexpr int kQCapQuad = 24;

template <int MODE>  // 0: no image shift, 1: per-cell shift, 2: full minimum image
UH_D float quad_dist2(const BoxT<float> &box, const float4 &ri, const float4 &rj, float sx, float sy, float sz) {
  real3f r12{rj.x - ri.x, rj.y - ri.y, rj.z - ri.z};
  if (MODE == 1) { r12.x += sx; r12.y += sy; r12.z += sz; }
  if (MODE == 2) r12 = box.apply_pbc(r12);
  return dot3(r12, r12);
}

template <int MODE, bool NT1, bool WE, bool WV>
UH_D void quad_scan(Acc &acc, PairQueue<uint, kQCapQuad, 64> &Q, bool &drainPBC, const float4 *__restrict__ P, int cs,
                    int ce, bool mine, const float4 &pi, const BoxT<float> &box, float sx, float sy, float sz,
                    float rc2, const LJParams &p1, const LJParams *tbl, int ntypes) {
  const int last = ce - 1;
  for (int s = cs; s < ce; s += 4) {  // s, cs, ce are wave-uniform
    if (__any(Q.n > kQCapQuad - 4)) {
      if (drainPBC) lj_drain<true, NT1, WE, WV>(acc, Q, P, pi, box, p1, tbl, ntypes);
      else lj_drain<false, NT1, WE, WV>(acc, Q, P, pi, box, p1, tbl, ntypes);
    }
    const int s1 = min(s + 1, last), s2 = min(s + 2, last), s3 = min(s + 3, last);
    const float4 c0 = P[s], c1 = P[s1], c2 = P[s2], c3 = P[s3];
    const float d0 = quad_dist2<MODE>(box, pi, c0, sx, sy, sz), d1 = quad_dist2<MODE>(box, pi, c1, sx, sy, sz);
    const float d2 = quad_dist2<MODE>(box, pi, c2, sx, sy, sz), d3 = quad_dist2<MODE>(box, pi, c3, sx, sy, sz);
    if (mine && d0 < rc2) { Q.slot[Q.n * 64] = (uint)s; ++Q.n; }
    if (mine && d1 < rc2 && s + 1 < ce) { Q.slot[Q.n * 64] = (uint)(s + 1); ++Q.n; }
    if (mine && d2 < rc2 && s + 2 < ce) { Q.slot[Q.n * 64] = (uint)(s + 2); ++Q.n; }
    if (mine && d3 < rc2 && s + 3 < ce) { Q.slot[Q.n * 64] = (uint)(s + 3); ++Q.n; }
  }
}

template <bool NT1, bool WE, bool WV>
__global__ void __launch_bounds__(256)
k_lj_quad(const float4 *__restrict__ P, const uint *__restrict__ sortHash, const uint *__restrict__ keyStart,
          const int *__restrict__ groupIndex, const unsigned char *__restrict__ keyOutside, GridT<float> grid,
          BoxT<float> box, const LJParams *__restrict__ tbl, int ntypes, float4 *__restrict__ oForce,
          float *__restrict__ oEnergy, float *__restrict__ oVirial, const int *__restrict__ globalIndex, uint nQuads) {
  // every global array is a separate __restrict__ parameter so that the uniform position loads can be
  // proven unclobbered and selected as scalar (SMEM) loads
  __shared__ uint queue[4 * kQCapQuad * 64];
  const Outputs out{oForce, oEnergy, oVirial, globalIndex};
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = (int)(threadIdx.x & 63);
  const uint q = blockIdx.x * 4u + (uint)wave;
  if (q >= nQuads) return;
  const uint key0 = q << 2;
  const int pStart = (int)keyStart[key0], pEnd = (int)keyStart[key0 + 4];
  if (pStart == pEnd) return;
  const int x0 = (int)compact10(key0), y0 = (int)compact10(key0 >> 1), z0 = (int)compact10(key0 >> 2);
  const int3 n = grid.cellDim;
  const bool smallGrid = (n.x < 5) || (n.y < 5) || (n.z < 5);
  const LJParams p1 = tbl[0];
  const float rc2 = NT1 ? p1.cutOff2 : lj_max_cutoff2(tbl, ntypes);
  const float hxL = 0.5f * box.boxSize.x, hyL = 0.5f * box.boxSize.y, hzL = 0.5f * box.boxSize.z;

  for (int i0 = pStart; i0 < pEnd; i0 += 64) {  // one trip unless the quad holds more than 64 particles
    const int i = i0 + lane;
    const bool active = i < pEnd;
    float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
    int lx = 0, ly = 0;
    if (active) {
      pi = P[i];
      const uint h = sortHash[i] - key0;
      lx = (int)(h & 1u);
      ly = (int)((h >> 1) & 1u);
    }
    const bool iIn = !active || (pi.x >= -hxL && pi.x < hxL && pi.y >= -hyL && pi.y < hyL && pi.z >= -hzL && pi.z < hzL);
    const bool allIIn = __all(iIn) != 0;
    bool drainPBC = false;
    Acc acc;
    PairQueue<uint, kQCapQuad, 64> Q{queue + wave * (kQCapQuad * 64) + lane, 0};
#pragma unroll 1
    for (int dz = -1; dz <= 1; ++dz) {
      int gz = z0 + dz;
      float sz = 0.0f;
      if (gz < 0) { gz += n.z; sz = -box.boxSize.z; } else if (gz >= n.z) { gz -= n.z; sz = box.boxSize.z; }
#pragma unroll 1
      for (int jy = -1; jy <= 2; ++jy) {
        int gy = y0 + jy;
        if (gy > n.y) continue;  // beyond the +1 neighbour of the last existing row
        float sy = 0.0f;
        if (gy < 0) { gy += n.y; sy = -box.boxSize.y; } else if (gy >= n.y) { gy -= n.y; sy = box.boxSize.y; }
        const bool mineY = active && (jy - ly <= 1) && (ly - jy <= 1);
#pragma unroll 1
        for (int jx = -1; jx <= 2; ++jx) {
          int gx = x0 + jx;
          if (gx > n.x) continue;
          float sx = 0.0f;
          if (gx < 0) { gx += n.x; sx = -box.boxSize.x; } else if (gx >= n.x) { gx -= n.x; sx = box.boxSize.x; }
          const uint hh = morton_hash(make_int3(gx, gy, gz));
          const int cs = (int)keyStart[hh], ce = (int)keyStart[hh + 1];
          if (cs == ce) continue;
          const bool mine = mineY && (jx - lx <= 1) && (lx - jx <= 1);
          const bool shiftOK = allIIn && !smallGrid && keyOutside[hh] == 0;
          if (!shiftOK) {
            drainPBC = true;
            quad_scan<2, NT1, WE, WV>(acc, Q, drainPBC, P, cs, ce, mine, pi, box, 0.f, 0.f, 0.f, rc2, p1, tbl, ntypes);
          } else if (sx != 0.0f || sy != 0.0f || sz != 0.0f) {
            drainPBC = true;  // queued pairs of this cell need the image; the full arithmetic gives the same bits
            quad_scan<1, NT1, WE, WV>(acc, Q, drainPBC, P, cs, ce, mine, pi, box, sx, sy, sz, rc2, p1, tbl, ntypes);
          } else {
            quad_scan<0, NT1, WE, WV>(acc, Q, drainPBC, P, cs, ce, mine, pi, box, 0.f, 0.f, 0.f, rc2, p1, tbl, ntypes);
          }
        }
      }
    }
    if (drainPBC) lj_drain<true, NT1, WE, WV>(acc, Q, P, pi, box, p1, tbl, ntypes);
    else lj_drain<false, NT1, WE, WV>(acc, Q, P, pi, box, p1, tbl, ntypes);
    if (active) {
      const int gi = groupIndex[i];
      const int ori = out.globalIndex ? out.globalIndex[gi] : gi;
      write_out(out, ori, acc);
    }
  }
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
template <int K, bool NT1, bool WE, bool WV>
static int launch_brick(const ListView &cl, const GridT<float> &grid, const BoxT<float> &box, const LJParams *tbl,
                        int ntypes, const Outputs &out, uint nKeys, hipStream_t st) {
  using Bk = Brick<K>;
  // LDS budget: positions + tables.  Sized for ~1.5x the mean halo population at liquid density.
  const int ldsBytes = (K == 3) ? 30 * 1024 : (K == 4) ? 50 * 1024 : (K == 5) ? 78 * 1024 : 140 * 1024;
  const int fixed = (int)(sizeof(int) * (3 * Bk::NH + 1 + 4) + sizeof(LJParams) * kMaxTypesLds * kMaxTypesLds +
                          sizeof(unsigned short) * kQCapBrick * Bk::THREADS + 16);
  const int capacity = (ldsBytes - fixed) / (int)sizeof(float4);
  const uint nBricks = (nKeys + Bk::NCELL - 1) >> K;
  auto kern = k_lj_brick<K, NT1, WE, WV>;
  static bool attrSet = false;
  if (!attrSet) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes);
    attrSet = true;
  }
  hipLaunchKernelGGL(kern, dim3(nBricks), dim3(Bk::THREADS), ldsBytes, st, cl, grid, box, tbl, ntypes, out, capacity,
                     (const int *)nullptr);
  return 0;
}

// lj_tile.hip: cell-pair tiles, distance test on the matrix pipe (same pairs, another summation order)
bool lj_tile_supported(const CellList *h, const BoxT<float> &box);
template <bool NT1, bool WE, bool WV>
int launch_lj_tile(CellList *h, const ListView &cl, const BoxT<float> &box, const LJParams *tbl, int ntypes, const Outputs &out,
                   int shape, hipStream_t st);

template <bool NT1, bool WE, bool WV>
static int dispatch_celllist(CellList *h, int algo, int brickBits, const BoxT<float> &box, const LJParams *tbl,
                             int ntypes, const Outputs &out, hipStream_t st) {
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
  if (h->numOwned != 0x7fffffff && algo != UAMMD_LJ_ALGO_AUTO && algo != UAMMD_LJ_ALGO_GENERAL && algo != UAMMD_LJ_ALGO_STAGED &&
      algo != UAMMD_LJ_ALGO_RING && algo != UAMMD_LJ_ALGO_RING_HALF && algo != UAMMD_LJ_ALGO_TILE && algo != UAMMD_LJ_ALGO_TILE1 && algo != UAMMD_LJ_ALGO_EXACT) {
    set_last_error("uammd_lj_transverse_celllist: the num_owned option is implemented by the general kernel only");
    return -3;
  }
  if ((algo == UAMMD_LJ_ALGO_TILE || algo == UAMMD_LJ_ALGO_TILE1) && !lj_tile_supported(h, box)) {
    set_last_error("uammd_lj_transverse_celllist: the tile kernel needs a tabulated list built on the potential's box with 1 or >= 3 "
                   "cells per dimension (>= 4 along a periodic x)");
    return -3;
  }
  if (algo == UAMMD_LJ_ALGO_TILE || algo == UAMMD_LJ_ALGO_TILE1 || (algo == UAMMD_LJ_ALGO_AUTO && lj_tile_supported(h, box)))
    return launch_lj_tile<NT1, WE, WV>(h, cl, box, tbl, ntypes, out, algo == UAMMD_LJ_ALGO_TILE1 ? 1 : 4, st);
  if (algo == UAMMD_LJ_ALGO_EXACT) algo = UAMMD_LJ_ALGO_AUTO;  // from here on AUTO = the fastest bit-exact kernel for the grid
  const GridT<float> &g = h->grid;
  const bool brickOK = h->haveKeyStart && g.box.px() && g.box.py() && g.box.pz() && g.cellDim.x >= 4 &&
                       g.cellDim.y >= 4 && g.cellDim.z >= 4 && (NT1 || ntypes <= kMaxTypesLds);
  if (algo == UAMMD_LJ_ALGO_BRICK && !brickOK) {
    set_last_error("uammd_lj_transverse_celllist: the LDS-tiled kernel needs a fully periodic grid with >= 4 cells "
                   "per dimension (cellDim = %d %d %d) and <= %d types", g.cellDim.x, g.cellDim.y, g.cellDim.z, kMaxTypesLds);
    return -3;
  }
  const bool sameBox = box.boxSize.x == g.box.boxSize.x && box.boxSize.y == g.box.boxSize.y &&
                       box.boxSize.z == g.box.boxSize.z && box.px() && box.py() && box.pz();
  const bool quadOK = brickOK && sameBox && h->keyOutside.ptr != nullptr;
  if (algo == UAMMD_LJ_ALGO_QUAD && !quadOK) {
    set_last_error("uammd_lj_transverse_celllist: the uniform-j kernel needs a fully periodic grid with >= 4 cells per "
                   "dimension built on the same box as the potential");
    return -3;
  }
  // AUTO currently resolves to the thread-per-particle kernel: measured fastest on MI355X at C2/C3
  // (profiles/r01_lj_kernels.md); the brick and uniform-j kernels stay selectable.
  if (quadOK && algo == UAMMD_LJ_ALGO_QUAD) {
    const uint nQuads = h->nKeys >> 2;
    hipLaunchKernelGGL((k_lj_quad<NT1, WE, WV>), dim3((nQuads + 3) / 4), dim3(256), 0, st, cl.sortPos, cl.sortHash,
                       cl.keyStart, cl.groupIndex, (const unsigned char *)h->keyOutside.ptr, g, box, tbl, ntypes,
                       out.force, out.energy, out.virial, out.globalIndex, nQuads);
    return 0;
  }
  if (brickOK && algo == UAMMD_LJ_ALGO_BRICK) {
    switch (brickBits) {
      case 3: return launch_brick<3, NT1, WE, WV>(cl, g, box, tbl, ntypes, out, h->nKeys, st);
      case 4: return launch_brick<4, NT1, WE, WV>(cl, g, box, tbl, ntypes, out, h->nKeys, st);
      case 5: return launch_brick<5, NT1, WE, WV>(cl, g, box, tbl, ntypes, out, h->nKeys, st);
      default: return launch_brick<6, NT1, WE, WV>(cl, g, box, tbl, ntypes, out, h->nKeys, st);
    }
  }
  if (algo == UAMMD_LJ_ALGO_CELLWAVE) {
    const int ncells = g.cellDim.x * g.cellDim.y * g.cellDim.z;
    hipLaunchKernelGGL((k_lj_cellwave<NT1, WE, WV>), dim3((ncells + 3) / 4), dim3(256), 0, st, cl, g, box, tbl, ntypes, out);
    return 0;
  }
  if (algo == UAMMD_LJ_ALGO_STAGED && !cl.cellRange) {
    set_last_error("uammd_lj_transverse_celllist: the wave-staged kernel needs the per-cell range table (a tabulated key space)");
    return -3;
  }
  const bool staged = cl.cellRange && algo == UAMMD_LJ_ALGO_STAGED;
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
  if (false)
    ;
  else if (staged)
    hipLaunchKernelGGL((k_lj_staged<NT1, WE, WV>), dim3((cl.N + 127) / 128), dim3(128), 0, st, cl, g, box, tbl, ntypes, out);
  else
    hipLaunchKernelGGL((k_lj_general<NT1, WE, WV>), dim3((cl.N + 127) / 128), dim3(128), 0, st, cl, g, box, tbl, ntypes, out);
  return 0;
}

template <bool NT1, bool WE, bool WV>
static int dispatch_nbody(const float4 *pos, int N, const BoxT<float> &box, const LJParams *tbl, int ntypes,
                          const Outputs &out, hipStream_t st) {
  hipLaunchKernelGGL((k_lj_nbody<NT1, WE, WV>), dim3((N + 127) / 128), dim3(128), 0, st, pos, N, box, tbl, ntypes, out);
  return 0;
}

int g_brick_bits = 5;  // tunable (uammd_hip_set_tunable("lj_brick_bits", k))

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
  if (name && std::string(name) == "lj_brick_bits" && value >= 3 && value <= 6) { g_brick_bits = value; return 0; }
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
  UH_DISPATCH_FEV(dispatch_celllist, h, algo, g_brick_bits, box, tbl, ntypes, out, (hipStream_t)stream);
  if (rc) return rc;
  UH_CHECK(hipGetLastError());
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
