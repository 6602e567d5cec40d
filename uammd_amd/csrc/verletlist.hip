// Path A — Verlet (explicit neighbour) list for gfx950.
//
// Reference behaviour:
//   BasicNeighbourListBase::update / fillBasicNeighbourList      Interactor/NeighbourList/BasicList/BasicListBase.cuh:41-215
//   BasicNeighbourList_ns::NeighbourIterator                      Interactor/NeighbourList/BasicList/NeighbourContainer.cuh:54-104
//   VerletListBase::update / needsRebuild / checkMaximumDrift     Interactor/NeighbourList/VerletList/VerletListBase.cuh:55-199
//   transverseWithNeighbourContainer                              Interactor/NeighbourList/common.cuh:10-34
// What must come out: a cell list built on STORED positions with cut-off 1.08 rc; for every sorted particle i the
// sorted indices j of the particles with |pbc(rj - ri)|^2 <= (1.08 rc)^2 — i itself included — in the order of the
// 27-cell walk, entry k at neighbourList[k*N + i] (lanes of a wave write/read consecutive addresses); capacity starts at
// 32 per particle and grows by 32 until no particle reaches it; the list is rebuilt when any particle has moved
// >= (1.08 rc - rc)/2 from its stored position; sortPos is refreshed from the current positions at every update.
#include "celllist.hpp"
#include "lj_common.hpp"

#include <string>

namespace uammd_hip {

struct VerletList {
  CellList cl;  // BasicNeighbourListBase::cl
  DeviceBuffer neighbourList, numberNeighbours, storedPos, sortPos, flags;
  int maxNeighboursPerParticle = 32;      // BasicListBase.cuh:127
  float verletRadiusMultiplier = 1.08f;   // VerletListBase.cuh:101
  float currentCutOff = 0.0f;
  float boxL[3] = {0, 0, 0};
  int boxPeriodic[3] = {0, 0, 0};
  bool haveBox = false;
  int storedN = -1;
  bool forceNextRebuild = true;
  int stepsSinceLastUpdate = 0;
  int N = 0;
  uint *hostFlag = nullptr;  // pinned: the drift / overflow flags are read back every update, as in the reference
  ~VerletList() {
    if (hostFlag) (void)hipHostFree(hostFlag);
  }
};

// K7.  One thread per sorted particle; the walk is the cell list's (x fastest, then y, z; particles ascending).
// Entry k of particle i lives at neighbourList[k*N + i]: a row k is contiguous over i, but the lanes of a wave reach a
// given k at different times, so storing each hit as it is found scatters every store instruction over up to 64 cache
// lines (measured: 973 us per build at C3, 2.4x the force traversal).  Instead each lane appends its hits to a small
// FIFO in LDS ([slot][lane], conflict free) and, when any lane's FIFO is nearly full, the wave flushes ROW BY ROW: for
// k from the smallest pending row to the largest, the lanes that hold an entry for row k store it — every store
// instruction writes one (partially masked) contiguous 256-byte row segment.
constexpr int kFillQCap = 24;
__global__ void __launch_bounds__(128) k_verlet_fill(const float4 *__restrict__ sortPos, const uint *__restrict__ cellStart,
                                                      const int *__restrict__ cellEnd, uint validCell, int N,
                                                      GridT<float> grid, BoxT<float> box, float cutOff2,
                                                      int maxNeighboursPerParticle, int *__restrict__ neighbourList,
                                                      int *__restrict__ numberNeighbours, int *__restrict__ tooManyFlag,
                                                      const unsigned char *__restrict__ cellOutside,
                                                      const uint2 *__restrict__ cellRange) {
  __shared__ int q[kFillQCap * 128];
  const int idRaw = (int)xcd_contiguous_block(blockIdx.x, gridDim.x) * 128 + threadIdx.x;
  const bool valid = idRaw < N;
  const int id = valid ? idRaw : N - 1;  // idle lanes of the last block shadow a real particle and never store
  int *myq = q + threadIdx.x;            // entry t at myq[t * 128]
  const float4 pi = sortPos[id];
  const int3 n = grid.cellDim;
  const int npx = n.x > 1 ? 3 : 1, npy = n.y > 1 ? 3 : 1, npz = n.z > 1 ? 3 : 1;
  const int numberNeighbourCells = npx * npy * npz;
  const int3 celli = grid.getCell(real3f{pi.x, pi.y, pi.z});
  int base = 0, cnt = 0;  // rows [0, base) are stored, rows [base, base + cnt) are in the FIFO
  bool overflow = false;
  // exact minimum-image skipping, as in the force traversal (lj.hip walk_global): for direct, unwrapped neighbour cells on a
  // >= 5-cell grid whose particles are stored inside the primary box the image offset is exactly zero
  const bool sameBox = box.boxSize.x == grid.box.boxSize.x && box.boxSize.y == grid.box.boxSize.y &&
                       box.boxSize.z == grid.box.boxSize.z && box.px() == grid.box.px() && box.py() == grid.box.py() &&
                       box.pz() == grid.box.pz();
  const bool smallGrid = n.x < 5 || n.y < 5 || n.z < 5 || !cellOutside || !sameBox;
  const float hx = 0.5f * box.boxSize.x, hy = 0.5f * box.boxSize.y, hz = 0.5f * box.boxSize.z;
  const bool iOut = !(pi.x >= -hx && pi.x < hx && pi.y >= -hy && pi.y < hy && pi.z >= -hz && pi.z < hz);
  int *mine = neighbourList + id;
  auto flush = [&]() {
    int kmin = cnt > 0 ? base : 0x7fffffff, kmax = cnt > 0 ? base + cnt : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      kmin = min(kmin, __shfl_xor(kmin, o, 64));
      kmax = max(kmax, __shfl_xor(kmax, o, 64));
    }
    for (int k = kmin; k < kmax; ++k) {
      const int t = k - base;
      if (valid && t >= 0 && t < cnt) mine[(size_t)k * (size_t)N] = myq[t * 128];
    }
    base += cnt;
    cnt = 0;
  };
  const int ncells = n.x * n.y * n.z;
  // with the per-cell range table (celllist.hip k_cell_tables): one 8-byte read per neighbour cell, issued one cell ahead
  // (577 -> 531 us per build at C3 together with the ballot loop below, which replaces a 6-step shuffle reduction per cell)
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
    rg = cellRange[exists ? grid.getCellIndex(cellj) : ncells];
  };
  uint2 rgAhead = make_uint2(0u, 0u);
  bool wrappedAhead = false;
  if (cellRange) fetch(0, rgAhead, wrappedAhead);
  for (int cc = 0; cc < numberNeighbourCells; ++cc) {
    int first = 0, last = 0;
    bool needPBC = false;
    if (cellRange) {
      const uint2 rg = rgAhead;
      const bool wrapped = wrappedAhead;
      if (cc + 1 < numberNeighbourCells) fetch(cc + 1, rgAhead, wrappedAhead);
      first = (int)rg.x;
      last = (int)(rg.y & 0x7fffffffu);
      needPBC = first < last && (smallGrid || iOut || wrapped || (rg.y >> 31) != 0u);
    } else {
      int3 cellj = celli;
      if (npx > 1) cellj.x += cc % 3 - 1;
      if (npy > 1) cellj.y += (cc / npx) % 3 - 1;
      if (npz > 1) cellj.z += cc / (npx * npy) - 1;
      const int3 raw = cellj;
      cellj.x = grid.pbc_x(cellj.x);
      cellj.y = grid.pbc_y(cellj.y);
      cellj.z = grid.pbc_z(cellj.z);
      // outside a non periodic box: no such cell (DESIGN.md "non-periodic neighbours")
      const bool exists = !(cellj.x < 0 || cellj.x >= n.x || cellj.y < 0 || cellj.y >= n.y || cellj.z < 0 || cellj.z >= n.z);
      if (exists) {
        const int icellj = grid.getCellIndex(cellj);
        const uint cs = cellStart[icellj];
        if (cs >= validCell) {
          first = (int)(cs - validCell);
          last = cellEnd[icellj];
          needPBC = smallGrid || iOut || raw.x != cellj.x || raw.y != cellj.y || raw.z != cellj.z || cellOutside[icellj] != 0;
        }
      }
    }
    const bool wavePBC = __any(needPBC);
    // all lanes iterate together (the FIFO flush is a wave-level operation): up to the longest cell of the wave
    const int len = last - first;
    for (int s = 0; __any(s < len); s += 4) {
      if (__any(cnt > kFillQCap - 4)) flush();
      const int j = first + s;
      if (j < last) {
        const float4 *__restrict__ pj = sortPos + j;  // reads past the cell are masked below (the array has 4 spare entries)
        const float4 c0 = pj[0], c1 = pj[1], c2 = pj[2], c3 = pj[3];
        float dd[4];
        if (wavePBC) {
          dd[0] = lj_dist2<true>(box, pi, c0); dd[1] = lj_dist2<true>(box, pi, c1);
          dd[2] = lj_dist2<true>(box, pi, c2); dd[3] = lj_dist2<true>(box, pi, c3);
        } else {
          dd[0] = lj_dist2<false>(box, pi, c0); dd[1] = lj_dist2<false>(box, pi, c1);
          dd[2] = lj_dist2<false>(box, pi, c2); dd[3] = lj_dist2<false>(box, pi, c3);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (j + u < last && dd[u] <= cutOff2 && !overflow) {
            // BasicListBase.cuh:65-70: nneigh++; if (nneigh >= max) { atomicMax(flag, nneigh); return; }
            if (base + cnt + 1 >= maxNeighboursPerParticle) {
              overflow = true;
            } else {
              myq[cnt * 128] = j + u;
              ++cnt;
            }
          }
        }
      }
    }
  }
  flush();
  if (!valid) return;
  if (overflow) atomicMax(tooManyFlag, maxNeighboursPerParticle);  // the host retries with 32 more slots
  else numberNeighbours[id] = base;
}

// K8.  Counts the particles that moved >= maxDistAllowed from their stored position.
__global__ void __launch_bounds__(256) k_verlet_drift(const float4 *__restrict__ currentPos, const float4 *__restrict__ storedPos,
                                                       float maxDistAllowed, uint *__restrict__ errorFlag, BoxT<float> box, int N) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  bool over = false;
  if (id < N) {
    const float4 c = currentPos[id], s = storedPos[id];
    const real3f rij = box.apply_pbc(real3f{c.x - s.x, c.y - s.y, c.z - s.z});
    over = dot3(rij, rij) >= (maxDistAllowed * maxDistAllowed);
  }
  const unsigned long long m = __ballot(over);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(errorFlag, (uint)__popcll(m));
}

__global__ void __launch_bounds__(256) k_verlet_sortpos(const float4 *__restrict__ pos, const int *__restrict__ groupIndex,
                                                         float4 *__restrict__ sortPos, int N) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id < N) sortPos[id] = pos[groupIndex[id]];
}

// K6 over the list.  Thread per sorted particle; neighbour k of the 64 lanes of a wave is one coalesced 256-B read of the
// list, followed by a gather of the neighbours' positions (L1/L2 hits: neighbouring particles share most neighbours).
// Every listed pair goes through the full minimum image and the exact per-pair cut-off, in list order: the float sums are
// the reference's.  79 % of the listed pairs are inside rc, so there is no scan/drain split here.
template <bool NT1, bool WE, bool WV>
__global__ void __launch_bounds__(128) k_lj_verlet(const float4 *__restrict__ sortPos, const int *__restrict__ groupIndex,
                                                    const int *__restrict__ neighbourList,
                                                    const int *__restrict__ numberNeighbours, int N, BoxT<float> box,
                                                    const LJParams *__restrict__ tbl, int ntypes, Outputs out) {
  const int id = (int)xcd_contiguous_block(blockIdx.x, gridDim.x) * 128 + threadIdx.x;
  if (id >= N) return;
  const int gi = groupIndex[id];
  const int ori = out.globalIndex ? out.globalIndex[gi] : gi;
  const float4 pi = sortPos[id];
  const LJParams p1 = tbl[0];
  const int nn = numberNeighbours[id];
  const int *mine = neighbourList + id;
  Acc acc;
  // Software pipeline, two stages deep: the list entries of iteration k + 2 and the positions of iteration k + 1 are requested before
  // the pairs of iteration k are evaluated (written as load - gather - evaluate per iteration the loop was two dependent round trips
  // per four neighbours, ~16 times per particle).  Entries past the particle's own count re-read its last one and carry no weight.
  const int l1 = max(nn - 1, 0);
  // (a particle without a single neighbour — a NaN position fails even its own r2 <= rc2 test — has no row 0: its prologue reads
  // nothing and gathers its own position)
  auto entries = [&](int k, int (&j)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) j[u] = nn > 0 ? mine[(size_t)min(k + u, l1) * N] : id;
  };
  int jb[4], jc[4];
  float4 cb[4];
  entries(0, jb);
  entries(4, jc);
#pragma unroll
  for (int u = 0; u < 4; ++u) cb[u] = sortPos[jb[u]];
  for (int k = 0; k < nn; k += 4) {
    float4 c[4];
    int jd[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = cb[u];
    entries(k + 8, jd);
#pragma unroll
    for (int u = 0; u < 4; ++u) cb[u] = sortPos[jc[u]];
#pragma unroll
    for (int u = 0; u < 4; ++u) jc[u] = jd[u];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      real3f r12;
      float fm, e;
      if (NT1) lj_eval<true, WE>(box, p1, pi, c[u], r12, fm, e);
      else lj_eval<true, WE>(box, lj_lookup(tbl, ntypes, (int)pi.w, (int)c[u].w), pi, c[u], r12, fm, e);
      const bool live = k + u < nn;
      lj_acc<WE, WV>(acc, r12, live ? fm : 0.0f, live ? e : 0.0f);
    }
  }
  write_out(out, ori, acc);
}

template <bool NT1, bool WE, bool WV>
static int dispatch_verlet(VerletList *v, const BoxT<float> &box, const LJParams *tbl, int ntypes, const Outputs &out,
                           hipStream_t st) {
  hipLaunchKernelGGL((k_lj_verlet<NT1, WE, WV>), dim3((v->N + 127) / 128), dim3(128), 0, st, (const float4 *)v->sortPos.ptr,
                     (const int *)v->cl.index.ptr, (const int *)v->neighbourList.ptr, (const int *)v->numberNeighbours.ptr,
                     v->N, box, tbl, ntypes, out);
  return 0;
}

static int verlet_read_flag(VerletList *v, int slot, hipStream_t st, uint *value) {
  UH_CHECK(hipMemcpyAsync(v->hostFlag, (const uint *)v->flags.ptr + slot, sizeof(uint), hipMemcpyDeviceToHost, st));
  UH_CHECK(hipStreamSynchronize(st));
  *value = *v->hostFlag;
  return 0;
}

// BasicNeighbourListBase::update (BasicListBase.cuh:132-142) on the stored positions
static int verlet_rebuild(VerletList *v, hipStream_t st) {
  const int N = v->N;
  const float rcut = v->currentCutOff * v->verletRadiusMultiplier;  // VerletListBase.cuh:153-156
  const float rc3[3] = {rcut, rcut, rcut};
  int cd[3], gper[3];
  float gL[3];
  if (int e = uammd_celllist_create_grid(v->boxL, v->boxPeriodic, rc3, cd, gL, gper)) return e;
  if (int e = v->cl.update((const float4 *)v->storedPos.ptr, N, gL, gper, cd, st)) return e;
  const BoxT<float> box = make_box<float>(v->boxL, v->boxPeriodic);
  if (int e = v->numberNeighbours.reserve(sizeof(int) * (size_t)N)) return e;
  for (;;) {  // fillBasicNeighbourList: retry with 32 more slots while some particle overflows
    if (int e = v->neighbourList.reserve(sizeof(int) * (size_t)N * (size_t)(v->maxNeighboursPerParticle + 1))) return e;
    UH_CHECK(hipMemsetAsync((uint *)v->flags.ptr + 1, 0, sizeof(uint), st));
    hipLaunchKernelGGL(k_verlet_fill, dim3((N + 127) / 128), dim3(128), 0, st, (const float4 *)v->cl.sortPos.ptr,
                       (const uint *)v->cl.cellStart.ptr, (const int *)v->cl.cellEnd.ptr, v->cl.validCell, N, v->cl.grid, box,
                       rcut * rcut, v->maxNeighboursPerParticle, (int *)v->neighbourList.ptr, (int *)v->numberNeighbours.ptr,
                       (int *)v->flags.ptr + 1, v->cl.haveCellOutside ? (const unsigned char *)v->cl.cellOutside.ptr : nullptr,
                       v->cl.haveCellOutside ? (const uint2 *)v->cl.cellRange.ptr : nullptr);
    UH_CHECK(hipGetLastError());
    uint flag = 0;
    if (int e = verlet_read_flag(v, 1, st, &flag)) return e;
    if (flag == 0) break;
    v->maxNeighboursPerParticle += 32;
  }
  return 0;
}

}  // namespace uammd_hip

using namespace uammd_hip;

extern "C" {

int uammd_verletlist_create(uammd_verletlist **out) {
  if (!out) { set_last_error("uammd_verletlist_create: null argument"); return -1; }
  VerletList *v = new VerletList();
  if (hipHostMalloc((void **)&v->hostFlag, sizeof(uint)) != hipSuccess) { v->hostFlag = nullptr; }
  if (!v->hostFlag || v->flags.reserve(2 * sizeof(uint))) {
    delete v;
    set_last_error("uammd_verletlist_create: allocation failed");
    return -2;
  }
  *out = reinterpret_cast<uammd_verletlist *>(v);
  return 0;
}

int uammd_verletlist_destroy(uammd_verletlist *h) {
  delete reinterpret_cast<VerletList *>(h);
  return 0;
}

int uammd_verletlist_force_next_update(uammd_verletlist *h) {
  if (!h) { set_last_error("uammd_verletlist_force_next_update: null handle"); return -1; }
  reinterpret_cast<VerletList *>(h)->forceNextRebuild = true;
  return 0;
}

int uammd_verletlist_set_cutoff_multiplier(uammd_verletlist *h, float newMultiplier) {
  if (!h) { set_last_error("uammd_verletlist_set_cutoff_multiplier: null handle"); return -1; }
  VerletList *v = reinterpret_cast<VerletList *>(h);
  v->forceNextRebuild = true;
  v->verletRadiusMultiplier = newMultiplier;
  return 0;
}

int uammd_verletlist_get_steps_since_last_update(uammd_verletlist *h, int *steps) {
  if (!h || !steps) { set_last_error("uammd_verletlist_get_steps_since_last_update: null argument"); return -1; }
  *steps = reinterpret_cast<VerletList *>(h)->stepsSinceLastUpdate - 1;  // VerletListBase.cuh:124
  return 0;
}

// VerletListBase::update (VerletListBase.cuh:107-117)
int uammd_verletlist_update(uammd_verletlist *h, const float *d_pos, int numberParticles, const float L[3],
                            const int periodic[3], float cutOff, void *stream, int *rebuilt) {
  if (!h || (numberParticles > 0 && !d_pos) || !L || !periodic) { set_last_error("uammd_verletlist_update: null argument"); return -1; }
  VerletList *v = reinterpret_cast<VerletList *>(h);
  hipStream_t st = (hipStream_t)stream;
  const int N = numberParticles;
  if (rebuilt) *rebuilt = 0;
  if (N <= 0) { v->N = 0; return 0; }
  const BoxT<float> box = make_box<float>(L, periodic);
  // needsRebuild, VerletListBase.cuh:158-175
  bool rebuild = false;
  if (v->forceNextRebuild) {
    v->forceNextRebuild = false;
    rebuild = true;
  } else if (!v->haveBox || L[0] != v->boxL[0] || L[1] != v->boxL[1] || L[2] != v->boxL[2] ||
             (periodic[0] != 0) != (v->boxPeriodic[0] != 0) || (periodic[1] != 0) != (v->boxPeriodic[1] != 0) ||
             (periodic[2] != 0) != (v->boxPeriodic[2] != 0) || cutOff != v->currentCutOff || N != v->storedN) {
    rebuild = true;
  } else {  // isParticleDriftOverThreshold, :177-199
    const float thresholdDistance = (float)((v->verletRadiusMultiplier * v->currentCutOff - v->currentCutOff) / 2.0);
    if (thresholdDistance <= 1e-6) {
      rebuild = true;
    } else {
      UH_CHECK(hipMemsetAsync(v->flags.ptr, 0, sizeof(uint), st));
      hipLaunchKernelGGL(k_verlet_drift, dim3((N + 255) / 256), dim3(256), 0, st, (const float4 *)d_pos,
                         (const float4 *)v->storedPos.ptr, thresholdDistance, (uint *)v->flags.ptr, box, N);
      UH_CHECK(hipGetLastError());
      uint flag = 0;
      if (int e = verlet_read_flag(v, 0, st, &flag)) return e;
      rebuild = flag > 0;
    }
  }
  if (rebuild) {
    v->stepsSinceLastUpdate = 0;
    for (int k = 0; k < 3; ++k) { v->boxL[k] = L[k]; v->boxPeriodic[k] = periodic[k] != 0; }
    v->haveBox = true;
    v->currentCutOff = cutOff;
    v->N = N;
    v->storedN = N;
    if (int e = v->storedPos.reserve(sizeof(float4) * (size_t)N)) return e;
    if (int e = v->sortPos.reserve(sizeof(float4) * (size_t)N)) return e;
    UH_CHECK(hipMemcpyAsync(v->storedPos.ptr, d_pos, sizeof(float4) * (size_t)N, hipMemcpyDeviceToDevice, st));  // storeCurrentPos
    if (int e = verlet_rebuild(v, st)) return e;
    if (rebuilt) *rebuilt = 1;
  }
  // updateSortedPositions, :140-151
  hipLaunchKernelGGL(k_verlet_sortpos, dim3((N + 255) / 256), dim3(256), 0, st, (const float4 *)d_pos,
                     (const int *)v->cl.index.ptr, (float4 *)v->sortPos.ptr, N);
  UH_CHECK(hipGetLastError());
  v->stepsSinceLastUpdate++;
  return 0;
}

int uammd_verletlist_get(uammd_verletlist *h, uammd_verletlist_data *out) {
  if (!h || !out) { set_last_error("uammd_verletlist_get: null argument"); return -1; }
  VerletList *v = reinterpret_cast<VerletList *>(h);
  out->d_neighbourList = (const int *)v->neighbourList.ptr;
  out->d_numberNeighbours = (const int *)v->numberNeighbours.ptr;
  out->d_sortPos = (const float *)v->sortPos.ptr;
  out->d_groupIndex = (const int *)v->cl.index.ptr;
  out->particleStride = v->N;  // BasicListBase.cuh:163-166: the stride is numberNeighbours.size()
  out->maxNeighboursPerParticle = v->maxNeighboursPerParticle;
  out->numberParticles = v->N;
  return 0;
}

int uammd_lj_transverse_verletlist(uammd_verletlist *h, const uammd_lj_pair_parameters *d_paramTable, int ntypes,
                                   const float boxL[3], const int boxPeriodic[3], float *d_force, float *d_energy,
                                   float *d_virial, const int *d_globalIndex, void *stream) {
  if (!h || !d_paramTable || ntypes < 1) { set_last_error("uammd_lj_transverse_verletlist: bad arguments"); return -1; }
  VerletList *v = reinterpret_cast<VerletList *>(h);
  if (v->N == 0) return 0;
  const BoxT<float> box = make_box<float>(boxL, boxPeriodic);
  Outputs out{reinterpret_cast<float4 *>(d_force), d_energy, d_virial, d_globalIndex};
  const LJParams *tbl = reinterpret_cast<const LJParams *>(d_paramTable);
  int rc = 0;
  const bool nt1 = ntypes == 1, we = d_energy != nullptr, wv = d_virial != nullptr;
  if (nt1 && !we && !wv) rc = dispatch_verlet<true, false, false>(v, box, tbl, ntypes, out, (hipStream_t)stream);
  else if (nt1) rc = dispatch_verlet<true, true, true>(v, box, tbl, ntypes, out, (hipStream_t)stream);
  else if (!we && !wv) rc = dispatch_verlet<false, false, false>(v, box, tbl, ntypes, out, (hipStream_t)stream);
  else rc = dispatch_verlet<false, true, true>(v, box, tbl, ntypes, out, (hipStream_t)stream);
  if (rc) return rc;
  UH_CHECK(hipGetLastError());
  return 0;
}

}  // extern "C"
