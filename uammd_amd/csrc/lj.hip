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

#include <string>

namespace uammd_hip {

struct LJParams { float cutOff2, sigma2, epsilonDivSigma2, shift; };

UH_D float lj_force(float r2, const LJParams &p) {  // |f|/r
  if (r2 >= p.cutOff2) return 0.0f;
  const float invr2 = p.sigma2 / r2;
  const float invr6 = invr2 * invr2 * invr2;
  return p.epsilonDivSigma2 * fmaf(-48.0f, invr6, 24.0f) * invr6 * invr2;
}
UH_D float lj_energy(float r2, const LJParams &p) {  // half the pair energy
  if (r2 >= p.cutOff2) return 0.0f;
  const float invr2 = p.sigma2 / r2;
  const float invr6 = invr2 * invr2 * invr2;
  const float E = fmaf(p.epsilonDivSigma2 * p.sigma2 * 4.0f * invr6, (invr6 - 1.0f), -p.shift);
  return 0.5f * E;
}
UH_D LJParams lj_lookup(const LJParams *tbl, int ntypes, int ti, int tj) {
  if (ti > tj) { const int t = ti; ti = tj; tj = t; }
  int typeIndex = ti + ntypes * tj;
  if (ti >= ntypes || tj >= ntypes) typeIndex = 0;
  return tbl[typeIndex];
}

struct Acc { float fx = 0.f, fy = 0.f, fz = 0.f, e = 0.f, v = 0.f; };

// One pair (compute + default accumulate).  PBC=false is only instantiated where the minimum image
// offset is provably zero.
template <bool PBC, bool WE, bool WV>
UH_D void lj_pair(Acc &a, const BoxT<float> &box, const LJParams &p, const float4 &ri, const float4 &rj) {
  real3f r12{rj.x - ri.x, rj.y - ri.y, rj.z - ri.z};
  if (PBC) r12 = box.apply_pbc(r12);
  const float r2 = dot3(r12, r12);
  if (r2 == 0.0f) return;
  if (WE) a.e += lj_energy(r2, p);
  const float fm = lj_force(r2, p);
  a.fx = fmaf(fm, r12.x, a.fx);
  a.fy = fmaf(fm, r12.y, a.fy);
  a.fz = fmaf(fm, r12.z, a.fz);
  if (WV) a.v += dot3(real3f{fm * r12.x, fm * r12.y, fm * r12.z}, r12);
}

struct ListView {
  const uint *cellStart;
  const int *cellEnd;
  const float4 *sortPos;
  const int *groupIndex;
  const uint *sortHash;
  const uint *keyStart;
  uint validCell;
  int N;
};

struct Outputs {
  float4 *force;
  float *energy;
  float *virial;
  const int *globalIndex;
};

UH_D void write_out(const Outputs &o, int ori, const Acc &a) {
  if (o.force) {
    float4 f = o.force[ori];
    f.x += a.fx; f.y += a.fy; f.z += a.fz; f.w += 0.0f;
    o.force[ori] = f;
  }
  if (o.energy) o.energy[ori] += a.e;
  if (o.virial) o.virial[ori] += a.v;
}

// ---- general walk (also the in-kernel fallback of the brick kernel) ------------------------------
template <bool NT1, bool WE, bool WV>
UH_D void walk_global(Acc &acc, const ListView &cl, const GridT<float> &grid, const BoxT<float> &box,
                      const LJParams *tbl, int ntypes, const LJParams &p1, const float4 &pi) {
  const int3 n = grid.cellDim;
  const int npx = n.x > 1 ? 3 : 1, npy = n.y > 1 ? 3 : 1, npz = n.z > 1 ? 3 : 1;
  const int numberNeighbourCells = npx * npy * npz;
  const int3 celli = grid.getCell(real3f{pi.x, pi.y, pi.z});
  for (int cc = 0; cc < numberNeighbourCells; ++cc) {
    int3 cellj = celli;
    if (npx > 1) cellj.x += cc % 3 - 1;
    if (npy > 1) cellj.y += (cc / npx) % 3 - 1;
    if (npz > 1) cellj.z += cc / (npx * npy) - 1;
    cellj.x = grid.pbc_x(cellj.x);
    cellj.y = grid.pbc_y(cellj.y);
    cellj.z = grid.pbc_z(cellj.z);
    // outside a non periodic box: no such cell (see DESIGN.md "non-periodic neighbours")
    if (cellj.x < 0 || cellj.x >= n.x || cellj.y < 0 || cellj.y >= n.y || cellj.z < 0 || cellj.z >= n.z) continue;
    const int icellj = grid.getCellIndex(cellj);
    const uint cs = cl.cellStart[icellj];
    if (cs < cl.validCell) continue;
    const int first = (int)(cs - cl.validCell), last = cl.cellEnd[icellj];
    for (int j = first; j < last; ++j) {
      const float4 pj = cl.sortPos[j];
      if (NT1) lj_pair<true, WE, WV>(acc, box, p1, pi, pj);
      else lj_pair<true, WE, WV>(acc, box, lj_lookup(tbl, ntypes, (int)pi.w, (int)pj.w), pi, pj);
    }
  }
}

template <bool NT1, bool WE, bool WV>
__global__ void __launch_bounds__(128) k_lj_general(ListView cl, GridT<float> grid, BoxT<float> box,
                                                     const LJParams *__restrict__ tbl, int ntypes, Outputs out) {
  const int id = blockIdx.x * 128 + threadIdx.x;
  if (id >= cl.N) return;
  const int gi = cl.groupIndex[id];
  const int ori = out.globalIndex ? out.globalIndex[gi] : gi;
  const float4 pi = cl.sortPos[id];
  LJParams p1 = tbl[0];
  Acc acc;
  walk_global<NT1, WE, WV>(acc, cl, grid, box, tbl, ntypes, p1, pi);
  write_out(out, ori, acc);
}

// ---- LDS-tiled brick kernel ------------------------------------------------------------------------
template <int K> struct Brick {
  static constexpr int BX = (K >= 4) ? 4 : 2;
  static constexpr int BY = (K >= 5) ? 4 : 2;
  static constexpr int BZ = (K >= 6) ? 4 : 2;
  static constexpr int HX = BX + 2, HY = BY + 2, HZ = BZ + 2;
  static constexpr int NH = HX * HY * HZ;
  static constexpr int NCELL = 1 << K;
  static constexpr int THREADS = (K == 3) ? 128 : 256;
};

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
  int *misc = gstart + Bk::NH;                                                         // [4]: total, allInBox
  LJParams *stbl = reinterpret_cast<LJParams *>(misc + 4);                             // [kMaxTypesLds^2] if !NT1

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
    off[t + 1] = c;  // counts, scanned in place below
  }
  if (tid == 0) off[0] = 0;
  __syncthreads();
  // 2. exclusive scan of NH (<= 216) counts: a single wave does it with shuffles.
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

  if (total > capacity) {
    // Too dense for the LDS tile (block-uniform): walk global memory like k_lj_general.
    for (int i = pStart + tid; i < pEnd; i += T) {
      const int gi = cl.groupIndex[i];
      const int ori = out.globalIndex ? out.globalIndex[gi] : gi;
      const float4 pi = cl.sortPos[i];
      Acc acc;
      walk_global<NT1, WE, WV>(acc, cl, grid, box, tbl, ntypes, p1, pi);
      write_out(out, ori, acc);
    }
    return;
  }

  // 3. stage the halo: 16-lane groups copy one cell at a time (a cell is ~13 contiguous float4).
  {
    const int g = tid >> 4, l = tid & 15;
    const float hxL = 0.5f * box.boxSize.x, hyL = 0.5f * box.boxSize.y, hzL = 0.5f * box.boxSize.z;
    bool inBox = true;
    for (int t = g; t < Bk::NH; t += T / 16) {
      const int s = gstart[t], o = off[t], c = off[t + 1] - o;
      for (int k = l; k < c; k += 16) {
        const float4 p = cl.sortPos[s + k];
        spos[o + k] = p;
        inBox = inBox && (p.x >= -hxL && p.x < hxL && p.y >= -hyL && p.y < hyL && p.z >= -hzL && p.z < hzL);
      }
    }
    if (!inBox) misc[1] = 0;  // benign race: every writer stores 0
  }
  __syncthreads();
  const bool allInBox = misc[1] != 0;
  const bool smallGrid = (n.x < 5) || (n.y < 5) || (n.z < 5);

  // 4. traversal: one i-particle per lane, 9 contiguous LDS rows each.
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
    if (active) {
      const int cbase = lx + Bk::HX * (ly + Bk::HY * lz);
#pragma unroll 1
      for (int r = 0; r < 9; ++r) {
        const int c0 = cbase + Bk::HX * ((r % 3) + Bk::HY * (r / 3));
        const int jb = off[c0], je = off[c0 + 3];
        if (needPBC) {
          for (int j = jb; j < je; ++j) {
            const float4 pj = spos[j];
            if (NT1) lj_pair<true, WE, WV>(acc, box, p1, pi, pj);
            else lj_pair<true, WE, WV>(acc, box, lj_lookup(stbl, ntypes, (int)pi.w, (int)pj.w), pi, pj);
          }
        } else {
          for (int j = jb; j < je; ++j) {
            const float4 pj = spos[j];
            if (NT1) lj_pair<false, WE, WV>(acc, box, p1, pi, pj);
            else lj_pair<false, WE, WV>(acc, box, lj_lookup(stbl, ntypes, (int)pi.w, (int)pj.w), pi, pj);
          }
        }
      }
      const int gi = cl.groupIndex[i];
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
  const int t = blockIdx.x * 128 + threadIdx.x;
  const bool active = t < N;
  const int id = active ? (out.globalIndex ? out.globalIndex[t] : t) : 0;
  const float4 pi = active ? pos[id] : make_float4(0.f, 0.f, 0.f, 0.f);
  const LJParams p1 = tbl[0];
  Acc acc;
  const int numTiles = (N + 127) / 128;
  for (int tileIdx = 0; tileIdx < numTiles; ++tileIdx) {
    const int iload = tileIdx * 128 + threadIdx.x;
    if (iload < N) tile[threadIdx.x] = pos[out.globalIndex ? out.globalIndex[iload] : iload];
    __syncthreads();
    if (active) {
      const int cnt = min(128, N - tileIdx * 128);
      for (int c = 0; c < cnt; ++c) {
        const float4 pj = tile[c];
        if (NT1) lj_pair<true, WE, WV>(acc, box, p1, pi, pj);
        else lj_pair<true, WE, WV>(acc, box, lj_lookup(tbl, ntypes, (int)pi.w, (int)pj.w), pi, pj);
      }
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
  const int ldsBytes = (K == 3) ? 24 * 1024 : (K == 4) ? 40 * 1024 : (K == 5) ? 52 * 1024 : 64 * 1024;
  const int fixed = (int)(sizeof(int) * (2 * Bk::NH + 1 + 4) + sizeof(LJParams) * kMaxTypesLds * kMaxTypesLds + 16);
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
  cl.validCell = h->validCell;
  cl.N = h->numberParticlesBuilt;
  const GridT<float> &g = h->grid;
  const bool brickOK = h->haveKeyStart && g.box.px() && g.box.py() && g.box.pz() && g.cellDim.x >= 4 &&
                       g.cellDim.y >= 4 && g.cellDim.z >= 4 && (NT1 || ntypes <= kMaxTypesLds);
  if (algo == UAMMD_LJ_ALGO_BRICK && !brickOK) {
    set_last_error("uammd_lj_transverse_celllist: the LDS-tiled kernel needs a fully periodic grid with >= 4 cells "
                   "per dimension (cellDim = %d %d %d) and <= %d types", g.cellDim.x, g.cellDim.y, g.cellDim.z, kMaxTypesLds);
    return -3;
  }
  if (brickOK && algo != UAMMD_LJ_ALGO_GENERAL) {
    switch (brickBits) {
      case 3: return launch_brick<3, NT1, WE, WV>(cl, g, box, tbl, ntypes, out, h->nKeys, st);
      case 4: return launch_brick<4, NT1, WE, WV>(cl, g, box, tbl, ntypes, out, h->nKeys, st);
      case 5: return launch_brick<5, NT1, WE, WV>(cl, g, box, tbl, ntypes, out, h->nKeys, st);
      default: return launch_brick<6, NT1, WE, WV>(cl, g, box, tbl, ntypes, out, h->nKeys, st);
    }
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
