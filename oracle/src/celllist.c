/* ORACLE — TEST INFRASTRUCTURE ONLY (see common.h).
 *
 * Path A, neighbour search: CPU restatement of
 *   Sorter::MortonHash                      utils/ParticleSorter.cuh:51-76
 *   Sorter::clz / maxbit                    utils/ParticleSorter.cuh:93-100, :264-266
 *   assignHash (K1)                         utils/ParticleSorter.cuh:102-111
 *   cub::DeviceRadixSort::SortPairs (K2)    utils/ParticleSorter.cuh:303-321  — a STABLE sort of
 *        (hash,index) pairs on key bits [0,maxbit); CUB itself is a third-party dependency that
 *        is not in /root/reference (CUDA toolkit >= 11, third_party/uammd_cub.cuh:4-8); a stable
 *        sort has exactly one answer, which is what is restated here.
 *   applyCurrentOrder (K3)                  utils/ParticleSorter.cuh:178-187
 *   CellList::createUpdateGrid              Interactor/NeighbourList/CellList.cuh:100-126
 *   CellListBase::updateCurrentValidCell    .../CellList/CellListBase.cuh:210-230
 *   CellList_ns::fillCellList (K4)          .../CellList/CellListBase.cuh:68-94
 */
#include "common.h"
#ifdef _OPENMP
#include <omp.h>
#endif
#include <limits.h>
#include <float.h>

/* utils/ParticleSorter.cuh:57-65 */
static inline uint encodeMorton(uint i) {
  uint x = i;
  x &= 0x3ff;
  x = (x | x << 16) & 0x30000ff;
  x = (x | x << 8) & 0x300f00f;
  x = (x | x << 4) & 0x30c30c3;
  x = (x | x << 2) & 0x9249249;
  return x;
}
/* utils/ParticleSorter.cuh:67-70 */
static inline uint mortonHash(int3 c) {
  return encodeMorton((uint)c.x) | (encodeMorton((uint)c.y) << 1) | (encodeMorton((uint)c.z) << 2);
}
/* utils/ParticleSorter.cuh:93-100 */
static int sorter_clz(uint n) {
  n |= (n >> 1);
  n |= (n >> 2);
  n |= (n >> 4);
  n |= (n >> 8);
  n |= (n >> 16);
  return 32 - __builtin_ffs((int)(n - (n >> 1)));
}

ORACLE_API uint oracle_morton_hash(int cx, int cy, int cz) { return mortonHash(mki3(cx, cy, cz)); }

ORACLE_API int oracle_sort_end_bit(uint maxHash) { /* ParticleSorter.cuh:264-266 */
  int maxbit = 32 - sorter_clz(maxHash);
  return maxbit < 32 ? maxbit : 32;
}

/* CellList::createUpdateGrid, CellList.cuh:100-126.  Outputs the grid's cellDim and the box the
 * list is built on (an infinite dimension becomes 64*cutOff and non periodic). */
ORACLE_API void oracle_celllist_create_grid(const real *L_in, const int *periodic_in, const real *cutOff,
                                            int *cellDim_out, real *L_out, int *periodic_out) {
  real3 L = mk3(L_in[0], L_in[1], L_in[2]);
#ifdef DOUBLE_PRECISION
  const real inf = DBL_MAX;
#else
  const real inf = FLT_MAX;
#endif
  const int maximumNumberOfCells = 64;
  if (L.x >= inf) L.x = maximumNumberOfCells * cutOff[0];
  if (L.y >= inf) L.y = maximumNumberOfCells * cutOff[1];
  if (L.z >= inf) L.z = maximumNumberOfCells * cutOff[2];
  Box in = box_from(L_in, periodic_in);
  Box updateBox = box_make(L);
  box_set_periodicity(&updateBox, box_px(&in) && L.x < inf, box_py(&in) && L.y < inf, box_pz(&in) && L.z < inf);
  Grid g = grid_make_mincell(updateBox, mk3(cutOff[0], cutOff[1], cutOff[2]));
  int3 cd = g.cellDim;
  if (cd.x <= 3) cd.x = 1;
  if (cd.y <= 3) cd.y = 1;
  if (cd.z <= 3) cd.z = 1;
  cellDim_out[0] = cd.x; cellDim_out[1] = cd.y; cellDim_out[2] = cd.z;
  L_out[0] = L.x; L_out[1] = L.y; L_out[2] = L.z;
  periodic_out[0] = box_px(&updateBox); periodic_out[1] = box_py(&updateBox); periodic_out[2] = box_pz(&updateBox);
}

/* CellListBase::updateCurrentValidCell, CellListBase.cuh:210-230.
 * state[0] = currentValidCell_counter (start at -1), state[1] = particles of the previous build.
 * Returns currentValidCell; *needs_clear = 1 when the reference zero-fills cellStart. */
ORACLE_API uint oracle_celllist_next_valid_cell(int numberParticles, long long *state, int *needs_clear) {
  long long counter = state[0];
  if ((long long)numberParticles != state[1]) counter = -1;
  const int isCounterUninitialized = (counter < 0);
  const unsigned long long nextStepMaximumValue = (unsigned long long)numberParticles * (unsigned long long)(counter + 2);
  const unsigned long long maximumStorableValue = (unsigned long long)UINT_MAX - 1ull;
  const int nextStepOverflows = (nextStepMaximumValue >= maximumStorableValue);
  uint currentValidCell;
  if (isCounterUninitialized || nextStepOverflows) {
    currentValidCell = (uint)numberParticles;
    counter = 1;
    *needs_clear = 1;
  } else {
    counter++;
    currentValidCell = (uint)numberParticles * (uint)counter;
    *needs_clear = 0;
  }
  state[0] = counter;
  state[1] = numberParticles;
  return currentValidCell;
}

/* bench.py's cpu_baseline runs the oracle on all host cores: with this flag the build below (hash, sort,
 * reorder, cell tables) and the IBM spreading use OpenMP.  The build's results are identical either way (the sort stays stable);
 * the spreading then adds with atomics in thread order (rounding-level differences), so tests leave the flag off. */
static int g_oracle_parallel = 0;
ORACLE_API void oracle_set_parallel(int on) { g_oracle_parallel = on; }
ORACLE_API int oracle_get_parallel(void) { return g_oracle_parallel; }

/* the same stable sort with every pass split over T contiguous chunks of the input: per-chunk digit histograms, one exclusive scan
 * over (digit, chunk) and a scatter in which a chunk's elements keep their order: bit-identical output */
static void stable_sort_pairs_parallel(uint *keys, int *vals, int N, int end_bit) {
  int T = 1;
#ifdef _OPENMP
  T = omp_get_max_threads();
#endif
  if (T > 64) T = 64;
  uint *k2 = (uint *)malloc(sizeof(uint) * (size_t)N);
  int *v2 = (int *)malloc(sizeof(int) * (size_t)N);
  size_t *hist = (size_t *)malloc(sizeof(size_t) * 256 * (size_t)T);
  uint *ka = keys, *kb = k2;
  int *va = vals, *vb = v2;
  for (int shift = 0; shift < end_bit; shift += 8) {
    int bits = end_bit - shift < 8 ? end_bit - shift : 8;
    uint mask = (1u << bits) - 1u;
    memset(hist, 0, sizeof(size_t) * 256 * (size_t)T);
#pragma omp parallel for schedule(static) num_threads(T)
    for (int t = 0; t < T; t++) {
      const size_t lo = (size_t)N * (size_t)t / (size_t)T, hi = (size_t)N * (size_t)(t + 1) / (size_t)T;
      size_t *h = hist + 256 * (size_t)t;
      for (size_t i = lo; i < hi; i++) h[(ka[i] >> shift) & mask]++;
    }
    size_t run = 0;
    for (int d = 0; d < 256; d++)
      for (int t = 0; t < T; t++) {
        const size_t c = hist[256 * (size_t)t + d];
        hist[256 * (size_t)t + d] = run;
        run += c;
      }
#pragma omp parallel for schedule(static) num_threads(T)
    for (int t = 0; t < T; t++) {
      const size_t lo = (size_t)N * (size_t)t / (size_t)T, hi = (size_t)N * (size_t)(t + 1) / (size_t)T;
      size_t *h = hist + 256 * (size_t)t;
      for (size_t i = lo; i < hi; i++) {
        const size_t dst = h[(ka[i] >> shift) & mask]++;
        kb[dst] = ka[i];
        vb[dst] = va[i];
      }
    }
    uint *tk = ka; ka = kb; kb = tk;
    int *tv = va; va = vb; vb = tv;
  }
  if (ka != keys) {
    memcpy(keys, ka, sizeof(uint) * (size_t)N);
    memcpy(vals, va, sizeof(int) * (size_t)N);
  }
  free(k2);
  free(v2);
  free(hist);
}

/* Stable LSD radix sort of (key,value) on key bits [0,end_bit) — the contract of
 * cub::DeviceRadixSort::SortPairs(…, begin_bit=0, end_bit) used at ParticleSorter.cuh:316-320. */
ORACLE_API void oracle_stable_sort_pairs(uint *keys, int *vals, int N, int end_bit) {
  if (end_bit <= 0 || N <= 1) return;
  if (g_oracle_parallel) { stable_sort_pairs_parallel(keys, vals, N, end_bit); return; }
  uint *k2 = (uint *)malloc(sizeof(uint) * (size_t)N);
  int *v2 = (int *)malloc(sizeof(int) * (size_t)N);
  uint *ka = keys, *kb = k2;
  int *va = vals, *vb = v2;
  for (int shift = 0; shift < end_bit; shift += 8) {
    int bits = end_bit - shift < 8 ? end_bit - shift : 8;
    uint mask = (1u << bits) - 1u;
    size_t count[257];
    memset(count, 0, sizeof(count));
    for (int i = 0; i < N; i++) count[((ka[i] >> shift) & mask) + 1]++;
    for (int d = 0; d < 256; d++) count[d + 1] += count[d];
    for (int i = 0; i < N; i++) {
      size_t dst = count[(ka[i] >> shift) & mask]++;
      kb[dst] = ka[i];
      vb[dst] = va[i];
    }
    uint *tk = ka; ka = kb; kb = tk;
    int *tv = va; va = vb; vb = tv;
  }
  if (ka != keys) {
    memcpy(keys, ka, sizeof(uint) * (size_t)N);
    memcpy(vals, va, sizeof(int) * (size_t)N);
  }
  free(k2);
  free(v2);
}

/* K1: hash[i] = Morton(getCell(pos[i])), index[i] = i   (ParticleSorter.cuh:102-111, :156-164) */
ORACLE_API void oracle_assign_hash(const real4 *pos, int N, const real *L, const int *periodic, const int *cellDim,
                                   uint *hash, int *index) {
  Box box = box_from(L, periodic);
  Grid grid = grid_make(box, mki3(cellDim[0], cellDim[1], cellDim[2]));
#pragma omp parallel for schedule(static) if (g_oracle_parallel)
  for (int i = 0; i < N; i++) {
    int3 c = grid_get_cell(&grid, mk3(pos[i].x, pos[i].y, pos[i].z));
    hash[i] = mortonHash(c);
    index[i] = i;
  }
}

/* Linear cell index of every position (used by tests to cross-check the tables). */
ORACLE_API void oracle_cell_of(const real4 *pos, int N, const real *L, const int *periodic, const int *cellDim,
                               int *icell) {
  Box box = box_from(L, periodic);
  Grid grid = grid_make(box, mki3(cellDim[0], cellDim[1], cellDim[2]));
  for (int i = 0; i < N; i++) icell[i] = grid_cell_index(&grid, grid_get_cell(&grid, mk3(pos[i].x, pos[i].y, pos[i].z)));
}

/* CellListBase::update (CellListBase.cuh:124-140) = K1 + K2 + K3 + K4.
 * cellStart/cellEnd are in/out: entries of cells that stay empty keep their previous contents,
 * exactly as on the device (the epoch trick).  Returns the error flag of fillCellList. */
ORACLE_API int oracle_celllist_build(const real4 *pos, int N, const real *L, const int *periodic, const int *cellDim,
                                     uint validCell, uint *hash, int *index, real4 *sortPos, uint *cellStart,
                                     int *cellEnd) {
  Box box = box_from(L, periodic);
  Grid grid = grid_make(box, mki3(cellDim[0], cellDim[1], cellDim[2]));
  oracle_assign_hash(pos, N, L, periodic, cellDim, hash, index);
  uint maxHash = mortonHash(mki3(cellDim[0] - 1, cellDim[1] - 1, cellDim[2] - 1)); /* :161 */
  oracle_stable_sort_pairs(hash, index, N, oracle_sort_end_bit(maxHash));
#pragma omp parallel for schedule(static) if (g_oracle_parallel)
  for (int i = 0; i < N; i++) sortPos[i] = pos[index[i]]; /* K3 */
  /* K4: fillCellList, CellListBase.cuh:68-94 (one "thread" per id; writes never collide) */
  int errorFlag = 0;
  const int ncells = grid_ncells(&grid);
#pragma omp parallel for schedule(static) reduction(| : errorFlag) if (g_oracle_parallel)
  for (int id = 0; id < N; id++) {
    uint icell, icell2;
    icell = (uint)grid_cell_index(&grid, grid_get_cell(&grid, mk3(sortPos[id].x, sortPos[id].y, sortPos[id].z)));
    if (id > 0)
      icell2 = (uint)grid_cell_index(&grid, grid_get_cell(&grid, mk3(sortPos[id - 1].x, sortPos[id - 1].y, sortPos[id - 1].z)));
    else
      icell2 = 0;
    if (icell >= (uint)ncells || icell2 >= (uint)ncells) { errorFlag = 1; continue; }
    if (icell != icell2 || id == 0) {
      cellStart[icell] = (uint)id + validCell;
      if (id > 0) cellEnd[icell2] = id;
    }
    if (id == N - 1) cellEnd[icell] = N;
  }
  return errorFlag;
}
