// Path A — cell-list construction for gfx950.
//
// What the reference does (behaviour to match bit-for-bit; NOT its implementation):
//   K1 assignHash            utils/ParticleSorter.cuh:102-111   hash = Morton(getCell(pos)), index = i
//   K2 stable radix sort     utils/ParticleSorter.cuh:303-321   on key bits [0, maxbit)
//   K3 reorder               utils/ParticleSorter.cuh:178-187   sortPos[i] = pos[index[i]]
//   K4 fillCellList          Interactor/NeighbourList/CellList/CellListBase.cuh:68-94
//   epoch trick              CellListBase.cuh:210-230           cellStart holds start + VALID_CELL
//
// MI355X design: the sorted order of a stable sort by Morton key is unique — (key, original index)
// lexicographic — so it does not have to come from a multi-pass radix sort.  The default build is
// a COUNTING sort over the (small) Morton key space:
//   hash+count  : one pass over pos; per-key histogram with a returning atomic (the returned value
//                 is a provisional, order-nondeterministic rank inside the key);
//   scan        : exclusive scan of the histogram (<= 2^maxbit entries, L2 resident) -> keyStart[];
//   rank+scatter: each particle re-derives its STABLE rank = #{same-key particles with a smaller
//                 original index} from the provisional per-key member list (~13 members, L2 hits),
//                 then writes index/hash/sortPos at keyStart[key] + rank.  Deterministic, bit-equal
//                 to the radix sort, ~3 streaming passes instead of ~8;
//   cell tables : cellStart/cellEnd come straight from keyStart (one thread per *cell*).
// When the key space is too large for that (very sparse / huge grids) the build falls back to
// rocPRIM's stable radix sort + a boundary-detection pass.
// keyStart[] is kept: the LDS-tiled traversal (lj.hip) uses it to find the particle range of an
// aligned 4x4x4 brick of cells in O(1) (64 consecutive Morton keys).
#include "celllist.hpp"
#include "gj_step.hpp"
#include <atomic>
#include <vector>

#include <cstring>
#include <string>
#include <rocprim/rocprim.hpp>

#include <cstring>
#include <cstdarg>
#include <cstdio>
#include <limits>

namespace uammd_hip {

thread_local char g_last_error[1024] = "";
void set_last_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}

// ---- small device buffer helper ----------------------------------------------------------------
int DeviceBuffer::reserve(size_t bytes) {
  if (bytes <= cap) return 0;
  if (ptr && owned) UH_CHECK(hipFree(ptr));
  owned = true;
  ptr = nullptr;
  cap = 0;
  size_t want = bytes + bytes / 8 + 256;
  UH_CHECK(hipMalloc(&ptr, want));
  cap = want;
  return 0;
}
DeviceBuffer::~DeviceBuffer() {
  if (ptr && owned) (void)hipFree(ptr);
}

// utils/ParticleSorter.cuh:93-100 + :264-266 (clz by smearing; maxbit = 32 - clz(maxHash))
static int sort_end_bit(uint maxHash) {
  int msb = -1;
  for (int b = 31; b >= 0; --b)
    if (maxHash & (1u << b)) { msb = b; break; }
  return msb + 1;  // 0 when maxHash == 0 (the sort is then a no-op)
}

// ---- kernels -------------------------------------------------------------------------------------
constexpr int kBlock = 256;

// K1 (+ histogram): one thread per particle.
template <bool COUNT>
__global__ void __launch_bounds__(kBlock) k_hash(const float4 *__restrict__ pos, int N, GridT<float> grid,
                                                 uint *__restrict__ hash, int *__restrict__ index,
                                                 uint *__restrict__ keyCount, uint *__restrict__ provRank,
                                                 int *__restrict__ errorFlag, unsigned char *__restrict__ keyOutside) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;  // exited lanes drop out of the ballots below
  const float4 p = pos[i];
  int3 c = grid.getCell(real3f{p.x, p.y, p.z});
  // A particle outside a non periodic box (or a NaN) has no cell: flag it (the reference raises the
  // same flag in fillCellList, CellListBase.cuh:82-85) and clamp so that no table is overrun.
  if (c.x < 0 || c.x >= grid.cellDim.x || c.y < 0 || c.y >= grid.cellDim.y || c.z < 0 || c.z >= grid.cellDim.z ||
      !(p.x == p.x && p.y == p.y && p.z == p.z)) {  // (int)NaN is a valid-looking cell 0: test it by name
    errorFlag[0] = 1;
    c.x = min(max(c.x, 0), grid.cellDim.x - 1);
    c.y = min(max(c.y, 0), grid.cellDim.y - 1);
    c.z = min(max(c.z, 0), grid.cellDim.z - 1);
  }
  const uint h = morton_hash(c);
  hash[i] = h;
  // Per-key flag "some particle of this cell is stored outside the primary box" (unwrapped coordinates):
  // the uniform-j traversal may only replace the minimum-image arithmetic by a per-cell shift when the
  // flag is clear for both cells of a pair.  Benign race: every writer stores 1.
  if (keyOutside) {
    const float hx = 0.5f * grid.box.boxSize.x, hy = 0.5f * grid.box.boxSize.y, hz = 0.5f * grid.box.boxSize.z;
    if (!(p.x >= -hx && p.x < hx && p.y >= -hy && p.y < hy && p.z >= -hz && p.z < hz)) keyOutside[h] = 1;
  }
  if (COUNT) {
    // One returning atomic per particle.  (A wave-aggregated variant — one atomic per distinct key per
    // wave — was measured SLOWER at C3: ParticleData::sortParticles only sorts on a coarse 10-sigma grid, so
    // a wave still holds ~64 distinct fine keys; profiles/r01_lj_kernels.md.)
    const uint rank = atomicAdd(&keyCount[h], 1u);
    provRank[i] = rank;
  } else {
    index[i] = i;
  }
}

// K1 + histogram with block-level aggregation.  One returning global atomic per particle caps the kernel at ~19 G atomics/s
// (52 us at C3).  ParticleData::sortParticles keeps the input roughly in Morton order of a COARSE grid, so the 1024 particles
// of a workgroup fall into ~100 distinct fine keys: they are counted in an LDS hash table first (LDS atomics run per CU) and
// each distinct key then costs ONE global atomic that reserves the whole group's range of provisional ranks.  Unsorted input
// degrades gracefully to one global atomic per particle.
// kAggPerThread particles per thread: 4 for the plain build (256 or 512 particles per workgroup instead of 1024: same wall time,
// measured), 1 with the fused half step, whose workgroups otherwise all stream, all compute and all wait for their atomics at the
// same time (35.3 -> 30.5 us at C3 with four times the workgroups; the launch is 3.8 workgroups per CU at 4).
// GJ1: the fused MD step (uammd_verletnvt_gj_lj_step) — VerletNVT::GronbechJensen's first half step (GronbechJensen.cu:28-57) is applied
// to each particle as it is loaded, the new position is stored and hashed: one pass over pos / vel / force instead of the integrator's
// own launch followed by this kernel re-reading the positions.
template <bool GJ1, int kAggPerThread>
__global__ void __launch_bounds__(kBlock) k_hash_agg(float4 *__restrict__ pos, int N, GridT<float> grid,
                                                     uint *__restrict__ hash, uint *__restrict__ keyCount,
                                                     uint *__restrict__ provRank, int *__restrict__ errorFlag,
                                                     unsigned char *__restrict__ keyOutside, GJFuse gj) {
  constexpr int kAggSlots = 2 * kBlock * kAggPerThread;  // 2 x particles per workgroup: the probe sequences stay short
  constexpr int kSlotShift = 32 - (kAggPerThread == 4 ? 11 : kAggPerThread == 2 ? 10 : 9);
  static_assert(kAggSlots == (1 << (32 - kSlotShift)), "kSlotShift is log2 of the table size");
  __shared__ uint tKey[kAggSlots], tCnt[kAggSlots];
  for (int s = threadIdx.x; s < kAggSlots; s += kBlock) { tKey[s] = 0xffffffffu; tCnt[s] = 0u; }
  __syncthreads();
  const int base = blockIdx.x * (kBlock * kAggPerThread);
  uint myKey[kAggPerThread], mySlot[kAggPerThread], myRank[kAggPerThread];
  float4 p[kAggPerThread];
#pragma unroll
  for (int u = 0; u < kAggPerThread; ++u) {  // all loads first: four independent requests in flight per thread
    const int i = base + u * kBlock + threadIdx.x;
    p[u] = (i < N) ? pos[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (GJ1) {
    float3 v[kAggPerThread];
    float4 f4[kAggPerThread];
    const int nRows = gj.nRows > 0 ? min(gj.nRows, N) : N;   // (rows beyond are ghosts: hashed, not integrated)
#pragma unroll
    for (int u = 0; u < kAggPerThread; ++u) {
      const int i = min(base + u * kBlock + (int)threadIdx.x, nRows - 1);
      v[u] = make_float3(gj.vel[3 * (size_t)i], gj.vel[3 * (size_t)i + 1], gj.vel[3 * (size_t)i + 2]);
      f4[u] = gj.force[i];
    }
#pragma unroll
    for (int u = 0; u < kAggPerThread; ++u) {
      const int i = base + u * kBlock + threadIdx.x;
      if (i < nRows && !(gj.skip && gj.skip[i])) {
        const float invMass = 1.0f / (gj.defaultMass > 0 ? gj.defaultMass : gj.mass[i]);
        gj_step1(p[u], v[u], f4[u], invMass, gj.dt, gj.friction, gj.noiseAmplitude, gj.is2D, gj.keys ? (uint)gj.keys[i] : (uint)i, gj.stepNum,
                 gj.seed);
        pos[i] = p[u];
        gj.vel[3 * (size_t)i] = v[u].x; gj.vel[3 * (size_t)i + 1] = v[u].y; gj.vel[3 * (size_t)i + 2] = v[u].z;
        if (!gj.keepForce) gj.force[i] = make_float4(0.f, 0.f, 0.f, 0.f);  // (keepForce: the traversal that follows overwrites it)
      }
    }
  }
#pragma unroll
  for (int u = 0; u < kAggPerThread; ++u) {
    const int i = base + u * kBlock + threadIdx.x;
    myKey[u] = 0xffffffffu;
    mySlot[u] = 0;
    myRank[u] = 0;
    if (i < N) {
      int3 c = grid.getCell(real3f{p[u].x, p[u].y, p[u].z});
      if (c.x < 0 || c.x >= grid.cellDim.x || c.y < 0 || c.y >= grid.cellDim.y || c.z < 0 || c.z >= grid.cellDim.z ||
          !(p[u].x == p[u].x && p[u].y == p[u].y && p[u].z == p[u].z)) {  // (int)NaN is a valid-looking cell 0: test it by name
        errorFlag[0] = 1;
        c.x = min(max(c.x, 0), grid.cellDim.x - 1);
        c.y = min(max(c.y, 0), grid.cellDim.y - 1);
        c.z = min(max(c.z, 0), grid.cellDim.z - 1);
      }
      const uint h = morton_hash(c);
      hash[i] = h;
      if (keyOutside) {
        const float hx = 0.5f * grid.box.boxSize.x, hy = 0.5f * grid.box.boxSize.y, hz = 0.5f * grid.box.boxSize.z;
        if (!(p[u].x >= -hx && p[u].x < hx && p[u].y >= -hy && p[u].y < hy && p[u].z >= -hz && p[u].z < hz)) keyOutside[h] = 1;
      }
      myKey[u] = h;
    }
  }
  bool first[kAggPerThread];
#pragma unroll
  for (int u = 0; u < kAggPerThread; ++u) {
    first[u] = false;
    if (myKey[u] != 0xffffffffu) {
      const uint h = myKey[u];
      uint s = (h * 2654435761u) >> kSlotShift;  // multiplicative hash -> log2(kAggSlots) bits
      for (;;) {
        const uint old = atomicCAS(&tKey[s], 0xffffffffu, h);
        if (old == 0xffffffffu) first[u] = true;  // this thread opened the slot: it makes the slot's global reservation below
        if (old == 0xffffffffu || old == h) break;
        s = (s + 1) & (kAggSlots - 1);
      }
      mySlot[u] = s;
      myRank[u] = atomicAdd(&tCnt[s], 1u);
    }
  }
  __syncthreads();
  // One global atomic per distinct key of the workgroup reserves the whole group's range of provisional ranks.  The thread that
  // opened the slot issues it, so a thread's (up to kAggPerThread) returning atomics are all in flight together; walking the
  // table instead costs every wave kAggSlots / kBlock serialised round trips (measured: 12 of the kernel's 29 us at C3).
  uint grp[kAggPerThread];
#pragma unroll
  for (int u = 0; u < kAggPerThread; ++u) grp[u] = first[u] ? atomicAdd(&keyCount[myKey[u]], tCnt[mySlot[u]]) : 0u;
#pragma unroll
  for (int u = 0; u < kAggPerThread; ++u)
    if (first[u]) tCnt[mySlot[u]] = grp[u];  // (only the opener has read this slot's count since the barrier)
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kAggPerThread; ++u) {
    const int i = base + u * kBlock + threadIdx.x;
    if (i < N) provRank[i] = tCnt[mySlot[u]] + myRank[u];
  }
}

constexpr int kScanChunks = 256;     // workgroups of the key scan (k_key_scan)
// Exclusive scan of the key counters in ONE launch (rocPRIM's look-back scan is two: state initialisation + scan).  Workgroup b owns
// the keys [b, b + 1) << chunkShift, at most kScanChunks workgroups: each publishes the total of its chunk tagged with the build's
// generation (nothing to reset between builds), then thread t < b waits for workgroup t's total — a direct sum over the predecessors,
// no chained look-back.  Workgroups are dispatched in index order, so the ones waited for are always running.  (A second histogram
// over chunks accumulated by the hash kernel was tried first: ~1e5 atomics on a few hundred addresses serialise in L2, 450 us.)
__global__ void __launch_bounds__(kBlock) k_key_scan(const uint *__restrict__ keyCount, uint nKeys, int chunkShift,
                                                     unsigned long long *__restrict__ flags, uint generation,
                                                     uint *__restrict__ keyStart, int *__restrict__ errorFlag) {
  __shared__ uint waveSum[kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto block_sum = [&](uint v) -> uint {  // every thread gets the workgroup's total
    v = wave_inclusive_scan(v);   // (lane 63: the wave's total)
    __syncthreads();  // (waveSum is reused)
    if (lane == 63) waveSum[wave] = v;
    __syncthreads();
    uint t = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) t += waveSum[w];
    return t;
  };
  const uint k0 = (uint)blockIdx.x << chunkShift, k1 = min(k0 + (1u << chunkShift), nKeys);
  auto load4 = [&](uint k, uint c[4]) {
    if (k + 3 < k1) {
      const uint4 q = *reinterpret_cast<const uint4 *>(keyCount + k);  // (k is a multiple of 4 and the counters are 16-byte aligned)
      c[0] = q.x; c[1] = q.y; c[2] = q.z; c[3] = q.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) c[j] = (k + j < k1) ? keyCount[k + j] : 0u;
    }
  };
  uint c[4];
  uint part = 0;
  for (uint t = k0; t < k1; t += 4 * kBlock) {
    load4(t + 4 * threadIdx.x, c);
    part += c[0] + c[1] + c[2] + c[3];
  }
  const uint myTotal = block_sum(part);
  if (threadIdx.x == 0)
    __hip_atomic_store(&flags[blockIdx.x], ((unsigned long long)generation << 32) | myTotal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint before = 0;
  if (threadIdx.x < blockIdx.x) {
    unsigned long long f;
    int spins = 0;
    do {
      f = __hip_atomic_load(&flags[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (the flag IS the data: no ordering needed)
    } while ((uint)(f >> 32) != generation && ++spins < (1 << 22));  // (bounded: a broken dispatch-order assumption must not hang the GPU)
    if ((uint)(f >> 32) != generation) errorFlag[0] = 4;
    before = (uint)f;
  }
  uint offset = block_sum(before);
  for (uint t = k0; t < k1; t += 4 * kBlock) {
    const uint k = t + 4 * threadIdx.x;
    load4(k, c);
    const uint mine = c[0] + c[1] + c[2] + c[3];
    const uint incl = wave_inclusive_scan(mine);  // inclusive scan inside the wave (six DPP additions: no ds_bpermute ladder)
    __syncthreads();
    if (lane == 63) waveSum[wave] = incl;
    __syncthreads();
    uint pre = offset, total = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) {
      if (w < wave) pre += waveSum[w];
      total += waveSum[w];
    }
    uint s = pre + incl - mine;
    if (k + 3 < k1) {
      *reinterpret_cast<uint4 *>(keyStart + k) = make_uint4(s, s + c[0], s + c[0] + c[1], s + c[0] + c[1] + c[2]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (k + j < k1) keyStart[k + j] = s;
        s += c[j];
      }
    }
    offset += total;
  }
  if (k1 == nKeys && k0 < nKeys && threadIdx.x == 0) keyStart[nKeys] = offset;
}

// Writes the provisional member list: members[keyStart[h] + provRank[i]] = i
__global__ void __launch_bounds__(kBlock) k_members(const uint *__restrict__ hash, const uint *__restrict__ provRank,
                                                    const uint *__restrict__ keyStart, int N,
                                                    int *__restrict__ members, uint *__restrict__ sortHash) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const uint h = hash[i];
  const uint dst = keyStart[h] + provRank[i];
  members[dst] = i;
  if (sortHash) sortHash[dst] = h;  // already FINAL: all members of a cell's slot range share the key, whatever their order
}

// ---- counting build, second half -----------------------------------------------------------------------------------
// Stable placement, one thread per PROVISIONAL slot: the slot's cell is known from sortHash (coalesced), its members are the
// neighbouring slots (cache hits), the only gather left is pos[i].  Final slot = first slot of the cell + #{members with a smaller
// index}: the order utils/ParticleSorter.cuh:156-164,303-321 (stable radix sort of the hashes) produces.  Blocks past the
// particle range write the per-cell tables (k_cell_tables' job) so that the build ends with this launch.
// (Round 3 tried to do without k_members: the hash kernel wrote the members into fixed-capacity rows, one per key, and half a wave per
// cell ranked its row through lane shuffles and scattered — lists identical, the C3 step the same to 0.2 % (0.19358 against 0.19391 ms:
// the hash kernel's scattered row stores cost what k_members saved), a plain build 5 us slower; a particle-major form of it lost 6 us
// to 16-byte scattered stores, and an atomicMax of every workgroup on one word cost 19 us.  Removed.)
__global__ void __launch_bounds__(kBlock) k_rank_scatter2(const float4 *__restrict__ pos, const uint *__restrict__ sortHash,
                                                          const uint *__restrict__ keyStart, const int *__restrict__ members,
                                                          int N, int particleBlocks, int *__restrict__ index,
                                                          float4 *__restrict__ sortPos,
                                                          unsigned char *__restrict__ keyOutside, uint *__restrict__ keyCount,
                                                          int3 cellDim, uint validCell, uint *__restrict__ cellStart,
                                                          int *__restrict__ cellEnd, unsigned char *__restrict__ cellOutside,
                                                          uint2 *__restrict__ cellRange) {
  if ((int)blockIdx.x >= particleBlocks) {
    const int c = (blockIdx.x - particleBlocks) * kBlock + threadIdx.x;
    const int ncells = cellDim.x * cellDim.y * cellDim.z;
    if (c >= ncells) return;
    int3 cc;
    cc.x = c % cellDim.x;
    cc.y = (c / cellDim.x) % cellDim.y;
    cc.z = c / (cellDim.x * cellDim.y);
    const uint h = morton_hash(cc);
    const uint s = keyStart[h], e = keyStart[h + 1];
    const bool out = keyOutside[h] != 0;
    // this key's counter and flag have done their job: left at zero, the next build of the same grid needs no memset launch (every
    // key that was touched belongs to a cell; the scan's input beyond the last key was never written)
    if (e > s) keyCount[h] = 0u;
    if (out) keyOutside[h] = 0;
    cellStart[c] = (e > s) ? s + validCell : 0u;
    cellEnd[c] = (int)e;
    cellOutside[c] = out;
    cellRange[c] = (e > s) ? make_uint2(s, e | (out ? 0x80000000u : 0u)) : make_uint2(0u, 0u);
    if (c == ncells - 1) cellRange[ncells] = make_uint2(0u, 0u);
    return;
  }
  const int m = blockIdx.x * kBlock + threadIdx.x;
  if (m >= N) return;
  const uint h = sortHash[m];
  const int i = members[m];
  const float4 p = pos[i];
  const uint s = keyStart[h], e = keyStart[h + 1];
  uint rank = 0;
  uint q = s;
  for (; q + 4 <= e; q += 4) {
    const int a = members[q], b = members[q + 1], c = members[q + 2], d = members[q + 3];
    rank += (a < i) + (b < i) + (c < i) + (d < i);
  }
  for (; q < e; ++q) rank += (members[q] < i) ? 1u : 0u;
  const uint dst = s + rank;
  index[dst] = i;
  sortPos[dst] = p;
}

// Cell tables from keyStart: one thread per cell (linear index).  Non-empty: start + VALID_CELL;
// empty: 0 (< VALID_CELL).  The reference leaves stale values in empty cells; both mean "empty".
__global__ void __launch_bounds__(kBlock) k_cell_tables(const uint *__restrict__ keyStart,
                                                        const unsigned char *__restrict__ keyOutside, int3 cellDim,
                                                        uint validCell, uint *__restrict__ cellStart,
                                                        int *__restrict__ cellEnd,
                                                        unsigned char *__restrict__ cellOutside,
                                                        uint2 *__restrict__ cellRange) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  const int ncells = cellDim.x * cellDim.y * cellDim.z;
  if (c >= ncells) return;
  int3 cc;
  cc.x = c % cellDim.x;
  cc.y = (c / cellDim.x) % cellDim.y;
  cc.z = c / (cellDim.x * cellDim.y);
  const uint h = morton_hash(cc);
  const uint s = keyStart[h], e = keyStart[h + 1];
  if (cellStart) {
    cellStart[c] = (e > s) ? s + validCell : 0u;
    cellEnd[c] = (int)e;
  }
  if (cellOutside) cellOutside[c] = keyOutside[h];
  // the traversals' fused view of a cell: {first, last | outside << 31}, {0, 0} when empty; entry ncells = "no such cell"
  if (cellRange) {
    cellRange[c] = (e > s) ? make_uint2(s, e | (keyOutside[h] ? 0x80000000u : 0u)) : make_uint2(0u, 0u);
    if (c == ncells - 1) cellRange[ncells] = make_uint2(0u, 0u);
  }
}

// Radix path: K3 + K4 fused.  One thread per sorted slot; the previous particle's cell comes from a
// wave shuffle (lane 0 recomputes it).
__global__ void __launch_bounds__(kBlock) k_reorder_fill(const float4 *__restrict__ pos, const int *__restrict__ index,
                                                         int N, GridT<float> grid, uint validCell,
                                                         float4 *__restrict__ sortPos, uint *__restrict__ cellStart,
                                                         int *__restrict__ cellEnd, int *__restrict__ errorFlag) {
  const int id = blockIdx.x * kBlock + threadIdx.x;
  const bool active = id < N;
  float4 p = make_float4(0, 0, 0, 0);
  uint icell = 0;
  if (active) {
    p = pos[index[id]];
    sortPos[id] = p;
    icell = (uint)grid.getCellIndex(grid.getCell(real3f{p.x, p.y, p.z}));
  }
  uint icell2 = __shfl_up(icell, 1, 64);
  if (!active) return;
  if ((threadIdx.x & 63) == 0) {
    if (id > 0) {
      const float4 q = pos[index[id - 1]];
      icell2 = (uint)grid.getCellIndex(grid.getCell(real3f{q.x, q.y, q.z}));
    } else
      icell2 = 0;
  }
  const uint ncells = (uint)grid.getNumberCells();
  if (!(p.x == p.x && p.y == p.y && p.z == p.z)) errorFlag[0] = 1;
  if (icell >= ncells || icell2 >= ncells) {
    errorFlag[0] = 1;
    return;
  }
  if (icell != icell2 || id == 0) {
    cellStart[icell] = (uint)id + validCell;
    if (id > 0) cellEnd[icell2] = id;
  }
  if (id == N - 1) cellEnd[icell] = N;
}

// keyStart from sorted keys (radix path): keyStart[h] = first sorted index with key >= h.
__global__ void __launch_bounds__(kBlock) k_key_start_from_sorted(const uint *__restrict__ sortHash, int N, uint nkeys,
                                                                  uint *__restrict__ keyStart) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  uint lo = 1u, hi = 0u;  // empty range for the lanes past N
  if (i <= N) {
    lo = (i == 0) ? 0u : sortHash[i - 1] + 1u;
    hi = min((i == N) ? nkeys : sortHash[i], nkeys);  // inclusive upper key that starts at i
  }
  // Morton keys of a non power-of-two grid leave gaps of thousands of unused keys: a short run is written by its own lane,
  // a long one by the whole wave (one lane walking a 2^15-key gap alone cost 400 us at C3).
  constexpr uint kShort = 4;
  for (uint h = lo; h <= hi && h < lo + kShort; ++h) keyStart[h] = (uint)i;
  const bool isLong = hi >= lo && hi - lo >= kShort;
  unsigned long long pending = __ballot(isLong);
  const int lane = threadIdx.x & 63;
  while (pending) {
    const int src = __ffsll((long long)pending) - 1;
    pending &= pending - 1;
    const uint l = __shfl(lo, src, 64) + kShort, hh = __shfl(hi, src, 64);
    const uint v = (uint)__shfl(i, src, 64);
    for (uint h = l + lane; h <= hh; h += 64) keyStart[h] = v;
  }
}

__global__ void __launch_bounds__(kBlock) k_iota(int *__restrict__ v, int n) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) v[i] = i;
}

template <int BYTES> struct Elem { char b[BYTES]; };
struct Elem24 { uint2 v[3]; };   // real3 and real4 of the DOUBLE_PRECISION build, moved as 8- / 16-byte words
struct Elem32 { uint4 v[2]; };
__global__ void __launch_bounds__(kBlock) k_f64_to_f32(const double *__restrict__ in, float *__restrict__ out, size_t count) {
  const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (t < count) out[t] = (float)in[t];
}
// thrust::fill over pg->getPropertyIterator(property): zero the members' elements only
__global__ void __launch_bounds__(kBlock) k_zero_indexed(uint *__restrict__ v, const int *__restrict__ index, int n, int words) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= n * words) return;
  v[(size_t)index[t / words] * words + t % words] = 0u;
}

template <class T>
__global__ void __launch_bounds__(kBlock) k_gather(const T *__restrict__ in, const int *__restrict__ index,
                                                   T *__restrict__ out, int n) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = in[index[i]];
}

template <class T>
__global__ void __launch_bounds__(kBlock) k_scatter(const T *__restrict__ in, const int *__restrict__ index,
                                                    T *__restrict__ out, int n) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[index[i]] = in[i];
}

static inline int nblocks(long long n) { return (int)((n + kBlock - 1) / kBlock); }

// GronbechJensen's first half step on the rows [0, nRows) that are not marked in gj.skip, noise keyed by gj.keys[row] (the fallback of
// uammd_celllist_update_gj1 where the build does not carry the half step in its hash kernel)
__global__ void __launch_bounds__(kBlock) k_gj1_rows(float4 *__restrict__ pos, int nRows, GJFuse gj) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= nRows || (gj.skip && gj.skip[i])) return;
  float4 p = pos[i];
  float3 v = make_float3(gj.vel[3 * (size_t)i], gj.vel[3 * (size_t)i + 1], gj.vel[3 * (size_t)i + 2]);
  const float4 f4 = gj.force[i];
  const float invMass = 1.0f / (gj.defaultMass > 0 ? gj.defaultMass : gj.mass[i]);
  gj_step1(p, v, f4, invMass, gj.dt, gj.friction, gj.noiseAmplitude, gj.is2D, gj.keys ? (uint)gj.keys[i] : (uint)i, gj.stepNum, gj.seed);
  pos[i] = p;
  gj.vel[3 * (size_t)i] = v.x; gj.vel[3 * (size_t)i + 1] = v.y; gj.vel[3 * (size_t)i + 2] = v.z;
  if (!gj.keepForce) gj.force[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- CellList ------------------------------------------------------------------------------------
int CellList::next_valid_cell(int numberParticles, bool *needsClear) {
  // CellListBase::updateCurrentValidCell, CellListBase.cuh:210-230
  if (numberParticles != lastN) validCounter = -1;
  const bool uninit = validCounter < 0;
  const unsigned long long nextMax = (unsigned long long)numberParticles * (unsigned long long)(validCounter + 2);
  const unsigned long long maxStorable = (unsigned long long)std::numeric_limits<uint>::max() - 1ull;
  if (uninit || nextMax >= maxStorable) {
    validCell = (uint)numberParticles;
    validCounter = 1;
    *needsClear = true;
  } else {
    validCounter++;
    validCell = (uint)numberParticles * (uint)validCounter;
    *needsClear = false;
  }
  lastN = numberParticles;
  return 0;
}

// Half-precision copy of the sorted positions for the traversals' prefilter (lj.hip, k_lj_ringh): entry i holds particles
// i and i + 1 as three half2 (x_i, x_i+1), (y_i, y_i+1), (z_i, z_i+1), relative to the centre of particle i's cell and in
// units of the largest cell edge (|value| <= 0.5 for a particle of that cell).  A lane that walks a cell reads entries
// first, first + 2, ... : two candidates per 12 bytes, both relative to the same centre; the second half of the last entry
// of a cell is +inf when particle i + 1 belongs to another cell, so it fails every distance test.
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(kBlock) k_pack_half(const float4 *__restrict__ sortPos, int N, GridT<float> grid,
                                                      float scale, uint3 *__restrict__ out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N + 8) return;
  uint3 w = make_uint3(0u, 0u, 0u);
  if (i < N) {
    const float4 p0 = sortPos[i];
    const float4 p1 = sortPos[i + 1 < N ? i + 1 : i];
    const int3 c = grid.getCell(real3f{p0.x, p0.y, p0.z});
    const real3f a = grid.distanceToCellCenter(real3f{p0.x, p0.y, p0.z}, c);
    real3f b = grid.distanceToCellCenter(real3f{p1.x, p1.y, p1.z}, c);
    // the second half of a cell's last entry is not a candidate of that cell: +inf fails every distance test (NaN stays NaN
    // only for a real candidate), so the scan needs no per-candidate range test
    const int3 c1 = grid.getCell(real3f{p1.x, p1.y, p1.z});
    if (i + 1 >= N || c1.x != c.x || c1.y != c.y || c1.z != c.z) b = real3f{__builtin_inff(), 0.f, 0.f};
    const half2_t hx = {(_Float16)(a.x * scale), (_Float16)(b.x * scale)};
    const half2_t hy = {(_Float16)(a.y * scale), (_Float16)(b.y * scale)};
    const half2_t hz = {(_Float16)(a.z * scale), (_Float16)(b.z * scale)};
    w = make_uint3(__builtin_bit_cast(uint, hx), __builtin_bit_cast(uint, hy), __builtin_bit_cast(uint, hz));
  }
  out[i] = w;
}

int CellList::ensure_pack(hipStream_t st) {
  const float hmax = fmaxf(grid.cellSize.x, fmaxf(grid.cellSize.y, grid.cellSize.z));
  const float hmin = fminf(grid.cellSize.x, fminf(grid.cellSize.y, grid.cellSize.z));
  if (!(hmin > 0.f) || !(hmax < 3.0e38f)) return 1;  // 2D / degenerate grids keep the full-precision scan
  if (packValid) return 0;
  const int N = numberParticlesBuilt;
  if (int e = packHalf.reserve(sizeof(uint3) * ((size_t)N + 8))) return e;
  packScale = 1.0f / hmax;
  hipLaunchKernelGGL(k_pack_half, dim3(nblocks(N + 8)), dim3(kBlock), 0, st, (const float4 *)sortPos.ptr, N, grid, packScale,
                     (uint3 *)packHalf.ptr);
  UH_CHECK(hipGetLastError());
  packValid = true;
  return 0;
}

int CellList::Profile::next(hipEvent_t *start, hipEvent_t *stop) {
  if (live == kRing) { if (int e = collect(kRing - 1)) return e; }
  const int slot = head % kRing;
  for (int k = 0; k < 2; ++k)
    if (!ev[slot][k]) UH_CHECK(hipEventCreate(&ev[slot][k]));
  *start = ev[slot][0];
  *stop = ev[slot][1];
  head++;
  live++;
  return 0;
}
int CellList::Profile::collect(int upTo) {
  while (live > upTo) {
    const int slot = (head - live) % kRing;
    UH_CHECK(hipEventSynchronize(ev[slot][1]));
    float ms = 0.f;
    UH_CHECK(hipEventElapsedTime(&ms, ev[slot][0], ev[slot][1]));
    totalMs += ms;
    launches++;
    live--;
  }
  return 0;
}
CellList::Profile::~Profile() {
  for (auto &p : ev)
    for (auto &e : p)
      if (e) (void)hipEventDestroy(e);
}

CellList::~CellList() {
  if (hostErr) (void)hipHostFree(hostErr);
}

int CellList::check_errors(hipStream_t st, bool sync) {
  if (!hostErr || !reportErrors) return 0;
  if (sync) UH_CHECK(hipStreamSynchronize(st));
  if (const int code = __atomic_load_n(hostErr, __ATOMIC_ACQUIRE)) {
    __atomic_store_n(hostErr, 0, __ATOMIC_RELEASE);
    if (code == 2) {
      ljTable = nullptr;  // the cached cut-off was stale: the next traversal reads the table again
      set_last_error("PairForces: the LJ parameter table was rewritten in place behind the list's cached copy (a cut-off larger than the "
                     "cell edge, or sigma / epsilon no longer 1 where the reduced-units kernel was chosen); the last traversal did not "
                     "compute forces — the table is read again on the next call (rebuild the list if the cut-off grew)");
      return -4;
    }
    if (code == 4) {
      set_last_error("CellList: the key scan gave up waiting for a preceding workgroup; the last list is incomplete");
      return -4;
    }
    set_last_error("CellList encountered NaN positions or particles outside a non-periodic box");  // CellListBase.cuh:262
    return -4;
  }
  return 0;
}

// A list keeps what it needs of the pair-parameter table (largest cut-off, reduced units or not) keyed by the table's address.  A table
// rewritten IN PLACE (Potential::LJ::setPotParameters between steps re-uploads into the same buffer; an allocator hands a freed address
// out again) is announced through uammd_lj_table_changed(): a process-wide count every list compares with the one it read at.
static std::atomic<unsigned> g_ljTableEpoch{1};
extern "C" int uammd_lj_table_changed(void) { g_ljTableEpoch.fetch_add(1); return 0; }

int CellList::lj_max_cutoff2(const void *d_table, int ntypes, hipStream_t st, float *out) {
  const unsigned epoch = g_ljTableEpoch.load();
  if (d_table != ljTable || ntypes != ljTableTypes || epoch != ljTableEpoch) {
    ljTableEpoch = epoch;
    std::vector<uammd_lj_pair_parameters> host((size_t)ntypes * (size_t)ntypes);
    UH_CHECK(hipMemcpyAsync(host.data(), d_table, host.size() * sizeof(host[0]), hipMemcpyDeviceToHost, st));
    UH_CHECK(hipStreamSynchronize(st));
    float m = 0.f;
    for (const auto &p : host) m = p.cutOff2 > m ? p.cutOff2 : m;
    ljTable = d_table; ljTableTypes = ntypes; ljTableMaxCut2 = m;
    ljTableUnit = ntypes == 1 && host[0].sigma2 == 1.0f && host[0].epsilonDivSigma2 == 1.0f;
  }
  *out = ljTableMaxCut2;
  return 0;
}

// gj (nullable): the fused MD step asks for GronbechJensen's first half step to be applied BEFORE the list is built — inside the hash
// kernel where the build takes the aggregated counting path, by the integrator's own kernel otherwise (gjDone says which happened).
int CellList::update(const float4 *d_pos, int numberParticles, const float L[3], const int periodic[3],
                     const int cellDim_[3], hipStream_t st, const GJFuse *gj) {
  gjDone = false;
  gjInHash = false;
  if (numberParticles < 0 || cellDim_[0] <= 0 || cellDim_[1] <= 0 || cellDim_[2] < 0) {
    set_last_error("CellList encountered an invalid grid and/or cutoff (N=%d cellDim=%d %d %d)", numberParticles,
                   cellDim_[0], cellDim_[1], cellDim_[2]);
    return -2;
  }
  if (!hostErr) {
    UH_CHECK(hipHostMalloc((void **)&hostErr, 64, hipHostMallocMapped));
    hostErr[0] = 0;
    UH_CHECK(hipHostGetDevicePointer((void **)&devErr, hostErr, 0));
  }
  if (int e = check_errors(st, false)) return e;  // raised by an earlier build
  if (buildStreamSet && buildStream != st) {  // a list's buffers are ordered by the stream it is used on: changing it drains the old one
    UH_CHECK(hipStreamSynchronize(buildStream));
    zeroBlockClean = false;
  }
  buildStream = st;
  buildStreamSet = true;
  const BoxT<float> box = make_box<float>(L, periodic);
  grid = make_grid<float>(box, make_int3(cellDim_[0], cellDim_[1], cellDim_[2]));
  for (int k = 0; k < 3; ++k) { boxL[k] = L[k]; boxPeriodic[k] = periodic[k] != 0; }
  const int N = numberParticles;
  const long long ncells = (long long)grid.cellDim.x * grid.cellDim.y * grid.cellDim.z;
  if (ncells > (1ll << 30)) {
    set_last_error("CellList: too many cells (%lld)", ncells);
    return -2;
  }
  bool needsClear = false;
  const bool cellsResized = (long long)nCellsAlloc != ncells;
  next_valid_cell(N, &needsClear);
  if (int e = cellStart.reserve(sizeof(uint) * (size_t)ncells)) return e;
  if (int e = cellEnd.reserve(sizeof(int) * (size_t)ncells)) return e;
  if (cellsResized || needsClear) UH_CHECK(hipMemsetAsync(cellStart.ptr, 0, sizeof(uint) * (size_t)ncells, st));
  nCellsAlloc = (int)ncells;
  if (int e = hash.reserve(sizeof(uint) * (size_t)(N + 1))) return e;
  if (int e = sortHash.reserve(sizeof(uint) * (size_t)(N + 1))) return e;
  if (int e = index.reserve(sizeof(int) * (size_t)(N + 1))) return e;
  // +4: the traversal kernels read candidates in groups of four from one base address (reads past a cell are masked)
  if (int e = sortPos.reserve(sizeof(float4) * (size_t)(N + 8))) return e;
  numberParticlesBuilt = N;
  packValid = false;

  const uint maxHash = morton_hash(make_int3(grid.cellDim.x - 1, grid.cellDim.y - 1, grid.cellDim.z - 1));
  endBit = sort_end_bit(maxHash);
  nKeys = (endBit >= 31) ? 0u : (1u << endBit);  // number of distinct keys the grid can produce (pow2 envelope)
  const bool tabulated = nKeys != 0 && nKeys <= (1u << 27);
  // Counting-sort build when the key table is comparable to the particle count.
  const bool counting = forceRadix ? false : (nKeys != 0 && (unsigned long long)nKeys <= 8ull * (unsigned long long)N + 4096ull);
  usedCounting = counting;
  if (gj && !(counting && aggregateHash && N > 0) && (gj->keys || gj->skip || gj->nRows > 0)) {  // the decomposed step's form of the same fallback
    const int nRows = gj->nRows > 0 ? std::min(gj->nRows, N) : N;
    if (nRows > 0) {
      hipLaunchKernelGGL(k_gj1_rows, dim3(nblocks(nRows)), dim3(kBlock), 0, st, const_cast<float4 *>(d_pos), nRows, *gj);
      UH_CHECK(hipGetLastError());
    }
    gjDone = true;
    gj = nullptr;
  }
  if (gj && !(counting && aggregateHash && N > 0)) {  // only the aggregated counting build carries the half step inside its hash kernel
    if (N > 0)
      if (int e = uammd_verletnvt_gj(1, (float *)const_cast<float4 *>(d_pos), gj->vel, (float *)gj->force, gj->mass, gj->defaultMass, nullptr,
                                     N, gj->dt, gj->friction, gj->is2D, gj->noiseAmplitude, gj->stepNum, gj->seed, (void *)st))
        return e;
    gjDone = true;
    gj = nullptr;
  }
  {  // everything that must start at zero sits in one block: error flag, per-key "outside" flags, per-key counters
    const size_t errB = 16, koB = tabulated ? (((size_t)nKeys + 16 + 15) & ~(size_t)15) : 0,
                 kcB = counting ? sizeof(uint) * ((size_t)nKeys + 4) : 0;
    if (int e = zeroBlock.reserve(errB + koB + kcB)) return e;
    char *base = (char *)zeroBlock.ptr;
    errorFlag.alias(base, errB);
    keyOutside.alias(tabulated ? base + errB : nullptr, koB);
    keyCount.alias(counting ? base + errB + koB : nullptr, kcB);
    if (N == 0) { UH_CHECK(hipMemsetAsync(base, 0, errB, st)); zeroBlockClean = false; return 0; }
    // the counting build hands the block back zeroed (k_rank_scatter2); anything else — first use, another layout, a radix build, a
    // build cut short by an error — clears it here
    const bool clean = zeroBlockClean && base == zeroBase && zeroLayout[0] == koB && zeroLayout[1] == kcB && counting;
    if (!clean) UH_CHECK(hipMemsetAsync(base, 0, errB + koB + kcB, st));
    zeroBlockClean = false;
    zeroBase = base;
    zeroLayout[0] = koB;
    zeroLayout[1] = kcB;
  }
  if (counting) {
    if (int e = keyStart.reserve(sizeof(uint) * ((size_t)nKeys + 2))) return e;
    if (int e = provRank.reserve(sizeof(uint) * (size_t)N)) return e;
    if (int e = members.reserve(sizeof(int) * (size_t)N)) return e;
    if (aggregateHash && gj) {
      hipLaunchKernelGGL((k_hash_agg<true, 1>), dim3((N + kBlock - 1) / kBlock), dim3(kBlock), 0, st,
                         const_cast<float4 *>(d_pos), N, grid, (uint *)hash.ptr, (uint *)keyCount.ptr, (uint *)provRank.ptr, devErr,
                         (unsigned char *)keyOutside.ptr, *gj);
      gjDone = true;
      gjInHash = true;
    } else if (aggregateHash)
      hipLaunchKernelGGL((k_hash_agg<false, 4>), dim3((N + kBlock * 4 - 1) / (kBlock * 4)), dim3(kBlock), 0, st,
                         const_cast<float4 *>(d_pos), N, grid, (uint *)hash.ptr, (uint *)keyCount.ptr, (uint *)provRank.ptr, devErr,
                         (unsigned char *)keyOutside.ptr, GJFuse{});
    else
      hipLaunchKernelGGL(k_hash<true>, dim3(nblocks(N)), dim3(kBlock), 0, st, d_pos, N, grid, (uint *)hash.ptr,
                         (int *)nullptr, (uint *)keyCount.ptr, (uint *)provRank.ptr, devErr,
                         (unsigned char *)keyOutside.ptr);
    if (int e = cellOutside.reserve((size_t)ncells + 16)) return e;
    if (int e = cellRange.reserve(sizeof(uint2) * ((size_t)ncells + 1))) return e;
    {
      const int chunkShift = endBit > 18 ? endBit - 8 : 10;  // <= kScanChunks chunks of >= 1024 keys
      const int nChunks = (int)((nKeys + (1u << chunkShift) - 1) >> chunkShift);
      if (!scanFlags.ptr) {
        if (int e = scanFlags.reserve(sizeof(unsigned long long) * kScanChunks)) return e;
        UH_CHECK(hipMemsetAsync(scanFlags.ptr, 0, sizeof(unsigned long long) * kScanChunks, st));
        scanGeneration = 0;
      }
      if (++scanGeneration == 0u) scanGeneration = 1u;  // (0 is the cleared state)
      hipLaunchKernelGGL(k_key_scan, dim3(nChunks), dim3(kBlock), 0, st, (const uint *)keyCount.ptr, nKeys, chunkShift,
                         (unsigned long long *)scanFlags.ptr, scanGeneration, (uint *)keyStart.ptr, devErr);
    }
    hipLaunchKernelGGL(k_members, dim3(nblocks(N)), dim3(kBlock), 0, st, (const uint *)hash.ptr, (const uint *)provRank.ptr,
                       (const uint *)keyStart.ptr, N, (int *)members.ptr, (uint *)sortHash.ptr);
    const int pb = nblocks(N);
    hipLaunchKernelGGL(k_rank_scatter2, dim3(pb + nblocks(ncells)), dim3(kBlock), 0, st, d_pos, (const uint *)sortHash.ptr,
                       (const uint *)keyStart.ptr, (const int *)members.ptr, N, pb, (int *)index.ptr, (float4 *)sortPos.ptr,
                       (unsigned char *)keyOutside.ptr, (uint *)keyCount.ptr, grid.cellDim, validCell, (uint *)cellStart.ptr,
                       (int *)cellEnd.ptr, (unsigned char *)cellOutside.ptr, (uint2 *)cellRange.ptr);
    zeroBlockClean = true;
    haveCellOutside = true;
  } else {
    if (int e = indexAlt.reserve(sizeof(int) * (size_t)N)) return e;
    hipLaunchKernelGGL(k_hash<false>, dim3(nblocks(N)), dim3(kBlock), 0, st, d_pos, N, grid, (uint *)hash.ptr,
                       (int *)indexAlt.ptr, (uint *)nullptr, (uint *)nullptr, devErr,
                       tabulated ? (unsigned char *)keyOutside.ptr : (unsigned char *)nullptr);
    if (endBit > 0) {
      size_t tmpBytes = 0;
      UH_CHECK(rocprim::radix_sort_pairs(nullptr, tmpBytes, (uint *)hash.ptr, (uint *)sortHash.ptr,
                                         (int *)indexAlt.ptr, (int *)index.ptr, (size_t)N, 0u, (unsigned)endBit, st));
      if (int e = scratch.reserve(tmpBytes)) return e;
      UH_CHECK(rocprim::radix_sort_pairs(scratch.ptr, tmpBytes, (uint *)hash.ptr, (uint *)sortHash.ptr,
                                         (int *)indexAlt.ptr, (int *)index.ptr, (size_t)N, 0u, (unsigned)endBit, st));
    } else {
      UH_CHECK(hipMemcpyAsync(sortHash.ptr, hash.ptr, sizeof(uint) * (size_t)N, hipMemcpyDeviceToDevice, st));
      UH_CHECK(hipMemcpyAsync(index.ptr, indexAlt.ptr, sizeof(int) * (size_t)N, hipMemcpyDeviceToDevice, st));
    }
    hipLaunchKernelGGL(k_reorder_fill, dim3(nblocks(N)), dim3(kBlock), 0, st, d_pos, (const int *)index.ptr, N, grid,
                       validCell, (float4 *)sortPos.ptr, (uint *)cellStart.ptr, (int *)cellEnd.ptr,
                       devErr);
    if (tabulated) {
      if (int e = keyStart.reserve(sizeof(uint) * ((size_t)nKeys + 2))) return e;
      hipLaunchKernelGGL(k_key_start_from_sorted, dim3(nblocks(N + 1)), dim3(kBlock), 0, st,
                         (const uint *)sortHash.ptr, N, nKeys, (uint *)keyStart.ptr);
      haveKeyStart = true;
      if (int e = cellOutside.reserve((size_t)ncells + 16)) return e;
      if (int e = cellRange.reserve(sizeof(uint2) * ((size_t)ncells + 1))) return e;
      hipLaunchKernelGGL(k_cell_tables, dim3(nblocks(ncells)), dim3(kBlock), 0, st, (const uint *)keyStart.ptr,
                         (const unsigned char *)keyOutside.ptr, grid.cellDim, validCell, (uint *)nullptr,
                         (int *)nullptr, (unsigned char *)cellOutside.ptr, (uint2 *)cellRange.ptr);
      haveCellOutside = true;
    } else {
      haveKeyStart = false;
      haveCellOutside = false;
    }
  }
  if (counting) haveKeyStart = true;
  UH_CHECK(hipGetLastError());
  if (strictErrors) return check_errors(st, true);
  return 0;
}

}  // namespace uammd_hip

using namespace uammd_hip;

// ---- C ABI -----------------------------------------------------------------------------------------
extern "C" {

int uammd_hip_abi_version(void) { return UAMMD_HIP_ABI_VERSION; }
const char *uammd_hip_last_error(void) { return g_last_error; }

int uammd_hip_device_count(int *count) {
  UH_CHECK(hipGetDeviceCount(count));
  return 0;
}
int uammd_hip_set_device(int device) {
  UH_CHECK(hipSetDevice(device));
  return 0;
}

int uammd_celllist_create(uammd_celllist **out) {
  if (!out) { set_last_error("uammd_celllist_create: null output"); return -1; }
  *out = reinterpret_cast<uammd_celllist *>(new CellList());
  return 0;
}
int uammd_celllist_destroy(uammd_celllist *h) {
  delete reinterpret_cast<CellList *>(h);
  return 0;
}

int uammd_celllist_create_grid(const float L_in[3], const int periodic_in[3], const float cutOff[3],
                               int cellDim_out[3], float L_out[3], int periodic_out[3]) {
  // Behaviour of CellList::createUpdateGrid (Interactor/NeighbourList/CellList.cuh:100-126)
  const float inf = std::numeric_limits<float>::max();
  for (int k = 0; k < 3; ++k) {
    float L = L_in[k];
    const bool inputPeriodic = periodic_in[k] && !(L == 0.0f || std::isinf(L));
    if (L >= inf) L = 64 * cutOff[k];
    L_out[k] = L;
    periodic_out[k] = (inputPeriodic && L < inf) ? 1 : 0;
    int cd = (int)(L / cutOff[k]);  // Grid(Box, real3 minCellSize): C truncation in float
    if (k == 2 && cd == 0) cd = 1;
    if (cd <= 3) cd = 1;
    cellDim_out[k] = cd;
  }
  return 0;
}

int uammd_celllist_update(uammd_celllist *h, const float *d_pos, int numberParticles, const float L[3],
                          const int periodic[3], const int cellDim[3], void *stream) {
  if (!h) { set_last_error("uammd_celllist_update: null handle"); return -1; }
  return reinterpret_cast<CellList *>(h)->update(reinterpret_cast<const float4 *>(d_pos), numberParticles, L,
                                                 periodic, cellDim, (hipStream_t)stream);
}

int uammd_celllist_set_option(uammd_celllist *h, const char *name, int value) {
  if (!h || !name) { set_last_error("uammd_celllist_set_option: null argument"); return -1; }
  CellList *cl = reinterpret_cast<CellList *>(h);
  if (std::string(name) == "force_radix") { cl->forceRadix = value != 0; return 0; }
  if (std::string(name) == "aggregate_hash") { cl->aggregateHash = value != 0; return 0; }
  if (std::string(name) == "strict_errors") { cl->strictErrors = value != 0; return 0; }
  if (std::string(name) == "report_errors") { cl->reportErrors = value != 0; return 0; }
  if (std::string(name) == "num_owned") { cl->numOwned = value < 0 ? 0x7fffffff : value; return 0; }
  set_last_error("uammd_celllist_set_option: unknown option %s", name);
  return -1;
}

int uammd_celllist_get(uammd_celllist *h, uammd_celllist_data *out) {
  if (!h || !out) { set_last_error("uammd_celllist_get: null argument"); return -1; }
  CellList *cl = reinterpret_cast<CellList *>(h);
  if (int e = cl->check_errors(nullptr, false)) return e;
  out->d_cellStart = (const unsigned int *)cl->cellStart.ptr;
  out->d_cellEnd = (const int *)cl->cellEnd.ptr;
  out->d_sortPos = (const float *)cl->sortPos.ptr;
  out->d_groupIndex = (const int *)cl->index.ptr;
  out->d_sortHash = (const unsigned int *)cl->sortHash.ptr;
  out->cellDim[0] = cl->grid.cellDim.x; out->cellDim[1] = cl->grid.cellDim.y; out->cellDim[2] = cl->grid.cellDim.z;
  for (int k = 0; k < 3; ++k) { out->boxSize[k] = cl->boxL[k]; out->periodic[k] = cl->boxPeriodic[k]; }
  out->VALID_CELL = cl->validCell;
  out->numberParticles = cl->numberParticlesBuilt;
  return 0;
}

int uammd_lj_profile_enable(uammd_celllist *h, int enable) {
  if (!h) { set_last_error("uammd_lj_profile_enable: null handle"); return -1; }
  CellList *cl = reinterpret_cast<CellList *>(h);
  if (int e = cl->prof.collect(0)) return e;
  cl->prof.enabled = enable != 0;
  cl->prof.totalMs = 0.0;
  cl->prof.launches = 0;
  return 0;
}
int uammd_lj_profile_read(uammd_celllist *h, double *totalMs, long long *launches) {
  if (!h || !totalMs || !launches) { set_last_error("uammd_lj_profile_read: null argument"); return -1; }
  CellList *cl = reinterpret_cast<CellList *>(h);
  if (int e = cl->prof.collect(0)) return e;  // waits for the launches still in flight
  *totalMs = cl->prof.totalMs;
  *launches = cl->prof.launches;
  return 0;
}

int uammd_lj_tile_stats(uammd_celllist *h, int enable, unsigned int out[16], void *stream) {
  if (!h) { set_last_error("uammd_lj_tile_stats: null handle"); return -1; }
  CellList *cl = reinterpret_cast<CellList *>(h);
  hipStream_t st = (hipStream_t)stream;
  if (out) {
    for (int k = 0; k < 16; ++k) out[k] = 0u;
    if (cl->tileStatsOn) {
      UH_CHECK(hipMemcpyAsync(out, cl->tileStats.ptr, 16 * sizeof(uint), hipMemcpyDeviceToHost, st));
      UH_CHECK(hipStreamSynchronize(st));
    }
  }
  if (enable) {
    if (int e = cl->tileStats.reserve(16 * sizeof(uint))) return e;
    UH_CHECK(hipMemsetAsync(cl->tileStats.ptr, 0, 16 * sizeof(uint), st));
  }
  cl->tileStatsOn = enable != 0;
  return 0;
}

int uammd_celllist_check_errors(uammd_celllist *h, void *stream) {
  if (!h) { set_last_error("uammd_celllist_check_errors: null handle"); return -1; }
  return reinterpret_cast<CellList *>(h)->check_errors((hipStream_t)stream, true);
}

int uammd_sort_pairs(unsigned int *d_keys, int *d_values, int n, int end_bit, void *stream) {
  if (n <= 1 || end_bit <= 0) return 0;
  if (end_bit > 32) end_bit = 32;
  hipStream_t st = (hipStream_t)stream;
  // Stable LSD radix sort on [0,end_bit) — the contract of the SortPairs call at
  // utils/ParticleSorter.cuh:316-320.  Temporary storage is per call (stream ordered).
  uint *keysAlt = nullptr;
  int *valsAlt = nullptr;
  void *tmp = nullptr;
  UH_CHECK(hipMallocAsync((void **)&keysAlt, sizeof(uint) * (size_t)n, st));
  UH_CHECK(hipMallocAsync((void **)&valsAlt, sizeof(int) * (size_t)n, st));
  size_t tmpBytes = 0;
  UH_CHECK(rocprim::radix_sort_pairs(nullptr, tmpBytes, d_keys, keysAlt, d_values, valsAlt, (size_t)n, 0u,
                                     (unsigned)end_bit, st));
  UH_CHECK(hipMallocAsync(&tmp, tmpBytes ? tmpBytes : 16, st));
  UH_CHECK(rocprim::radix_sort_pairs(tmp, tmpBytes, d_keys, keysAlt, d_values, valsAlt, (size_t)n, 0u,
                                     (unsigned)end_bit, st));
  UH_CHECK(hipMemcpyAsync(d_keys, keysAlt, sizeof(uint) * (size_t)n, hipMemcpyDeviceToDevice, st));
  UH_CHECK(hipMemcpyAsync(d_values, valsAlt, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, st));
  UH_CHECK(hipFreeAsync(tmp, st));
  UH_CHECK(hipFreeAsync(keysAlt, st));
  UH_CHECK(hipFreeAsync(valsAlt, st));
  return 0;
}

int uammd_gather(const void *d_in, const int *d_index, void *d_out, int n, int elem_bytes, void *stream) {
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g(nblocks(n)), b(kBlock);
  switch (elem_bytes) {
    case 4: hipLaunchKernelGGL(k_gather<uint>, g, b, 0, st, (const uint *)d_in, d_index, (uint *)d_out, n); break;
    case 8: hipLaunchKernelGGL(k_gather<uint2>, g, b, 0, st, (const uint2 *)d_in, d_index, (uint2 *)d_out, n); break;
    case 12: hipLaunchKernelGGL(k_gather<Elem<12>>, g, b, 0, st, (const Elem<12> *)d_in, d_index, (Elem<12> *)d_out, n); break;
    case 16: hipLaunchKernelGGL(k_gather<uint4>, g, b, 0, st, (const uint4 *)d_in, d_index, (uint4 *)d_out, n); break;
    case 24: hipLaunchKernelGGL(k_gather<Elem24>, g, b, 0, st, (const Elem24 *)d_in, d_index, (Elem24 *)d_out, n); break;   // real3, real = double
    case 32: hipLaunchKernelGGL(k_gather<Elem32>, g, b, 0, st, (const Elem32 *)d_in, d_index, (Elem32 *)d_out, n); break;   // real4, real = double
    default: set_last_error("uammd_gather: unsupported element size %d", elem_bytes); return -1;
  }
  UH_CHECK(hipGetLastError());
  return 0;
}

int uammd_scatter(const void *d_in, const int *d_index, void *d_out, int n, int elem_bytes, void *stream) {
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g(nblocks(n)), b(kBlock);
  switch (elem_bytes) {
    case 4: hipLaunchKernelGGL(k_scatter<uint>, g, b, 0, st, (const uint *)d_in, d_index, (uint *)d_out, n); break;
    case 8: hipLaunchKernelGGL(k_scatter<uint2>, g, b, 0, st, (const uint2 *)d_in, d_index, (uint2 *)d_out, n); break;
    case 12: hipLaunchKernelGGL(k_scatter<Elem<12>>, g, b, 0, st, (const Elem<12> *)d_in, d_index, (Elem<12> *)d_out, n); break;
    case 16: hipLaunchKernelGGL(k_scatter<uint4>, g, b, 0, st, (const uint4 *)d_in, d_index, (uint4 *)d_out, n); break;
    case 24: hipLaunchKernelGGL(k_scatter<Elem24>, g, b, 0, st, (const Elem24 *)d_in, d_index, (Elem24 *)d_out, n); break;   // real3, real = double
    case 32: hipLaunchKernelGGL(k_scatter<Elem32>, g, b, 0, st, (const Elem32 *)d_in, d_index, (Elem32 *)d_out, n); break;   // real4, real = double
    default: set_last_error("uammd_scatter: unsupported element size %d", elem_bytes); return -1;
  }
  UH_CHECK(hipGetLastError());
  return 0;
}

// ---- slab decomposition: halo packing -------------------------------------------------------------------------------------
// out[k] = pos[idx[k]] with z shifted into the receiver's frame, both faces in one launch (new design, SURVEY 8e: the reference
// is single GPU).  Replaces two index_select + two in-place adds of the Python glue.
__global__ void __launch_bounds__(kBlock) k_halo_pack(const float4 *__restrict__ pos, const int *__restrict__ idxUp, int nUp,
                                                      const int *__restrict__ idxDown, int nDown, float dzUp, float dzDown,
                                                      float4 *__restrict__ outUp, float4 *__restrict__ outDown) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t < nUp) {
    float4 p = pos[idxUp[t]];
    p.z += dzUp;
    outUp[t] = p;
  } else if (t < nUp + nDown) {
    float4 p = pos[idxDown[t - nUp]];
    p.z += dzDown;
    outDown[t - nUp] = p;
  }
}

int uammd_halo_pack(const float *d_pos, const int *d_idxUp, int nUp, const int *d_idxDown, int nDown, float dzUp, float dzDown,
                    float *d_outUp, float *d_outDown, void *stream) {
  if (nUp < 0 || nDown < 0) { set_last_error("uammd_halo_pack: negative count"); return -1; }
  if (nUp + nDown == 0) return 0;
  hipLaunchKernelGGL(k_halo_pack, dim3(nblocks(nUp + nDown)), dim3(kBlock), 0, (hipStream_t)stream, (const float4 *)d_pos, d_idxUp, nUp,
                     d_idxDown, nDown, dzUp, dzDown, (float4 *)d_outUp, (float4 *)d_outDown);
  UH_CHECK(hipGetLastError());
  return 0;
}

// uammd_halo_pack that first applies GronbechJensen's first half step to the rows it packs (each listed row once: the caller's up and
// down lists are disjoint — a slab wider than two reaches) — the decomposed step then needs no integrator launch of its own before the
// exchange: the unlisted rows take their half step inside the list build's hash kernel (uammd_celllist_update_gj1 with these rows masked)
__global__ void __launch_bounds__(kBlock) k_halo_pack_gj1(float4 *__restrict__ pos, const int *__restrict__ idxUp, int nUp,
                                                          const int *__restrict__ idxDown, int nDown, float dzUp, float dzDown,
                                                          float4 *__restrict__ outUp, float4 *__restrict__ outDown, GJFuse gj) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= nUp + nDown) return;
  const bool up = t < nUp;
  const int i = up ? idxUp[t] : idxDown[t - nUp];
  float4 p = pos[i];
  float3 v = make_float3(gj.vel[3 * (size_t)i], gj.vel[3 * (size_t)i + 1], gj.vel[3 * (size_t)i + 2]);
  const float4 f4 = gj.force[i];
  const float invMass = 1.0f / (gj.defaultMass > 0 ? gj.defaultMass : gj.mass[i]);
  gj_step1(p, v, f4, invMass, gj.dt, gj.friction, gj.noiseAmplitude, gj.is2D, gj.keys ? (uint)gj.keys[i] : (uint)i, gj.stepNum, gj.seed);
  pos[i] = p;
  gj.vel[3 * (size_t)i] = v.x; gj.vel[3 * (size_t)i + 1] = v.y; gj.vel[3 * (size_t)i + 2] = v.z;
  gj.force[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  p.z += up ? dzUp : dzDown;
  if (up) outUp[t] = p; else outDown[t - nUp] = p;
}

int uammd_halo_pack_gj1(float *d_pos, float *d_vel, float *d_force, const float *d_mass, float defaultMass, const int *d_keys,
                        const int *d_idxUp, int nUp, const int *d_idxDown, int nDown, float dzUp, float dzDown, float *d_outUp,
                        float *d_outDown, float dt, float friction, int is2D, float noiseAmplitude, unsigned int stepNum, unsigned int seed,
                        void *stream) {
  if (nUp < 0 || nDown < 0 || !d_pos || !d_vel || !d_force) { set_last_error("uammd_halo_pack_gj1: bad arguments"); return -1; }
  if (!d_mass && !(defaultMass > 0)) { set_last_error("uammd_halo_pack_gj1: no mass array and defaultMass <= 0"); return -1; }
  if (nUp + nDown == 0) return 0;
  GJFuse gj{d_vel, (float4 *)d_force, d_mass, defaultMass, dt, friction, noiseAmplitude, is2D, stepNum, seed, 0, d_keys, nullptr, 0};
  hipLaunchKernelGGL(k_halo_pack_gj1, dim3(nblocks(nUp + nDown)), dim3(kBlock), 0, (hipStream_t)stream, (float4 *)d_pos, d_idxUp, nUp,
                     d_idxDown, nDown, dzUp, dzDown, (float4 *)d_outUp, (float4 *)d_outDown, gj);
  UH_CHECK(hipGetLastError());
  return 0;
}

// uammd_celllist_update on owned + ghost rows with GronbechJensen's first half step of the owned, unmasked rows applied as they are
// loaded (k_hash_agg<GJ1>; where the build takes another path, one plain launch first): the new positions are what is hashed.
int uammd_celllist_update_gj1(uammd_celllist *hh, float *d_pos, int numberParticles, const float L[3], const int periodic[3],
                              const int cellDim[3], float *d_vel, float *d_force, const float *d_mass, float defaultMass,
                              const int *d_keys, const unsigned char *d_skip, int numberOwned, float dt, float friction, int is2D,
                              float noiseAmplitude, unsigned int stepNum, unsigned int seed, void *stream) {
  if (!hh || !d_pos || !d_vel || !d_force || numberOwned < 0 || numberOwned > numberParticles) {
    set_last_error("uammd_celllist_update_gj1: bad arguments");
    return -1;
  }
  if (!d_mass && !(defaultMass > 0)) { set_last_error("uammd_celllist_update_gj1: no mass array and defaultMass <= 0"); return -1; }
  CellList *h = reinterpret_cast<CellList *>(hh);
  if (numberOwned == 0) return h->update((const float4 *)d_pos, numberParticles, L, periodic, cellDim, (hipStream_t)stream, nullptr);
  const GJFuse gj{d_vel, (float4 *)d_force, d_mass, defaultMass, dt, friction, noiseAmplitude, is2D, stepNum, seed, 0, d_keys, d_skip, numberOwned};
  return h->update((const float4 *)d_pos, numberParticles, L, periodic, cellDim, (hipStream_t)stream, &gj);
}

int uammd_fill_zero(void *d_ptr, size_t bytes, void *stream) {
  UH_CHECK(hipMemsetAsync(d_ptr, 0, bytes, (hipStream_t)stream));
  return 0;
}

// positions of a DOUBLE_PRECISION ParticleData rounded to single precision: the keys of ParticleData::sortParticles (a memory-locality
// order, not a result) come from the single-precision cell list
int uammd_convert_f64_to_f32(const double *d_in, float *d_out, size_t count, void *stream) {
  if (count == 0) return 0;
  if (!d_in || !d_out) { set_last_error("uammd_convert_f64_to_f32: null argument"); return -1; }
  hipLaunchKernelGGL(k_f64_to_f32, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, d_in, d_out, count);
  UH_CHECK(hipGetLastError());
  return 0;
}

int uammd_fill_zero_indexed(void *d_ptr, const int *d_index, int n, int elem_bytes, void *stream) {
  if (n <= 0) return 0;
  if (!d_ptr || !d_index || elem_bytes % 4 != 0 || elem_bytes <= 0) { set_last_error("uammd_fill_zero_indexed: bad arguments"); return -1; }
  const int words = elem_bytes / 4;
  hipLaunchKernelGGL(k_zero_indexed, dim3(nblocks(n * words)), dim3(kBlock), 0, (hipStream_t)stream, (uint *)d_ptr, d_index, n, words);
  UH_CHECK(hipGetLastError());
  return 0;
}

}  // extern "C"
