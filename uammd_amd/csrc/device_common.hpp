// Value types shared by every gfx950 kernel of libuammd_hip: real3/real4, Box, Grid.
//
// These mirror the *semantics* of UAMMD's utils/Box.cuh:16-58 and utils/Grid.cuh:21-131
// (minimum-image convention by floor(r*(-1/L)+0.5), cell of a position by C truncation with the
// cell==cellDim -> 0 guard) because bit-exact cell/neighbour indexing is part of the contract.
//
// Floating-point contract: the whole library is compiled with -ffp-contract=off and every fused
// multiply-add is spelled fmaf()/fma() explicitly, at the same places as the CPU oracle
// (oracle/src/common.h).  Division and sqrt are IEEE correctly rounded (hipcc default).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>

namespace uammd_hip {

typedef unsigned int uint;

#define UH_HD __host__ __device__ __forceinline__
#define UH_D __device__ __forceinline__

template <class T> struct vec3 { T x, y, z; };
using real3f = vec3<float>;
using real3d = vec3<double>;

UH_HD float fma_(float a, float b, float c) { return fmaf(a, b, c); }
UH_HD double fma_(double a, double b, double c) { return fma(a, b, c); }
UH_HD float floor_(float a) { return floorf(a); }
UH_HD double floor_(double a) { return floor(a); }
UH_HD float sqrt_(float a) { return sqrtf(a); }
UH_HD double sqrt_(double a) { return sqrt(a); }

template <class T> UH_HD T dot3(vec3<T> a, vec3<T> b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }

// ---- Box (reference: utils/Box.cuh:16-58) ------------------------------------------------------
template <class T> struct BoxT {
  vec3<T> boxSize, minusInvBoxSize;
  UH_HD bool px() const { return minusInvBoxSize.x != T(0); }
  UH_HD bool py() const { return minusInvBoxSize.y != T(0); }
  UH_HD bool pz() const { return minusInvBoxSize.z != T(0); }
  UH_HD vec3<T> apply_pbc(vec3<T> r) const {
    const T ox = floor_(fma_(r.x, minusInvBoxSize.x, T(0.5)));
    const T oy = floor_(fma_(r.y, minusInvBoxSize.y, T(0.5)));
    const T oz = floor_(fma_(r.z, minusInvBoxSize.z, T(0.5)));
    const T tx = ox * boxSize.x, ty = oy * boxSize.y, tz = oz * boxSize.z;
    r.x += px() ? tx : T(0);
    r.y += py() ? ty : T(0);
    r.z += pz() ? tz : T(0);
    return r;
  }
};

template <class T> inline BoxT<T> make_box(const T *L, const int *periodic) {
  BoxT<T> b;
  b.boxSize = {L[0], L[1], L[2]};
  b.minusInvBoxSize = {T(-1.0) / L[0], T(-1.0) / L[1], T(-1.0) / L[2]};
  if (L[0] == T(0) || std::isinf(L[0]) || !periodic[0]) b.minusInvBoxSize.x = T(0);
  if (L[1] == T(0) || std::isinf(L[1]) || !periodic[1]) b.minusInvBoxSize.y = T(0);
  if (L[2] == T(0) || std::isinf(L[2]) || !periodic[2]) b.minusInvBoxSize.z = T(0);
  return b;
}

// ---- Grid (reference: utils/Grid.cuh:21-131) ---------------------------------------------------
template <class T> struct GridT {
  int3 cellDim;
  vec3<T> cellSize, invCellSize;
  BoxT<T> box;
  T cellVolume;

  UH_HD int3 getCell(vec3<T> r) const {
    const vec3<T> p = box.apply_pbc(r);
    int3 c;
    c.x = (int)((p.x + T(0.5) * box.boxSize.x) * invCellSize.x);
    c.y = (int)((p.y + T(0.5) * box.boxSize.y) * invCellSize.y);
    c.z = (int)((p.z + T(0.5) * box.boxSize.z) * invCellSize.z);
    if (c.x == cellDim.x) c.x = 0;
    if (c.y == cellDim.y) c.y = 0;
    if (c.z == cellDim.z) c.z = 0;
    return c;
  }
  UH_HD int getCellIndex(int3 c) const { return c.x + cellDim.x * (c.y + cellDim.y * c.z); }
  UH_HD int pbc_x(int c) const { const int n = box.px() ? cellDim.x : 0; return c <= -1 ? c + n : (c >= n ? c - n : c); }
  UH_HD int pbc_y(int c) const { const int n = box.py() ? cellDim.y : 0; return c <= -1 ? c + n : (c >= n ? c - n : c); }
  UH_HD int pbc_z(int c) const { const int n = box.pz() ? cellDim.z : 0; return c <= -1 ? c + n : (c >= n ? c - n : c); }
  UH_HD int getNumberCells() const { return cellDim.x * cellDim.y * cellDim.z; }
  // distance from pos to the centre of `cell` (cell centres at (c+0.5)h from the lower corner)
  UH_HD vec3<T> distanceToCellCenter(vec3<T> pos, int3 c) const {
    vec3<T> d;
    d.x = fma_(-cellSize.x, T(c.x) + T(0.5), pos.x + box.boxSize.x * T(0.5));
    d.y = fma_(-cellSize.y, T(c.y) + T(0.5), pos.y + box.boxSize.y * T(0.5));
    d.z = fma_(-cellSize.z, T(c.z) + T(0.5), pos.z + box.boxSize.z * T(0.5));
    return box.apply_pbc(d);
  }
};

template <class T> inline GridT<T> make_grid(const BoxT<T> &box, int3 cellDim) {
  GridT<T> g;
  g.box = box;
  if (cellDim.z == 0) cellDim.z = 1;
  g.cellDim = cellDim;
  g.cellSize = {box.boxSize.x / T(cellDim.x), box.boxSize.y / T(cellDim.y), box.boxSize.z / T(cellDim.z)};
  g.invCellSize = {T(1.0) / g.cellSize.x, T(1.0) / g.cellSize.y, T(1.0) / g.cellSize.z};
  if (box.boxSize.z == T(0)) g.invCellSize.z = T(0);
  g.cellVolume = g.cellSize.x * g.cellSize.y;
  if (cellDim.z > 1) g.cellVolume *= g.cellSize.z;
  return g;
}

// ---- Morton key of a cell: 3 x 10 bits interleaved, x lowest (utils/ParticleSorter.cuh:51-76) ---
UH_HD uint spread10(uint i) {
  uint x = i & 0x3ffu;
  x = (x | x << 16) & 0x30000ffu;
  x = (x | x << 8) & 0x300f00fu;
  x = (x | x << 4) & 0x30c30c3u;
  x = (x | x << 2) & 0x9249249u;
  return x;
}
UH_HD uint compact10(uint x) {  // inverse of spread10
  x &= 0x9249249u;
  x = (x | x >> 2) & 0x30c30c3u;
  x = (x | x >> 4) & 0x300f00fu;
  x = (x | x >> 8) & 0x30000ffu;
  x = (x | x >> 16) & 0x3ffu;
  return x;
}
// MI355X deals consecutive workgroups round-robin to its 8 XCDs, each with its own 4 MB L2.  Kernels whose neighbouring
// workgroups share data (Morton-sorted particles, adjacent grid tiles) remap the block index so that every XCD gets one
// CONTIGUOUS range of the work: the shared lines are then fetched into one L2 instead of eight.
UH_D uint xcd_contiguous_block(uint b, uint nb) {
  const uint per = nb >> 3;
  if (b >= (per << 3)) return b;  // tail of an incomplete round keeps its place
  return (b & 7u) * per + (b >> 3);
}
// inclusive prefix sum over the 64 lanes of a wave in six DPP additions (no LDS round trips: a __shfl_up ladder is six ds_bpermute, each
// with its address arithmetic and ~100 cycles of latency): Hillis-Steele inside each row of 16 lanes, then the rows' totals carried across
// (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3)
UH_D uint wave_inclusive_scan(uint x) {
  x += (uint)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);   // row_shr:1
  x += (uint)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);   // row_shr:2
  x += (uint)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);   // row_shr:4
  x += (uint)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);   // row_shr:8
  x += (uint)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15
  x += (uint)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31
  return x;
}
// the sum of x over the 64 lanes of a wave, valid in lane 63 (read it with wave_total), in six DPP additions — a __shfl_xor ladder is six
// ds_bpermute round trips per value (~100 cycles each): pairs and quads by quad_perm, the row of 16 by the two mirrors, the rows by the
// two broadcasts.  The order of the additions is fixed: the same bits on every run.
template <int CTRL, int ROWS, bool BOUND> UH_D float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWS, 0xf, BOUND));
}
UH_D float wave_sum_to_last(float x) {
  x += dpp_move<0xB1, 0xf, true>(x);    // quad_perm [1, 0, 3, 2]
  x += dpp_move<0x4E, 0xf, true>(x);    // quad_perm [2, 3, 0, 1]
  x += dpp_move<0x141, 0xf, true>(x);   // row_half_mirror: eight lanes
  x += dpp_move<0x140, 0xf, true>(x);   // row_mirror: the row of sixteen
  x += dpp_move<0x142, 0xa, false>(x);  // row_bcast:15 into rows 1 and 3
  x += dpp_move<0x143, 0xc, false>(x);  // row_bcast:31 into rows 2 and 3
  return x;
}
UH_D float wave_total(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_sum_to_last(x)), 63)); }
UH_HD uint morton_hash(int3 c) { return spread10((uint)c.x) | (spread10((uint)c.y) << 1) | (spread10((uint)c.z) << 2); }

// Global -> LDS copies go through registers U at a time: written as `buf[f(i)] = g[h(i)]` in a loop of run-time length the compiler
// emits load, s_waitcnt vmcnt(0), ds_write per iteration — every element one dependent round trip (8 to 16 of them per workgroup in
// each of these kernels).  Here the U loads of a round are all issued before the first is used.
template <int U, class T, class LD, class ST> UH_D void staged_copy(int begin, int end, int step, LD ld, ST st) {
  for (int i0 = begin; i0 < end; i0 += step * U) {
    T t[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * step;
      if (i < end) t[u] = ld(i);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * step;
      if (i < end) st(i, t[u]);
    }
  }
}

// ---- error plumbing ------------------------------------------------------------------------------
void set_last_error(const char *fmt, ...);
#define UH_CHECK(expr)                                                                         \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      uammd_hip::set_last_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return (int)e_ ? (int)e_ : -1;                                                           \
    }                                                                                          \
  } while (0)

}  // namespace uammd_hip
