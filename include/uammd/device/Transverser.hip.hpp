// Transverser.hip.hpp — user-defined DEVICE functors on top of the cell list built by libuammd_hip (MI355X / gfx950).
//
// A functor cannot cross a C ABI, so — exactly as in the reference, where Transversers are template arguments compiled by
// nvcc with the user's translation unit — this header is compiled by hipcc with the user's code.  It provides
//   NeighbourContainer / NeighbourIterator / Neighbour   Interactor/NeighbourList/CellList/NeighbourContainer.cuh:54-191
//   transverseWithNeighbourContainer                     Interactor/NeighbourList/common.cuh:10-34
//   the optional-member dispatch of the Transverser concept (zero / accumulate / getInfo / prepare / getSharedMemorySize)
//                                                        utils/TransverserUtils.cuh:34-51, :151-274
//   the same container concept over the Verlet list      Interactor/NeighbourList/BasicList/NeighbourContainer.cuh:42-125
//   NBodyBase::transverse (all pairs, 128-wide LDS tiles) Interactor/NBodyBase.cuh:46-159
// on the POD that uammd_celllist_get() returns (uammd_celllist_data = CellListBase::CellListData, CellListBase.cuh:145-160).
// The fused Lennard-Jones path (uammd_lj_transverse_celllist) does not go through here; this is for everything else.
//
//   hipcc --offload-arch=gfx950 -std=c++17 -ffp-contract=off -Iinclude my_sim.hip -Luammd_amd/lib -luammd_hip
#ifndef UAMMD_MI355X_TRANSVERSER_HIP_HPP
#define UAMMD_MI355X_TRANSVERSER_HIP_HPP
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "../../uammd_hip.h"

namespace uammd {
namespace device {

using real = float;
using real4 = float4;
using real3 = float3;

// ---- grid arithmetic of the list (utils/Box.cuh:51-58, utils/Grid.cuh:49-106), same FMA placement as the library --------
struct ListGrid {
  int3 cellDim;
  float3 L, minusInvL, invCellSize;
  __host__ explicit ListGrid(const uammd_celllist_data &d) {
    cellDim = make_int3(d.cellDim[0], d.cellDim[1], d.cellDim[2]);
    L = make_float3(d.boxSize[0], d.boxSize[1], d.boxSize[2]);
    minusInvL = make_float3(d.periodic[0] ? -1.0f / L.x : 0.0f, d.periodic[1] ? -1.0f / L.y : 0.0f, d.periodic[2] ? -1.0f / L.z : 0.0f);
    const float3 cs = make_float3(L.x / (float)cellDim.x, L.y / (float)cellDim.y, L.z / (float)cellDim.z);
    invCellSize = make_float3(1.0f / cs.x, 1.0f / cs.y, 1.0f / cs.z);
  }
  __device__ float3 apply_pbc(float3 r) const {
    const float ox = floorf(fmaf(r.x, minusInvL.x, 0.5f)), oy = floorf(fmaf(r.y, minusInvL.y, 0.5f)), oz = floorf(fmaf(r.z, minusInvL.z, 0.5f));
    return make_float3(r.x + ox * L.x, r.y + oy * L.y, r.z + oz * L.z);
  }
  __device__ int3 getCell(float3 r) const {
    const float3 p = apply_pbc(r);
    int3 c = make_int3((int)(fmaf(0.5f, L.x, p.x) * invCellSize.x), (int)(fmaf(0.5f, L.y, p.y) * invCellSize.y),
                       (int)(fmaf(0.5f, L.z, p.z) * invCellSize.z));
    if (c.x == cellDim.x) c.x = 0;
    if (c.y == cellDim.y) c.y = 0;
    if (c.z == cellDim.z) c.z = 0;
    return c;
  }
  __device__ int pbc(int c, int n, float periodicFlag) const {
    if (periodicFlag == 0.0f) return c;  // non periodic: left unwrapped (the iterator skips it)
    return c < 0 ? c + n : (c >= n ? c - n : c);
  }
};

struct Neighbour {
  int internal_i;
  const int *groupIndex;
  const real4 *sortPos;
  __device__ int getInternalIndex() const { return internal_i; }
  __device__ int getGroupIndex() const { return groupIndex[internal_i]; }
  __device__ real4 getPos() const { return sortPos[internal_i]; }
};

class NeighbourContainer;
// forward iterator over the particles of the 27 (9 / 3 / 1) surrounding cells: x fastest, then y, then z; ascending inside a cell
class NeighbourIterator {
  friend class NeighbourContainer;
  const uammd_celllist_data *d;
  const ListGrid *g;
  int3 celli;
  int currentCell, numberCells, npx, npy, j, last;
  __device__ void nextCell() {
    while (j >= last) {
      if (++currentCell >= numberCells) { j = -1; return; }
      int3 c = celli;
      if (npx > 1) c.x += currentCell % 3 - 1;
      if (npy > 1) c.y += (currentCell / npx) % 3 - 1;
      if (numberCells > npx * npy) c.z += currentCell / (npx * npy) - 1;
      c.x = g->pbc(c.x, g->cellDim.x, g->minusInvL.x);
      c.y = g->pbc(c.y, g->cellDim.y, g->minusInvL.y);
      c.z = g->pbc(c.z, g->cellDim.z, g->minusInvL.z);
      if (c.x < 0 || c.x >= g->cellDim.x || c.y < 0 || c.y >= g->cellDim.y || c.z < 0 || c.z >= g->cellDim.z) continue;
      const int ic = c.x + g->cellDim.x * (c.y + g->cellDim.y * c.z);
      const unsigned cs = d->d_cellStart[ic];
      if (cs < d->VALID_CELL) continue;  // empty cell
      j = (int)(cs - d->VALID_CELL);
      last = d->d_cellEnd[ic];
    }
  }
  __device__ NeighbourIterator(const uammd_celllist_data *d_, const ListGrid *g_, int i) : d(d_), g(g_) {
    const real4 p = reinterpret_cast<const real4 *>(d->d_sortPos)[i];
    celli = g->getCell(make_float3(p.x, p.y, p.z));
    npx = g->cellDim.x > 1 ? 3 : 1;
    npy = g->cellDim.y > 1 ? 3 : 1;
    numberCells = npx * npy * (g->cellDim.z > 1 ? 3 : 1);
    currentCell = -1;
    j = 0;
    last = 0;
    nextCell();
  }
public:
  __device__ explicit operator bool() const { return j >= 0; }
  __device__ Neighbour operator*() const { return Neighbour{j, d->d_groupIndex, reinterpret_cast<const real4 *>(d->d_sortPos)}; }
  __device__ NeighbourIterator &operator++() { ++j; nextCell(); return *this; }
  __device__ NeighbourIterator operator++(int) { NeighbourIterator t = *this; ++(*this); return t; }
};

class NeighbourContainer {
  uammd_celllist_data d;
  ListGrid g;
  int my_i = -1;
public:
  __host__ explicit NeighbourContainer(const uammd_celllist_data &d_) : d(d_), g(d_) {}
  __device__ void set(int i) { my_i = i; }
  __device__ NeighbourIterator begin() const { return NeighbourIterator(&d, &g, my_i); }
  __host__ __device__ const real4 *getSortedPositions() const { return (const real4 *)d.d_sortPos; }
  __host__ __device__ const int *getGroupIndexes() const { return d.d_groupIndex; }
};

// ---- optional members of the Transverser concept (docs/Transverser.rst:25-54) ---------------------------------------------
namespace detail {
template <class...> using void_t = void;
template <class T, class = void> struct has_getInfo : std::false_type {};
template <class T> struct has_getInfo<T, void_t<decltype(std::declval<T &>().getInfo(0))>> : std::true_type {};
template <class T, class = void> struct has_zero : std::false_type {};
template <class T> struct has_zero<T, void_t<decltype(std::declval<T &>().zero())>> : std::true_type {};
template <class T, class Q, class = void> struct has_accumulate : std::false_type {};
template <class T, class Q>
struct has_accumulate<T, Q, void_t<decltype(std::declval<T &>().accumulate(std::declval<Q &>(), std::declval<const Q &>()))>> : std::true_type {};

template <class T, class = void> struct has_getSharedMemorySize : std::false_type {};
template <class T> struct has_getSharedMemorySize<T, void_t<decltype(std::declval<T &>().getSharedMemorySize())>> : std::true_type {};
// SharedMemorySizeDelegator (TransverserUtils.cuh:42-51): dynamic LDS the functor asks for, 0 when it has no such member
template <class T> inline std::enable_if_t<has_getSharedMemorySize<T>::value, size_t> sharedMemorySize(T &tr) { return tr.getSharedMemorySize(); }
template <class T> inline std::enable_if_t<!has_getSharedMemorySize<T>::value, size_t> sharedMemorySize(T &) { return 0; }
// TransverserAdaptor::prepare (TransverserUtils.cuh:249-263): host-side hook called before every launch, if the functor has it
template <class T, class... A> inline auto prepare_impl(int, T &tr, A &&...a) -> decltype(tr.prepare(std::forward<A>(a)...), void()) { tr.prepare(std::forward<A>(a)...); }
template <class T, class... A> inline void prepare_impl(long, T &, A &&...) {}
template <class T, class... A> inline void prepare(T &tr, A &&...a) { prepare_impl(0, tr, std::forward<A>(a)...); }

template <class T, bool = has_getInfo<T>::value> struct Adaptor;
template <class T> struct Adaptor<T, true> {  // compute(pi, pj, infoi, infoj)
  decltype(std::declval<T &>().getInfo(0)) infoi;
  __device__ void load(T &tr, int ori) { infoi = tr.getInfo(ori); }
  __device__ auto compute(T &tr, int j, const real4 &pi, const real4 &pj) -> decltype(tr.compute(pi, pj, infoi, infoi)) {
    return tr.compute(pi, pj, infoi, tr.getInfo(j));
  }
};
template <class T> struct Adaptor<T, false> {  // compute(pi, pj)
  __device__ void load(T &, int) {}
  __device__ auto compute(T &tr, int, const real4 &pi, const real4 &pj) -> decltype(tr.compute(pi, pj)) { return tr.compute(pi, pj); }
};
template <class T, class Q> __device__ std::enable_if_t<has_zero<T>::value, Q> zero(T &tr) { return tr.zero(); }
template <class T, class Q> __device__ std::enable_if_t<!has_zero<T>::value, Q> zero(T &) { return Q(); }
template <class T, class Q> __device__ std::enable_if_t<has_accumulate<T, Q>::value> accumulate(T &tr, Q &total, const Q &cur) { tr.accumulate(total, cur); }
template <class T, class Q> __device__ std::enable_if_t<!has_accumulate<T, Q>::value> accumulate(T &, Q &total, const Q &cur) { total = total + cur; }
}  // namespace detail

// common.cuh:10-34: one thread per sorted particle.  globalIndex: group -> ParticleData index (nullptr = identity)
template <class Transverser, class Container>
__global__ void __launch_bounds__(128) transverseWithNeighbourContainer(Transverser tr, const int *globalIndex, Container ni, int N) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= N) return;
  const int gi = ni.getGroupIndexes()[id];
  const int ori = globalIndex ? globalIndex[gi] : gi;
  const real4 pi = ni.getSortedPositions()[id];
  detail::Adaptor<Transverser> adaptor;
  adaptor.load(tr, ori);
  using Q = decltype(adaptor.compute(tr, 0, pi, pi));
  Q quantity = detail::zero<Transverser, Q>(tr);
  ni.set(id);
  auto it = ni.begin();
  while (it) {
    const auto n = *it++;
    const int gj = n.getGroupIndex();
    detail::accumulate<Transverser, Q>(tr, quantity, adaptor.compute(tr, globalIndex ? globalIndex[gj] : gj, pi, n.getPos()));
  }
  tr.set(ori, quantity);
}

// CellList::transverseList(tr, st) (NeighbourList/CellList.cuh:165-182) on a built list.  `prepareArgs` go to tr.prepare(...) when the
// functor has that member (the reference passes the ParticleData); tr.getSharedMemorySize() bytes of dynamic LDS are requested
// when it has that one.
template <class Transverser, class... PrepareArgs>
inline int transverseList(uammd_celllist *list, Transverser &tr, hipStream_t st = 0, const int *d_globalIndex = nullptr, PrepareArgs &&...prepareArgs) {
  uammd_celllist_data d;
  if (int e = uammd_celllist_get(list, &d)) return e;
  if (d.numberParticles <= 0) return 0;
  detail::prepare(tr, std::forward<PrepareArgs>(prepareArgs)...);
  hipLaunchKernelGGL((transverseWithNeighbourContainer<Transverser, NeighbourContainer>), dim3((d.numberParticles + 127) / 128), dim3(128),
                     detail::sharedMemorySize(tr), st, tr, d_globalIndex, NeighbourContainer(d), d.numberParticles);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- the Verlet list as a NeighbourContainer (BasicList/NeighbourContainer.cuh:42-125) ---------------------------------------------------
// entry k of sorted particle i at neighbourList[k * particleStride + i]; positions = the CURRENT ones in the order of the last build
class VerletNeighbourContainer;
class VerletNeighbourIterator {
  friend class VerletNeighbourContainer;
  const uammd_verletlist_data *nl;
  int i, k, n, cur;
  __device__ void load() { cur = k < n ? nl->d_neighbourList[(size_t)k * nl->particleStride + i] : -1; }
  __device__ VerletNeighbourIterator(const uammd_verletlist_data *nl_, int i_) : nl(nl_), i(i_), k(0), n(nl_->d_numberNeighbours[i_]) { load(); }
public:
  __device__ explicit operator bool() const { return cur >= 0; }
  __device__ Neighbour operator*() const { return Neighbour{cur, nl->d_groupIndex, reinterpret_cast<const real4 *>(nl->d_sortPos)}; }
  __device__ VerletNeighbourIterator &operator++() { ++k; load(); return *this; }
  __device__ VerletNeighbourIterator operator++(int) { VerletNeighbourIterator t = *this; ++(*this); return t; }
};
class VerletNeighbourContainer {
  uammd_verletlist_data nl;
  int my_i = -1;
public:
  __host__ explicit VerletNeighbourContainer(const uammd_verletlist_data &d) : nl(d) {}
  __device__ void set(int i) { my_i = i; }
  __device__ VerletNeighbourIterator begin() const { return VerletNeighbourIterator(&nl, my_i); }
  __host__ __device__ const real4 *getSortedPositions() const { return (const real4 *)nl.d_sortPos; }
  __host__ __device__ const int *getGroupIndexes() const { return nl.d_groupIndex; }
};
// VerletList::transverseList(tr, st) (NeighbourList/VerletList.cuh:149-168) on an updated list
template <class Transverser, class... PrepareArgs>
inline int transverseList(uammd_verletlist *list, Transverser &tr, hipStream_t st = 0, const int *d_globalIndex = nullptr, PrepareArgs &&...prepareArgs) {
  uammd_verletlist_data d;
  if (int e = uammd_verletlist_get(list, &d)) return e;
  if (d.numberParticles <= 0) return 0;
  detail::prepare(tr, std::forward<PrepareArgs>(prepareArgs)...);
  hipLaunchKernelGGL((transverseWithNeighbourContainer<Transverser, VerletNeighbourContainer>), dim3((d.numberParticles + 127) / 128), dim3(128),
                     detail::sharedMemorySize(tr), st, tr, d_globalIndex, VerletNeighbourContainer(d), d.numberParticles);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- all pairs: NBodyBase::transverse (Interactor/NBodyBase.cuh:46-159) -----------------------------------------------------------------
// Thread per particle; the wave-sized workgroup stages 128 positions (and their Info) per tile in LDS and every thread walks the tile
// in index order, so each particle sees j = 0 .. N-1 in the reference's order.  indices: thread -> particle (nullptr = identity).
namespace detail {
template <class T, bool = has_getInfo<T>::value> struct TileInfo;
template <class T> struct TileInfo<T, true> {
  using Info = decltype(std::declval<T &>().getInfo(0));
  static constexpr size_t bytes = sizeof(Info);
  static __device__ void fill(T &tr, void *sh, int slot, int j) { reinterpret_cast<Info *>(sh)[slot] = tr.getInfo(j); }
  static __device__ auto compute(T &tr, const Info &infoi, const real4 &pi, const real4 &pj, const void *sh, int slot)
      -> decltype(tr.compute(pi, pj, infoi, infoi)) { return tr.compute(pi, pj, infoi, reinterpret_cast<const Info *>(sh)[slot]); }
};
template <class T> struct TileInfo<T, false> {
  struct Info {};
  static constexpr size_t bytes = 0;
  static __device__ void fill(T &, void *, int, int) {}
  static __device__ auto compute(T &tr, const Info &, const real4 &pi, const real4 &pj, const void *, int) -> decltype(tr.compute(pi, pj)) {
    return tr.compute(pi, pj);
  }
};
template <class T> __device__ std::enable_if_t<has_getInfo<T>::value, typename TileInfo<T>::Info> load_info(T &tr, int i) { return tr.getInfo(i); }
template <class T> __device__ std::enable_if_t<!has_getInfo<T>::value, typename TileInfo<T>::Info> load_info(T &, int) { return {}; }
}  // namespace detail

template <class Transverser>
__global__ void __launch_bounds__(128) transverseNBodyKernel(const real4 *pos, const int *indices, Transverser tr, int N, size_t userShMem) {
  extern __shared__ char shMem[];  // [user bytes | 128 positions | 128 Infos]
  real4 *shPos = reinterpret_cast<real4 *>(shMem + userShMem);
  void *shInfo = shMem + userShMem + 128 * sizeof(real4);
  using TI = detail::TileInfo<Transverser>;
  const int tid = blockIdx.x * 128 + threadIdx.x;
  const bool active = tid < N;
  const int id = active ? (indices ? indices[tid] : tid) : 0;
  const real4 pi = active ? pos[id] : real4{0, 0, 0, 0};
  typename TI::Info infoi = detail::load_info(tr, id);
  using Q = decltype(TI::compute(tr, infoi, pi, pi, shInfo, 0));
  Q quantity = detail::zero<Transverser, Q>(tr);
  const int numTiles = (N + 127) / 128;
  for (int tile = 0; tile < numTiles; ++tile) {
    const int iload = tile * 128 + threadIdx.x;
    if (iload < N) {
      const int j = indices ? indices[iload] : iload;
      shPos[threadIdx.x] = pos[j];
      TI::fill(tr, shInfo, threadIdx.x, j);
    }
    __syncthreads();
    if (active) {
      const int cnt = min(128, N - tile * 128);
      for (int c = 0; c < cnt; ++c) detail::accumulate<Transverser, Q>(tr, quantity, TI::compute(tr, infoi, pi, shPos[c], shInfo, c));
    }
    __syncthreads();
  }
  if (active) tr.set(id, quantity);
}

template <class Transverser, class... PrepareArgs>
inline int transverseNBody(const real4 *d_pos, const int *d_indices, Transverser &tr, int numberParticles, hipStream_t st = 0,
                           PrepareArgs &&...prepareArgs) {
  if (numberParticles <= 0) return 0;
  detail::prepare(tr, std::forward<PrepareArgs>(prepareArgs)...);
  const size_t user = (detail::sharedMemorySize(tr) + 15) & ~(size_t)15;
  const size_t sh = user + 128 * (sizeof(real4) + detail::TileInfo<Transverser>::bytes);
  hipLaunchKernelGGL((transverseNBodyKernel<Transverser>), dim3((numberParticles + 127) / 128), dim3(128), sh, st, d_pos, d_indices, tr,
                     numberParticles, user);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace device
}  // namespace uammd
#endif
