// utils/ParticleSorter.cuh — the sorter as a class of its own, for library-mode use (the reference's src/utils/ParticleSorter.cuh:128-323 and
// its test, test/utils/ParticleSorter.cu; ParticleData::sortParticles and the neighbour lists reach the same machinery through the C ABI).
// Built on the C ABI: the order is uammd_sort_pairs' — a STABLE sort of (hash, index) on the bits [0, highest bit of maxHash], so elements
// of equal hash keep their input order — and properties are moved by uammd_gather when they sit behind plain pointers.  What a C ABI
// cannot take — a hash ITERATOR (a transform over positions, a device functor) or properties behind thrust iterators — is evaluated by the
// three small kernels below, compiled with the caller's TU: this header needs hipcc, as the reference's needs nvcc.
#ifndef UAMMD_MI355X_UTILS_PARTICLESORTER_CUH
#define UAMMD_MI355X_UTILS_PARTICLESORTER_CUH
#include "../uammd.h"
#if defined(__HIPCC__)
#include <limits>
#include <thrust/iterator/transform_iterator.h>
// (what the reference's header brings into a program's scope through its own includes, and programs use without including:
// test/utils/ParticleSorter.cu:28-31,42-43)
#include <thrust/copy.h>
#include <thrust/device_vector.h>
#include <thrust/sequence.h>

namespace uammd {
namespace Sorter {
// three 10-bit cell coordinates interleaved, x lowest: the key the cell lists sort by (ParticleSorter.cuh:51-75)
struct MortonHash {
  Grid grid;
  MortonHash(Grid grid) : grid(grid) {}
  // bit b of a 10-bit number moves to bit 3 b (halving strides: 16, 8, 4, 2)
  inline __host__ __device__ uint encodeMorton(const uint &i) const {
    uint v = i & 1023u;
    v = (v ^ (v << 16)) & 0xff0000ffu;
    v = (v ^ (v << 8)) & 0x0300f00fu;
    v = (v ^ (v << 4)) & 0x030c30c3u;
    v = (v ^ (v << 2)) & 0x09249249u;
    return v;
  }
  inline __host__ __device__ uint hash(int3 cell) const { return encodeMorton(cell.x) | encodeMorton(cell.y) << 1 | encodeMorton(cell.z) << 2; }
  template <class VectorType> inline __host__ __device__ uint operator()(VectorType pos) const { return hash(grid.getCell(pos)); }
};
// the cell's linear index as the key (ParticleSorter.cuh:78-90)
struct CellIndexHash {
  Grid grid;
  CellIndexHash(Grid grid) : grid(grid) {}
  inline __host__ __device__ uint hash(int3 cell) const { return grid.getCellIndex(cell); }
  template <class VectorType> inline __host__ __device__ uint operator()(VectorType pos) const { return hash(grid.getCell(pos)); }
};
inline int clz(uint n) { return n ? __builtin_clz(n) : 32; }   // leading zeros of a key (ParticleSorter.cuh:92-99)

namespace detail {
template <class KeyIterator> __global__ void k_keys_and_identity(KeyIterator keys, uint *__restrict__ keyOut, int *__restrict__ order, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { keyOut[i] = keys[i]; order[i] = i; }
}
__global__ inline void k_identity_from(int *__restrict__ order, int first, int n) {
  const int i = first + blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) order[i] = i;
}
template <class In, class Out> __global__ void k_permute(In unsorted, const int *__restrict__ order, Out sorted, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sorted[i] = unsorted[order[i]];
}
inline dim3 blocksFor(int n) { return dim3((unsigned)((n + 255) / 256)); }
// plain pointers to elements uammd_gather moves: the C ABI's copy; anything else: the template kernel
template <class T> struct GatherBytes { static constexpr int value = (sizeof(T) == 4 || sizeof(T) == 8 || sizeof(T) == 12 || sizeof(T) == 16 || sizeof(T) == 24 || sizeof(T) == 32) ? (int)sizeof(T) : 0; };
}  // namespace detail
}  // namespace Sorter

class ParticleSorter {
  uninitialized_cached_vector<uint> keys;      // the hashes, sorted
  uninitialized_cached_vector<int> order;      // order[i] = input position of the element that sorts to place i
  uninitialized_cached_vector<int> placeOfId;  // placeOfId[id] = where the particle with that id sits (updateOrderById)
  bool sortedOnce = false, placeOfIdStale = true;

  static int significantBits(uint maxHash) { return std::min(32 - Sorter::clz(maxHash), 32); }
  template <class T> void permute(const T *unsorted, T *sorted, int N, hipStream_t st, std::true_type) {
    uammd::detail::check(uammd_gather(unsorted, order.raw(), sorted, N, Sorter::detail::GatherBytes<T>::value, (void *)st));
  }
  template <class In, class Out> void permute(In unsorted, Out sorted, int N, hipStream_t st, std::false_type) {
    if (N <= 0) return;
    hipLaunchKernelGGL((Sorter::detail::k_permute<In, Out>), Sorter::detail::blocksFor(N), dim3(256), 0, st, unsorted, (const int *)order.raw(), sorted, N);
    uammd::detail::hipCheck(hipGetLastError(), "ParticleSorter::applyCurrentOrder");
  }
  template <class A, class B> struct BothPlainPointers : std::false_type {};
  template <class T> struct BothPlainPointers<T *, T *> : std::integral_constant<bool, (Sorter::detail::GatherBytes<T>::value > 0) && std::is_trivially_copyable<T>::value> {};
  template <class T> struct BothPlainPointers<const T *, T *> : BothPlainPointers<T *, T *> {};
public:
  // the current order from the hashes an iterator hands out (ParticleSorter.cuh:131-152, :243-272).  maxHash, when the caller knows it,
  // limits the sort to the bits in use.
  template <class HashIterator>
  void updateOrderWithCustomHash(HashIterator hasher, uint N, uint maxHash = std::numeric_limits<uint>::max(), hipStream_t st = 0) {
    sortedOnce = true;
    placeOfIdStale = true;
    keys.resize(N);
    order.resize(N);
    if (N == 0) return;
    hipLaunchKernelGGL((Sorter::detail::k_keys_and_identity<HashIterator>), Sorter::detail::blocksFor((int)N), dim3(256), 0, st, hasher, keys.raw(), order.raw(), (int)N);
    uammd::detail::hipCheck(hipGetLastError(), "ParticleSorter::updateOrderWithCustomHash");
    uammd::detail::check(uammd_sort_pairs(keys.raw(), order.raw(), (int)N, significantBits(maxHash), (void *)st));
  }
  // ... from the hash of the cell each position falls in (ParticleSorter.cuh:156-164)
  template <class CellHasher = Sorter::MortonHash, class InputIterator>
  void updateOrderByCellHash(InputIterator pos, uint N, Box box, int3 cellDim, hipStream_t st = 0) {
    const CellHasher hasher{Grid(box, cellDim)};
    updateOrderWithCustomHash(thrust::make_transform_iterator(pos, hasher), N, hasher.hash(cellDim - make_int3(1, 1, 1)), st);
  }
  // ... with the ids as keys: afterwards placeOfId[id] is the row of the particle with that id (ParticleSorter.cuh:167-175, :222-241)
  void updateOrderById(int *id, int N, hipStream_t st = 0) {
    placeOfId.resize(N);
    if (N <= 0) return;
    uninitialized_cached_vector<uint> idKeys(N);
    hipLaunchKernelGGL((Sorter::detail::k_keys_and_identity<const int *>), Sorter::detail::blocksFor(N), dim3(256), 0, st, (const int *)id, idKeys.raw(), placeOfId.raw(), N);
    uammd::detail::hipCheck(hipGetLastError(), "ParticleSorter::updateOrderById");
    uammd::detail::check(uammd_sort_pairs(idKeys.raw(), placeOfId.raw(), N, 32, (void *)st));
    uammd::detail::hipCheck(hipStreamSynchronize(st), "hipStreamSynchronize");   // (idKeys goes back to the pool with this scope)
  }
  // sorted[i] = unsorted[order[i]]; the two must not alias (ParticleSorter.cuh:178-187)
  template <class InputIterator, class OutputIterator>
  void applyCurrentOrder(InputIterator d_property_unsorted, OutputIterator d_property_sorted, int N, hipStream_t st = 0) {
    permute(d_property_unsorted, d_property_sorted, N, st, BothPlainPointers<InputIterator, OutputIterator>());
  }
  // the order as an index array of N entries; beyond the last sort's size it is the identity (ParticleSorter.cuh:275-284)
  int *getSortedIndexArray(int N) {
    const int known = (int)order.size();
    if (N != known) {
      order.resize(N);   // (keeps the first min(N, known) entries)
      if (N > known) {
        hipLaunchKernelGGL(Sorter::detail::k_identity_from, Sorter::detail::blocksFor(N - known), dim3(256), 0, 0, order.raw(), known, N);
        uammd::detail::hipCheck(hipGetLastError(), "ParticleSorter::getSortedIndexArray");
      }
    }
    return order.raw();
  }
  uint *getSortedHashes() { return keys.raw(); }
  // the row of each id after the last sort (the ids themselves before any sort: nothing has moved, ParticleSorter.cuh:286-301)
  int *getIndexArrayById(int *id, int N, hipStream_t st = 0) {
    if (!sortedOnce) return id;
    if (placeOfIdStale) {
      updateOrderById(id, N, st);
      placeOfIdStale = false;
    }
    const int known = (int)placeOfId.size();
    if (N != known) {
      placeOfId.resize(N);
      if (N > known)   // ids the last sort did not see sit where they are (the reference copies from id + 0 here: the first rows' ids)
        uammd::detail::hipCheck(hipMemcpyAsync(placeOfId.raw() + known, id + known, sizeof(int) * (size_t)(N - known), hipMemcpyDeviceToDevice, st), "hipMemcpyAsync");
    }
    return placeOfId.raw();
  }
};
}  // namespace uammd
#else
#error "utils/ParticleSorter.cuh: hash iterators are device iterators — compile this translation unit with hipcc"
#endif
#endif
