// utils/ParticleSorter.cuh (reference: src/utils/ParticleSorter.cuh:128-323) — the sorter as a class of its own, for library-mode use
// (test/utils/ParticleSorter.cu; ParticleData::sortParticles and the lists use the same machinery through the C ABI).  The order is a
// STABLE sort of the hashes on bits [0, last bit of maxHash) — uammd_sort_pairs (uammd_hip.h), the contract of the SortPairs call at
// ParticleSorter.cuh:316-320 — so elements of equal hash keep their input order, as in the reference.
// Needs a translation unit compiled by hipcc (hash iterators are thrust iterators, as they are CUDA-side in the reference).
#ifndef UAMMD_MI355X_UTILS_PARTICLESORTER_CUH
#define UAMMD_MI355X_UTILS_PARTICLESORTER_CUH
#include "../uammd.h"
#if defined(__HIPCC__)
#include <limits>
#include <thrust/copy.h>
#include <thrust/device_vector.h>
#include <thrust/execution_policy.h>
#include <thrust/iterator/counting_iterator.h>
#include <thrust/iterator/permutation_iterator.h>
#include <thrust/iterator/transform_iterator.h>
#include <thrust/sequence.h>

namespace uammd {
namespace Sorter {
// Z-order hash of a position's cell: three 10-bit cell coordinates interleaved (ParticleSorter.cuh:51-75)
struct MortonHash {
  Grid grid;
  MortonHash(Grid grid) : grid(grid) {}
  inline __host__ __device__ uint encodeMorton(const uint &i) const {
    uint x = i & 0x3ffu;
    x = (x | x << 16) & 0x30000ffu;
    x = (x | x << 8) & 0x300f00fu;
    x = (x | x << 4) & 0x30c30c3u;
    x = (x | x << 2) & 0x9249249u;
    return x;
  }
  inline __host__ __device__ uint hash(int3 cell) const { return encodeMorton(cell.x) | (encodeMorton(cell.y) << 1) | (encodeMorton(cell.z) << 2); }
  template <class VectorType> inline __host__ __device__ uint operator()(VectorType pos) const { return hash(grid.getCell(pos)); }
};
// the cell's linear index as the hash (ParticleSorter.cuh:78-90)
struct CellIndexHash {
  Grid grid;
  CellIndexHash(Grid grid) : grid(grid) {}
  inline __host__ __device__ uint hash(int3 cell) const { return grid.getCellIndex(cell); }
  template <class VectorType> inline __host__ __device__ uint operator()(VectorType pos) const { return hash(grid.getCell(pos)); }
};
inline int clz(uint n) { return n ? __builtin_clz(n) : 32; }   // (ParticleSorter.cuh:92-99)
}  // namespace Sorter

class ParticleSorter {
  bool init = false, originalOrderNeedsUpdate = true;
  thrust::device_vector<int> original_index, index;
  thrust::device_vector<uint> hash;
  static void sortByKey(uint *keys, int *values, int N, hipStream_t st, int end_bit = 32) {
    detail::check(uammd_sort_pairs(keys, values, N, end_bit, (void *)st));
  }
public:
  // the current order from the hashes the iterator hands out (ParticleSorter.cuh:131-152, :243-272)
  template <class HashIterator>
  void updateOrderWithCustomHash(HashIterator hasher, uint N, uint maxHash = std::numeric_limits<uint>::max(), hipStream_t st = 0) {
    init = true;
    hash.resize(N);
    index.resize(N);
    thrust::copy_n(thrust::hip::par.on(st), hasher, N, hash.begin());
    thrust::sequence(thrust::hip::par.on(st), index.begin(), index.end(), 0);
    const int maxbit = std::min(32 - Sorter::clz(maxHash), 32);
    sortByKey(thrust::raw_pointer_cast(hash.data()), thrust::raw_pointer_cast(index.data()), (int)N, st, maxbit);
    originalOrderNeedsUpdate = true;
  }
  // ... from the hash of the cell each position falls in (ParticleSorter.cuh:156-164)
  template <class CellHasher = Sorter::MortonHash, class InputIterator>
  void updateOrderByCellHash(InputIterator pos, uint N, Box box, int3 cellDim, hipStream_t st = 0) {
    Grid grid(box, cellDim);
    CellHasher hasher(grid);
    auto hashIterator = thrust::make_transform_iterator(pos, hasher);
    const uint maxHash = hasher.hash(make_int3(cellDim.x - 1, cellDim.y - 1, cellDim.z - 1));
    updateOrderWithCustomHash(hashIterator, N, maxHash, st);
  }
  // ... with the ids as hashes: original_index[id] = where the particle with that id sits now (ParticleSorter.cuh:167-175, :222-241)
  void updateOrderById(int *id, int N, hipStream_t st = 0) {
    original_index.resize(N);
    thrust::device_vector<uint> keys(N);
    thrust::sequence(thrust::hip::par.on(st), original_index.begin(), original_index.end(), 0);
    thrust::copy_n(thrust::hip::par.on(st), id, N, keys.begin());
    sortByKey(thrust::raw_pointer_cast(keys.data()), thrust::raw_pointer_cast(original_index.data()), N, st);
  }
  // sorted[i] = unsorted[index[i]]; the two must not alias (ParticleSorter.cuh:178-187)
  template <class InputIterator, class OutputIterator>
  void applyCurrentOrder(InputIterator d_property_unsorted, OutputIterator d_property_sorted, int N, hipStream_t st = 0) {
    auto pi = thrust::make_permutation_iterator(d_property_unsorted, index.begin());
    thrust::copy_n(thrust::hip::par.on(st), pi, N, d_property_sorted);
  }
  // (an index array longer than the last sort is completed with the identity, ParticleSorter.cuh:275-284)
  int *getSortedIndexArray(int N) {
    const int lastN = (int)index.size();
    if (lastN != N) {
      index.resize(N);
      if (N > lastN) thrust::sequence(index.begin() + lastN, index.end(), lastN);
    }
    return thrust::raw_pointer_cast(index.data());
  }
  uint *getSortedHashes() { return thrust::raw_pointer_cast(hash.data()); }
  int *getIndexArrayById(int *id, int N, hipStream_t st = 0) {   // ParticleSorter.cuh:286-301
    if (!init) return id;
    if (originalOrderNeedsUpdate) {
      updateOrderById(id, N, st);
      originalOrderNeedsUpdate = false;
    }
    const int lastN = (int)original_index.size();
    if (lastN != N) {
      original_index.resize(N);
      if (N > lastN) thrust::copy(thrust::hip::par.on(st), id + lastN, id + N, original_index.begin() + lastN);
    }
    return thrust::raw_pointer_cast(original_index.data());
  }
};
}  // namespace uammd
#else
#error "utils/ParticleSorter.cuh: hash iterators are thrust iterators — compile this translation unit with hipcc"
#endif
#endif
