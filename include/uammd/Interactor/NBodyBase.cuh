// Interactor/NBodyBase.cuh — the include path and interface of the reference's src/Interactor/NBodyBase.cuh:30-170: every element of a list
// against every other through a Transverser, O(N^2), usable outside the UAMMD classes (any random-access iterator of per-particle values,
// any iterator of indices).  Device code of the user's translation unit: hipcc.
//   NBodyBase::transverse(list, indices, tr, N, stream)     element indices[t] for thread t, the values list[indices[..]]
//   NBodyBase::transverse(list, tr, N, stream)              indices = 0, 1, 2, ...
// A workgroup of 128 threads stages 128 values (and their Infos, when the Transverser has getInfo) per tile in LDS and every thread walks
// the tile in index order: each element meets j = indices[0], indices[1], ... in that order, as in the reference (NBodyBase.cuh:46-110).
// The Transverser's optional members (zero, accumulate, getInfo, getSharedMemorySize; TransverserUtils.cuh) are found by detection.
#pragma once
#include "../uammd.h"
#if defined(__HIPCC__)
#include "../device/Transverser.hip.hpp"
#include <thrust/iterator/counting_iterator.h>
#include <iterator>
namespace uammd {
namespace nbody_ns {
template <class T, class V, bool = device::detail::has_getInfo<T>::value> struct Pair;
template <class T, class V> struct Pair<T, V, true> {   // compute(vi, vj, infoi, infoj)
  using Info = decltype(std::declval<T &>().getInfo(0));
  static constexpr size_t infoBytes = sizeof(Info);
  static __device__ Info load(T &tr, int i) { return tr.getInfo(i); }
  static __device__ void stage(T &tr, void *sh, int slot, int j) { reinterpret_cast<Info *>(sh)[slot] = tr.getInfo(j); }
  static __device__ auto compute(T &tr, const Info &infoi, const V &vi, const V &vj, const void *sh, int slot) -> decltype(tr.compute(vi, vj, infoi, infoi)) {
    return tr.compute(vi, vj, infoi, reinterpret_cast<const Info *>(sh)[slot]);
  }
};
template <class T, class V> struct Pair<T, V, false> {  // compute(vi, vj)
  struct Info {};
  static constexpr size_t infoBytes = 0;
  static __device__ Info load(T &, int) { return {}; }
  static __device__ void stage(T &, void *, int, int) {}
  static __device__ auto compute(T &tr, const Info &, const V &vi, const V &vj, const void *, int) -> decltype(tr.compute(vi, vj)) { return tr.compute(vi, vj); }
};

template <class Transverser, class Iterator, class IndexIterator>
__global__ void __launch_bounds__(128) transverseGPU(const Iterator list, IndexIterator threadId2Index, Transverser tr, int N, size_t valueOffset,
                                                     size_t infoOffset) {
  using V = typename std::iterator_traits<Iterator>::value_type;
  using P = Pair<Transverser, V>;
  extern __shared__ char shMem[];   // [the Transverser's own bytes | 128 values | 128 Infos]
  V *shValue = reinterpret_cast<V *>(shMem + valueOffset);
  void *shInfo = shMem + infoOffset;
  const int tid = blockIdx.x * 128 + threadIdx.x;
  const bool active = tid < N;
  const int id = active ? (int)threadId2Index[tid] : 0;
  const V vi = active ? V(list[id]) : V();
  const typename P::Info infoi = P::load(tr, id);
  using Q = decltype(P::compute(tr, infoi, vi, vi, shInfo, 0));
  Q quantity = device::detail::zero<Transverser, Q>(tr);
  for (int tile = 0; tile * 128 < N; ++tile) {
    const int iload = tile * 128 + threadIdx.x;
    if (iload < N) {
      const int j = (int)threadId2Index[iload];
      shValue[threadIdx.x] = V(list[j]);
      P::stage(tr, shInfo, threadIdx.x, j);
    }
    __syncthreads();
    if (active) {
      const int count = min(128, N - tile * 128);
      for (int c = 0; c < count; ++c) device::detail::accumulate<Transverser, Q>(tr, quantity, P::compute(tr, infoi, vi, shValue[c], shInfo, c));
    }
    __syncthreads();
  }
  if (active) tr.set(id, quantity);
}
}  // namespace nbody_ns

class NBodyBase {
public:
  template <class Iterator, class IndexIterator, class Transverser>
  static inline void transverse(const Iterator &particle_list, const IndexIterator &indices, Transverser &a_tr, int numberParticles, hipStream_t st = 0) {
    if (numberParticles <= 0) return;
    using V = typename std::iterator_traits<Iterator>::value_type;
    using P = nbody_ns::Pair<Transverser, V>;
    auto align = [](size_t bytes, size_t to) { return (bytes + to - 1) / to * to; };
    const size_t valueOffset = align(device::detail::sharedMemorySize(a_tr), 16), infoOffset = align(valueOffset + 128 * sizeof(V), 16);
    const size_t shared = infoOffset + 128 * P::infoBytes;
    hipLaunchKernelGGL((nbody_ns::transverseGPU<Transverser, Iterator, IndexIterator>), dim3((numberParticles + 127) / 128), dim3(128), shared, st,
                       particle_list, indices, a_tr, numberParticles, valueOffset, infoOffset);
    CudaCheckError();
  }
  template <class Iterator, class Transverser>
  static inline void transverse(const Iterator &particle_list, Transverser &a_tr, int numberParticles, hipStream_t st = 0) {
    transverse(particle_list, thrust::make_counting_iterator<int>(0), a_tr, numberParticles, st);
  }
};
}  // namespace uammd
#endif
