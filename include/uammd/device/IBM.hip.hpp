// device/IBM.hip.hpp — the device side of IBM<Kernel, Grid, Index3D> (reference: src/misc/IBM.cuh:63-203, IBM.cu, IBM_utils.cuh) for ANY
// kernel, quantity and iterator types: compiled with the user's translation unit by hipcc, as nvcc compiles the reference's templates.
//
// The Kernel concept (IBM_utils.cuh:8-65):  support   a public member `support` (int or int3), or getSupport(real3 pos, int3 cell) when it
//                                                      depends on the position, with getMaxSupport() sizing the weights' storage;
//                                           window    phi(real r, real3 pos), or phiX / phiY / phiZ when the axes differ; any return type
//                                                      the weight computation can multiply (the reference's tests use int and real).
// WeightCompute(value, thrust::tuple<phiX, phiY, phiZ>) -> what is added to a node (spread) or summed for a particle (gather);
// QuadratureWeights(cell, grid) -> the quadrature weight of a node in the gather.  Defaults: value * phiX * phiY * phiZ, the cell volume.
//
// Execution (own design, as csrc/ibm.hip's fixed-window kernels): ONE 64-LANE WAVE PER PARTICLE, four particles per workgroup.  The
// sx + sy + sz one-dimensional weights are evaluated once by the wave's first lanes into the wave's own slice of LDS; the lanes then walk
// the sx sy sz nodes of the stencil, x fastest (consecutive lanes -> consecutive addresses).  Spreading adds with the hardware's atomics
// (f32 / f64 / int in L2); the gather reduces the lanes' partial sums with shuffles and lane 0 adds the total to the particle's entry.
// No workgroup barrier anywhere: a wave whose particle does not exist leaves at once.
#ifndef UAMMD_MI355X_DEVICE_IBM_HIP_HPP
#define UAMMD_MI355X_DEVICE_IBM_HIP_HPP
#if !defined(__HIPCC__)
#error "device/IBM.hip.hpp holds device code: compile this translation unit with hipcc"
#endif
#include <hip/hip_runtime.h>
#include <thrust/device_ptr.h>
#include <thrust/memory.h>
#include <thrust/tuple.h>

#include <iterator>
#include <type_traits>
#include <utility>

#include "../utils/vector.cuh"

namespace uammd {
namespace IBM_ns {

struct DefaultQuadratureWeights {   // IBM.cuh:80-86
  template <class Grid> inline __host__ __device__ real operator()(int3 cellj, const Grid &grid) const { return grid.getCellVolume(cellj); }
};
struct DefaultWeightCompute {       // IBM.cuh:88-97
  template <class T1, class T2> inline __device__ auto operator()(T1 value, thrust::tuple<T2, T2, T2> kernel) const
      -> decltype(value * thrust::get<0>(kernel) * thrust::get<1>(kernel) * thrust::get<2>(kernel)) {
    return value * thrust::get<0>(kernel) * thrust::get<1>(kernel) * thrust::get<2>(kernel);
  }
};

namespace detail {
template <class...> using void_t = void;
// ---- the Kernel concept, by detection ------------------------------------------------------------------------------------------------
template <class K, class = void> struct has_getSupport : std::false_type {};
template <class K> struct has_getSupport<K, void_t<decltype(std::declval<K &>().getSupport(real3(), int3()))>> : std::true_type {};
template <class K, class = void> struct has_getMaxSupport : std::false_type {};
template <class K> struct has_getMaxSupport<K, void_t<decltype(std::declval<K &>().getMaxSupport())>> : std::true_type {};
template <class K, class = void> struct has_phiX : std::false_type {};
template <class K> struct has_phiX<K, void_t<decltype(std::declval<K &>().phiX(real(), real3()))>> : std::true_type {};
template <class K, class = void> struct has_phiY : std::false_type {};
template <class K> struct has_phiY<K, void_t<decltype(std::declval<K &>().phiY(real(), real3()))>> : std::true_type {};
template <class K, class = void> struct has_phiZ : std::false_type {};
template <class K> struct has_phiZ<K, void_t<decltype(std::declval<K &>().phiZ(real(), real3()))>> : std::true_type {};

inline __host__ __device__ int3 asInt3(int s) { return make_int3(s, s, s); }
inline __host__ __device__ int3 asInt3(int3 s) { return s; }
template <class K, bool = has_getSupport<K>::value> struct GetSupport {
  static __host__ __device__ int3 get(K &kernel, real3 pos, int3 cell) { return asInt3(kernel.getSupport(pos, cell)); }
};
template <class K> struct GetSupport<K, false> {
  static __host__ __device__ int3 get(K &kernel, real3, int3) { return asInt3(kernel.support); }
};
template <class K, bool = has_getMaxSupport<K>::value> struct GetMaxSupport {
  static int3 get(K &kernel) { return asInt3(kernel.getMaxSupport()); }
};
template <class K> struct GetMaxSupport<K, false> {
  static int3 get(K &kernel) { return GetSupport<K>::get(kernel, real3(), int3()); }
};
template <class K> __host__ __device__ inline auto phiX(K &k, real r, real3 pos, std::true_type) -> decltype(k.phiX(r, pos)) { return k.phiX(r, pos); }
template <class K> __host__ __device__ inline auto phiX(K &k, real r, real3 pos, std::false_type) -> decltype(k.phi(r, pos)) { return k.phi(r, pos); }
template <class K> __host__ __device__ inline auto phiY(K &k, real r, real3 pos, std::true_type) -> decltype(k.phiY(r, pos)) { return k.phiY(r, pos); }
template <class K> __host__ __device__ inline auto phiY(K &k, real r, real3 pos, std::false_type) -> decltype(k.phi(r, pos)) { return k.phi(r, pos); }
template <class K> __host__ __device__ inline auto phiZ(K &k, real r, real3 pos, std::true_type) -> decltype(k.phiZ(r, pos)) { return k.phiZ(r, pos); }
template <class K> __host__ __device__ inline auto phiZ(K &k, real r, real3 pos, std::false_type) -> decltype(k.phi(r, pos)) { return k.phi(r, pos); }
template <class K> __host__ __device__ inline auto phiX(K &k, real r, real3 pos) -> decltype(phiX(k, r, pos, has_phiX<K>())) { return phiX(k, r, pos, has_phiX<K>()); }
template <class K> __host__ __device__ inline auto phiY(K &k, real r, real3 pos) -> decltype(phiY(k, r, pos, has_phiY<K>())) { return phiY(k, r, pos, has_phiY<K>()); }
template <class K> __host__ __device__ inline auto phiZ(K &k, real r, real3 pos) -> decltype(phiZ(k, r, pos, has_phiZ<K>())) { return phiZ(k, r, pos, has_phiZ<K>()); }
template <class T> __host__ __device__ T &lvalueOf();   // (std::declval is a host function: unusable in a type computed inside a kernel)
template <class K> using KernelValue = decltype(phiX(lvalueOf<K>(), real(), real3()));

// ---- adding to a node: the hardware's returning-nothing atomics (utils/atomics.cuh's role) ------------------------------------------------
__device__ inline void nodeAdd(int &dst, int v) { atomicAdd(&dst, v); }
__device__ inline void nodeAdd(unsigned int &dst, unsigned int v) { atomicAdd(&dst, v); }
__device__ inline void nodeAdd(float &dst, float v) { unsafeAtomicAdd(&dst, v); }
__device__ inline void nodeAdd(double &dst, double v) { unsafeAtomicAdd(&dst, v); }
__device__ inline void nodeAdd(::float2 &dst, const ::float2 &v) { unsafeAtomicAdd(&dst.x, v.x); unsafeAtomicAdd(&dst.y, v.y); }
__device__ inline void nodeAdd(::float3 &dst, const ::float3 &v) { unsafeAtomicAdd(&dst.x, v.x); unsafeAtomicAdd(&dst.y, v.y); unsafeAtomicAdd(&dst.z, v.z); }
__device__ inline void nodeAdd(::float4 &dst, const ::float4 &v) {
  unsafeAtomicAdd(&dst.x, v.x); unsafeAtomicAdd(&dst.y, v.y); unsafeAtomicAdd(&dst.z, v.z); unsafeAtomicAdd(&dst.w, v.w);
}
__device__ inline void nodeAdd(::double2 &dst, const ::double2 &v) { unsafeAtomicAdd(&dst.x, v.x); unsafeAtomicAdd(&dst.y, v.y); }
__device__ inline void nodeAdd(::double3 &dst, const ::double3 &v) { unsafeAtomicAdd(&dst.x, v.x); unsafeAtomicAdd(&dst.y, v.y); unsafeAtomicAdd(&dst.z, v.z); }
__device__ inline void nodeAdd(::double4 &dst, const ::double4 &v) {
  unsafeAtomicAdd(&dst.x, v.x); unsafeAtomicAdd(&dst.y, v.y); unsafeAtomicAdd(&dst.z, v.z); unsafeAtomicAdd(&dst.w, v.w);
}

// ---- the sum of a value over the 64 lanes of a wave (every lane gets it) -----------------------------------------------------------------
template <class T> __device__ inline typename std::enable_if<std::is_arithmetic<T>::value, T>::type waveSum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline ::float2 waveSum(::float2 v) { return ::float2(waveSum(v.x), waveSum(v.y)); }
__device__ inline ::float3 waveSum(::float3 v) { return ::float3(waveSum(v.x), waveSum(v.y), waveSum(v.z)); }
__device__ inline ::float4 waveSum(::float4 v) { return ::float4(waveSum(v.x), waveSum(v.y), waveSum(v.z), waveSum(v.w)); }
__device__ inline ::double2 waveSum(::double2 v) { return ::double2(waveSum(v.x), waveSum(v.y)); }
__device__ inline ::double3 waveSum(::double3 v) { return ::double3(waveSum(v.x), waveSum(v.y), waveSum(v.z)); }
__device__ inline ::double4 waveSum(::double4 v) { return ::double4(waveSum(v.x), waveSum(v.y), waveSum(v.z), waveSum(v.w)); }

constexpr int kWavesPerBlock = 4;

// The stencil of one particle, identical in every lane: the particle's cell, how far the stencil starts to its left, its extent.
// (IBM.cu:10-31: with an even support the particle's cell is off centre and the stencil is shifted by one node when the leftmost
// node's centre would be farther away than support h / 2.)
template <class Grid> struct Footprint {
  real3 pos;
  int3 cell, left, support;
};
template <bool is2D, class Grid, class Kernel> __device__ inline Footprint<Grid> footprint(const Grid &grid, Kernel &kernel, real3 pos) {
  Footprint<Grid> f;
  f.pos = pos;
  f.cell = grid.getCell(pos);
  f.support = GetSupport<Kernel>::get(kernel, pos, f.cell);
  f.left = make_int3(f.support.x / 2, f.support.y / 2, f.support.z / 2);
  const real3 toLeftmost = grid.distanceToCellCenter(pos, make_int3(f.cell.x - f.left.x, f.cell.y - f.left.y, f.cell.z - f.left.z));
  const real3 h = grid.getCellSize(f.cell);
  if (h.x > 0 && fabs(toLeftmost.x) > f.support.x * h.x / real(2.0)) f.left.x -= 1;
  if (h.y > 0 && fabs(toLeftmost.y) > f.support.y * h.y / real(2.0)) f.left.y -= 1;
  if (h.z > 0 && fabs(toLeftmost.z) > f.support.z * h.z / real(2.0)) f.left.z -= 1;
  if (is2D) { f.left.z = 0; f.support.z = 1; }
  return f;
}
// weights[0, sx) along x, [sx, sx + sy) along y, [sx + sy, sx + sy + sz) along z; a node outside a non-periodic grid gets a zero weight
// (it is skipped by the node loop anyway)
template <class Grid, class Kernel, class KV> __device__ inline void fillWeights(KV *weights, const Footprint<Grid> &f, const Grid &grid, Kernel &kernel, int lane) {
  const int sx = f.support.x, sy = f.support.y, sz = f.support.z;
  for (int t = lane; t < sx + sy + sz; t += 64) {
    KV w = KV();
    if (t < sx) {
      const int c = grid.template pbc_cell_coord<0>(f.cell.x + t - f.left.x);
      if (c >= 0) w = phiX(kernel, grid.distanceToCellCenter(f.pos, make_int3(c, f.cell.y, f.cell.z)).x, f.pos);
    } else if (t < sx + sy) {
      const int c = grid.template pbc_cell_coord<1>(f.cell.y + (t - sx) - f.left.y);
      if (c >= 0) w = phiY(kernel, grid.distanceToCellCenter(f.pos, make_int3(f.cell.x, c, f.cell.z)).y, f.pos);
    } else {
      const int c = grid.template pbc_cell_coord<2>(f.cell.z + (t - sx - sy) - f.left.z);
      if (c >= 0) w = phiZ(kernel, grid.distanceToCellCenter(f.pos, make_int3(f.cell.x, f.cell.y, c)).z, f.pos);
    }
    weights[t] = w;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();   // (LDS serves a wave's requests in order: the reads below see the writes above)
}
// node `i` of the stencil (x fastest) -> its cell, wrapped; false when it falls outside a non-periodic grid
template <bool is2D, class Grid> __device__ inline bool stencilNode(const Footprint<Grid> &f, const Grid &grid, int i, int3 &offset, int3 &cell) {
  const int sx = f.support.x, sxy = f.support.x * f.support.y;
  offset.z = is2D ? 0 : i / sxy;
  const int rem = i - offset.z * sxy;
  offset.y = rem / sx;
  offset.x = rem - offset.y * sx;
  cell = grid.pbc_cell(make_int3(f.cell.x + offset.x - f.left.x, f.cell.y + offset.y - f.left.y, is2D ? 0 : (f.cell.z + offset.z - f.left.z)));
  return cell.x >= 0 && cell.y >= 0 && cell.z >= 0 && cell.x < grid.cellDim.x && cell.y < grid.cellDim.y && cell.z < grid.cellDim.z;
}

// S: gridQuantity[node] += weightCompute(quantity[i], phi(node - pos[i]))   (particles2GridD, IBM.cu:83-147)
template <bool is2D, class Grid, class Index3D, class Kernel, class PosIterator, class QuantityIterator, class GridIterator, class WeightCompute>
__global__ void __launch_bounds__(64 * kWavesPerBlock)
spreadWavePerParticle(const PosIterator pos, const QuantityIterator quantity, GridIterator gridQuantity, int numberParticles, Grid grid,
                      Index3D cell2index, Kernel kernel, WeightCompute weightCompute, int weightsPerWave) {
  using KV = KernelValue<Kernel>;
  extern __shared__ unsigned char ibmSharedWeights[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int id = blockIdx.x * kWavesPerBlock + wave;
  if (id >= numberParticles) return;   // (the whole wave)
  KV *weights = reinterpret_cast<KV *>(ibmSharedWeights) + wave * weightsPerWave;
  const typename std::iterator_traits<PosIterator>::value_type pi = pos[id];   // (by value: an iterator may hand out a proxy reference)
  const typename std::iterator_traits<QuantityIterator>::value_type vi = quantity[id];
  const Footprint<Grid> f = footprint<is2D>(grid, kernel, make_real3(pi));
  fillWeights(weights, f, grid, kernel, lane);
  const int nodes = f.support.x * f.support.y * f.support.z;
  for (int i = lane; i < nodes; i += 64) {
    int3 o, cell;
    if (!stencilNode<is2D>(f, grid, i, o, cell)) continue;
    const auto w = weightCompute(vi, thrust::make_tuple(weights[o.x], weights[f.support.x + o.y], weights[f.support.x + f.support.y + o.z]));
    nodeAdd(*thrust::raw_pointer_cast(&gridQuantity[cell2index(cell)]), w);
  }
}

// J: particleQuantity[i] += sum_node qw(node) weightCompute(gridQuantity[node], phi(node - pos[i]))   (grid2ParticlesDTPP, IBM.cu:164-235)
template <bool is2D, class Grid, class Index3D, class Kernel, class PosIterator, class ResultIterator, class GridIterator, class WeightCompute,
          class QuadratureWeights>
__global__ void __launch_bounds__(64 * kWavesPerBlock)
gatherWavePerParticle(const PosIterator pos, ResultIterator particleQuantity, const GridIterator gridQuantity, int numberParticles, Grid grid,
                      Index3D cell2index, Kernel kernel, WeightCompute weightCompute, QuadratureWeights qw, int weightsPerWave) {
  using KV = KernelValue<Kernel>;
  using Result = typename std::iterator_traits<ResultIterator>::value_type;
  extern __shared__ unsigned char ibmSharedWeights[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int id = blockIdx.x * kWavesPerBlock + wave;
  if (id >= numberParticles) return;
  KV *weights = reinterpret_cast<KV *>(ibmSharedWeights) + wave * weightsPerWave;
  const typename std::iterator_traits<PosIterator>::value_type pi = pos[id];
  const Footprint<Grid> f = footprint<is2D>(grid, kernel, make_real3(pi));
  fillWeights(weights, f, grid, kernel, lane);
  Result mine = Result();
  const int nodes = f.support.x * f.support.y * f.support.z;
  for (int i = lane; i < nodes; i += 64) {
    int3 o, cell;
    if (!stencilNode<is2D>(f, grid, i, o, cell)) continue;
    const real dV = qw(cell, grid);
    const typename std::iterator_traits<GridIterator>::value_type gq = gridQuantity[cell2index(cell)];
    const auto w = weightCompute(gq, thrust::make_tuple(weights[o.x], weights[f.support.x + o.y], weights[f.support.x + f.support.y + o.z]));
    mine += dV * w;
  }
  const Result total = waveSum(mine);
  if (lane == 0) *thrust::raw_pointer_cast(&particleQuantity[id]) += total;
}

template <bool is2D, class Kernel, class Grid, class Index3D, class Pos, class Q, class G, class WC>
inline void launchSpread(Kernel &kernel, const Grid &grid, const Index3D &cell2index, Pos pos, Q v, G gridData, WC wc, int numberParticles, hipStream_t st) {
  if (numberParticles <= 0) return;
  const int3 support = GetMaxSupport<Kernel>::get(kernel);
  const int weightsPerWave = support.x + support.y + (is2D ? 1 : support.z);
  const size_t shared = sizeof(KernelValue<Kernel>) * (size_t)weightsPerWave * kWavesPerBlock;
  hipLaunchKernelGGL((spreadWavePerParticle<is2D, Grid, Index3D, Kernel, Pos, Q, G, WC>), dim3((numberParticles + kWavesPerBlock - 1) / kWavesPerBlock),
                     dim3(64 * kWavesPerBlock), shared, st, pos, v, gridData, numberParticles, grid, cell2index, kernel, wc, weightsPerWave);
}
template <bool is2D, class Kernel, class Grid, class Index3D, class Pos, class R, class G, class WC, class QW>
inline void launchGather(Kernel &kernel, const Grid &grid, const Index3D &cell2index, Pos pos, R Jq, G gridData, QW qw, WC wc, int numberParticles, hipStream_t st) {
  if (numberParticles <= 0) return;
  const int3 support = GetMaxSupport<Kernel>::get(kernel);
  const int weightsPerWave = support.x + support.y + (is2D ? 1 : support.z);
  const size_t shared = sizeof(KernelValue<Kernel>) * (size_t)weightsPerWave * kWavesPerBlock;
  hipLaunchKernelGGL((gatherWavePerParticle<is2D, Grid, Index3D, Kernel, Pos, R, G, WC, QW>), dim3((numberParticles + kWavesPerBlock - 1) / kWavesPerBlock),
                     dim3(64 * kWavesPerBlock), shared, st, pos, Jq, gridData, numberParticles, grid, cell2index, kernel, wc, qw, weightsPerWave);
}
}  // namespace detail
}  // namespace IBM_ns
}  // namespace uammd
#endif
