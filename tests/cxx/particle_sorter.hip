// The checks of the reference's test/utils/ParticleSorter.cu (gtest there; plain asserts here) on include/uammd/utils/ParticleSorter.cuh,
// plus the cell-hash order against CellList's own (the same stable Morton order: ParticleSorter.cuh:156-164 is what CellListBase::update calls).
#include "utils/ParticleSorter.cuh"
#include "utils/container.h"
#include <cassert>
#include <cstdio>
#include <random>
#include <vector>
using namespace uammd;

struct TrivialHash {
  inline __host__ __device__ uint operator()(uint i) const { return i; }
};

int main() {
  {  // "SortsCorrectly": a reversed sequence comes out ascending
    auto sorter = std::make_shared<ParticleSorter>();
    const int n = 163840;
    thrust::device_vector<uint> vec(n), svec(n);
    thrust::sequence(vec.rbegin(), vec.rend(), 0);
    hipStream_t st;
    (void)hipStreamCreate(&st);
    auto hash = thrust::make_transform_iterator(vec.begin(), TrivialHash());
    sorter->updateOrderWithCustomHash(hash, n, (uint)n, st);
    sorter->applyCurrentOrder(vec.begin(), svec.begin(), n, st);
    (void)hipStreamSynchronize(st);
    std::vector<uint> h(n);
    thrust::copy(svec.begin(), svec.end(), h.begin());
    for (int i = 0; i < n; ++i) assert(h[i] == (uint)i);
    // equal hashes keep their input order (stable)
    thrust::device_vector<uint> dup(n);
    std::vector<uint> hd(n);
    for (int i = 0; i < n; ++i) hd[i] = (uint)((i * 7919) % 97);
    dup = hd;
    sorter->updateOrderWithCustomHash(dup.begin(), n, 96u, st);
    (void)hipStreamSynchronize(st);
    std::vector<int> idx(n);
    (void)hipMemcpy(idx.data(), sorter->getSortedIndexArray(n), sizeof(int) * n, hipMemcpyDeviceToHost);
    for (int i = 1; i < n; ++i) assert(hd[idx[i - 1]] < hd[idx[i]] || (hd[idx[i - 1]] == hd[idx[i]] && idx[i - 1] < idx[i]));
    (void)hipStreamDestroy(st);
  }
  {  // updateOrderByCellHash == the order CellList builds (stable Morton order on the same grid), getIndexArrayById inverts the ids
    const int n = 20000;
    const real L = 32;
    std::mt19937 gen(7);
    std::uniform_real_distribution<float> u(-L / 2, L / 2);
    std::vector<real4> hp(n);
    for (auto &p : hp) p = make_real4(u(gen), u(gen), u(gen), 0);
    thrust::device_vector<real4> pos = hp;
    Box box(L);
    const int3 cellDim = make_int3(12, 12, 12);
    ParticleSorter sorter;
    sorter.updateOrderByCellHash(thrust::raw_pointer_cast(pos.data()), n, box, cellDim);
    std::vector<int> idx(n);
    (void)hipMemcpy(idx.data(), sorter.getSortedIndexArray(n), sizeof(int) * n, hipMemcpyDeviceToHost);
    CellListBase cl;
    cl.update(thrust::raw_pointer_cast(pos.data()), n, Grid(box, cellDim));
    auto d = cl.getCellList();
    std::vector<int> gi(n);
    (void)hipMemcpy(gi.data(), d.groupIndex, sizeof(int) * n, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) assert(idx[i] == gi[i]);
    thrust::device_vector<int> id(n);
    thrust::sequence(id.begin(), id.end(), 0);
    thrust::device_vector<int> sortedId(n);
    sorter.applyCurrentOrder(id.begin(), sortedId.begin(), n);
    int *byId = sorter.getIndexArrayById(thrust::raw_pointer_cast(sortedId.data()), n);
    std::vector<int> hb(n);
    (void)hipMemcpy(hb.data(), byId, sizeof(int) * n, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) assert(idx[hb[i]] == i);   // the particle with id i sits at row hb[i] of the sorted arrays
  }
  std::printf("particle_sorter: ok\n");
  return 0;
}
