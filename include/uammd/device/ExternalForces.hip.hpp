// ExternalForces.hip.hpp — ExternalForces<Functor> (reference: src/Interactor/ExternalForces.cuh:77-190): a force / energy / virial that
// acts on every particle on its own (walls, gravity, traps), given by a user functor — device code, compiled by hipcc with the user's
// translation unit as it is by nvcc in the reference (examples/misc/LJ.cu:35-66 `HarmonicWall`).
//
// The functor provides
//     __device__ ForceEnergyVirial sum(Interactor::Computables comp, <one value per array>);
//     auto getArrays(ParticleData *pd);      // one device pointer, or a std::tuple of them: what sum() receives, element by element
// and may derive from ParameterUpdatable: the Interactor forwards updateSimulationTime / updateBox / ... to it.  For member i of the
// group: force[i] += sum(...).force (w = 0), energy[i] += .energy, virial[i] += .virial, each only where asked for.
#ifndef UAMMD_MI355X_EXTERNALFORCES_HIP_HPP
#define UAMMD_MI355X_EXTERNALFORCES_HIP_HPP

#include "../uammd.h"
#include "ForceEnergyVirial.hpp"

#include <tuple>
#include <type_traits>
#include <utility>

namespace uammd {
namespace ExternalForces_ns {
// arrays travel to the kernel as plain pointer arguments (no tuple on the device side)
template <class Functor, class... Ptr>
__global__ void __launch_bounds__(128) computeSumGPU(Functor f, int numberParticlesInGroup, const int *__restrict__ groupIndex, Interactor::Computables comp,
                                                     real4 *__restrict__ force, real *__restrict__ energy, real *__restrict__ virial, Ptr... arrays) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= numberParticlesInGroup) return;
  const int i = groupIndex ? groupIndex[id] : id;
  const ForceEnergyVirial res = f.sum(comp, arrays[i]...);
  if (comp.force) force[i] += make_real4(res.force);
  if (comp.energy) energy[i] += res.energy;
  if (comp.virial) virial[i] += res.virial;
}
template <class T> std::tuple<T *> asTuple(T *p) { return std::make_tuple(p); }
template <class... T> std::tuple<T...> asTuple(std::tuple<T...> t) { return t; }
// delegate the ParameterUpdatable calls to the functor when it is one (misc/ParameterUpdatable.h: ParameterUpdatableDelegate)
template <class F> std::enable_if_t<std::is_base_of<ParameterUpdatable, F>::value, ParameterUpdatable *> updatable(F *f) { return f; }
template <class F> std::enable_if_t<!std::is_base_of<ParameterUpdatable, F>::value, ParameterUpdatable *> updatable(F *) { return nullptr; }
}  // namespace ExternalForces_ns

template <class Functor> class ExternalForces : public Interactor {
  std::shared_ptr<Functor> tr;
  ParameterUpdatable *delegate() { return ExternalForces_ns::updatable(tr.get()); }
  template <class... Ptr, size_t... I>
  void launch(Computables comp, hipStream_t st, std::tuple<Ptr...> arrays, std::index_sequence<I...>) {
    const int n = subgroup ? subgroup->getNumberParticles() : pd->getNumParticles();
    if (n <= 0) return;
    const int *groupIndex = subgroup ? subgroup->getIndicesRawPtr(access::gpu) : nullptr;
    real4 *force = comp.force ? pd->getForce(access::gpu, access::readwrite).raw() : nullptr;
    real *energy = comp.energy ? pd->getEnergy(access::gpu, access::readwrite).raw() : nullptr;
    real *virial = comp.virial ? pd->getVirial(access::gpu, access::readwrite).raw() : nullptr;
    hipLaunchKernelGGL((ExternalForces_ns::computeSumGPU<Functor, Ptr...>), dim3((n + 127) / 128), dim3(128), 0, st, *tr, n, groupIndex, comp, force,
                       energy, virial, std::get<I>(arrays)...);
    detail::hipCheck(hipGetLastError(), "ExternalForces");
  }
public:
  ExternalForces(shared_ptr<ParticleGroup> pg, std::shared_ptr<Functor> tr = std::make_shared<Functor>()) : Interactor(pg, "ExternalForces"), tr(tr) {}
  ExternalForces(shared_ptr<ParticleData> pd, std::shared_ptr<Functor> tr = std::make_shared<Functor>()) : Interactor(pd, "ExternalForces"), tr(tr) {}
  void sum(Computables comp, hipStream_t st = 0) override {
    auto arrays = ExternalForces_ns::asTuple(tr->getArrays(pd.get()));
    launch(comp, st, arrays, std::make_index_sequence<std::tuple_size<decltype(arrays)>::value>());
  }
  void updateTimeStep(real v) override { if (auto *d = delegate()) d->updateTimeStep(v); }
  void updateSimulationTime(real v) override { if (auto *d = delegate()) d->updateSimulationTime(v); }
  void updateBox(Box v) override { if (auto *d = delegate()) d->updateBox(v); }
  void updateTemperature(real v) override { if (auto *d = delegate()) d->updateTemperature(v); }
  void updateViscosity(real v) override { if (auto *d = delegate()) d->updateViscosity(v); }
};

}  // namespace uammd
#endif
