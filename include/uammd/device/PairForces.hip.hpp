// PairForces.hip.hpp — PairForces<MyPotential, NeighbourList> for ANY potential, and Potential::Radial<Functor>, compiled by hipcc with
// the user's translation unit (a potential is a device functor: it needs hipcc here as it needs nvcc in the reference).
//
//   PairForces<MyPotential, NL>::sum            Interactor/PairForces.cu:43-78   neighbour list unless the box is <= 3 rc in every direction,
//                                                                                 then all pairs; the Transverser comes from
//                                                                                 pot->getTransverser(comp, box, pd)
//   Potential::Radial<Functor>                  Interactor/Potential/RadialPotential.cuh:49-154
//   BasicParameterHandler<Functor>              Interactor/Potential/ParameterHandler.cuh:8-66 (type-pair table; type = pos.w)
//
// It sits on the host classes of uammd.h (ParticleData, Box, CellList, VerletList, Interactor: the C ABI underneath) and on the device
// layer of Transverser.hip.hpp (NeighbourContainer, transverseWithNeighbourContainer, the all-pairs tiles).  uammd.h keeps the
// specialisation PairForces<Potential::LJ, NL> — the library's fused Lennard-Jones path, no hipcc needed; everything else lands here:
//
//   struct Yukawa {                                   // a user's radial functor, exactly as the reference documents it
//     struct InputPairParameters { real cutOff, kappa, strength; };
//     struct PairParameters { real cutOff2, kappa, strength; };
//     static PairParameters processPairParameters(InputPairParameters in) { return {in.cutOff * in.cutOff, in.kappa, in.strength}; }
//     __device__ real force(real r2, PairParameters p) { ... |f| / r ... }
//     __device__ real energy(real r2, PairParameters p) { ... }
//   };
//   auto pot = std::make_shared<Potential::Radial<Yukawa>>();
//   pot->setPotParameters(0, 0, {2.5, 1.0, 3.0});
//   PairForces<Potential::Radial<Yukawa>>::Parameters par; par.box = box;
//   auto pf = std::make_shared<PairForces<Potential::Radial<Yukawa>>>(pd, par, pot);
//   integrator->addInteractor(pf);
//
//   hipcc --offload-arch=gfx950 -std=c++17 -ffp-contract=off -Iinclude/uammd -Iinclude sim.hip -Luammd_amd/lib -luammd_hip
#ifndef UAMMD_MI355X_PAIRFORCES_HIP_HPP
#define UAMMD_MI355X_PAIRFORCES_HIP_HPP

#include "../uammd.h"
#include "ForceEnergyVirial.hpp"
#include "Transverser.hip.hpp"

#include <algorithm>
#include <vector>

namespace uammd {

namespace Potential {

// the table of per-type-pair parameters: symmetric, grows with the largest type seen, entry (0, 0) answers for types it does not hold
template <class Functor> class BasicParameterHandler {
  using PairParameters = typename Functor::PairParameters;
  std::vector<PairParameters> host;
  detail::DeviceArray<PairParameters> device;
  int ntypes = 1;
  real cutOff = 0;
  bool stale = true;
public:
  BasicParameterHandler() : host(1) {}
  void add(int ti, int tj, typename Functor::InputPairParameters p) {
    cutOff = std::max((real)p.cutOff, cutOff);
    const int grown = std::max(ntypes, std::max(ti, tj) + 1);
    if (grown != ntypes) {
      std::vector<PairParameters> wider((size_t)grown * grown);
      for (int j = 0; j < ntypes; ++j)
        for (int i = 0; i < ntypes; ++i) wider[i + (size_t)grown * j] = host[i + (size_t)ntypes * j];
      host.swap(wider);
      ntypes = grown;
    }
    host[ti + (size_t)ntypes * tj] = host[tj + (size_t)ntypes * ti] = Functor::processPairParameters(p);
    stale = true;
  }
  real getCutOff() const { return cutOff; }
  struct Iterator {
    const PairParameters *table;
    int ntypes;
    __device__ PairParameters operator()(int ti, int tj) const {
      if (ntypes == 1) return table[0];
      if (ti > tj) { const int t = ti; ti = tj; tj = t; }
      return table[(ti >= ntypes || tj >= ntypes) ? 0 : ti + ntypes * tj];
    }
  };
  Iterator getIterator() {
    if (stale) {
      device.resize(host.size());
      detail::hipCheck(hipMemcpy(device.d, host.data(), sizeof(PairParameters) * host.size(), hipMemcpyHostToDevice), "hipMemcpy");
      stale = false;
    }
    return Iterator{device.d, ntypes};
  }
};

template <class PotentialFunctor, class ParameterHandle = BasicParameterHandler<PotentialFunctor>> class Radial {
public:
  using InputPairParameters = typename PotentialFunctor::InputPairParameters;
protected:
  std::shared_ptr<PotentialFunctor> pot;
  std::shared_ptr<ParameterHandle> pairParameters;
public:
  Radial() : Radial(std::make_shared<PotentialFunctor>()) {}
  explicit Radial(std::shared_ptr<PotentialFunctor> functor) : pot(functor), pairParameters(std::make_shared<ParameterHandle>()) {}
  void setPotParameters(int ti, int tj, InputPairParameters p) { pairParameters->add(ti, tj, p); }
  real getCutOff() { return pairParameters->getCutOff(); }

  // RadialPotential.cuh:86-127: r12 = rj - ri under the box's minimum image; nothing for a pair at zero distance; force = (|f| / r) r12,
  // virial = f . r12, each only where its output array was asked for; set() adds the particle's total to the arrays
  struct Transverser {
    typename ParameterHandle::Iterator typeParameters;
    device::ListGrid box;   // (apply_pbc with the library's FMA placement)
    PotentialFunctor pot;
    real4 *force;
    real *energy, *virial;
    __device__ real getCutOff2BetweenTypes(int ti, int tj) { return typeParameters(ti, tj).cutOff2; }
    __device__ ForceEnergyVirial compute(const real4 &ri, const real4 &rj) {
      const real3 r12 = box.apply_pbc(make_real3(rj) - make_real3(ri));
      const auto params = typeParameters((int)ri.w, (int)rj.w);
      const real r2 = dot(r12, r12);
      if (r2 == real(0.0)) return ForceEnergyVirial{real3(0, 0, 0), 0, 0};
      const real E = energy ? pot.energy(r2, params) : real(0);
      const real3 F = (force || virial) ? pot.force(r2, params) * r12 : real3(0, 0, 0);
      const real V = virial ? dot(F, r12) : real(0);
      return ForceEnergyVirial{F, E, V};
    }
    __device__ void set(int i, ForceEnergyVirial total) {
      if (force) force[i] += make_real4(total.force, 0);
      if (energy) energy[i] += total.energy;
      if (virial) virial[i] += total.virial;
    }
  };
  Transverser getTransverser(Interactor::Computables comp, Box box, shared_ptr<ParticleData> pd) {
    uammd_celllist_data geometry{};  // (ListGrid takes the box from a list POD: only the box fields matter here)
    geometry.cellDim[0] = geometry.cellDim[1] = geometry.cellDim[2] = 1;
    float L[3]; int per[3];
    box.toArrays(L, per);
    for (int d = 0; d < 3; ++d) { geometry.boxSize[d] = L[d]; geometry.periodic[d] = per[d]; }
    real4 *f = comp.force ? pd->getForce(access::gpu, access::readwrite).raw() : nullptr;
    real *e = comp.energy ? pd->getEnergy(access::gpu, access::readwrite).raw() : nullptr;
    real *v = comp.virial ? pd->getVirial(access::gpu, access::readwrite).raw() : nullptr;
    return Transverser{pairParameters->getIterator(), device::ListGrid(geometry), *pot, f, e, v};
  }
};

}  // namespace Potential

namespace pairforces_detail {
template <class List> shared_ptr<List> makeList(shared_ptr<ParticleData> pd, shared_ptr<ParticleGroup> pg, List *) {
  return pg ? make_shared<List>(pg) : make_shared<List>(pd);  // both lists take a group (CellList.cuh:132, VerletList.cuh:96)
}
template <class Tr> int transverseList(CellList &nl, Tr &tr, const int *globalIndex, hipStream_t st) {
  return device::transverseList(nl.handle(), tr, st, globalIndex);
}
template <class Tr> int transverseList(VerletList &nl, Tr &tr, const int *globalIndex, hipStream_t st) {
  return device::transverseList(nl.handle(), tr, st, globalIndex);
}
}  // namespace pairforces_detail

// The primary template (uammd.h declares it and specialises it for Potential::LJ): any MyPotential with getCutOff() and
// getTransverser(Computables, Box, shared_ptr<ParticleData>), on the CellList or the VerletList.
namespace pairforces_detail {
// ParameterUpdatableDelegate<Potential> (misc/ParameterUpdatable.h:84-103, PairForces.cuh:25,40-44, PairForces.cu:40): every parameter
// update PairForces receives goes on to its Potential when the Potential is ParameterUpdatable — a Potential whose strength follows the
// simulation time or the temperature (examples/advanced/ParameterUpdatable.cu:126-201) depends on it
template <class P> std::enable_if_t<std::is_base_of<ParameterUpdatable, P>::value, ParameterUpdatable *> updatable(P *p) { return p; }
template <class P> std::enable_if_t<!std::is_base_of<ParameterUpdatable, P>::value, ParameterUpdatable *> updatable(P *) { return nullptr; }
}  // namespace pairforces_detail

template <class MyPotential, class NL> class PairForces : public Interactor {
  Box box;
  shared_ptr<MyPotential> pot;
  shared_ptr<NL> nl;
  ParameterUpdatable *delegate() { return pairforces_detail::updatable(pot.get()); }
public:
  struct Parameters { Box box; shared_ptr<NL> nl = nullptr; };
  PairForces(shared_ptr<ParticleData> pd, Parameters par, shared_ptr<MyPotential> pot = make_shared<MyPotential>())
      : Interactor(pd, "PairForces"), box(par.box), pot(pot), nl(par.nl) {}
  PairForces(shared_ptr<ParticleGroup> pg, Parameters par, shared_ptr<MyPotential> pot = make_shared<MyPotential>())
      : Interactor(pg, "PairForces"), box(par.box), pot(pot), nl(par.nl) {}
  void updateBox(Box b) override { box = b; if (auto *d = delegate()) d->updateBox(b); }
  void updateTimeStep(real v) override { if (auto *d = delegate()) d->updateTimeStep(v); }
  void updateSimulationTime(real v) override { if (auto *d = delegate()) d->updateSimulationTime(v); }
  void updateTemperature(real v) override { if (auto *d = delegate()) d->updateTemperature(v); }
  void updateViscosity(real v) override { if (auto *d = delegate()) d->updateViscosity(v); }
  shared_ptr<MyPotential> getPotential() { return pot; }
  template <class Transverser> void sumTransverser(Transverser &tr, hipStream_t st) {  // PairForces.cu:43-68
    const real rcut = pot->getCutOff();
    const bool useNeighbourList = !(box.boxSize.x <= 3 * rcut && box.boxSize.y <= 3 * rcut && box.boxSize.z <= 3 * rcut);
    const int *globalIndex = subgroup ? subgroup->getIndicesRawPtr(access::gpu) : nullptr;
    if (useNeighbourList) {
      if (!nl) nl = pairforces_detail::makeList(pd, subgroup, (NL *)nullptr);
      nl->update(box, rcut, st);
      if (pairforces_detail::transverseList(*nl, tr, globalIndex, st) != 0)
        throw cuda_generic_error(std::string("PairForces: traversal failed: ") + uammd_hip_last_error(), -1);
    } else {
      const int N = subgroup ? subgroup->getNumberParticles() : pd->getNumParticles();
      auto pos = pd->getPos(access::gpu, access::read);
      if (device::transverseNBody(pos.raw(), globalIndex, tr, N, st) != 0) throw cuda_generic_error("PairForces: all-pairs traversal failed", -1);
    }
  }
  void sum(Computables comp, hipStream_t st = 0) override {  // PairForces.cu:70-78
    auto tr = pot->getTransverser(comp, box, pd);
    sumTransverser(tr, st);
  }
};

}  // namespace uammd
#endif
