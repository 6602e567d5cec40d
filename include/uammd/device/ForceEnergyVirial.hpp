// what the Transverser of a pair potential, or an external-force functor, returns per particle (Interactor/Potential/PotentialBase.cuh,
// Interactor/ExternalForces.cuh): user functors write `return {force, energy, virial};`
#ifndef UAMMD_MI355X_FORCEENERGYVIRIAL_HPP
#define UAMMD_MI355X_FORCEENERGYVIRIAL_HPP
#include "../utils/vector.cuh"
namespace uammd {
struct ForceEnergyVirial {
  real3 force;
  real energy, virial;
};
UAMMD_HD ForceEnergyVirial operator+(const ForceEnergyVirial &a, const ForceEnergyVirial &b) {
  return {a.force + b.force, a.energy + b.energy, a.virial + b.virial};
}
}  // namespace uammd
#endif
