// Interactor/DoublyPeriodic/DPPoissonSlab.cuh — NOT part of the MI355X library: the doubly periodic Poisson solver (Chebyshev in z, a
// boundary value problem per wave number) is outside the scope of this build.  This header exists because
// the reference's triply periodic quadrupole test shares its scaffolding with the doubly periodic one
// (test/Potentials/Poisson/common/quadrupole_test_base.cuh:4,45-47 includes this path and overloads getL on DPPoissonSlab::Parameters): it
// declares the parameter struct that scaffolding names and a class whose construction fails with a message that says so.
#pragma once
#include "../../uammd.h"
#include <stdexcept>
namespace uammd {
class DPPoissonSlab : public Interactor {
public:
  struct Permitivity { real top = 1, bottom = 1, inside = 1; };
  struct Parameters {   // (the fields of the reference's struct that programs set: DPPoissonSlab.cuh:21-38)
    real upsampling = 1.2;
    real2 Lxy = real2();
    int Nxy = -1;
    real H = 0;
    Permitivity permitivity;
    real tolerance = 1e-4;
    real gw = -1;
    int support = 12;
    real numberStandardDeviations = 4;
    real split = -1;
    bool printK0Mode = false;
  };
  DPPoissonSlab(shared_ptr<ParticleGroup> pg, Parameters) : Interactor(pg, "DPPoissonSlab") {
    throw std::runtime_error("[DPPoissonSlab] the doubly periodic Poisson solver is not part of the MI355X build of the library (out of scope: DESIGN.md, section 8)");
  }
  DPPoissonSlab(shared_ptr<ParticleData> pd, Parameters par) : DPPoissonSlab(std::make_shared<ParticleGroup>(pd, "All"), par) {}
  void sum(Computables, hipStream_t = 0) override {}
  std::vector<real4> computeFieldPotentialAtParticles() { return {}; }
};
}  // namespace uammd
