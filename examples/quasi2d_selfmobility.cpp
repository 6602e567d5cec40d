// BDHI::Quasi2D and BDHI::True2D pulling one particle: the self mobilities of the reference's test/BDHI/quasi2D/quasi2d_test.cu
// (computeSelfMobility, :56-138).  Plain g++ (C++14), linked against libuammd_hip.so.
#include "uammd.cuh"
#include "Integrator/Hydro/BDHI_quasi2D.cuh"
#include "Integrator/BDHI/FIB.cuh"
#include "Integrator/Hydro/ICM.cuh"
#include <cmath>
#include <cstdio>
using namespace uammd;

struct Pull : public Interactor {
  using Interactor::Interactor;
  void sum(Computables, hipStream_t) override {
    auto f = pd->getForce(access::cpu, access::write);
    f[0] = make_real4(1, 0, 0, 0);
  }
};

template <class Scheme> static double selfMobility(shared_ptr<System> sys, real lbox, real a) {
  auto pd = std::make_shared<ParticleData>(1, sys);
  typename Scheme::Parameters par;
  par.temperature = 0; par.viscosity = 1.12312; par.dt = 0.1; par.hydrodynamicRadius = a; par.box = Box(make_real3(lbox, lbox, 0));
  auto bdhi = std::make_shared<Scheme>(pd, par);
  bdhi->addInteractor(std::make_shared<Pull>(pd, "puller"));
  double M = 0;
  const int ntest = 20;
  for (int i = 0; i < ntest; ++i) {
    const real4 p0 = make_real4(sys->rng().uniform(-0.5, 0.5) * lbox, sys->rng().uniform(-0.5, 0.5) * lbox, 0, 0);
    { auto pos = pd->getPos(access::cpu, access::write); pos[0] = p0; }
    bdhi->forwardTime();
    auto pos = pd->getPos(access::cpu, access::read);
    M += (double)pos[0].x - (double)p0.x;
  }
  return par.viscosity * M / (ntest * par.dt);
}

int main(int argc, char *argv[]) {
  auto sys = std::make_shared<System>(argc, argv);
  const real a = 1.21312;
  int bad = 0;
  for (real lbox : {32.0, 128.0}) {
    const double mq = selfMobility<BDHI::Quasi2D>(sys, lbox * a, a), tq = 1.0 / (6 * M_PI * a) / (1 + 4.41 / lbox);
    const double mt = selfMobility<BDHI::True2D>(sys, lbox * a, a), tt = (std::log(lbox) - 1.3105329259115095183) / (4 * M_PI);
    std::printf("L/a = %3.0f  Quasi2D %.5f (theory %.5f)   True2D %.5f (theory %.5f)\n", (double)lbox, mq, tq, mt, tt);
    bad += !(std::abs(mq - tq) < 1e-3) + !(std::abs(mt - tt) < 1e-3);
  }
  {  // BDHI::FIB: the same pull in a cubic box, against FIB::getSelfMobility() (+- 1 %, FIB.cuh:35-37)
    auto pd = std::make_shared<ParticleData>(1, sys);
    BDHI::FIB::Parameters par;
    par.temperature = 0; par.viscosity = 1.0; par.dt = 0.01; par.hydrodynamicRadius = 1.0; par.box = Box(64.0);
    auto fib = std::make_shared<BDHI::FIB>(pd, par);
    fib->addInteractor(std::make_shared<Pull>(pd, "puller"));
    double M = 0;
    for (int i = 0; i < 20; ++i) {
      const real4 p0 = make_real4(sys->rng().uniform(-32, 32), sys->rng().uniform(-32, 32), sys->rng().uniform(-32, 32), 0);
      { auto pos = pd->getPos(access::cpu, access::write); pos[0] = p0; }
      fib->forwardTime();
      auto pos = pd->getPos(access::cpu, access::read);
      M += ((double)pos[0].x - (double)p0.x) / (20 * par.dt);
    }
    std::printf("FIB  a = %.4f  mobility %.5f (getSelfMobility %.5f)\n", (double)fib->getHydrodynamicRadius(), M, (double)fib->getSelfMobility());
    bad += !(std::abs(M / fib->getSelfMobility() - 1) < 0.02);
  }
  {  // Hydro::ICM: a fluid at rest stays at rest, a tethered-free particle in it does not move, and the API answers
    auto pd = std::make_shared<ParticleData>(8, sys);
    Hydro::ICM::Parameters par;
    par.temperature = 0; par.viscosity = 1.0; par.density = 1.0; par.dt = 0.01; par.hydrodynamicRadius = 1.0; par.box = Box(32.0);
    auto icm = std::make_shared<Hydro::ICM>(pd, par);
    icm->addInteractor(std::make_shared<Pull>(pd, "puller"));
    for (int i = 0; i < 50; ++i) icm->forwardTime();
    const auto n = icm->getNumberFluidCells();
    const real3 *v = icm->getFluidVelocities(access::cpu);
    double vmax = 0, mom = 0;
    for (int i = 0; i < n.x * n.y * n.z; ++i) { vmax = std::max(vmax, (double)std::abs(v[i].x)); mom += v[i].x; }
    real4 p0;
    { auto pos = pd->getPos(access::cpu, access::read); p0 = pos[0]; }
    std::printf("ICM  a = %.4f  cells %d^3  particle 0 moved to x = %.5f, max |v_x| = %.3e, mean v_x = %.1e\n",
                (double)icm->getHydrodynamicRadius(), n.x, (double)p0.x, vmax, mom / (n.x * n.y * n.z));
    bad += !(p0.x > 0 && vmax > 0 && std::abs(mom / (n.x * n.y * n.z)) < 1e-6 * vmax + 1e-9);
  }
  sys->finish();
  return bad;
}
