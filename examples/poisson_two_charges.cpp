// Poisson Interactor on the configuration of the reference's SingleSimulationTest
// (test/Potentials/Poisson/TriplyPeriodic/test_poisson.cu:189-222): a unit charge and two half charges of opposite sign at
// distance r in a box of L = 100; the force and the field on the first charge are compared with the free-space field of a
// Gaussian source.  Plain g++ (C++14), linked against libuammd_hip.so.
#include "uammd.cuh"
#include "Interactor/SpectralEwaldPoisson.cuh"
#include <cmath>
#include <cstdio>
using namespace uammd;

int main(int argc, char *argv[]) {
  auto sys = std::make_shared<System>(argc, argv);
  const real L = 100.0, r = 2.0, gw = 0.001;
  auto pd = std::make_shared<ParticleData>(3, sys);
  {
    auto pos = pd->getPos(access::cpu, access::write);
    auto charge = pd->getCharge(access::cpu, access::write);
    const real ox = 13.7, oy = -31.2, oz = 44.9;
    pos[0] = make_real4(-r * 0.5 + ox, oy, oz, 0);
    pos[1] = make_real4(r * 0.5 + ox, oy, oz, 0);
    pos[2] = make_real4(r * 0.5 + ox, oy, oz, 0);
    charge[0] = 1.0;
    charge[1] = -0.5;
    charge[2] = -0.5;
  }
  Poisson::Parameters par;
  par.box = Box(L);
  par.epsilon = 1;
  par.gw = gw;
  par.tolerance = 1e-7;
  par.split = 0.2;
  auto poisson = std::make_shared<Poisson>(pd, par);
  {
    auto force = pd->getForce(access::cpu, access::write);
    for (auto &f : force) f = make_real4(0);
  }
  Interactor::Computables comp;
  comp.force = true;
  poisson->sum(comp, 0);
  real4 f0;
  { auto force = pd->getForce(access::cpu, access::read); f0 = force[0]; }
  auto field = poisson->computeFieldPotentialAtParticles();
  const double pi = M_PI;
  const double theory = -std::exp(-r * r / (4.0 * gw * gw)) / (4 * pi * std::sqrt(pi) * gw * r) - std::erf(r / (2.0 * gw)) / (4 * pi * r * r);
  std::printf("grid %d^3, support %d, near cut-off %.4f\n", poisson->getCells().x, poisson->getSupport(), poisson->getNearFieldCutOff());
  std::printf("force.x %.8f field.x %.8f |theory| %.8f\n", f0.x, field[0].x, std::abs(theory));
  int bad = 0;
  bad += !(f0.x > 0 && std::abs(1.0 - std::abs(f0.x / theory)) < 1e-3 && std::abs(f0.y) < 1e-6 && std::abs(f0.z) < 1e-6);
  bad += !(field[0].x > 0 && std::abs(1.0 - std::abs(field[0].x / theory)) < 1e-3);
  // error behaviour: std::invalid_argument as in the reference's constructor
  try {
    Poisson::Parameters p2 = par;
    p2.box = Box(8.0); p2.gw = 0.5; p2.tolerance = 1e-6;
    Poisson bad_one(pd, p2);
    bad++;
  } catch (const std::invalid_argument &e) {
    std::printf("expected error: %s\n", e.what());
  }
  sys->finish();
  return bad;
}
