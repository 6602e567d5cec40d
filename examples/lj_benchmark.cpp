// LJ liquid under VerletNVT::GronbechJensen with a CellList — the set-up of the reference's examples/misc/benchmark.cu
// (the C3 workload of BASELINE.json), written against the same class names.
#include "uammd.cuh"
#include "Interactor/NeighbourList/CellList.cuh"
#include "Interactor/NeighbourList/VerletList.cuh"
#include "Interactor/PairForces.cuh"
#include "Interactor/Potential/Potential.cuh"
#include "Integrator/VerletNVT.cuh"
#include <chrono>
#include <cstdio>
using namespace uammd;

int main(int argc, char *argv[]) {
  const int N = argc > 1 ? std::atoi(argv[1]) : 1 << 20;
  const int nsteps = argc > 2 ? std::atoi(argv[2]) : 100;
  const real3 boxSize = make_real3(argc > 3 ? std::atof(argv[3]) : 128);
  auto sys = std::make_shared<System>(argc, argv);
  sys->rng().setSeed(0xf31337Bada55D00dULL);
  auto pd = std::make_shared<ParticleData>(N, sys);
  Box box(boxSize);
  box.setPeriodicity(true, true, true);
  {
    auto pos = pd->getPos(access::cpu, access::write);
    auto initial = initLatticeSC(boxSize, N);
    std::copy(initial.begin(), initial.end(), pos.begin());
  }
  VerletNVT::GronbechJensen::Parameters par;
  par.temperature = 1.0;
  par.dt = 0.005;
  par.friction = 1.0;
  auto verlet = std::make_shared<VerletNVT::GronbechJensen>(pd, par);
#ifdef USE_CELLLIST
  using NeighbourList = CellList;
#else
  using NeighbourList = VerletList;  // examples/misc/benchmark.cu:82-84
#endif
  using PairForces = PairForces<Potential::LJ, NeighbourList>;
  auto pot = std::make_shared<Potential::LJ>();
  {
    Potential::LJ::InputPairParameters p;
    p.epsilon = 1.0; p.shift = false; p.sigma = 1; p.cutOff = 2.5 * p.sigma;
    pot->setPotParameters(0, 0, p);
  }
  PairForces::Parameters params;
  params.box = box;
  auto pairforces = std::make_shared<PairForces>(pd, params, pot);
  verlet->addInteractor(pairforces);
  for (int i = 0; i < 20; i++) verlet->forwardTime();  // warm up
  sys->finish();
  const auto t0 = std::chrono::steady_clock::now();
  for (int j = 0; j < nsteps; j++) {
    verlet->forwardTime();
    if (j % 500 == 0) pd->sortParticles();
  }
  sys->finish();
  const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  double ekin = 0;
  {
    auto vel = pd->getVel(access::cpu, access::read);
    for (auto v : vel) ekin += 0.5 * ((double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z);
  }
  const double T = 2 * ekin / (3.0 * N);
  std::printf("N %d steps %d ms_per_step %.4f particle_steps_per_s %.4g kinetic_temperature %.4f\n", N, nsteps, 1e3 * s / nsteps,
              (double)N * nsteps / s, T);
  // the lattice releases potential energy while it melts: early on T overshoots the thermostat value
  return (std::isfinite(T) && T > 0.5 && T < 3.0) ? 0 : 1;
}
