// The reference README's first program (README.md "Brownian Dynamics of non-interacting particles"),
// compiled against the MI355X library with a plain C++14 compiler:
//   g++ -std=c++14 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I../include/uammd bd_readme.cpp \
//       -L../uammd_amd/lib -luammd_hip -L/opt/rocm/lib -lamdhip64 -o bd_readme
#include "uammd.cuh"
#include "Integrator/BrownianDynamics.cuh"
#include <cstdio>
using namespace uammd;

int main(int argc, char *argv[]) {
  int numberParticles = argc > 1 ? std::atoi(argv[1]) : 100000;
  auto sys = std::make_shared<System>(argc, argv);
  sys->rng().setSeed(1234);
  auto pd = std::make_shared<ParticleData>(numberParticles, sys);
  {
    auto pos = pd->getPos(access::cpu, access::write);
    std::generate(pos.begin(), pos.end(), [&]() { return make_real4(sys->rng().uniform3(-0.5, 0.5), 0); });
  }
  BD::EulerMaruyama::Parameters par;
  par.temperature = 1.0;
  par.viscosity = 1.0;
  par.hydrodynamicRadius = 1.0;
  par.dt = 0.1;
  auto bd = std::make_shared<BD::EulerMaruyama>(pd, par);
  const int nsteps = 100;
  for (int i = 0; i < nsteps; i++) bd->forwardTime();
  // free diffusion: <dr^2> = 6 D t + <r0^2>, D = T/(6 pi eta a)
  double msd = 0;
  {
    auto pos = pd->getPos(access::cpu, access::read);
    for (auto p : pos) msd += (double)p.x * p.x + (double)p.y * p.y + (double)p.z * p.z;
  }
  msd /= numberParticles;
  const double D = par.temperature / (6 * M_PI * par.viscosity * par.hydrodynamicRadius);
  const double expected = 6 * D * nsteps * par.dt + 0.25;
  std::printf("msd %.6f expected %.6f ratio %.4f\n", msd, expected, msd / expected);
  sys->finish();
  return std::abs(msd / expected - 1) < 0.05 ? 0 : 1;
}
