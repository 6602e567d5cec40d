// BD::EulerMaruyama / MidPoint / AdamsBashforth / Leimkuhler in the DOUBLE_PRECISION build (Integrator/BrownianDynamics.cuh:57-183 with
// real = double, as test/BD/Makefile:2 builds the reference's BD test): a particle pulled by a constant force at T = 0 moves by
// dt M F per step to rounding, a sheared box advects it, and without forces the mean square displacement per axis after n steps is
// 2 T M dt n (the quantity test/BD/test.bash fits) for all four schemes.
#include "uammd.cuh"
#include "Integrator/BrownianDynamics.cuh"
#include <cstdio>
#include <random>
#include <vector>
using namespace uammd;
static_assert(std::is_same<real, double>::value, "this test is the DOUBLE_PRECISION build's");

struct Pull : public Interactor {
  using Interactor::Interactor;
  void sum(Computables, hipStream_t) override {
    auto f = pd->getForce(access::cpu, access::readwrite);
    for (auto &v : f) v = v + make_real4(0.25, -0.5, 1.0, 0);
  }
};

static BD::Parameters parameters(real T) {
  BD::Parameters par;
  par.temperature = T;
  par.viscosity = 1.0 / (6.0 * M_PI);   // self mobility 1 at radius 1
  par.hydrodynamicRadius = 1.0;
  par.dt = 0.125;
  return par;
}

template <class Scheme> static int pulled(shared_ptr<System> sys, const char *name) {
  const int N = 100, steps = 6;
  auto pd = std::make_shared<ParticleData>(N, sys);
  std::vector<real4> start(N);
  {
    std::mt19937 gen(77);
    std::uniform_real_distribution<double> u(-4, 4);
    auto pos = pd->getPos(access::cpu, access::write);
    for (int i = 0; i < N; ++i) start[i] = pos[i] = make_real4(u(gen), u(gen), u(gen), i % 3);
  }
  auto par = parameters(0);
  auto bd = std::make_shared<Scheme>(pd, par);
  bd->addInteractor(std::make_shared<Pull>(pd, "pull"));
  for (int s = 0; s < steps; ++s) bd->forwardTime();
  double worst = 0;
  bool typesKept = true;
  {
    auto pos = pd->getPos(access::cpu, access::read);
    const double d = steps * par.dt;
    for (int i = 0; i < N; ++i) {
      worst = std::max(worst, std::abs(pos[i].x - (start[i].x + 0.25 * d)));
      worst = std::max(worst, std::abs(pos[i].y - (start[i].y - 0.5 * d)));
      worst = std::max(worst, std::abs(pos[i].z - (start[i].z + 1.0 * d)));
      typesKept = typesKept && pos[i].w == start[i].w;
    }
  }
  std::printf("%s: %d steps under a constant force, worst |x - (x0 + n dt M F)| = %.2e\n", name, steps, worst);
  return (worst < 1e-14 && typesKept) ? 0 : 1;   // (double: a float update would leave 1e-7)
}

static int sheared(shared_ptr<System> sys) {
  auto pd = std::make_shared<ParticleData>(1, sys);
  { auto pos = pd->getPos(access::cpu, access::write); pos[0] = make_real4(1.0, 2.0, -3.0, 0); }
  auto par = parameters(0);
  par.K[0] = make_real3(0, 0.5, 0);   // dx/dt = 0.5 y
  BD::EulerMaruyama bd(pd, par);
  bd.forwardTime();
  real4 p;
  { auto pos = pd->getPos(access::cpu, access::read); p = pos[0]; }
  std::printf("EulerMaruyama: one sheared step, x = %.17g (expected %.17g)\n", (double)p.x, 1.0 + 0.125 * 0.5 * 2.0);
  return (std::abs(p.x - (1.0 + 0.125 * 0.5 * 2.0)) < 1e-15 && p.y == 2.0 && p.z == -3.0) ? 0 : 1;
}

template <class Scheme> static int diffusing(shared_ptr<System> sys, const char *name) {
  const int N = 16384, steps = 8;
  const real T = 0.7;
  auto pd = std::make_shared<ParticleData>(N, sys);
  { auto pos = pd->getPos(access::cpu, access::write); for (int i = 0; i < N; ++i) pos[i] = make_real4(0, 0, 0, 0); }
  auto par = parameters(T);
  auto bd = std::make_shared<Scheme>(pd, par);
  for (int s = 0; s < steps; ++s) bd->forwardTime();
  double msd = 0;
  {
    auto pos = pd->getPos(access::cpu, access::read);
    for (int i = 0; i < N; ++i) msd += (pos[i].x * pos[i].x + pos[i].y * pos[i].y + pos[i].z * pos[i].z) / (3.0 * N);
  }
  const double expected = 2 * T * 1.0 * par.dt * steps;
  // Leimkuhler's noise at step n is (dW_n + dW_(n-1)) / 2: successive steps share a draw, and n steps from rest give 2 T M dt (n - 1/2)
  const double expectedL = 2 * T * 1.0 * par.dt * (steps - 0.5);
  const bool leimkuhler = std::string(name) == "Leimkuhler";
  const double ratio = msd / (leimkuhler ? expectedL : expected);
  std::printf("%s: <x^2> per axis after %d free steps of %d particles = %.5f, expected %.5f (ratio %.4f)\n", name, steps, N, msd,
              leimkuhler ? expectedL : expected, ratio);
  return (ratio > 0.97 && ratio < 1.03) ? 0 : 1;   // (3 N = 49152 samples of a sum of 8 steps: the estimate's own deviation is ~0.7 %)
}

int main(int argc, char *argv[]) {
  auto sys = std::make_shared<System>(argc, argv);
  int bad = 0;
  bad += pulled<BD::EulerMaruyama>(sys, "EulerMaruyama");
  bad += pulled<BD::MidPoint>(sys, "MidPoint");
  bad += pulled<BD::AdamsBashforth>(sys, "AdamsBashforth");
  bad += pulled<BD::Leimkuhler>(sys, "Leimkuhler");
  bad += sheared(sys);
  bad += diffusing<BD::EulerMaruyama>(sys, "EulerMaruyama");
  bad += diffusing<BD::MidPoint>(sys, "MidPoint");
  bad += diffusing<BD::AdamsBashforth>(sys, "AdamsBashforth");
  bad += diffusing<BD::Leimkuhler>(sys, "Leimkuhler");
  std::printf(bad ? "dp_bd: FAILED\n" : "dp_bd: ok\n");
  return bad;
}
