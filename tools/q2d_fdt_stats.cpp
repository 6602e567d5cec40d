// The fluctuation-dissipation check of the reference's test/BDHI/quasi2D/quasi2d_test.cu:131-186 with more samples: the test takes
// 50 000 one-step displacements of one particle and asks the variance to match 2 T dt M within 1 % — 1.6 standard deviations of its own
// estimator, so a correct sampler fails a run's four components about 38 % of the time.  Here: NAVG samples (default 800 000: 0.16 % per
// component) to tell a bias from that scatter.   usage: q2d_fdt_stats [navg]      (built with -DDOUBLE_PRECISION like the test)
#include "uammd.cuh"
#include "Integrator/Hydro/BDHI_quasi2D.cuh"
#include <cstdio>
#include <cstdlib>
#include <random>
using namespace uammd;

struct Puller : public Interactor {
  using Interactor::Interactor;
  void sum(Computables, hipStream_t) override { pd->getForce(access::cpu, access::write)[0] = make_real4(1, 0, 0, 0); }
};

template <class Scheme> static std::shared_ptr<Scheme> create(std::shared_ptr<ParticleData> pd, real T, real dt, real viscosity, real a, real lbox) {
  typename Scheme::Parameters par;
  par.temperature = T; par.viscosity = viscosity; par.dt = dt; par.hydrodynamicRadius = a;
  par.box = Box(make_real3(lbox, lbox, 0));
  return std::make_shared<Scheme>(pd, par);
}

template <class Scheme> static void run(const char *name, long navg) {
  const real a = 1.21312, T = 1.012312, dt = 0.9, lbox = 128 * a;
  std::mt19937 gen(20260930);
  std::uniform_real_distribution<> dis(-lbox * 0.5, lbox * 0.5);
  double M = 0;
  {
    auto pd = std::make_shared<ParticleData>(1);
    auto bdhi = create<Scheme>(pd, 0, 0.1, 1.12312, a, lbox);
    bdhi->addInteractor(std::make_shared<Puller>(pd, "puller"));
    const int ntest = 2000;
    for (int i = 0; i < ntest; ++i) {
      const real4 p0 = make_real4(dis(gen), dis(gen), 0, 0);
      pd->getPos(access::cpu, access::write)[0] = p0;
      bdhi->forwardTime();
      M += (pd->getPos(access::cpu, access::read)[0].x - p0.x) * 1.12312 / (ntest * 0.1);
    }
  }
  auto pd = std::make_shared<ParticleData>(1);
  auto bdhi = create<Scheme>(pd, T, dt, 1, a, lbox);
  double sx = 0, sy = 0, mx = 0, my = 0;
  for (long i = 0; i < navg; ++i) {
    const real4 p0 = make_real4(dis(gen), dis(gen), 0, 0);
    pd->getPos(access::cpu, access::write)[0] = p0;
    bdhi->forwardTime();
    const real4 p = pd->getPos(access::cpu, access::read)[0];
    const double dx = p.x - p0.x, dy = p.y - p0.y;
    sx += dx * dx; sy += dy * dy; mx += dx; my += dy;
  }
  const double dxx = sx / navg / (2 * T * dt), dyy = sy / navg / (2 * T * dt), sigma = std::sqrt(2.0 / navg);
  std::printf("%s: M (2000 pulls) = %.6f;  <dx^2> / (2 T dt) = %.6f (%+.3f %%), <dy^2> / (2 T dt) = %.6f (%+.3f %%);  estimator's sigma %.3f %%;  mean step %.2e %.2e\n",
              name, M, dxx, 100 * (dxx / M - 1), dyy, 100 * (dyy / M - 1), 100 * sigma, mx / navg, my / navg);
}

int main(int argc, char **argv) {
  const long navg = argc > 1 ? std::atol(argv[1]) : 800000;
  auto sys = std::make_shared<System>();
  (void)sys;
  run<BDHI::Quasi2D>("Quasi2D", navg);
  run<BDHI::True2D>("True2D", navg);
  return 0;
}
