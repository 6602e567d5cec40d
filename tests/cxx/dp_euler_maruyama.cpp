// BDHI::EulerMaruyama<Method> in the DOUBLE_PRECISION build (Method = PSE, FCM: Integrator/BDHI/BDHI_EulerMaruyama.cuh:55-110 with
// real = double): one particle pulled at T = 0 moves by dt M0 F, and at T > 0 without forces a dilute suspension's one-step mean square
// displacement per axis is 2 T M0 dt (test/BDHI/PSE/pse_test.cu:200-260 and test/BDHI/FCM/fcm_test.cu:140-200 check the same two
// quantities through the methods' own computeMF / computeBdW; here they pass through the integrator's position update).
// PSE's near-field noise carries the reference's scaling: NearField::computeStochasticDisplacements multiplies by sqrt(2 T)
// (PSE/NearField.cuh:276) and the integrator by sqrt(2 T dt) again (BDHI_EulerMaruyama.cu:110,150), so that part of the variance is
// 2 T times the fluctuation-dissipation value — invisible at T = 0.5, where the free-diffusion check below runs; at T = 0.7 the excess
// (the near field's share of the self mobility times 0.4) is checked to be there, as it is in the reference.
#include "uammd.cuh"
#include "Integrator/BDHI/BDHI_EulerMaruyama.cuh"
#include "Integrator/BDHI/BDHI_FCM.cuh"
#include "Integrator/BDHI/BDHI_PSE.cuh"
#include <cstdio>
#include <random>
#include <vector>
using namespace uammd;
static_assert(std::is_same<real, double>::value, "this test is the DOUBLE_PRECISION build's");

struct PullX : public Interactor {
  using Interactor::Interactor;
  void sum(Computables, hipStream_t) override {
    auto f = pd->getForce(access::cpu, access::write);
    for (auto &v : f) v = make_real4(0);
    f[0] = make_real4(1, 0, 0, 0);
  }
};

template <class Method> static typename BDHI::EulerMaruyama<Method>::Parameters parameters(real T, real L) {
  typename BDHI::EulerMaruyama<Method>::Parameters par;
  par.temperature = T;
  par.viscosity = 1.0;
  par.hydrodynamicRadius = 1.0;
  par.dt = 0.25;
  par.box = Box(L);
  par.tolerance = 1e-4;
  return par;
}

template <class Method> static int pulled(shared_ptr<System> sys, const char *name) {
  auto pd = std::make_shared<ParticleData>(1, sys);
  { auto pos = pd->getPos(access::cpu, access::write); pos[0] = make_real4(0.3, -1.2, 2.1, 0); }
  auto par = parameters<Method>(0, 32.0);
  auto bdhi = std::make_shared<BDHI::EulerMaruyama<Method>>(pd, par);
  bdhi->addInteractor(std::make_shared<PullX>(pd, "pull"));
  bdhi->forwardTime();
  real4 after;
  { auto pos = pd->getPos(access::cpu, access::read); after = pos[0]; }
  const double M = (after.x - 0.3) / par.dt, M0 = bdhi->getSelfMobility();
  std::printf("%s: pulled particle, mobility %.8f expected %.8f, |dy| + |dz| = %.2e\n", name, M, M0, std::abs(after.y + 1.2) + std::abs(after.z - 2.1));
  return (std::abs(M / M0 - 1) < 1e-3 && std::abs(after.y + 1.2) + std::abs(after.z - 2.1) < 1e-9) ? 0 : 1;
}

template <class Method> static int diffusing(shared_ptr<System> sys, const char *name, real T, double lowest, double highest) {
  const int N = 2048;
  const real L = 128.0;
  auto pd = std::make_shared<ParticleData>(N, sys);
  std::vector<real4> start(N);
  {
    std::mt19937 gen(1234);
    std::uniform_real_distribution<double> u(-0.5 * L, 0.5 * L);
    auto pos = pd->getPos(access::cpu, access::write);
    for (int i = 0; i < N; ++i) start[i] = pos[i] = make_real4(u(gen), u(gen), u(gen), 0);
  }
  auto par = parameters<Method>(T, L);
  auto bdhi = std::make_shared<BDHI::EulerMaruyama<Method>>(pd, par);
  bdhi->forwardTime();
  double msd = 0, mean = 0;
  {
    auto pos = pd->getPos(access::cpu, access::read);
    for (int i = 0; i < N; ++i) {
      const double dx = pos[i].x - start[i].x, dy = pos[i].y - start[i].y, dz = pos[i].z - start[i].z;
      msd += (dx * dx + dy * dy + dz * dz) / (3.0 * N);
      mean += (dx + dy + dz) / (3.0 * N);
    }
  }
  const double expected = 2 * T * bdhi->getSelfMobility() * par.dt;
  std::printf("%s: one free step of %d particles at T = %.1f, <dx^2> = %.6f, 2 T M0 dt = %.6f, <dx> = %.2e\n", name, N, (double)T, msd, expected, mean);
  // 6144 samples: the variance estimate's own deviation is ~ sqrt(2 / 6144) = 1.8 %; the mean's is sqrt(<dx^2> / 6144)
  return (msd / expected > lowest && msd / expected < highest && std::abs(mean) < 5 * std::sqrt(msd / (3.0 * N))) ? 0 : 1;
}

// BDHI::FCMIntegrator with real = double (BDHI_FCM.cuh:155-199): the same two checks through the integrator class of the FCM module
static int fcmIntegrator(shared_ptr<System> sys) {
  int bad = 0;
  BDHI::FCMIntegrator::Parameters par;
  par.viscosity = 1.0; par.hydrodynamicRadius = 1.0; par.dt = 0.25; par.tolerance = 1e-4;
  {
    auto pd = std::make_shared<ParticleData>(1, sys);
    { auto pos = pd->getPos(access::cpu, access::write); pos[0] = make_real4(0.3, -1.2, 2.1, 0); }
    par.temperature = 0; par.box = Box(32.0);
    auto fcm = std::make_shared<BDHI::FCMIntegrator>(pd, par);
    fcm->addInteractor(std::make_shared<PullX>(pd, "pull"));
    fcm->forwardTime();
    real4 after;
    { auto pos = pd->getPos(access::cpu, access::read); after = pos[0]; }
    const double M = (after.x - 0.3) / par.dt, M0 = fcm->getSelfMobility();
    std::printf("FCMIntegrator: pulled particle, mobility %.8f expected %.8f\n", M, M0);
    bad += !(std::abs(M / M0 - 1) < 1e-3 && std::abs(after.y + 1.2) + std::abs(after.z - 2.1) < 1e-9);
  }
  {
    const int N = 2048;
    const real L = 128.0, T = 0.5;
    auto pd = std::make_shared<ParticleData>(N, sys);
    std::vector<real4> start(N);
    {
      std::mt19937 gen(4321);
      std::uniform_real_distribution<double> u(-0.5 * L, 0.5 * L);
      auto pos = pd->getPos(access::cpu, access::write);
      for (int i = 0; i < N; ++i) start[i] = pos[i] = make_real4(u(gen), u(gen), u(gen), 0);
    }
    par.temperature = T; par.box = Box(L);
    auto fcm = std::make_shared<BDHI::FCMIntegrator>(pd, par);
    fcm->forwardTime();
    double msd = 0;
    {
      auto pos = pd->getPos(access::cpu, access::read);
      for (int i = 0; i < N; ++i) {
        const double dx = pos[i].x - start[i].x, dy = pos[i].y - start[i].y, dz = pos[i].z - start[i].z;
        msd += (dx * dx + dy * dy + dz * dz) / (3.0 * N);
      }
    }
    const double expected = 2 * T * fcm->getSelfMobility() * par.dt;
    std::printf("FCMIntegrator: one free step of %d particles at T = %.1f, <dx^2> = %.6f, 2 T M0 dt = %.6f\n", N, (double)T, msd, expected);
    bad += !(msd / expected > 0.92 && msd / expected < 1.08);
  }
  return bad;
}

int main(int argc, char *argv[]) {
  auto sys = std::make_shared<System>(argc, argv);
  int bad = 0;
  bad += fcmIntegrator(sys);
  bad += pulled<BDHI::PSE>(sys, "EulerMaruyama<PSE>");
  bad += pulled<BDHI::FCM>(sys, "EulerMaruyama<FCM>");
  bad += diffusing<BDHI::PSE>(sys, "EulerMaruyama<PSE>", 0.5, 0.92, 1.08);
  bad += diffusing<BDHI::FCM>(sys, "EulerMaruyama<FCM>", 0.5, 0.92, 1.08);
  bad += diffusing<BDHI::FCM>(sys, "EulerMaruyama<FCM>", 0.7, 0.92, 1.08);
  bad += diffusing<BDHI::PSE>(sys, "EulerMaruyama<PSE>", 0.7, 1.08, 1.25);   // (the reference's near-field scaling, above)
  std::printf(bad ? "dp_euler_maruyama: FAILED\n" : "dp_euler_maruyama: ok\n");
  return bad;
}
