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

// a constant torque on particle 0 (the reference's Interactors write pd->getTorque the same way)
struct TwistZ : public Interactor {
  using Interactor::Interactor;
  void sum(Computables, hipStream_t) override {
    auto t = pd->getTorque(access::cpu, access::readwrite);
    t[0] = make_real4(0, 0, 1, 0);
  }
};

// rotational self mobility with orientations allocated: dir = rotVec2Quaternion(w dt) * dir, w = tau/(8 pi eta a^3)
// up to the periodic correction (small at a/L = 1/32); also run with the six-point window FCM_impl.cuh:42 names.
template <class Integrator> static int rotation(shared_ptr<System> sys, const char *name, double tolerance) {
  auto pd = std::make_shared<ParticleData>(1, sys);
  { auto pos = pd->getPos(access::cpu, access::write); pos[0] = make_real4(0.3, -1.2, 2.1, 0); }
  { auto dir = pd->getDir(access::cpu, access::write); dir[0] = make_real4(1, 0, 0, 0); }
  typename Integrator::Parameters par;
  par.temperature = 0; par.viscosity = 1.0; par.hydrodynamicRadius = 1.0; par.dt = 0.5; par.box = Box(32.0); par.tolerance = 1e-3;
  auto fcm = std::make_shared<Integrator>(pd, par);
  fcm->addInteractor(std::make_shared<TwistZ>(pd, "twist"));
  fcm->forwardTime();
  real4 q;
  { auto dir = pd->getDir(access::cpu, access::read); q = dir[0]; }
  const double a = fcm->getFCM_impl()->getHydrodynamicRadius();
  const double w = 2 * std::atan2((double)q.w, (double)q.x) / par.dt;  // rotation about z: q = (cos(phi/2), 0, 0, sin(phi/2))
  const double w0 = 1.0 / (8 * M_PI * par.viscosity * a * a * a);
  std::printf("%s: angular velocity %.6f, isolated sphere %.6f (a = %.5f), |q| = %.7f\n", name, w, w0, a,
              std::sqrt((double)q.x * q.x + (double)q.y * q.y + (double)q.z * q.z + (double)q.w * q.w));
  return (std::abs(w / w0 - 1) < tolerance && std::abs(q.y) < 1e-6 && std::abs(q.z) < 1e-6) ? 0 : 1;
}

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
  int bad = std::abs(M / M0 - 1) < 2e-3 ? 0 : 1;
  {  // library-mode use of FCM_impl with the reference's own signature (FCM_impl.cuh:126-129): owning containers by value
    auto pos = pd->getPos(access::gpu, access::read);
    auto force = pd->getForce(access::gpu, access::read);
    auto disp = fcm->getFCM_impl()->computeHydrodynamicDisplacements(const_cast<real4 *>(pos.raw()), const_cast<real4 *>(force.raw()), nullptr,
                                                                      pd->getNumParticles(), 0, 0, 0);
    real3 v0;
    detail::hipCheck(hipMemcpy(&v0, disp.first.data().get(), sizeof(real3), hipMemcpyDeviceToHost), "hipMemcpy");
    std::printf("by-value displacements: %zu linear, %zu angular entries, v0.x = %.6f\n", disp.first.size(), disp.second.size(), v0.x);
    bad += disp.first.size() != (size_t)pd->getNumParticles() || !disp.second.empty() || std::abs(v0.x / M0 - 1) > 2e-3;
  }
  bad += rotation<BDHI::FCMIntegrator>(sys, "Gaussian", 0.02);
  bad += rotation<BDHI::FCMIntegratorT<BDHI::FCM_ns::Kernels::GaussianFlexible::sixPoint>>(sys, "sixPoint", 0.15);
  sys->finish();
  return bad;
}
