// test/BDHI/PSE/pse_test.cu "SelfMobilityIsCorrectUpToTolerance" against the C++ interface: pull one particle with BDHI::PSE
// and compare with the Hasimoto-corrected mobility; then one BDHI::EulerMaruyama<BDHI::PSE> step at T > 0.
#include "uammd.cuh"
#include "Integrator/BDHI/BDHI_EulerMaruyama.cuh"
#include "Integrator/BDHI/BDHI_Lanczos.cuh"
#include "Integrator/BDHI/BDHI_PSE.cuh"
#include <cstdio>
using namespace uammd;

int main(int argc, char *argv[]) {
  auto sys = std::make_shared<System>(argc, argv);
  sys->rng().setSeed(1234);
  const real rh = 1.012312;
  BDHI::PSE::Parameters par;
  par.viscosity = 1.12321;
  par.tolerance = 1e-4;
  par.dt = 1;
  par.box = Box(make_real3(32 * rh));
  par.hydrodynamicRadius = rh;
  par.psi = 1.0;
  par.temperature = 0;
  auto pd = std::make_shared<ParticleData>(1, sys);
  { auto pos = pd->getPos(access::cpu, access::write); pos[0] = make_real4(3.1, -7.2, 11.3, 0); }
  auto pse = std::make_shared<BDHI::PSE>(pd, par);
  detail::DeviceArray<real4> force(1);
  detail::DeviceArray<real3> MF(1);
  const real4 fx = make_real4(0, 1, 0, 0);
  (void)hipMemcpy(force.d, &fx, sizeof(fx), hipMemcpyHostToDevice);
  pse->computeHydrodynamicDisplacements(force.d, MF.d, 0, 0);
  real3 v;
  (void)hipMemcpy(&v, MF.d, sizeof(v), hipMemcpyDeviceToHost);
  const double m0 = pse->getSelfMobility();
  std::printf("PSE self mobility %.7f expected %.7f (off-diagonal %.2e %.2e)\n", v.y, m0, v.x, v.z);
  bool ok1 = std::abs(v.y - m0) < par.tolerance && std::abs(v.x) < par.tolerance && std::abs(v.z) < par.tolerance;
  {  // computeMF = computeMFFarField + computeMFNearField (BDHI_PSE.cuh:92-120), the forces taken from the ParticleData
    { auto f = pd->getForce(access::cpu, access::write); f[0] = fx; }
    detail::DeviceArray<real3> whole(1), parts(1);
    pse->computeMF(whole.d, 0);
    detail::check(uammd_fill_zero(parts.d, sizeof(real3), nullptr));
    pse->computeMFFarField(parts.d, 0);
    pse->computeMFNearField(parts.d, 0);
    pse->computeDivM(parts.d, 0);
    real3 a, b;
    (void)hipMemcpy(&a, whole.d, sizeof(a), hipMemcpyDeviceToHost);
    (void)hipMemcpy(&b, parts.d, sizeof(b), hipMemcpyDeviceToHost);
    std::printf("computeMF %.7f = far + near %.7f\n", a.y, b.y);
    ok1 = ok1 && std::abs(a.y - b.y) < 1e-6 && std::abs(a.y - m0) < par.tolerance;
  }
  // a thermal step: the particle must move, finitely
  par.temperature = 1.0;
  par.dt = 0.01;
  auto bd = std::make_shared<BDHI::EulerMaruyama<BDHI::PSE>>(pd, par);
  bd->forwardTime();
  real4 p;
  { auto pos = pd->getPos(access::cpu, access::read); p = pos[0]; }
  const double d2 = (p.x - 3.1) * (p.x - 3.1) + (p.y + 7.2) * (p.y + 7.2) + (p.z - 11.3) * (p.z - 11.3);
  std::printf("thermal displacement^2 %.3e (6 D dt = %.3e)\n", d2, 6 * m0 * par.dt);
  // open-boundary RPY with the Lanczos noise: same integrator template, another Method
  BDHI::Lanczos::Parameters lpar;
  lpar.viscosity = par.viscosity; lpar.hydrodynamicRadius = rh; lpar.temperature = 1.0; lpar.dt = 0.01; lpar.tolerance = 1e-3;
  auto bdl = std::make_shared<BDHI::EulerMaruyama<BDHI::Lanczos>>(pd, lpar);
  bdl->forwardTime();

  // ... and the dense Cholesky variant (rocSOLVER potrf behind the C ABI).  Only with --cholesky: outside a process that already
  // holds rocBLAS (PyTorch does), loading rocBLAS + rocSOLVER from a cold disk takes minutes on the test boxes.
  bool withCholesky = false;
  for (int i = 1; i < argc; ++i) withCholesky = withCholesky || std::string(argv[i]) == "--cholesky";
  if (!withCholesky) { sys->finish(); return (ok1 && std::isfinite(d2) && d2 > 0 && d2 < 100 * 6 * m0 * par.dt) ? 0 : 1; }
  auto bdc = std::make_shared<BDHI::EulerMaruyama<BDHI::Cholesky>>(pd, lpar);
  bdc->forwardTime();

  { auto pos = pd->getPos(access::cpu, access::read); p = pos[0]; }
  if (!std::isfinite(p.x + p.y + p.z)) return 1;
  sys->finish();
  return (ok1 && std::isfinite(d2) && d2 > 0 && d2 < 100 * 6 * m0 * par.dt) ? 0 : 1;
}
