// FCM on z slabs of the grid, one process per GPU: uammd::DistributedFCM over uammd::Comm (RCCL behind the C ABI).
//   usage:  fcm_slab <rank> <world> <id file> [cells per axis of ONE slab] [particles per rank]
// The grid is c x c x (world c) with spacing 1; with world = 1 the rank is its own neighbour through the periodic z faces and the result is
// compared with BDHI::FCM_impl of uammd.h on the same particles (1e-5 relative, T = 0 and T > 0: identical noise field).
#include "uammd.cuh"
#include "Distributed.h"
#include "Integrator/BDHI/BDHI_FCM.cuh"
#include <cmath>
#include <cstdio>
#include <cstdlib>
using namespace uammd;

int main(int argc, char *argv[]) {
  if (argc < 4) { std::fprintf(stderr, "usage: %s rank world idfile [cells] [particlesPerRank]\n", argv[0]); return 2; }
  const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]);
  const std::string idfile = argv[3];
  const int c = argc > 4 ? std::atoi(argv[4]) : 64, n = argc > 5 ? std::atoi(argv[5]) : 20000;
  int ndev = 1;
  detail::check(uammd_hip_device_count(&ndev));
  detail::check(uammd_hip_set_device(rank % ndev));
  auto comm = std::make_shared<Comm>(rank, world, exchangeUniqueIdThroughFile(idfile, rank));
  DistributedFCM::Parameters par;
  par.boxSize = make_real3(c, c, (real)c * world);
  par.cells.x = c; par.cells.y = c; par.cells.z = c * world;
  par.viscosity = 1.3;
  par.tolerance = 1e-3;
  par.seed = 1234;
  DistributedFCM fcm(comm, par);
  std::vector<real4> pos(n), force(n);
  Xorshift128plus rng;
  rng.setSeed(77 + rank);
  for (int i = 0; i < n; ++i) {  // z in the window frame: relative to the centre of the owned slab
    pos[i] = make_real4((real)(rng.uniform(-0.5, 0.5) * c), (real)(rng.uniform(-0.5, 0.5) * c), (real)(rng.uniform(-0.5, 0.5) * c), 0);
    force[i] = make_real4((real)rng.uniform(-1, 1), (real)rng.uniform(-1, 1), (real)rng.uniform(-1, 1), 0);
  }
  detail::DeviceArray<real4> dpos(n), dforce(n);
  detail::DeviceArray<real3> dvel(n);
  detail::hipCheck(hipMemcpy(dpos.d, pos.data(), sizeof(real4) * n, hipMemcpyHostToDevice), "hipMemcpy");
  detail::hipCheck(hipMemcpy(dforce.d, force.data(), sizeof(real4) * n, hipMemcpyHostToDevice), "hipMemcpy");
  std::vector<real3> v0(n), v1(n);
  fcm.computeHydrodynamicDisplacements(dpos.d, dforce.d, n, 0, 0, dvel.d);
  detail::hipCheck(hipDeviceSynchronize(), "hipDeviceSynchronize");
  detail::hipCheck(hipMemcpy(v0.data(), dvel.d, sizeof(real3) * n, hipMemcpyDeviceToHost), "hipMemcpy");
  fcm.computeHydrodynamicDisplacements(dpos.d, dforce.d, n, real(0.7), real(3.0), dvel.d);
  detail::hipCheck(hipDeviceSynchronize(), "hipDeviceSynchronize");
  detail::hipCheck(hipMemcpy(v1.data(), dvel.d, sizeof(real3) * n, hipMemcpyDeviceToHost), "hipMemcpy");
  double s0 = 0, s1 = 0;
  for (int i = 0; i < n; ++i) { s0 += (double)v0[i].x * v0[i].x + (double)v0[i].y * v0[i].y + (double)v0[i].z * v0[i].z;
                                s1 += (double)v1[i].x * v1[i].x + (double)v1[i].y * v1[i].y + (double)v1[i].z * v1[i].z; }
  std::printf("rank %d of %d: halo %d planes, |v(T=0)| = %.6e, |v(T=0.7)| = %.6e\n", rank, world, fcm.haloPlanes(), std::sqrt(s0), std::sqrt(s1));
  if (!(s0 > 0) || !std::isfinite(s0) || !std::isfinite(s1)) return 1;
  if (world == 1) {
    BDHI::FCM_impl<>::Parameters fp;
    fp.box = Box(par.boxSize);
    fp.cells = par.cells;
    fp.viscosity = par.viscosity;
    fp.tolerance = par.tolerance;
    fp.seed = par.seed;
    // FCM_impl takes its two windows ready-made, as the reference's does (FCM_impl.cuh:80-84; test/BDHI/FCM/fcm_test.cu:32-44)
    const real h = par.boxSize.x / par.cells.x;
    fp.kernel = std::make_shared<BDHI::FCM_ns::Kernels::Gaussian>(h, fp.tolerance);
    fp.hydrodynamicRadius = fp.kernel->fixHydrodynamicRadius(0, h);
    fp.kernelTorque = std::make_shared<BDHI::FCM_ns::Kernels::GaussianTorque>(fp.hydrodynamicRadius / real(std::pow(6 * std::sqrt(M_PI), 1 / 3.)), h, fp.tolerance);
    BDHI::FCM_impl<> ref(fp);
    double worst = 0;
    for (int call = 0; call < 2; ++call) {
      const real T = call ? real(0.7) : real(0), pf = call ? real(3.0) : real(0);
      ref.computeHydrodynamicDisplacements(dpos.d, dforce.d, dvel.d, n, T, pf, 0);
      std::vector<real3> r(n);
      detail::hipCheck(hipDeviceSynchronize(), "hipDeviceSynchronize");
      detail::hipCheck(hipMemcpy(r.data(), dvel.d, sizeof(real3) * n, hipMemcpyDeviceToHost), "hipMemcpy");
      const std::vector<real3> &v = call ? v1 : v0;
      double num = 0, den = 0;
      for (int i = 0; i < n; ++i) {
        const double dx = v[i].x - r[i].x, dy = v[i].y - r[i].y, dz = v[i].z - r[i].z;
        num += dx * dx + dy * dy + dz * dz;
        den += (double)r[i].x * r[i].x + (double)r[i].y * r[i].y + (double)r[i].z * r[i].z;
      }
      worst = std::max(worst, std::sqrt(num / den));
    }
    std::printf("world 1: relative L2 difference from BDHI::FCM_impl = %.3e\n", worst);
    if (!(worst <= 1e-5)) return 1;
  }
  return 0;
}
