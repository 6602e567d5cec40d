// LJ liquid on z slabs, one process per GPU: uammd::DistributedLJ over uammd::Comm (RCCL behind the C ABI).
//   usage:  lj_slab <rank> <world> <id file> [particles per rank] [steps]
// Start `world` processes with ranks 0 .. world - 1 and the same id file (rank 0 creates it); rank r uses GPU r % (GPUs of the node).
// With world = 1 the rank is its own neighbour through the periodic z faces (how tests/test_cxx_interface.py runs it on a one-GPU box)
// and the result is compared with the single-domain integrator of uammd.h on the same particles.
#include "uammd.cuh"
#include "Distributed.h"
#include "Interactor/NeighbourList/CellList.cuh"
#include "Interactor/PairForces.cuh"
#include "Interactor/Potential/Potential.cuh"
#include "Integrator/VerletNVT.cuh"
#include <cmath>
#include <cstdio>
#include <cstdlib>
using namespace uammd;

int main(int argc, char *argv[]) {
  if (argc < 4) { std::fprintf(stderr, "usage: %s rank world idfile [particlesPerRank] [steps]\n", argv[0]); return 2; }
  const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]);
  const std::string idfile = argv[3];
  const int nPer = argc > 4 ? std::atoi(argv[4]) : 32768, nsteps = argc > 5 ? std::atoi(argv[5]) : 40;
  int ndev = 1;
  detail::check(uammd_hip_device_count(&ndev));
  detail::check(uammd_hip_set_device(rank % ndev));
  auto comm = std::make_shared<Comm>(rank, world, exchangeUniqueIdThroughFile(idfile, rank));
  // every rank owns a cube of side l at rho* = 0.8: the global box is l x l x (world l)
  const int m = (int)std::lround(std::cbrt((double)nPer));
  const int n = m * m * m;
  const real l = (real)std::cbrt(n / 0.8);
  DistributedLJ::Parameters par;
  par.boxSize = make_real3(l, l, l * world);
  par.temperature = 0.0;   // deterministic: comparable with the single-domain run below
  par.dt = 0.004;
  par.skin = 0.3;
  par.exchangeEvery = 5;
  std::vector<real4> pos(n);
  std::vector<real3> vel(n);
  std::vector<int> ids(n);
  Xorshift128plus rng;
  rng.setSeed(1234 + rank);
  for (int i = 0; i < n; ++i) {
    const int ix = i % m, iy = (i / m) % m, iz = i / (m * m);
    pos[i] = make_real4((ix + real(0.5)) / m * l - l / 2 + real(0.1) * (real)(rng.uniform(-1, 1)),
                        (iy + real(0.5)) / m * l - l / 2 + real(0.1) * (real)(rng.uniform(-1, 1)),
                        (iz + real(0.5)) / m * l - l / 2 + real(0.1) * (real)(rng.uniform(-1, 1)), 0);
    vel[i] = make_real3((real)rng.uniform(-1, 1), (real)rng.uniform(-1, 1), (real)rng.uniform(-1, 1));
    ids[i] = rank * n + i;
  }
  DistributedLJ sim(comm, par, pos, vel, ids);
  for (int s = 0; s < nsteps; ++s) sim.forwardTime();
  sim.checkSkin();
  const long total = sim.totalParticles();
  std::vector<real4> p;
  std::vector<real3> v;
  std::vector<int> id;
  sim.download(p, v, id);
  std::printf("rank %d of %d: %d owned, %d ghosts, %ld particles in total (expected %ld)\n", rank, world, sim.numberOwned(), sim.numberGhosts(),
              total, (long)n * world);
  if (total != (long)n * world) { std::fprintf(stderr, "particles were lost or duplicated\n"); return 1; }
  if (world == 1) {
    // the same box through the plain classes: particle by particle (ids), positions agree to the accumulated rounding of two summation orders
    auto sys = std::make_shared<System>(argc, argv);
    auto pd = std::make_shared<ParticleData>(n, sys);
    {
      auto pp = pd->getPos(access::cpu, access::write);
      auto vv = pd->getVel(access::cpu, access::write);
      std::copy(pos.begin(), pos.end(), pp.begin());
      std::copy(vel.begin(), vel.end(), vv.begin());
    }
    VerletNVT::GronbechJensen::Parameters vp;
    vp.temperature = 0; vp.dt = par.dt; vp.friction = 1.0; vp.initVelocities = false;
    auto verlet = std::make_shared<VerletNVT::GronbechJensen>(pd, vp);
    auto pot = std::make_shared<Potential::LJ>();
    pot->setPotParameters(0, 0, Potential::LJ::InputPairParameters{2.5, 1, 1, false});
    PairForces<Potential::LJ, CellList>::Parameters fp;
    fp.box = Box(par.boxSize);
    verlet->addInteractor(std::make_shared<PairForces<Potential::LJ, CellList>>(pd, fp, pot));
    for (int s = 0; s < nsteps; ++s) verlet->forwardTime();
    auto ref = pd->getPos(access::cpu, access::read);
    double worst = 0;
    for (int k = 0; k < (int)p.size(); ++k) {
      const real4 a = p[k], b = ref.begin()[id[k]];
      double d[3] = {a.x - b.x, a.y - b.y, a.z - b.z};
      for (int c = 0; c < 3; ++c) { d[c] -= std::round(d[c] / l) * l; worst = std::max(worst, std::fabs(d[c])); }
    }
    std::printf("world 1: max |x_slab - x_single domain| after %d steps = %.3e\n", nsteps, worst);
    if (!(worst <= 5e-4)) return 1;
  }
  return 0;
}
