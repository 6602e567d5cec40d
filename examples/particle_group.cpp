// ParticleGroup on the C++ interface (ParticleData/ParticleGroup.cuh): a group selected by type, PairForces and VerletNVT acting on
// the group only, membership kept across ParticleData::sortParticles.  The forces among the members must equal, bit for bit, the
// forces of a ParticleData that holds just those particles; everybody else must feel nothing and stay put.
#include "uammd.cuh"
#include "Interactor/PairForces.cuh"
#include "Integrator/VerletNVT.cuh"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace uammd;

int main(int argc, char *argv[]) {
  auto sys = std::make_shared<System>(argc, argv);
  // optional arguments: N L.  "600 7" puts the box below 3 rc in every direction, so PairForces takes its all-pairs branch
  // (PairForces.cu:49-53) on the group — the branch that must read pos[globalIndex[.]] from the UN-gathered array.
  const int N = argc > 1 ? std::atoi(argv[1]) : 20000;
  const real L = argc > 2 ? std::atof(argv[2]) : 32;
  auto pd = std::make_shared<ParticleData>(N, sys);
  std::vector<real4> members;
  {
    auto pos = pd->getPos(access::cpu, access::write);
    // a jittered simple-cubic lattice (uniformly random positions overlap: forces ~1e12 would throw the members out of any box
    // within the five integration steps below, which the cell list now reports — CellListBase.cuh:82-85)
    int m = 1;
    while (m * m * m < N) ++m;
    for (int i = 0; i < N; ++i) {
      const int type = (i % 3 == 0) ? 1 : 0;  // a third of the particles are type 1
      const int ix = i % m, iy = (i / m) % m, iz = i / (m * m);
      const real a = L / m;
      pos[i] = make_real4((ix + 0.5 + sys->rng().uniform(-0.1, 0.1)) * a - L / 2, (iy + 0.5 + sys->rng().uniform(-0.1, 0.1)) * a - L / 2,
                          (iz + 0.5 + sys->rng().uniform(-0.1, 0.1)) * a - L / 2, type);
      if (type == 1) members.push_back(pos[i]);
    }
  }
  auto pg = std::make_shared<ParticleGroup>(particle_selector::Type(1), pd, "type1");
  int bad = pg->getNumberParticles() != (int)members.size();
  auto pot = std::make_shared<Potential::LJ>();
  Potential::LJ::InputPairParameters pp;
  pp.cutOff = 2.5; pp.sigma = 1; pp.epsilon = 1; pp.shift = false;
  pot->setPotParameters(0, 0, pp); pot->setPotParameters(0, 1, pp); pot->setPotParameters(1, 1, pp);
  using PF = PairForces<Potential::LJ>;
  PF::Parameters par;
  par.box = Box(L);
  Interactor::Computables c;
  c.force = true;
  // reference answer: the members alone
  auto pdm = std::make_shared<ParticleData>((int)members.size(), sys);
  { auto pos = pdm->getPos(access::cpu, access::write); for (size_t i = 0; i < members.size(); ++i) pos[i] = members[i]; }
  { auto f = pdm->getForce(access::cpu, access::write); for (auto &v : f) v = make_real4(0); }
  std::make_shared<PF>(pdm, par, pot)->sum(c, 0);
  // group forces, before and after a sort of the whole ParticleData
  auto pf = std::make_shared<PF>(pg, par, pot);
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) pd->sortParticles();
    { auto f = pd->getForce(access::cpu, access::write); for (auto &v : f) v = make_real4(0); }
    pf->sum(c, 0);
    auto f = pd->getForce(access::cpu, access::read);
    auto fm = pdm->getForce(access::cpu, access::read);
    auto pos = pd->getPos(access::cpu, access::read);
    auto id = pd->getId(access::cpu, access::read);
    auto idx = pg->getIndexIterator(access::cpu);
    int mismatches = 0, touched = 0;
    for (int k = 0; k < pg->getNumberParticles(); ++k) {
      const int i = idx[k];
      mismatches += std::memcmp(&f[i], &fm[k], sizeof(real4)) != 0;   // members are id ordered = the order of `members`
      mismatches += (int)pos[i].w != 1 || id[i] % 3 != 0;
    }
    for (int i = 0; i < N; ++i) if ((int)pos[i].w == 0) touched += f[i].x != 0 || f[i].y != 0 || f[i].z != 0;
    std::printf("pass %d: %d members, %d mismatching member forces, %d non-members with a force\n", pass, pg->getNumberParticles(),
                mismatches, touched);
    bad += mismatches + touched;
  }
  // an integrator on the group moves the members only
  std::vector<real4> before(N);
  { auto pos = pd->getPos(access::cpu, access::read); for (int i = 0; i < N; ++i) before[i] = pos[i]; }
  VerletNVT::GronbechJensen::Parameters vp;
  vp.temperature = 1.0; vp.dt = 0.001; vp.friction = 1.0;
  auto verlet = std::make_shared<VerletNVT::GronbechJensen>(pg, vp);
  verlet->addInteractor(pf);
  for (int s = 0; s < 5; ++s) verlet->forwardTime();
  int movedOthers = 0, movedMembers = 0;
  {
    auto pos = pd->getPos(access::cpu, access::read);
    for (int i = 0; i < N; ++i) {
      const bool moved = pos[i].x != before[i].x || pos[i].y != before[i].y || pos[i].z != before[i].z;
      if ((int)pos[i].w == 1) movedMembers += moved; else movedOthers += moved;
    }
  }
  std::printf("integrator on the group: %d members moved, %d others moved\n", movedMembers, movedOthers);
  bad += movedOthers + (movedMembers != pg->getNumberParticles());
  // id-range and explicit-list constructors
  auto pr = std::make_shared<ParticleGroup>(particle_selector::IDRange(10, 19), pd, "range");
  std::vector<int> ids = {5, 3, 9};
  auto pl = std::make_shared<ParticleGroup>(ids.begin(), ids.end(), pd, "list");
  bad += pr->getNumberParticles() != 10 || pl->getNumberParticles() != 3;
  {
    auto id = pd->getId(access::cpu, access::read);
    auto it = pl->getIndexIterator(access::cpu);
    bad += id[it[0]] != 3 || id[it[1]] != 5 || id[it[2]] != 9;
  }
  {  // a group emptied and filled again by hand (ParticleGroup.cuh:221-227): by id from the host, by current index from the device;
     // the membership is by id, so it follows the particles through a sort
    pl->clear();
    bad += pl->getNumberParticles() != 0;
    const int byId[2] = {42, 7};
    pl->addParticlesById(access::cpu, byId, 2);
    int target = -1;
    { auto id = pd->getId(access::cpu, access::read); for (int i = 0; i < N; ++i) if (id[i] == 1234 % N) target = i; }
    int *d_idx = nullptr;
    detail::hipCheck(hipMalloc((void **)&d_idx, sizeof(int)), "hipMalloc");
    detail::hipCheck(hipMemcpy(d_idx, &target, sizeof(int), hipMemcpyHostToDevice), "hipMemcpy");
    pl->addParticlesByCurrentIndex(access::gpu, d_idx, 1);
    (void)hipFree(d_idx);
    bad += pl->getNumberParticles() != 3;
    pd->sortParticles();
    auto id = pd->getId(access::cpu, access::read);
    auto it = pl->getIndexIterator(access::cpu);
    const bool ok = id[it[0]] == 42 && id[it[1]] == 7 && id[it[2]] == 1234 % N;
    std::printf("group refilled by id and by current index, after a sort: members' ids %d %d %d\n", id[it[0]], id[it[1]], id[it[2]]);
    bad += !ok;
  }
  sys->finish();
  return bad;
}
