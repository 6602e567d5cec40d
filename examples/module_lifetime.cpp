// Three things the reference's host interface promises and round 4 of this build did not keep ("What's missing" 1-4,
// "What's weak" 11):
//  (1) modules hold CONNECTIONS to ParticleData's signals and drop them when they die (CellList.cuh:88,142, VerletList.cuh:88,108-109,
//      ParticleGroup.cuh:188,212): destroying a list / interactor / integrator while the ParticleData lives, then writing positions or
//      sorting, must not call into the dead object.  Checked by counting slots and by a canary the dead object's slot would overwrite.
//  (2) the BDHI modules take a ParticleGroup (BDHI_EulerMaruyama.cuh:67, BDHI_FCM.cuh:98,167, BDHI_PSE.cuh:88, BDHI_Lanczos.cuh:31,
//      BrownianDynamics.cuh:113): on a proper subgroup they must move the members exactly as they move a ParticleData that holds the
//      members alone, and leave everybody else where they are.
//  (3) library mode: CellListBase / BasicNeighbourListBase on a raw position array, no ParticleData (CellListBase.cuh:124-172,
//      BasicListBase.cuh:131-165) — the Basic list against all pairs on the host.
#include "uammd.cuh"
#include "Integrator/BDHI/BDHI_EulerMaruyama.cuh"
#include "Integrator/BDHI/BDHI_FCM.cuh"
#include "Integrator/BDHI/BDHI_PSE.cuh"
#include "Integrator/BrownianDynamics.cuh"
#include "Integrator/VerletNVT.cuh"
#include "Interactor/NeighbourList/BasicList/BasicListBase.cuh"
#include "Interactor/NeighbourList/CellList/CellListBase.cuh"
#include "Interactor/PairForces.cuh"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <set>
#include <vector>
using namespace uammd;

static int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++fails; std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

// a constant force on every particle of the ParticleData (members and not): what moves the particles in part (2)
struct Pull : public Interactor {
  real3 f;
  Pull(std::shared_ptr<ParticleData> pd, real3 f) : Interactor(pd, "Pull"), f(f) {}
  void sum(Computables, hipStream_t) override {
    auto force = pd->getForce(access::cpu, access::readwrite);
    auto pos = pd->getPos(access::cpu, access::read);
    for (size_t i = 0; i < force.size(); ++i) {  // (depends on the particle, so a mixed-up index shows)
      const real s = real(1.0) + real(0.25) * std::sin(pos[i].x + 2 * pos[i].y);
      force[i] = force[i] + make_real4(f.x * s, f.y * s, f.z * s, 0);
    }
  }
};

static std::vector<real4> lattice(int N, real L, Xorshift128plus &rng) {
  std::vector<real4> p(N);
  int m = 1;
  while (m * m * m < N) ++m;
  const real a = L / m;
  for (int i = 0; i < N; ++i) {
    const int ix = i % m, iy = (i / m) % m, iz = i / (m * m);
    p[i] = make_real4((ix + 0.5 + rng.uniform(-0.2, 0.2)) * a - L / 2, (iy + 0.5 + rng.uniform(-0.2, 0.2)) * a - L / 2,
                      (iz + 0.5 + rng.uniform(-0.2, 0.2)) * a - L / 2, 0);
  }
  return p;
}

template <class Make> static void lifetime(const char *what, std::shared_ptr<ParticleData> pd, Make make) {
  const int w0 = pd->getPosWriteRequestedSignal()->slot_count(), r0 = pd->getReorderSignal()->slot_count();
  {
    auto obj = make();
    const int w1 = pd->getPosWriteRequestedSignal()->slot_count(), r1 = pd->getReorderSignal()->slot_count();
    std::printf("%-34s holds %d position-write and %d reorder connections\n", what, w1 - w0, r1 - r0);
    CHECK(w1 + r1 > w0 + r0, "%s listens to nothing", what);
  }
  CHECK(pd->getPosWriteRequestedSignal()->slot_count() == w0, "%s left a position-write slot behind", what);
  CHECK(pd->getReorderSignal()->slot_count() == r0, "%s left a reorder slot behind", what);
  { auto pos = pd->getPos(access::cpu, access::readwrite); pos[0].x += 0; }  // round 4: a call into freed memory
  pd->sortParticles();
}

template <class Integ, class Par> static void groupParity(const char *what, Par par, real L, std::shared_ptr<System> sys, real tol) {
  const int N = 6000;
  Xorshift128plus rng(0x5EED);
  const auto p0 = lattice(N, L, rng);
  std::vector<int> ids;
  for (int i = 0; i < N; ++i) if (i % 3 != 1) ids.push_back(i);  // two thirds of the particles, interleaved with the rest
  const int M = (int)ids.size();
  auto pd = std::make_shared<ParticleData>(N, sys);
  { auto pos = pd->getPos(access::cpu, access::write); std::copy(p0.begin(), p0.end(), pos.begin()); }
  pd->sortParticles();  // the members are scattered in memory and the group index is not the identity on any prefix
  auto pg = std::make_shared<ParticleGroup>(ids.begin(), ids.end(), pd, "two thirds");
  auto pdm = std::make_shared<ParticleData>(M, sys);
  { auto pos = pdm->getPos(access::cpu, access::write); for (int k = 0; k < M; ++k) pos[k] = p0[ids[k]]; }
  sys->rng().setSeed(1234);
  auto a = std::make_shared<Integ>(pg, par);
  sys->rng().setSeed(1234);
  auto b = std::make_shared<Integ>(pdm, par);
  a->addInteractor(std::make_shared<Pull>(pd, make_real3(0.3, -0.2, 0.1)));
  b->addInteractor(std::make_shared<Pull>(pdm, make_real3(0.3, -0.2, 0.1)));
  for (int step = 0; step < 3; ++step) { a->forwardTime(); b->forwardTime(); }
  auto pos = pd->getPos(access::cpu, access::read);
  auto posm = pdm->getPos(access::cpu, access::read);
  const int *id2index = pd->getIdOrderedIndices(access::cpu);
  double num = 0, den = 0, moved = 0;
  std::vector<char> member(N, 0);
  for (int k = 0; k < M; ++k) {
    member[ids[k]] = 1;
    const real4 x = pos[id2index[ids[k]]], y = posm[k], o = p0[ids[k]];
    num += (double)(x.x - y.x) * (x.x - y.x) + (double)(x.y - y.y) * (x.y - y.y) + (double)(x.z - y.z) * (x.z - y.z);
    den += (double)(y.x - o.x) * (y.x - o.x) + (double)(y.y - o.y) * (y.y - o.y) + (double)(y.z - o.z) * (y.z - o.z);
  }
  for (int i = 0; i < N; ++i)
    if (!member[i]) {
      const real4 x = pos[id2index[i]], o = p0[i];
      moved += std::fabs(x.x - o.x) + std::fabs(x.y - o.y) + std::fabs(x.z - o.z);
    }
  const double err = std::sqrt(num / den);
  std::printf("%-34s group of %d in %d vs the members alone: |dx| relative L2 %.2e (displacement rms %.3e), others moved by %g\n", what, M, N,
              err, std::sqrt(den / M), moved);
  CHECK(den > 0, "%s did not move the members", what);
  CHECK(err <= tol, "%s: group run differs from the members-alone run: %g > %g", what, err, (double)tol);
  CHECK(moved == 0, "%s moved particles outside its group", what);
}

int main(int argc, char *argv[]) {
  auto sys = std::make_shared<System>(argc, argv);
  const real L = 32;
  // ---- (1) lifetimes -----------------------------------------------------------------------------------------------------------------
  {
    const int N = 4096;
    auto pd = std::make_shared<ParticleData>(N, sys);
    Xorshift128plus rng(7);
    { auto p = lattice(N, L, rng); auto pos = pd->getPos(access::cpu, access::write); std::copy(p.begin(), p.end(), pos.begin()); }
    lifetime("CellList", pd, [&]() { auto l = std::make_shared<CellList>(pd); l->update(Box(L), real(2.5)); return l; });
    lifetime("VerletList", pd, [&]() { auto l = std::make_shared<VerletList>(pd); l->update(Box(L), real(2.5)); return l; });
    lifetime("ParticleGroup", pd, [&]() { return std::make_shared<ParticleGroup>(particle_selector::IDRange(10, 500), pd, "some"); });
    lifetime("PairForces<LJ, CellList>", pd, [&]() {
      auto pot = std::make_shared<Potential::LJ>();
      Potential::LJ::InputPairParameters pp; pp.cutOff = 2.5; pp.sigma = 1; pp.epsilon = 1; pp.shift = false;
      pot->setPotParameters(0, 0, pp);
      PairForces<Potential::LJ>::Parameters par; par.box = Box(L);
      auto pf = std::make_shared<PairForces<Potential::LJ>>(pd, par, pot);
      Interactor::Computables c; c.force = true;
      pf->sum(c, 0);
      return pf;
    });
    lifetime("BDHI::FCMIntegrator", pd, [&]() {
      BDHI::FCMIntegrator::Parameters par; par.temperature = 0; par.viscosity = 1; par.hydrodynamicRadius = 1; par.dt = 0.01; par.box = Box(L);
      auto i = std::make_shared<BDHI::FCMIntegrator>(pd, par);
      i->forwardTime();
      return i;
    });
    lifetime("BDHI::EulerMaruyama<BDHI::PSE>", pd, [&]() {
      BDHI::PSE::Parameters par; par.temperature = 0; par.viscosity = 1; par.hydrodynamicRadius = 1; par.dt = 0.01; par.box = Box(L); par.tolerance = 1e-3;
      auto i = std::make_shared<BDHI::EulerMaruyama<BDHI::PSE>>(pd, par);
      i->forwardTime();
      return i;
    });
    // a user's own connection, the reference's way (examples/advanced/signals.cu): connect, emit, disconnect
    int seen = 0;
    connection mine = pd->getVelWriteRequestedSignal()->connect([&]() { ++seen; });
    { auto v = pd->getVel(access::cpu, access::read); }
    CHECK(seen == 0, "a read raised the write signal");
    { auto v = pd->getVel(access::cpu, access::write); }
    { auto v = pd->getVel(access::gpu, access::readwrite); }
    CHECK(seen == 2, "velocity write signal seen %d times, expected 2", seen);
    mine.disconnect();
    { auto v = pd->getVel(access::cpu, access::write); }
    CHECK(seen == 2, "a disconnected slot was called");
  }
  // ---- (2) groups -----------------------------------------------------------------------------------------------------------------------
  {
    BD::Parameters par; par.temperature = 0; par.viscosity = 1; par.hydrodynamicRadius = 1; par.dt = 0.01;
    groupParity<BD::EulerMaruyama>("BD::EulerMaruyama", par, L, sys, 1e-7);
  }
  {
    BDHI::FCM::Parameters par; par.temperature = 0; par.viscosity = 1; par.hydrodynamicRadius = 1; par.dt = 0.01; par.box = Box(L); par.tolerance = 1e-3;
    groupParity<BDHI::EulerMaruyama<BDHI::FCM>>("BDHI::EulerMaruyama<BDHI::FCM>", par, L, sys, 1e-5);
    groupParity<BDHI::FCMIntegrator>("BDHI::FCMIntegrator", par, L, sys, 1e-5);
    par.temperature = 0.7;  // the noise is keyed by grid node and the seeds are drawn in the same order: the same fluctuating field
    groupParity<BDHI::FCMIntegrator>("BDHI::FCMIntegrator, T > 0", par, L, sys, 1e-5);
  }
  {
    BDHI::PSE::Parameters par; par.temperature = 0; par.viscosity = 1; par.hydrodynamicRadius = 1; par.dt = 0.01; par.box = Box(L); par.tolerance = 1e-3;
    groupParity<BDHI::EulerMaruyama<BDHI::PSE>>("BDHI::EulerMaruyama<BDHI::PSE>", par, L, sys, 1e-5);
  }
  // ---- (3) library mode ------------------------------------------------------------------------------------------------------------------
  {
    const int N = 3000;
    const real rc = 2.5;
    Xorshift128plus rng(99);
    std::vector<real4> p(N);
    for (auto &v : p) { const real3 u = make_real3(rng.uniform3(-0.5, 0.5)); v = make_real4(u.x * L, u.y * L, u.z * L, 0); }
    uninitialized_cached_vector<real4> d_pos(N);
    CudaSafeCall(hipMemcpy(d_pos.data().get(), p.data(), sizeof(real4) * N, hipMemcpyHostToDevice));
    Box box(L);
    BasicNeighbourListBase nl;
    nl.update(d_pos.data().get(), N, box, rc);
    auto data = nl.getBasicNeighbourList();
    std::vector<int> nneigh(N), group(N);
    CudaSafeCall(hipMemcpy(nneigh.data(), data.numberNeighbours, sizeof(int) * N, hipMemcpyDeviceToHost));
    CudaSafeCall(hipMemcpy(group.data(), data.groupIndex, sizeof(int) * N, hipMemcpyDeviceToHost));
    const int stride = data.particleStride[0];
    const int maxn = *std::max_element(nneigh.begin(), nneigh.end());
    std::vector<int> list((size_t)stride * maxn);
    CudaSafeCall(hipMemcpy(list.data(), data.neighbourList, sizeof(int) * list.size(), hipMemcpyDeviceToHost));
    long long pairs = 0, wrong = 0;
    for (int s = 0; s < N; ++s) {
      const int i = group[s];
      std::set<int> got, want;
      for (int k = 0; k < nneigh[s]; ++k) got.insert(group[list[(size_t)data.particleStride[s] * k + s]]);
      for (int j = 0; j < N; ++j) {
        const real3 r = box.apply_pbc(make_real3(p[j]) - make_real3(p[i]));
        if (dot(r, r) <= rc * rc) want.insert(j);  // (the particle itself is one of its neighbours, BasicListBase.cuh:60)
      }
      pairs += (long long)want.size();
      wrong += got != want;
    }
    std::printf("BasicNeighbourListBase (library mode)  %lld pairs over %d particles, stride %d, %lld particles with a wrong list\n", pairs, N, stride, wrong);
    CHECK(wrong == 0, "library-mode neighbour list differs from all pairs");
    CellListBase cl;
    cl.update(d_pos.data().get(), N, Grid(box, rc));
    auto cld = cl.getCellList();
    const int ncells = cld.grid.getNumberCells();
    std::vector<uint> cellStart(ncells);
    std::vector<int> cellEnd(ncells);
    std::vector<real4> sortPos(N);
    CudaSafeCall(hipMemcpy(cellStart.data(), cld.cellStart, sizeof(uint) * ncells, hipMemcpyDeviceToHost));
    CudaSafeCall(hipMemcpy(cellEnd.data(), cld.cellEnd, sizeof(int) * ncells, hipMemcpyDeviceToHost));
    CudaSafeCall(hipMemcpy(sortPos.data(), cld.sortPos, sizeof(real4) * N, hipMemcpyDeviceToHost));
    long long counted = 0, misplaced = 0;
    for (int c = 0; c < ncells; ++c) {
      if (cellStart[c] < cld.VALID_CELL) continue;
      for (int s = (int)(cellStart[c] - cld.VALID_CELL); s < cellEnd[c]; ++s) {
        ++counted;
        misplaced += cld.grid.getCellIndex(cld.grid.getCell(sortPos[s])) != c;
      }
    }
    std::printf("CellListBase (library mode)            %d cells, %lld particles listed, %lld in the wrong cell\n", ncells, counted, misplaced);
    CHECK(counted == N && misplaced == 0, "library-mode cell list is wrong");
  }
  // ---- (4) Potential parameters changed between steps (legal in the reference, Potential.cuh:60-82; ADVICE r04: the list's cached view of
  // the table was keyed by the table's address alone, and setPotParameters re-uploads into the same buffer) ----------------------------
  {
    const int N = 32768;
    const real Lb = 34;
    auto pd = std::make_shared<ParticleData>(N, sys);
    Xorshift128plus rng(3);
    { auto p = lattice(N, Lb, rng); auto pos = pd->getPos(access::cpu, access::write); std::copy(p.begin(), p.end(), pos.begin()); }
    auto pot = std::make_shared<Potential::LJ>();
    Potential::LJ::InputPairParameters pp; pp.cutOff = 2.5; pp.sigma = 1; pp.epsilon = 1; pp.shift = false;
    pot->setPotParameters(0, 0, pp);
    PairForces<Potential::LJ>::Parameters par; par.box = Box(Lb);
    auto pf = std::make_shared<PairForces<Potential::LJ>>(pd, par, pot);
    Interactor::Computables c; c.force = true;
    auto forces = [&]() {
      { auto f = pd->getForce(access::gpu, access::write); CudaSafeCall(hipMemset(f.raw(), 0, sizeof(real4) * N)); }
      pf->sum(c, 0);
      auto f = pd->getForce(access::cpu, access::read);
      return std::vector<real4>(f.begin(), f.end());
    };
    const auto f1 = forces();  // sigma = epsilon = 1: the reduced-units kernel
    pp.epsilon = 2;
    pot->setPotParameters(0, 0, pp);  // same table size: the same device buffer is rewritten
    const auto f2 = forces();
    pp.epsilon = 1;
    pot->setPotParameters(0, 0, pp);
    const auto f3 = forces();
    long long notDouble = 0, notBack = 0;
    double maxF = 0;
    for (int i = 0; i < N; ++i) {
      notDouble += !(f2[i].x == 2 * f1[i].x && f2[i].y == 2 * f1[i].y && f2[i].z == 2 * f1[i].z);
      notBack += !(f3[i].x == f1[i].x && f3[i].y == f1[i].y && f3[i].z == f1[i].z);
      maxF = std::max(maxF, (double)std::fabs(f1[i].x));
    }
    std::printf("epsilon 1 -> 2 -> 1 between steps       max|Fx| %.3g: %lld forces not exactly doubled, %lld not restored\n", maxF, notDouble, notBack);
    CHECK(maxF > 0 && notDouble == 0 && notBack == 0, "PairForces did not follow a parameter change between steps");
  }
  sys->finish();
  std::printf(fails ? "FAILED (%d)\n" : "module lifetime / groups / library mode: ok\n", fails);
  return fails ? 1 : 0;
}
