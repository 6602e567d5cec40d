// Public members of the host classes that no other program of the suite names (found by listing every member of uammd.h against the
// sources of the programs that run): each is called here once, and checked against what the reference's definition says —
// Box::isInside / getVolume (utils/Box.cuh:60-77), Grid::getCellCenter / distanceToCellUpperLeftCorner (utils/Grid.cuh:129-137),
// ParticleData::get<Name>IfAllocated / is<Name>Allocated / hintSortByHash (ParticleData.cuh:247-259,389-394),
// ParticleGroup::getPropertyIterator (ParticleGroup.cuh:300-320), Interactor::getName, VerletList::getNumberOfStepsSinceLastUpdate
// (VerletList.cuh:160-170), BDHI::PSE::setShearStrain (BDHI_PSE.cuh:156-161), lanczos::Solver::setIterationHardLimit.
#include "uammd.cuh"
#include "Integrator/BDHI/BDHI_PSE.cuh"
#include "Interactor/NeighbourList/VerletList.cuh"
#include "Interactor/PairForces.cuh"
#include "misc/LanczosAlgorithm.cuh"
#include <cstdio>
#include <random>
using namespace uammd;

static int bad = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++bad; } } while (0)

int main(int argc, char *argv[]) {
  auto sys = std::make_shared<System>(argc, argv);
  {  // Box
    Box box(make_real3(4, 6, 8));
    CHECK(box.getVolume() == real(192) && Box(make_real3(4, 6, 0)).getVolume() == real(24));
    CHECK(box.isInside(make_real3(1.9, -2.9, 3.9)) && !box.isInside(make_real3(2.1, 0, 0)) && !box.isInside(make_real3(0, 0, -4.0)) && box.isInside(make_real3(2.0, 3.0, 4.0)));
  }
  {  // Grid
    Grid grid(Box(make_real3(8, 8, 8)), make_int3(4, 4, 4));
    const real3 c = grid.getCellCenter(make_int3(1, 0, 3));
    CHECK(c.x == real(3) && c.y == real(1) && c.z == real(7));
    const real3 d = grid.distanceToCellUpperLeftCorner(make_real3(-1.5, -3.5, 2.25), make_int3(1, 0, 3));   // the cell's corner is at (2, 0, 6) from the box's
    CHECK(std::abs(d.x - 0.5) < 1e-6 && std::abs(d.y - 0.5) < 1e-6 && std::abs(d.z - 0.25) < 1e-6);
  }
  const int N = 4096;
  auto pd = std::make_shared<ParticleData>(N, sys);
  {  // properties that were never asked for do not exist; asking through IfAllocated does not create them
    CHECK(!pd->isPosAllocated() && !pd->isMassAllocated() && !pd->isForceAllocated());   // (the constructor makes the ids only: ParticleData.cuh:471-490)
    CHECK(pd->getPosIfAllocated(access::cpu, access::read).raw() == nullptr);
    CHECK(pd->getVelIfAllocated(access::cpu, access::read).raw() == nullptr && pd->getForceIfAllocated(access::gpu, access::read).raw() == nullptr);
    CHECK(pd->getEnergyIfAllocated(access::cpu, access::read).raw() == nullptr && pd->getVirialIfAllocated(access::cpu, access::read).raw() == nullptr);
    CHECK(pd->getChargeIfAllocated(access::cpu, access::read).raw() == nullptr && !pd->isForceAllocated());
    { auto p = pd->getPos(access::cpu, access::write); (void)p; }
    CHECK(pd->isPosAllocated() && pd->getPosIfAllocated(access::cpu, access::read).raw() != nullptr);
  }
  {
    std::mt19937 gen(2);
    std::uniform_real_distribution<double> u(-8, 8);
    auto pos = pd->getPos(access::cpu, access::write);
    for (int i = 0; i < N; ++i) pos[i] = make_real4(u(gen), u(gen), u(gen), i % 4);
  }
  {  // a group's view of a property: element k is the property of member k, through a sort
    auto pg = std::make_shared<ParticleGroup>(particle_selector::Type(3), pd, "threes");
    pd->hintSortByHash(Box(make_real3(16)), make_real3(2.0));
    pd->sortParticles();
    auto pos = pd->getPos(access::cpu, access::read);
    auto it = pg->getPropertyIterator(pos);
    auto id = pd->getId(access::cpu, access::read);
    auto index = pg->getIndexIterator(access::cpu);
    bool all3 = pg->getNumberParticles() == N / 4;
    for (int k = 0; k < pg->getNumberParticles(); ++k) all3 = all3 && (int)it[k].w == 3 && id[index[k]] % 4 == 3 && it[k].x == pos[index[k]].x;
    CHECK(all3);
    CHECK(pg->getName() == "threes");
  }
  {  // the Verlet list counts the updates it answered without rebuilding
    Box box(make_real3(16));
    auto nl = std::make_shared<VerletList>(pd);
    nl->update(box, real(2.0));
    const int s0 = nl->getNumberOfStepsSinceLastUpdate();
    // (an update with no position write in between does not reach the list at all — VerletList.cuh:112-124,190-200 — so the particles are
    // nudged, by much less than the list's skin: the stored list stays valid and the update is counted)
    auto nudge = [&]() { auto pos = pd->getPos(access::cpu, access::readwrite); for (int i = 0; i < N; ++i) pos[i].x += real(1e-4); };
    nudge();
    nl->update(box, real(2.0));
    const int s1 = nl->getNumberOfStepsSinceLastUpdate();
    nudge();
    nl->update(box, real(2.0));
    const int s2 = nl->getNumberOfStepsSinceLastUpdate();
    std::printf("VerletList: updates answered since the last rebuild %d %d %d\n", s0, s1, s2);
    CHECK(s0 == 0 && s1 == 1 && s2 == 2);   // (VerletListBase.cuh:105-123: the count of update() calls since the rebuild, minus one)
    auto data = nl->getVerletList();
    CHECK(data.maxNeighboursPerParticle > 0 && data.neighbourList != nullptr && data.numberNeighbours != nullptr);
    struct Named : public Interactor { using Interactor::Interactor; void sum(Computables, hipStream_t) override {} };
    CHECK(Named(pd, "a name").getName() == "a name");
  }
  {  // PSE: a sheared lattice sum is another operator; back at zero strain it is the first one again
    BDHI::PSE::Parameters par;
    par.viscosity = 1; par.hydrodynamicRadius = 1; par.dt = 1; par.temperature = 0; par.tolerance = 1e-3; par.psi = 0.5;
    par.box = Box(make_real3(16));
    auto pse = std::make_shared<BDHI::PSE>(pd, par);
    { auto f = pd->getForce(access::cpu, access::write); for (int i = 0; i < N; ++i) f[i] = make_real4(i % 2 ? 1 : -1, 0.5, 0, 0); }
    detail::DeviceArray<real3> a(N), b(N), c(N);
    pse->computeMF(a.d, 0);
    pse->setShearStrain(real(0.2));
    pse->computeMF(b.d, 0);
    pse->setShearStrain(real(0.0));
    pse->computeMF(c.d, 0);
    std::vector<real3> ha(N), hb(N), hc(N);
    (void)hipMemcpy(ha.data(), a.d, sizeof(real3) * N, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hb.data(), b.d, sizeof(real3) * N, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hc.data(), c.d, sizeof(real3) * N, hipMemcpyDeviceToHost);
    double dab = 0, dac = 0, scale = 0;
    for (int i = 0; i < N; ++i) { dab = std::max(dab, (double)std::abs(ha[i].x - hb[i].x)); dac = std::max(dac, (double)std::abs(ha[i].x - hc[i].x)); scale = std::max(scale, (double)std::abs(ha[i].x)); }
    std::printf("PSE::setShearStrain: max |MF(0.2) - MF(0)| = %.2e, max |MF(0 again) - MF(0)| = %.2e of %.2e\n", dab, dac, scale);
    CHECK(dab > 1e-4 * scale && dac < 2e-5 * scale);
  }
  {  // Lanczos: with a hard limit of two iterations a solve that needs more says so (LanczosAlgorithm.cuh:69-72)
    lanczos::Solver solver;
    solver.setIterationHardLimit(2);
    const int n = 300;
    detail::DeviceArray<real> v(n), out(n);
    std::vector<real> hv(n);
    for (int i = 0; i < n; ++i) hv[i] = real(1 + (i % 7));
    (void)hipMemcpy(v.d, hv.data(), sizeof(real) * n, hipMemcpyHostToDevice);
    struct Diagonal : public lanczos::MatrixDot {   // M = diag(1 .. n): sqrt(M) v needs many Krylov vectors
      int n;
      void operator()(real *in, real *res) override {
        std::vector<real> h(n);
        (void)hipMemcpy(h.data(), in, sizeof(real) * n, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; ++i) h[i] *= real(i + 1);
        (void)hipMemcpy(res, h.data(), sizeof(real) * n, hipMemcpyHostToDevice);
      }
    } dot;
    dot.n = n;
    bool threw = false;
    try { solver.run(dot, out.d, v.d, real(1e-6), n); } catch (const std::exception &e) { threw = true; std::printf("lanczos at the hard limit: %s\n", e.what()); }
    CHECK(threw);
  }
  std::printf(bad ? "api_corners: FAILED\n" : "api_corners: ok\n");
  return bad;
}
