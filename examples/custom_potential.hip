// PairForces<Potential::Radial<Functor>, NeighbourList> with a functor written by the "user" (the reference's docs/Potential.rst shape):
// a Lennard-Jones functor must reproduce the library's own PairForces<Potential::LJ> (forces, energies, virials; on the CellList, on the
// VerletList, and through the all-pairs branch of a small box), and a Yukawa functor with two particle types runs an NVT step through
// VerletNVT::GronbechJensen like any other Interactor.
#include <Interactor/PairForces.cuh>   // under hipcc this brings device/PairForces.hip.hpp
#include <Integrator/VerletNVT.cuh>
#include <Interactor/ExternalForces.cuh>
#include <utils/InitialConditions.cuh>
#include <cstdio>
#include <random>
using namespace uammd;

struct UserLJ {
  struct InputPairParameters { real cutOff, sigma, epsilon; };
  struct PairParameters { real cutOff2, sigma2, epsilonDivSigma2; };
  static PairParameters processPairParameters(InputPairParameters in) { return {in.cutOff * in.cutOff, in.sigma * in.sigma, in.epsilon / (in.sigma * in.sigma)}; }
  __device__ real force(real r2, PairParameters p) {   // |f| / r
    if (r2 >= p.cutOff2) return 0;
    const real invr2 = p.sigma2 / r2, invr6 = invr2 * invr2 * invr2;
    return p.epsilonDivSigma2 * (real(-48.0) * invr6 + real(24.0)) * invr6 * invr2;
  }
  __device__ real energy(real r2, PairParameters p) {  // half the pair energy goes to each partner
    if (r2 >= p.cutOff2) return 0;
    const real invr2 = p.sigma2 / r2, invr6 = invr2 * invr2 * invr2;
    return real(0.5) * real(4.0) * p.epsilonDivSigma2 * p.sigma2 * invr6 * (invr6 - real(1.0));
  }
};

struct Yukawa {
  struct InputPairParameters { real cutOff, kappa, strength; };
  struct PairParameters { real cutOff2, kappa, strength; };
  static PairParameters processPairParameters(InputPairParameters in) { return {in.cutOff * in.cutOff, in.kappa, in.strength}; }
  __device__ real force(real r2, PairParameters p) {   // U = A exp(-kappa r) / r  ->  |f| / r = A exp(-kappa r) (1 + kappa r) / r^3
    if (r2 >= p.cutOff2) return 0;
    const real r = sqrtf(r2);
    return -p.strength * expf(-p.kappa * r) * (real(1.0) + p.kappa * r) / (r2 * r);
  }
  __device__ real energy(real r2, PairParameters p) {
    if (r2 >= p.cutOff2) return 0;
    const real r = sqrtf(r2);
    return real(0.5) * p.strength * expf(-p.kappa * r) / r;
  }
};

// the reference's examples/misc/LJ.cu:35-66 shape: a wall acting on one species, arrays chosen by the functor
// a soft repulsion whose strength follows the simulation time (the pattern of the reference's examples/advanced/ParameterUpdatable.cu:126-201)
struct GrowingRepulsion : public ParameterUpdatable {
  real temperature = -1, dt = -1, time = -1, boxL = -1, viscosity = -1;
  int timeCalls = 0;
  real getCutOff() { return real(2.0); }
  struct Transverser {
    real4 *force;
    Box box;
    real strength;
    __device__ real3 compute(real4 pi, real4 pj) {
      const real3 rij = box.apply_pbc(make_real3(pj) - make_real3(pi));
      const real r2 = dot(rij, rij);
      if (r2 > 0 && r2 < real(4.0)) return rij * (-strength * (real(4.0) - r2));
      return make_real3(0);
    }
    __device__ void set(int id, real3 total) { force[id] += make_real4(total, 0); }
  };
  Transverser getTransverser(Interactor::Computables, Box box, std::shared_ptr<ParticleData> pd) {
    return Transverser{pd->getForce(access::gpu, access::readwrite).raw(), box, real(1.0) + (time > 0 ? time : real(0))};
  }
  void updateTemperature(real v) override { temperature = v; }
  void updateTimeStep(real v) override { dt = v; }
  void updateSimulationTime(real v) override { time = v; ++timeCalls; }
  void updateBox(Box b) override { boxL = b.boxSize.x; }
  void updateViscosity(real v) override { viscosity = v; }
};

struct HarmonicWall : public ParameterUpdatable {
  real zwall, k = 0.1;
  real lastTime = -1;
  explicit HarmonicWall(real zwall) : zwall(zwall) {}
  __device__ ForceEnergyVirial sum(Interactor::Computables comp, real4 pos) {
    const real fz = -k * (pos.z - zwall);
    return {{0.0f, 0.0f, fz}, comp.energy ? real(0.5) * k * (pos.z - zwall) * (pos.z - zwall) : real(0), 0};
  }
  auto getArrays(ParticleData *pd) {
    auto pos = pd->getPos(access::gpu, access::read);
    return pos.begin();
  }
  void updateSimulationTime(real t) override { lastTime = t; }
};
struct DragOnIds {  // several arrays through a tuple
  real gamma = 0.5;
  __device__ ForceEnergyVirial sum(Interactor::Computables, real4, real3 vel, int id) {
    const real g = (id % 2) ? gamma : real(0);
    return {{-g * vel.x, -g * vel.y, -g * vel.z}, 0, 0};
  }
  auto getArrays(ParticleData *pd) {
    auto pos = pd->getPos(access::gpu, access::read);
    auto vel = pd->getVel(access::gpu, access::read);
    auto id = pd->getId(access::gpu, access::read);
    return std::make_tuple(pos.begin(), vel.begin(), id.begin());
  }
};

template <class PF> static void evaluate(std::shared_ptr<ParticleData> pd, std::shared_ptr<PF> pf, std::vector<real4> &f, std::vector<real> &e, std::vector<real> &v) {
  const int N = pd->getNumParticles();
  {
    auto force = pd->getForce(access::cpu, access::write);
    auto energy = pd->getEnergy(access::cpu, access::write);
    auto virial = pd->getVirial(access::cpu, access::write);
    std::fill(force.begin(), force.end(), real4());
    std::fill(energy.begin(), energy.end(), real(0));
    std::fill(virial.begin(), virial.end(), real(0));
  }
  pf->sum({.force = true, .energy = true, .virial = true});
  auto force = pd->getForce(access::cpu, access::read);
  auto energy = pd->getEnergy(access::cpu, access::read);
  auto virial = pd->getVirial(access::cpu, access::read);
  f.assign(force.begin(), force.begin() + N);
  e.assign(energy.begin(), energy.begin() + N);
  v.assign(virial.begin(), virial.begin() + N);
}

static int compare(const char *what, const std::vector<real4> &f, const std::vector<real> &e, const std::vector<real> &v, const std::vector<real4> &rf,
                   const std::vector<real> &re, const std::vector<real> &rv) {
  double df = 0, fmax = 0, de = 0, emax = 0, dv = 0, vmax = 0;
  for (size_t i = 0; i < f.size(); ++i) {
    df = std::max(df, (double)std::max(std::fabs(f[i].x - rf[i].x), std::max(std::fabs(f[i].y - rf[i].y), std::fabs(f[i].z - rf[i].z))));
    fmax = std::max(fmax, (double)std::max(std::fabs(rf[i].x), std::max(std::fabs(rf[i].y), std::fabs(rf[i].z))));
    de = std::max(de, (double)std::fabs(e[i] - re[i])); emax = std::max(emax, (double)std::fabs(re[i]));
    dv = std::max(dv, (double)std::fabs(v[i] - rv[i])); vmax = std::max(vmax, (double)std::fabs(rv[i]));
  }
  std::printf("%-44s |dF| %.2e of %.2e, |dE| %.2e of %.2e, |dV| %.2e of %.2e\n", what, df, fmax, de, emax, dv, vmax);
  return (df <= 2e-5 * fmax && de <= 2e-5 * emax && dv <= 2e-5 * vmax && fmax > 0) ? 0 : 1;
}

int main() {
  int fails = 0;
  for (int small = 0; small < 2; ++small) {
    // a liquid-density box on 20^3 cells, and a box of 3 rc where PairForces takes all pairs (PairForces.cu:50-53)
    const real L = small ? real(7.4) : real(50.0);
    const int N = small ? 320 : 100000;
    auto pd = std::make_shared<ParticleData>(N);
    {
      auto pos = pd->getPos(access::cpu, access::write);
      auto lattice = initLattice(make_real3(L), N, sc);
      std::mt19937 gen(99);
      std::uniform_real_distribution<real> jitter(-0.1, 0.1);
      for (int i = 0; i < N; ++i) pos[i] = lattice[i] + make_real4(jitter(gen), jitter(gen), jitter(gen), 0);
    }
    const Box box(L);
    auto ljLibrary = std::make_shared<Potential::LJ>();
    ljLibrary->setPotParameters(0, 0, {real(2.5), real(1.0), real(1.0), false});
    PairForces<Potential::LJ>::Parameters parL; parL.box = box;
    auto reference = std::make_shared<PairForces<Potential::LJ>>(pd, parL, ljLibrary);
    std::vector<real4> rf, f; std::vector<real> re, rv, e, v;
    evaluate(pd, reference, rf, re, rv);
    using UserPot = Potential::Radial<UserLJ>;
    auto user = std::make_shared<UserPot>();
    user->setPotParameters(0, 0, {real(2.5), real(1.0), real(1.0)});
    {
      PairForces<UserPot>::Parameters par; par.box = box;
      auto pf = std::make_shared<PairForces<UserPot>>(pd, par, user);
      evaluate(pd, pf, f, e, v);
      fails += compare(small ? "user LJ, all pairs (box of 3 rc)" : "user LJ on the CellList vs PairForces<LJ>", f, e, v, rf, re, rv);
    }
    if (!small) {
      PairForces<UserPot, VerletList>::Parameters par; par.box = box;
      auto pf = std::make_shared<PairForces<UserPot, VerletList>>(pd, par, user);
      evaluate(pd, pf, f, e, v);
      fails += compare("user LJ on the VerletList vs PairForces<LJ>", f, e, v, rf, re, rv);
    }
  }
  {  // two particle types with their own parameters, inside an integrator
    const int N = 16384;
    const real L = 32;
    auto pd = std::make_shared<ParticleData>(N);
    {
      auto pos = pd->getPos(access::cpu, access::write);
      auto lattice = initLattice(make_real3(L), N, fcc);
      for (int i = 0; i < N; ++i) { pos[i] = lattice[i]; pos[i].w = real(i % 2); }
    }
    using Pot = Potential::Radial<Yukawa>;
    auto pot = std::make_shared<Pot>();
    pot->setPotParameters(0, 0, {real(3.0), real(1.0), real(2.0)});
    pot->setPotParameters(0, 1, {real(3.0), real(1.0), real(-1.0)});
    pot->setPotParameters(1, 1, {real(2.0), real(1.5), real(2.0)});
    if (pot->getCutOff() != real(3.0)) { ++fails; std::printf("FAIL getCutOff\n"); }
    PairForces<Pot>::Parameters par; par.box = Box(L);
    auto pf = std::make_shared<PairForces<Pot>>(pd, par, pot);
    VerletNVT::GronbechJensen::Parameters ipar;
    ipar.temperature = 1.0; ipar.dt = 0.002; ipar.friction = 1.0;
    auto nvt = std::make_shared<VerletNVT::GronbechJensen>(pd, ipar);
    nvt->addInteractor(pf);
    for (int s = 0; s < 20; ++s) nvt->forwardTime();
    auto pos = pd->getPos(access::cpu, access::read);
    auto force = pd->getForce(access::cpu, access::read);
    double fsum[3] = {0, 0, 0}, fabsmax = 0;
    bool finite = true;
    for (int i = 0; i < N; ++i) {
      finite = finite && std::isfinite(pos[i].x) && std::isfinite(force[i].x);
      fsum[0] += force[i].x; fsum[1] += force[i].y; fsum[2] += force[i].z;
      fabsmax = std::max(fabsmax, (double)std::fabs(force[i].x));
    }
    std::printf("two-type Yukawa, 20 NVT steps: finite %d, total force (%.2e %.2e %.2e) against max |f| %.2e\n", (int)finite, fsum[0], fsum[1], fsum[2], fabsmax);
    if (!finite || fabsmax <= 0 || std::fabs(fsum[0]) > 1e-3 * fabsmax * std::sqrt((double)N)) { ++fails; std::printf("FAIL Yukawa\n"); }  // Newton's third law
  }
  {  // ExternalForces<Functor> on a ParticleGroup selected by type, and with several arrays on everybody
    const int N = 5000;
    auto pd = std::make_shared<ParticleData>(N);
    std::vector<real4> hpos(N);
    std::vector<real3> hvel(N);
    {
      auto pos = pd->getPos(access::cpu, access::write);
      auto vel = pd->getVel(access::cpu, access::write);
      std::mt19937 gen(5);
      std::uniform_real_distribution<real> u(-10, 10);
      for (int i = 0; i < N; ++i) {
        hpos[i] = pos[i] = make_real4(u(gen), u(gen), u(gen), real(i % 3 == 0 ? 1 : 0));
        hvel[i] = vel[i] = make_real3(u(gen), u(gen), u(gen));
      }
      auto force = pd->getForce(access::cpu, access::write);
      auto energy = pd->getEnergy(access::cpu, access::write);
      std::fill(force.begin(), force.end(), real4());
      std::fill(energy.begin(), energy.end(), real(0));
    }
    auto species1 = std::make_shared<ParticleGroup>(particle_selector::Type(1), pd, "species 1");
    auto wall = std::make_shared<HarmonicWall>(real(2.5));
    auto ext = std::make_shared<ExternalForces<HarmonicWall>>(species1, wall);
    ext->updateSimulationTime(real(3.25));
    ext->sum({.force = true, .energy = true, .virial = false});
    auto drag = std::make_shared<ExternalForces<DragOnIds>>(pd);
    drag->sum({.force = true, .energy = false, .virial = false});
    auto force = pd->getForce(access::cpu, access::read);
    auto energy = pd->getEnergy(access::cpu, access::read);
    double worst = 0, worstE = 0;
    for (int i = 0; i < N; ++i) {
      const bool in = i % 3 == 0;
      const real g = (i % 2) ? real(0.5) : real(0);
      const real fz = (in ? real(-0.1) * (hpos[i].z - real(2.5)) : real(0)) - g * hvel[i].z;
      worst = std::max(worst, (double)std::fabs(force[i].z - fz));
      worst = std::max(worst, (double)std::fabs(force[i].x + g * hvel[i].x));
      const real e = in ? real(0.5) * real(0.1) * (hpos[i].z - real(2.5)) * (hpos[i].z - real(2.5)) : real(0);
      worstE = std::max(worstE, (double)std::fabs(energy[i] - e));
    }
    std::printf("ExternalForces: wall on the %d particles of type 1 + drag on odd ids: max |dF| %.2e, max |dE| %.2e, functor saw t = %.2f\n",
                species1->getNumberParticles(), worst, worstE, wall->lastTime);
    if (worst > 1e-5 || worstE > 1e-5 || wall->lastTime != real(3.25) || species1->getNumberParticles() != (N + 2) / 3) { ++fails; std::printf("FAIL ExternalForces\n"); }
  }
  {  // a Potential that is ParameterUpdatable hears what its PairForces hears (ParameterUpdatableDelegate, PairForces.cuh:25,40-44): through
     // an integrator — time every step, temperature and time step once — and directly; its Transverser follows the time
    const int N = 512;
    auto pd = std::make_shared<ParticleData>(N);
    {
      auto pos = pd->getPos(access::cpu, access::write);
      std::mt19937 gen(9);
      std::uniform_real_distribution<real> u(-8, 8);
      for (int i = 0; i < N; ++i) pos[i] = make_real4(u(gen), u(gen), u(gen), 0);
    }
    using PF = PairForces<GrowingRepulsion>;
    PF::Parameters par;
    par.box = Box(real(16));
    auto pot = std::make_shared<GrowingRepulsion>();
    auto pf = std::make_shared<PF>(pd, par, pot);
    VerletNVT::GronbechJensen::Parameters ipar;
    ipar.temperature = real(0.75); ipar.dt = real(0.002); ipar.friction = real(1.0);
    auto verlet = std::make_shared<VerletNVT::GronbechJensen>(pd, ipar);
    verlet->addInteractor(pf);
    for (int s = 0; s < 5; ++s) verlet->forwardTime();
    const bool heard = pot->temperature == real(0.75) && pot->dt == real(0.002) && pot->timeCalls == 5 && std::fabs(pot->time - real(4 * 0.002)) < 1e-7;
    pf->updateBox(Box(real(20)));
    pf->updateViscosity(real(3.5));
    pf->updateSimulationTime(real(2.0));   // strength(t) = 1 + t
    { auto f = pd->getForce(access::cpu, access::write); std::fill(f.begin(), f.end(), real4()); }
    pf->sum({.force = true, .energy = false, .virial = false});
    double f3 = 0;
    { auto f = pd->getForce(access::cpu, access::read); for (int i = 0; i < N; ++i) f3 += std::fabs(f[i].x) + std::fabs(f[i].y) + std::fabs(f[i].z); }
    pf->updateSimulationTime(real(0.0));
    { auto f = pd->getForce(access::cpu, access::write); std::fill(f.begin(), f.end(), real4()); }
    pf->sum({.force = true, .energy = false, .virial = false});
    double f1 = 0;
    { auto f = pd->getForce(access::cpu, access::read); for (int i = 0; i < N; ++i) f1 += std::fabs(f[i].x) + std::fabs(f[i].y) + std::fabs(f[i].z); }
    std::printf("ParameterUpdatable Potential behind PairForces: heard T = %.2f, dt = %.3f, %d time updates (last %.4f), box %.0f, viscosity %.1f; sum |f| at strength 3 / at strength 1 = %.5f\n",
                (double)pot->temperature, (double)pot->dt, pot->timeCalls, (double)pot->time, (double)pot->boxL, (double)pot->viscosity, f3 / f1);
    if (!heard || pot->boxL != real(20) || pot->viscosity != real(3.5) || !(f1 > 0) || std::fabs(f3 / f1 - 3.0) > 1e-4) { ++fails; std::printf("FAIL ParameterUpdatable Potential\n"); }
  }
  std::printf(fails ? "FAILED\n" : "ok\n");
  return fails;
}
