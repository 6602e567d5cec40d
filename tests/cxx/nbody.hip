// Interactor/NBody.cuh and NBodyBase.cuh (reference: src/Interactor/NBody.cuh:33-58, NBodyBase.cuh:112-166) with Transversers of the
// program's own: (1) the bare form — compute(pi, pj), sums through operator+ — counting the members closer than a cut-off, on a subgroup;
// (2) the full form — zero, accumulate, getInfo, getSharedMemorySize, prepare(pd) — a softened gravity-like force weighted by the
// partner's mass; (3) NBodyBase outside the classes, on a thrust vector of floats with a permutation as index iterator.  Each against
// the same double loop on the host.
#include "uammd.cuh"
#include "Interactor/NBody.cuh"
#include <thrust/device_vector.h>
#include <thrust/host_vector.h>
#include <cstdio>
#include <random>
#include <vector>
using namespace uammd;

struct CountCloser {   // bare Transverser
  int *count;
  real rc2;
  __device__ int compute(real4 pi, real4 pj) {
    const real3 d = make_real3(pj) - make_real3(pi);
    return dot(d, d) < rc2 ? 1 : 0;    // (the particle itself is counted too: i meets every j, itself included)
  }
  __device__ void set(int i, int total) { count[i] = total; }
};

struct MassWeighted {   // the optional members
  real4 *force;
  const real *mass = nullptr;
  int prepared = 0;
  void prepare(std::shared_ptr<ParticleData> pd) { mass = pd->getMass(access::gpu, access::read).raw(); ++prepared; }
  size_t getSharedMemorySize() { return 24; }   // (not used by the functor: the launcher must make room and keep its own blocks aligned)
  __device__ real getInfo(int i) { return mass[i]; }
  __device__ real3 zero() { return make_real3(0); }
  __device__ real3 compute(real4 pi, real4 pj, real mi, real mj) {
    const real3 d = make_real3(pj) - make_real3(pi);
    const real r2 = dot(d, d) + real(0.01);
    return d * (mi * mj / (r2 * sqrt(r2)));
  }
  __device__ void accumulate(real3 &total, const real3 &cur) { total += cur; }
  __device__ void set(int i, real3 total) { force[i] = make_real4(total, 0); }
};

struct SumOfProducts {   // NBodyBase on plain floats
  float *out;
  __device__ float compute(float a, float b) { return a * b; }
  __device__ void set(int i, float total) { out[i] = total; }
};

int main(int argc, char *argv[]) {
  auto sys = std::make_shared<System>(argc, argv);
  const int N = 1000;   // (not a multiple of the 128-wide tile)
  auto pd = std::make_shared<ParticleData>(N, sys);
  std::vector<real4> hp(N);
  std::vector<real> hm(N);
  {
    std::mt19937 gen(31);
    std::uniform_real_distribution<double> u(-5, 5), m(0.5, 2);
    auto pos = pd->getPos(access::cpu, access::write);
    auto mass = pd->getMass(access::cpu, access::write);
    for (int i = 0; i < N; ++i) { hp[i] = pos[i] = make_real4(u(gen), u(gen), u(gen), i % 2); hm[i] = mass[i] = m(gen); }
  }
  int bad = 0;
  {   // (1) subgroup of the odd particles
    auto pg = std::make_shared<ParticleGroup>(particle_selector::Type(1), pd, "odd");
    thrust::device_vector<int> count(N, -1);
    CountCloser tr{thrust::raw_pointer_cast(count.data()), real(9.0)};
    NBody(pg).transverse(tr);
    thrust::host_vector<int> got = count;
    int wrong = 0;
    for (int i = 0; i < N; ++i) {
      int expect = -1;
      if (i % 2) { expect = 0; for (int j = 1; j < N; j += 2) { const real3 d = make_real3(hp[j]) - make_real3(hp[i]); expect += dot(d, d) < real(9.0); } }
      wrong += got[i] != expect;
    }
    std::printf("NBody on a subgroup, bare Transverser: %d of %d counts differ (%d members)\n", wrong, N, pg->getNumberParticles());
    bad += wrong != 0;
  }
  {   // (2) everybody, optional members
    thrust::device_vector<real4> force(N);
    MassWeighted tr;
    tr.force = thrust::raw_pointer_cast(force.data());
    NBody nb(pd);
    nb.transverse(tr);
    thrust::host_vector<real4> got = force;
    double worst = 0, scale = 0;
    for (int i = 0; i < N; ++i) {
      double fx = 0, fy = 0, fz = 0;
      for (int j = 0; j < N; ++j) {
        const double dx = hp[j].x - hp[i].x, dy = hp[j].y - hp[i].y, dz = hp[j].z - hp[i].z, r2 = dx * dx + dy * dy + dz * dz + (double)real(0.01);
        const double w = hm[i] * hm[j] / (r2 * std::sqrt(r2));
        fx += dx * w; fy += dy * w; fz += dz * w;
      }
      worst = std::max(worst, std::max(std::abs(got[i].x - fx), std::max(std::abs(got[i].y - fy), std::abs(got[i].z - fz))));
      scale = std::max(scale, std::abs(fx));
    }
    std::printf("NBody, zero / accumulate / getInfo / getSharedMemorySize / prepare: worst deviation %.2e of %.2e, prepare called %d time(s)\n", worst, scale, tr.prepared);
    bad += !(worst < 2e-5 * scale) || tr.prepared != 1;
  }
  {   // (3) NBodyBase on floats with a permutation
    const int M = 300;
    std::vector<float> hv(M);
    std::vector<int> perm(M);
    for (int i = 0; i < M; ++i) { hv[i] = float(i % 17) * 0.25f - 1.0f; perm[i] = (7 * i + 3) % M; }   // 7 and 300 are coprime: a permutation
    thrust::device_vector<float> v = hv, out(M, -1.0f);
    thrust::device_vector<int> p = perm;
    SumOfProducts tr{thrust::raw_pointer_cast(out.data())};
    NBodyBase::transverse(v.begin(), p.begin(), tr, M);
    thrust::host_vector<float> got = out;
    int wrong = 0;
    for (int i = 0; i < M; ++i) {
      float expect = 0;
      for (int t = 0; t < M; ++t) expect += hv[i] * hv[perm[t]];   // the order the kernel meets them in: bit-equal sums
      wrong += got[i] != expect;
    }
    std::printf("NBodyBase on a thrust vector of floats through a permutation: %d of %d sums differ\n", wrong, M);
    bad += wrong != 0;
    thrust::device_vector<float> out2(M, -1.0f);
    SumOfProducts tr2{thrust::raw_pointer_cast(out2.data())};
    NBodyBase::transverse(thrust::raw_pointer_cast(v.data()), tr2, M);
    thrust::host_vector<float> got2 = out2;
    wrong = 0;
    for (int i = 0; i < M; ++i) { float expect = 0; for (int t = 0; t < M; ++t) expect += hv[i] * hv[t]; wrong += got2[i] != expect; }
    std::printf("NBodyBase without indices: %d of %d sums differ\n", wrong, M);
    bad += wrong != 0;
  }
  std::printf(bad ? "nbody: FAILED\n" : "nbody: ok\n");
  return bad;
}
