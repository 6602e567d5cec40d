// Two user-defined Transversers on the library's cell list, compiled by hipcc with this translation unit (the reference's
// docs/Transverser.rst examples): a neighbour counter, and a Lennard-Jones force functor written by the "user" that must
// reproduce the library's fused path bit for bit (same pairs, same order, same FMA placement).
#include "uammd/device/Transverser.hip.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace uammd::device;

struct NeighbourCounter {
  int *count;
  float rc2;
  float3 L;
  __device__ int compute(const real4 &pi, const real4 &pj) {
    float3 r = make_float3(pj.x - pi.x, pj.y - pi.y, pj.z - pi.z);
    r.x -= floorf(r.x / L.x + 0.5f) * L.x; r.y -= floorf(r.y / L.y + 0.5f) * L.y; r.z -= floorf(r.z / L.z + 0.5f) * L.z;
    const float r2 = r.x * r.x + r.y * r.y + r.z * r.z;
    return (r2 < rc2 && r2 > 0.0f) ? 1 : 0;
  }
  __device__ void set(int i, int total) { count[i] = total; }
};

struct UserLJ {  // Radial<LJFunctor>::Transverser (RadialPotential.cuh:107-127) as a user would write it: compute returns
                 // (|f|/r, r12), accumulate applies the library's FMA contract, set adds to the force array
  float4 *force;
  float rc2;
  float3 L, minusInvL;
  struct Cur { float fm, x, y, z; };
  __device__ Cur zero() { return Cur{0.f, 0.f, 0.f, 0.f}; }  // total: fm unused, (x,y,z) = force
  __device__ Cur compute(const real4 &pi, const real4 &pj) {
    float3 r = make_float3(pj.x - pi.x, pj.y - pi.y, pj.z - pi.z);
    r.x += floorf(fmaf(r.x, minusInvL.x, 0.5f)) * L.x;
    r.y += floorf(fmaf(r.y, minusInvL.y, 0.5f)) * L.y;
    r.z += floorf(fmaf(r.z, minusInvL.z, 0.5f)) * L.z;
    const float r2 = fmaf(r.z, r.z, fmaf(r.y, r.y, r.x * r.x));
    float fm = 0.0f;
    if (r2 != 0.0f && !(r2 >= rc2)) {
      const float invr2 = 1.0f / r2, invr6 = invr2 * invr2 * invr2;
      fm = 1.0f * fmaf(-48.0f, invr6, 24.0f) * invr6 * invr2;
    }
    return Cur{fm, r.x, r.y, r.z};
  }
  __device__ void accumulate(Cur &t, const Cur &c) { t.x = fmaf(c.fm, c.x, t.x); t.y = fmaf(c.fm, c.y, t.y); t.z = fmaf(c.fm, c.z, t.z); }
  __device__ void set(int i, Cur t) { float4 f = force[i]; f.x += t.x; f.y += t.y; f.z += t.z; f.w += 0.0f; force[i] = f; }
};

// A functor using every optional member of the concept (docs/Transverser.rst:25-54): per-particle Info (a "charge"), prepare() on
// the host before each launch, and dynamic LDS it asks for with getSharedMemorySize() (here a 4-entry table every thread fills
// with the same values in zero(), as the reference's own transversers fill theirs).
struct WeightedCounter {
  float *out;
  const float *charge;
  float rc2, scale;
  float3 L;
  using Info = float;
  void prepare(float newScale) { scale = newScale; }
  size_t getSharedMemorySize() { return 4 * sizeof(float); }
  __device__ Info getInfo(int i) { return charge[i]; }
  __device__ float zero() {
    extern __shared__ float table[];
    for (int k = 0; k < 4; ++k) table[k] = scale * (float)(k + 1);
    return 0.0f;
  }
  __device__ float compute(const real4 &pi, const real4 &pj, Info qi, Info qj) {
    extern __shared__ float table[];
    float3 r = make_float3(pj.x - pi.x, pj.y - pi.y, pj.z - pi.z);
    r.x -= floorf(r.x / L.x + 0.5f) * L.x; r.y -= floorf(r.y / L.y + 0.5f) * L.y; r.z -= floorf(r.z / L.z + 0.5f) * L.z;
    const float r2 = r.x * r.x + r.y * r.y + r.z * r.z;
    return (r2 < rc2 && r2 > 0.0f) ? table[0] * qi * qj : 0.0f;   // table[0] = scale
  }
  __device__ void set(int i, float total) { out[i] = total; }
};

#define CK(x) do { if ((x) != 0) { std::fprintf(stderr, "failed: %s (%s)\n", #x, uammd_hip_last_error()); return 2; } } while (0)

int main() {
  const int n = 20000;
  const float Lb = 30.0f, rc = 2.5f;
  std::vector<float4> pos(n);
  std::srand(7);
  const int m = 28;  // jittered lattice
  for (int i = 0; i < n; ++i) {
    const int ix = i % m, iy = (i / m) % m, iz = i / (m * m);
    auto jit = []() { return 0.2f * (std::rand() / (float)RAND_MAX - 0.5f); };
    pos[i] = make_float4((ix + 0.5f) / m * Lb - Lb / 2 + jit(), (iy + 0.5f) / m * Lb - Lb / 2 + jit(), (iz + 0.5f) / m * Lb - Lb / 2 + jit(), 0.f);
  }
  float4 *d_pos, *d_f1, *d_f2;
  int *d_count;
  hipMalloc(&d_pos, sizeof(float4) * n); hipMalloc(&d_f1, sizeof(float4) * n); hipMalloc(&d_f2, sizeof(float4) * n); hipMalloc(&d_count, sizeof(int) * n);
  hipMemcpy(d_pos, pos.data(), sizeof(float4) * n, hipMemcpyHostToDevice);
  hipMemset(d_f1, 0, sizeof(float4) * n); hipMemset(d_f2, 0, sizeof(float4) * n);
  const float L[3] = {Lb, Lb, Lb}, rc3[3] = {rc, rc, rc};
  const int per[3] = {1, 1, 1};
  int cd[3], po[3];
  float Lo[3];
  uammd_celllist *cl;
  CK(uammd_celllist_create(&cl));
  CK(uammd_celllist_create_grid(L, per, rc3, cd, Lo, po));
  CK(uammd_celllist_update(cl, (const float *)d_pos, n, Lo, po, cd, nullptr));
  // 1. neighbour counter
  NeighbourCounter nc{d_count, rc * rc, make_float3(Lb, Lb, Lb)};
  CK(transverseList(cl, nc));
  // 2. the user's LJ vs the library's fused LJ
  UserLJ lj{d_f1, rc * rc, make_float3(Lb, Lb, Lb), make_float3(-1.0f / Lb, -1.0f / Lb, -1.0f / Lb)};
  CK(transverseList(cl, lj));
  uammd_lj_pair_parameters p, *d_p;
  CK(uammd_lj_process_pair_parameters(rc, 1.0f, 1.0f, 0, &p));
  hipMalloc(&d_p, sizeof(p));
  hipMemcpy(d_p, &p, sizeof(p), hipMemcpyHostToDevice);
  CK(uammd_lj_transverse_celllist(cl, d_p, 1, L, per, (float *)d_f2, nullptr, nullptr, nullptr, UAMMD_LJ_ALGO_EXACT, nullptr));  // EXACT: the reference's summation order (AUTO sums in tile order)
  hipDeviceSynchronize();
  std::vector<int> count(n);
  std::vector<float4> f1(n), f2(n);
  hipMemcpy(count.data(), d_count, sizeof(int) * n, hipMemcpyDeviceToHost);
  hipMemcpy(f1.data(), d_f1, sizeof(float4) * n, hipMemcpyDeviceToHost);
  hipMemcpy(f2.data(), d_f2, sizeof(float4) * n, hipMemcpyDeviceToHost);
  // exact check of the counter against an O(N) host loop for the first 200 particles
  int countMismatch = 0;
  double mean = 0;
  for (int c : count) mean += c;
  mean /= n;
  for (int i = 0; i < 200; ++i) {
    int c = 0;
    for (int j = 0; j < n; ++j) {
      float dx = pos[j].x - pos[i].x, dy = pos[j].y - pos[i].y, dz = pos[j].z - pos[i].z;
      dx -= std::floor(dx / Lb + 0.5f) * Lb; dy -= std::floor(dy / Lb + 0.5f) * Lb; dz -= std::floor(dz / Lb + 0.5f) * Lb;
      const float r2 = dx * dx + dy * dy + dz * dz;
      c += (r2 < rc * rc && r2 > 0.0f) ? 1 : 0;
    }
    countMismatch += (c != count[i]);
  }
  int differing = 0;
  double maxerr = 0, maxf = 0;
  for (int i = 0; i < n; ++i) {
    differing += (f1[i].x != f2[i].x) + (f1[i].y != f2[i].y) + (f1[i].z != f2[i].z);
    maxerr = std::fmax(maxerr, std::fabs(f1[i].x - f2[i].x));
    maxf = std::fmax(maxf, std::fabs(f2[i].x));
  }
  std::printf("mean neighbours %.3f, %d of 200 counts differ from the host loop; user LJ vs fused LJ: %d differing words of %d, max |dF| %.3e of %.3e\n",
              mean, countMismatch, differing, 3 * n, maxerr, maxf);
  // 3. the same counter through the Verlet list's NeighbourContainer and through the all-pairs transverse: identical counts
  int *d_count2, *d_count3;
  hipMalloc(&d_count2, sizeof(int) * n); hipMalloc(&d_count3, sizeof(int) * n);
  uammd_verletlist *vl;
  CK(uammd_verletlist_create(&vl));
  CK(uammd_verletlist_update(vl, (const float *)d_pos, n, L, per, rc, nullptr, nullptr));
  NeighbourCounter ncv{d_count2, rc * rc, make_float3(Lb, Lb, Lb)}, ncn{d_count3, rc * rc, make_float3(Lb, Lb, Lb)};
  CK(transverseList(vl, ncv));
  CK(transverseNBody(d_pos, nullptr, ncn, n));
  // 4. every optional member at once: Info + prepare + dynamic LDS, on the cell list and on all pairs
  float *d_q, *d_w1, *d_w2;
  hipMalloc(&d_q, sizeof(float) * n); hipMalloc(&d_w1, sizeof(float) * n); hipMalloc(&d_w2, sizeof(float) * n);
  std::vector<float> q(n);
  for (int i = 0; i < n; ++i) q[i] = 1.0f + (i % 3);
  hipMemcpy(d_q, q.data(), sizeof(float) * n, hipMemcpyHostToDevice);
  WeightedCounter w1{d_w1, d_q, rc * rc, 0.0f, make_float3(Lb, Lb, Lb)}, w2{d_w2, d_q, rc * rc, 0.0f, make_float3(Lb, Lb, Lb)};
  CK(transverseList(cl, w1, nullptr, nullptr, 0.5f));            // prepare(0.5f)
  CK(transverseNBody(d_pos, nullptr, w2, n, nullptr, 0.5f));
  hipDeviceSynchronize();
  std::vector<int> count2(n), count3(n);
  std::vector<float> wa(n), wb(n);
  hipMemcpy(count2.data(), d_count2, sizeof(int) * n, hipMemcpyDeviceToHost);
  hipMemcpy(count3.data(), d_count3, sizeof(int) * n, hipMemcpyDeviceToHost);
  hipMemcpy(wa.data(), d_w1, sizeof(float) * n, hipMemcpyDeviceToHost);
  hipMemcpy(wb.data(), d_w2, sizeof(float) * n, hipMemcpyDeviceToHost);
  int verletDiff = 0, nbodyDiff = 0, weightedDiff = 0;
  double wsum = 0;
  for (int i = 0; i < n; ++i) {
    verletDiff += count2[i] != count[i];
    nbodyDiff += count3[i] != count[i];
    weightedDiff += wa[i] != wb[i];            // sums of small multiples of 0.5: exact in any order
    wsum += wa[i];
  }
  std::printf("Verlet container: %d counts differ; all-pairs transverse: %d differ; Info + prepare + LDS functor: %d differ (sum %.1f)\n",
              verletDiff, nbodyDiff, weightedDiff, wsum);
  uammd_verletlist_destroy(vl);
  uammd_celllist_destroy(cl);
  return (countMismatch <= 1 && maxerr <= 1e-5 * maxf && verletDiff == 0 && nbodyDiff == 0 && weightedDiff == 0 && wsum > 0) ? 0 : 1;  // one count may differ: pairs within an ulp of the cut-off (fma vs mul)
}
