// Path B, spectral Ewald variant — PSE near field for gfx950.
//
// Reference behaviour:
//   RPYPSE_near::FandG                                   Integrator/BDHI/PSE/RPY_PSE.cuh:45-128 (host, double, closed form)
//   TabulatedFunction<real2> (linear interpolation)     misc/TabulatedFunction.cuh:63-157
//   NearField::initializeDeterministicPart               Integrator/BDHI/PSE/NearField.cuh:65-99
//   RPYNearTransverser::compute/set over a CellList      NearField.cuh:120-196, NeighbourList/common.cuh:10-34
//   NearField::Mdot / computeStochasticDisplacements     NearField.cuh:239-285  (Saru noise -> lanczos::Solver)
// M_near v is a sparse matrix-vector product: (M v)_i = sum_j F(r) v_j + (G(r) - F(r)) (r.v_j) r / r^2 over the 27 cells,
// F and G read from a table of 2^14..2^22 real2 points.  The reference gathers v_j through the group index for every
// neighbour (getInfo); here v is gathered ONCE into cell order (same values, contiguous with the positions).
#include "lanczos_fused.hpp"
#include "celllist.hpp"
#include "saru.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

namespace uammd_hip {

struct PSENear {
  CellList cl;
  uammd_lanczos *lanczos = nullptr;
  DeviceBuffer table, table4, sortV, noise;
  int nearKernel = 1;  // option "near_kernel": 1 = eight lanes per particle (k_pse_near8, default: 55 us at the bench size), 0 = wave per cell with
                       // LDS-staged candidates (k_pse_near_cell: 73 us there — fewer loads, but latency bound at 3 waves per SIMD)
  int nPointsTable = 0;
  float rcut = 0.f, shear = 0.f, tolerance = 0.f;
  float boxL[3] = {0, 0, 0};
  unsigned int seed = 0;
  int N = 0;
  hipStream_t cbStream = 0;
  bool exactOrder = false;  // option "exact_order": the thread-per-particle walk (the reference's summation order, bit for bit)
  // option "lazy_list": like CellList::update (CellList.cuh:134-136,192-204) the list is rebuilt only after the positions were
  // written (uammd_pse_near_positions_changed, wired to ParticleData's write signal by the host layer) or when the array / N changes
  bool lazyList = false, listValid = false;
  const void *listPos = nullptr;
  hipStream_t listStream = nullptr;
  DeviceBuffer sortedOut;   // Lanczos result in cell order
  // Pair records (option "pair_list", default 1; used with "lazy_list"): the near field is applied ~8 times per step (the deterministic
  // product + the Lanczos iterations) to positions that do not move in between, so the pairs inside the cut-off — and everything about
  // a pair that does not depend on the vector: F, (G - F) / r^2, the distance vector — are found ONCE per list build
  // (k_pse_pairs_build: k_pse_near8's scan, the hits written out instead of summed) and every product is a stream over 24-byte records
  // (k_pse_near_pairs: two dependent loads per particle instead of ~17).
  bool pairList = true, pairsValid = false, pairsUnfit = false;
  DeviceBuffer recA, recB, pairRange, pairCursor;  // float4 (F, (G - F) / r2, rx, ry) | float2 (rz, j) per record; int2 (first, count) per particle
  size_t pairCap = 0;        // records allocated
  int pairRegions = 1;       // ... cut into this many regions by the last build (k_pse_pairs_build)
  int *pairTotalHost = nullptr;  // pinned, mapped: the build's status block {., hit list overflow, |displacement|^2 bits, candidate overflow, ..., region cursors}
  int *pairTotalDev = nullptr;   // its device address
  bool statusCopyDeferred = false;   // the copy of the status block is left to the solve's noise kernel (k_pse_noise_sorted_norm)
  bool statusZeroed = false;         // k_pse_refresh_sorted has zeroed the status block for the coming build
  DeviceBuffer zparts;               // |noise|^2 partials of k_pse_noise_sorted_norm
  // a build is two halves: the launch (kernel + copy of the two counters + pairsEvent) and the read of the counters.  A caller with other
  // work for the stream queues it between the two (uammd_pse_near_prepare, then the far field, then the near products): the read then
  // finds the event complete and the GPU busy — waited for on the spot it was a bubble of ~20 us per step
  bool pairsPending = false;
  hipEvent_t pairsEvent = nullptr;
  // option "optimistic_records" (1): uammd_pse_near_stochastic streams the records of a build whose counters it has not read yet and reads
  // them AFTER the Lanczos run (which has waited for the GPU by then): no wait at the start of the solve; a build that did not fit — the
  // kernel then wrote no records for the particles past the capacity — is repeated larger and the solve run again from the same noise,
  // with the solver's adaptive schedule put back.  `optimistic` is set only inside that call.
  bool optimisticRecords = true, optimistic = false;
  // Candidate lists kept over several steps (option "list_skin_percent", default 40; with "lazy_list" + "pair_list" and no shear).  A
  // Brownian step moves a particle by a small fraction of the cut-off, and what the list build spends its time on is the 27-cell scan
  // (~195 candidates per particle for ~29 neighbours at the bench's size) plus the cell list before it.  With skin = percent / 100 x rc:
  // a full build bins on a grid of rc + skin, and its scan also writes every particle's candidates within rc + skin (k_pse_pairs_build
  // MODE 1); the steps after it keep the particle ORDER and the candidate lists, refresh the sorted positions (k_pse_refresh_sorted)
  // and make the records from the candidates alone (MODE 2) — exact as long as nobody has moved more than skin / 2 since the full build,
  // which the refresh measures and the host reads with the records' counters: a broken bound repeats the build from a fresh list (and,
  // under "optimistic_records", the solve), exactly as a build that outgrew its arrays.  The host schedules a full build BEFORE the
  // bound is expected to break (last reading + twice the largest one-step increase seen), so a repeat is the exception.  Lists that
  // do not survive two steps three times running turn the mechanism off for the handle.  Results: the same pairs, summed in the
  // candidate list's order instead of the scan's (rounding-level differences).
  int skinPercent = 40;
  bool candEnabled = true;    // (auto-off: candShortLived)
  bool candValid = false;     // candList / candCount / refPos belong to cl's current order
  bool candBuild = false, candScan = false;   // what the last launched pair build was: MODE 1 / MODE 2 (neither: MODE 0)
  DeviceBuffer candList, candCount, refPos, dispParts;
  int candCap = 0, candGrow = 0;
  int stepsSinceFull = 0, candShortLived = 0;
  float skin = 0.f, lastDisp = 0.f, maxInc = 0.f;
  long long fullBuilds = 0, candBuilds = 0, candRepeats = 0;   // diagnostics (uammd_pse_near_pair_records' sibling option reads)
  size_t pairCapFirst = 0;   // option "pair_capacity": the first allocation in records (tests: a build that has to grow); 0 = 48 per particle
  uammd_interleave_fn interleave = nullptr;   // uammd_pse_near_set_interleave (one-shot, handed to the next solve)
  void *interleaveCtx = nullptr;
  uammd_interleave_fn interleaveEarly = nullptr;   // uammd_pse_near_set_interleave_early
  void *interleaveEarlyCtx = nullptr;
  // uammd_pse_near_set_mdot_rider (one-shot): the next uammd_pse_near_stochastic also adds M_near F to riderMF — as a second right-hand
  // side of the solve's FIRST product (the records are streamed once for both), F gathered into the list's order by the noise kernel,
  // the result added to riderMF by the kernel that unsorts the solve's result.  riderLaunched: the product of the current solve took it.
  const float *riderForce = nullptr;
  float *riderMF = nullptr;
  bool riderArmed = false, riderLaunched = false;
  DeviceBuffer riderFs, riderMFs;   // F and M_near F in the list's order (real3)
  ~PSENear() {
    if (lanczos) uammd_lanczos_destroy(lanczos);
    if (pairTotalHost) (void)hipHostFree(pairTotalHost);
    if (pairsEvent) (void)hipEventDestroy(pairsEvent);
  }
};

// the closed form of eq. A3-A4 of Fiore et al. 2017 as the reference evaluates it (coefficient sets f0..f7, g0..g7)
void rpy_near_FandG(double r, double rh, double psi, double rcut, double *F, double *G) {  // (also used by the double-precision build, f64.hip)
  *F = *G = 0.0;
  if (r >= rcut) return;
  const double spi = std::sqrt(M_PI);
  if (r <= 0.0) {
    *F = (1.0 / (4 * spi * psi * rh)) * (1 - std::exp(-4 * rh * rh * psi * psi) + 4 * spi * rh * psi * std::erfc(2 * rh * psi));
    return;
  }
  const double r2 = r * r, r3 = r2 * r, r4 = r3 * r;
  const double a2mr = 2 * rh - r, a2pr = 2 * rh + r;
  const double rh2 = rh * rh, rh4 = rh2 * rh2;
  const double psi2 = psi * psi, psi3 = psi2 * psi, psi4 = psi2 * psi2;
  double f[8], g[8];
  if (r > 2 * rh) {
    f[0] = (64.0 * rh4 * psi4 + 96.0 * rh2 * r2 * psi4 - 128.0 * rh * r3 * psi4 + 36.0 * r4 * psi4 - 3.0) / (128.0 * rh * r3 * psi4);
    f[4] = (3.0 - 4.0 * psi4 * a2mr * a2mr * (4.0 * rh2 + 4.0 * rh * r + 9.0 * r2)) / (256.0 * rh * r3 * psi4);
    f[5] = 0;
    g[0] = (-64.0 * rh4 * psi4 + 96.0 * rh2 * r2 * psi4 - 64.0 * rh * r3 * psi4 + 12.0 * r4 * psi4 + 3.0) / (64.0 * rh * r3 * psi4);
    g[4] = (4.0 * psi4 * a2mr * a2mr * a2mr * (2.0 * rh + 3.0 * r) - 3.0) / (128.0 * rh * r3 * psi4);
    g[5] = 0;
  } else {
    f[0] = (-16.0 * rh4 - 24.0 * rh2 * r2 + 32.0 * rh * r3 - 9.0 * r4) / (32.0 * rh * r3);
    f[4] = 0;
    f[5] = (4.0 * psi4 * a2mr * a2mr * (4.0 * rh2 + 4.0 * rh * r + 9.0 * r2) - 3.0) / (256.0 * rh * r3 * psi4);
    g[0] = a2mr * a2mr * a2mr * (2.0 * rh + 3.0 * r) / (16.0 * rh * r3);
    g[4] = 0;
    g[5] = (3.0 - 4.0 * psi4 * a2mr * a2mr * a2mr * (2.0 * rh + 3.0 * r)) / (128.0 * rh * r3 * psi4);
  }
  f[1] = (-2.0 * psi2 * a2pr * (4.0 * rh2 - 4.0 * rh * r + 9.0 * r2) + 2.0 * rh - 3.0 * r) / (128.0 * rh * r3 * psi3 * spi);
  f[2] = (2.0 * psi2 * a2mr * (4.0 * rh2 + 4.0 * rh * r + 9.0 * r2) - 2.0 * rh - 3.0 * r) / (128.0 * rh * r3 * psi3 * spi);
  f[3] = 3.0 * (6.0 * r2 * psi2 + 1.0) / (64.0 * spi * rh * r2 * psi3);
  f[6] = (4.0 * psi4 * a2pr * a2pr * (4.0 * rh2 - 4.0 * rh * r + 9.0 * r2) - 3.0) / (256.0 * rh * r3 * psi4);
  f[7] = 3.0 * (1.0 - 12.0 * r4 * psi4) / (128.0 * rh * r3 * psi4);
  g[1] = (2.0 * psi2 * a2pr * a2pr * (2.0 * rh - 3.0 * r) - 2.0 * rh + 3.0 * r) / (64.0 * spi * rh * r3 * psi3);
  g[2] = (-2.0 * psi2 * a2mr * a2mr * (2.0 * rh + 3.0 * r) + 2.0 * rh + 3.0 * r) / (64.0 * spi * rh * r3 * psi3);
  g[3] = (3.0 * (2.0 * r2 * psi2 - 1.0)) / (32.0 * spi * rh * r2 * psi3);
  g[6] = (3.0 - 4.0 * psi4 * (2.0 * rh - 3.0 * r) * a2pr * a2pr * a2pr) / (128.0 * rh * r3 * psi4);
  g[7] = -3.0 * (4.0 * r4 * psi4 + 1.0) / (64.0 * rh * r3 * psi4);
  const double e[8] = {1.0,
                       std::exp(-psi2 * a2pr * a2pr),
                       std::exp(-a2mr * a2mr * psi2),
                       std::exp(-psi2 * r2),
                       std::erfc(a2mr * psi),
                       std::erfc(-a2mr * psi),
                       std::erfc(a2pr * psi),
                       std::erfc(r * psi)};
  // the reference sums the eight terms left to right (RPY_PSE.cuh:124-127)
  double sf = f[0], sg = g[0];
  for (int t = 1; t < 8; ++t) { sf += f[t] * e[t]; sg += g[t] * e[t]; }
  *F = sf;
  *G = sg;
}

struct TableView {
  const float2 *table;
  int Ntable;
  float rmax, interval, dr;
};

// TabulatedFunction::operator() with LinearInterpolation and lerp (TabulatedFunction.cuh:36-45, :63-75, :148-157), rmin = 0
UH_D float2 table_get(const TableView &t, float rs) {
  const float r = rs * t.interval;
  if (rs >= t.rmax) return make_float2(0.f, 0.f);
  if (r <= 0.0f) return t.table[0];
  const int i = (int)(r * (float)t.Ntable);
  const float r0 = (float)i * t.dr;
  const float2 v0 = t.table[i], v1 = t.table[i + 1];
  const float w = (r - r0) * (float)t.Ntable;
  return make_float2(fmaf(w, v1.x, fmaf(-w, v0.x, v0.x)), fmaf(w, v1.y, fmaf(-w, v0.y, v0.y)));
}

// RPYNearTransverser::computeShearedDistancePBC (NearField.cuh:134-152)
UH_D real3f sheared_distance(const float4 &pi, const float4 &pj, real3f L, float shear) {
  real3f rij{pj.x - pi.x, pj.y - pi.y, pj.z - pi.z};
  rij.x = fmaf(shear, rij.y, rij.x);
  const float s1 = roundf(rij.y / L.y);
  rij.x = fmaf(-(shear * L.y), s1, rij.x);
  rij.y = fmaf(-L.y, s1, rij.y);
  rij.z = fmaf(-L.z, roundf(rij.z / L.z), rij.z);
  rij.x = fmaf(-L.x, roundf(rij.x / L.x), rij.x);
  return rij;
}

template <int VSTRIDE>
__global__ void __launch_bounds__(256) k_pse_gather_v(const float *__restrict__ v, const int *__restrict__ groupIndex,
                                                       float4 *__restrict__ sortV, int N) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= N) return;
  const float *p = v + (size_t)VSTRIDE * groupIndex[id];
  sortV[id] = make_float4(p[0], p[1], p[2], 0.f);
}

// thread per sorted particle, 27-cell walk in the reference's order; Mv[ori] += total
__global__ void __launch_bounds__(128) k_pse_near(const float4 *__restrict__ sortPos, const float4 *__restrict__ sortV,
                                                   const int *__restrict__ groupIndex, const uint *__restrict__ cellStart,
                                                   const int *__restrict__ cellEnd, uint validCell, int N, GridT<float> grid,
                                                   real3f L, float shear, float rcut2, TableView tab, float *__restrict__ Mv) {
  const int id = (int)xcd_contiguous_block(blockIdx.x, gridDim.x) * 128 + threadIdx.x;
  if (id >= N) return;
  const float4 pi = sortPos[id];
  const int3 n = grid.cellDim;
  const int npx = n.x > 1 ? 3 : 1, npy = n.y > 1 ? 3 : 1, npz = n.z > 1 ? 3 : 1;
  const int numberNeighbourCells = npx * npy * npz;
  const int3 celli = grid.getCell(real3f{pi.x, pi.y, pi.z});
  float tx = 0.f, ty = 0.f, tz = 0.f;
  for (int cc = 0; cc < numberNeighbourCells; ++cc) {
    int3 cellj = celli;
    if (npx > 1) cellj.x += cc % 3 - 1;
    if (npy > 1) cellj.y += (cc / npx) % 3 - 1;
    if (npz > 1) cellj.z += cc / (npx * npy) - 1;
    cellj.x = grid.pbc_x(cellj.x);
    cellj.y = grid.pbc_y(cellj.y);
    cellj.z = grid.pbc_z(cellj.z);
    if (cellj.x < 0 || cellj.x >= n.x || cellj.y < 0 || cellj.y >= n.y || cellj.z < 0 || cellj.z >= n.z) continue;
    const int icellj = grid.getCellIndex(cellj);
    const uint cs = cellStart[icellj];
    if (cs < validCell) continue;
    const int first = (int)(cs - validCell), last = cellEnd[icellj];
    for (int j = first; j < last; ++j) {
      const real3f rij = sheared_distance(pi, sortPos[j], L, shear);
      const float r2 = dot3(rij, rij);
      if (r2 >= rcut2) continue;
      const float4 vj = sortV[j];
      const float2 fg = table_get(tab, sqrtf(r2));
      const float f = fg.x, g = fg.y;
      float rx, ry, rz;
      if (r2 == 0.0f) {
        rx = f * vj.x; ry = f * vj.y; rz = f * vj.z;
      } else {
        const float invr2 = 1.0f / r2;
        const float gmfv = (g - f) * dot3(rij, real3f{vj.x, vj.y, vj.z}) * invr2;
        rx = fmaf(gmfv, rij.x, f * vj.x);
        ry = fmaf(gmfv, rij.y, f * vj.y);
        rz = fmaf(gmfv, rij.z, f * vj.z);
      }
      tx += rx; ty += ry; tz += rz;
    }
  }
  float *o = Mv + 3 * (size_t)groupIndex[id];
  o[0] += tx; o[1] += ty; o[2] += tz;
}

// ---- AUTO: eight lanes per sorted particle --------------------------------------------------------------------------------------
// The thread-per-particle walk above is one dependent chain of ~200 candidates (load, sheared image, compare, and for a hit a square
// root, two dependent table reads and a 3 x 3 product) on 1.5 waves per SIMD at N = 1e5: 201 us per product, latency from end to end.
// Here a group of 8 lanes shares a particle.  Scan: the 27 (first, last) ranges are fetched up front (lane q of the group takes cells
// q, q + 8, ...), then the group tests 8 candidates per step with a cheap SUPERSET test (reciprocal multiplications instead of the
// reference's divisions, 1e-5 of slack) and appends the hits, in the reference's visiting order, to the group's list in LDS (wave ballot,
// the group's byte of the mask).  Drain: lane k evaluates hits k, k + 8, ... exactly as the reference does (sheared_distance with its
// divisions and roundf, `r2 >= rcut2 -> nothing`, the same table arithmetic), so the SAME pairs contribute the SAME terms; the eight
// partial sums meet in a 3-step butterfly.  Another summation order: results agree with the walk to rounding (tests: 1e-6 of max|Mv|).
// k_pse_pairs_build's record space is cut into kPairRegions equal regions, workgroup b reserving in region b % kPairRegions through that
// region's own cursor (64 bytes apart: status[16 + 16 r]): 3125 returning atomics on ONE address are served one after the other by the
// memory side (~20 ns each: the whole 64 us of the kernel as it was), 49 on each of 64 addresses are not.
#ifndef UAMMD_PSE_PAIR_REGIONS
#define UAMMD_PSE_PAIR_REGIONS 64
#endif
constexpr int kPairRegions = UAMMD_PSE_PAIR_REGIONS, kPairStatusInts = 16 + 16 * kPairRegions;
#ifndef UAMMD_PSE_CAND_ROWS
#define UAMMD_PSE_CAND_ROWS 2
#endif
#ifndef UAMMD_PSE_REC_ROWS
#define UAMMD_PSE_REC_ROWS 1
#endif
#ifndef UAMMD_PSE_BUILD_ROWS
#define UAMMD_PSE_BUILD_ROWS 2
#endif
constexpr int kNearGroup = 8, kNearBlock = 256, kNearCap = 96;  // hits a group can hold before it drains (12 KB of LDS per workgroup)

// The sheared minimum image with the image counts from a reciprocal multiplication and round-to-even instead of the reference's
// division and roundf.  The distance vector depends on the COUNTS only (rij - L * count, one fma per component, as in
// sheared_distance): it is the reference's, bit for bit, whenever the counts agree — and they can only differ for a component within an
// ulp of half a box length, i.e. for pairs far outside any near-field cut-off (cut-off <= cell edge <= L / 3).
template <bool SHEAR>
UH_D real3f scan_rij(const float4 &pi, const float4 &pj, real3f L, real3f invL, float shear) {
  float x = pj.x - pi.x, y = pj.y - pi.y, z = pj.z - pi.z;
  if (SHEAR) x = fmaf(shear, y, x);
  const float s1 = __builtin_rintf(y * invL.y);
  if (SHEAR) x = fmaf(-(shear * L.y), s1, x);
  y = fmaf(-L.y, s1, y);
  z = fmaf(-L.z, __builtin_rintf(z * invL.z), z);
  x = fmaf(-L.x, __builtin_rintf(x * invL.x), x);
  return real3f{x, y, z};
}
template <bool SHEAR>
UH_D float scan_distance2(const float4 &pi, const float4 &pj, real3f L, real3f invL, float shear) {
  const real3f r = scan_rij<SHEAR>(pi, pj, L, invL, shear);
  return fmaf(r.z, r.z, fmaf(r.y, r.y, r.x * r.x));
}

template <int VSTRIDE, bool INDIRECT, bool ACCUM, bool SHEAR>
__global__ void __launch_bounds__(kNearBlock) k_pse_near8(const float4 *__restrict__ sortPos, const float *__restrict__ v,
                                                           const int *__restrict__ groupIndex, const uint *__restrict__ cellStart,
                                                           const int *__restrict__ cellEnd, uint validCell, int N, GridT<float> grid,
                                                           real3f L, float shear, float rcut2, TableView tab, float *__restrict__ Mv) {
  __shared__ int hitList[kNearBlock / kNearGroup][kNearCap];
  __shared__ int2 ranges[kNearBlock / kNearGroup][29];  // {first particle, offset in the group's flat candidate sequence} + sentinel
  const int lane = threadIdx.x & 63, sub = threadIdx.x & (kNearGroup - 1), gbase = lane & ~(kNearGroup - 1);
  const int grp = threadIdx.x / kNearGroup;
  const int id = (int)xcd_contiguous_block(blockIdx.x, gridDim.x) * (kNearBlock / kNearGroup) + grp;
  const bool active = id < N;
  const float4 pi = sortPos[active ? id : 0];
  const int3 n = grid.cellDim;
  const int npx = n.x > 1 ? 3 : 1, npy = n.y > 1 ? 3 : 1, npz = n.z > 1 ? 3 : 1;
  const int numberNeighbourCells = npx * npy * npz;
  const int3 celli = grid.getCell(real3f{pi.x, pi.y, pi.z});
  const real3f invL{1.0f / L.x, 1.0f / L.y, 1.0f / L.z};
  const float rcut2s = rcut2 * 1.00001f + 1e-30f;
  // ranges of the neighbour cells, four per lane
  int first4[4], last4[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int cc = sub + kNearGroup * q;
    first4[q] = 0; last4[q] = 0;
    if (active && cc < numberNeighbourCells) {
      int3 cellj = celli;
      if (npx > 1) cellj.x += cc % 3 - 1;
      if (npy > 1) cellj.y += (cc / npx) % 3 - 1;
      if (npz > 1) cellj.z += cc / (npx * npy) - 1;
      cellj.x = grid.pbc_x(cellj.x);
      cellj.y = grid.pbc_y(cellj.y);
      cellj.z = grid.pbc_z(cellj.z);
      if (!(cellj.x < 0 || cellj.x >= n.x || cellj.y < 0 || cellj.y >= n.y || cellj.z < 0 || cellj.z >= n.z)) {
        const int icellj = grid.getCellIndex(cellj);
        const uint cs = cellStart[icellj];
        if (cs >= validCell) { first4[q] = (int)(cs - validCell); last4[q] = cellEnd[icellj]; }
      }
    }
  }
  float tx = 0.f, ty = 0.f, tz = 0.f;
  int cnt = 0;  // hits in this group's list (the same number in its eight lanes)
  auto drain = [&]() {
    for (int k = sub; k < cnt; k += kNearGroup) {
      const int j = hitList[grp][k];
      const real3f rij = scan_rij<SHEAR>(pi, sortPos[j], L, invL, shear);  // (= sheared_distance for every pair that can pass the next line)
      const float r2 = dot3(rij, rij);
      if (r2 >= rcut2) continue;  // (the scan's test is a superset)
      const float *vp = v + (size_t)VSTRIDE * (INDIRECT ? groupIndex[j] : j);
      const float vx = vp[0], vy = vp[1], vz = vp[2];
      const float2 fg = table_get(tab, sqrtf(r2));
      const float f = fg.x, g = fg.y;
      float rx, ry, rz;
      if (r2 == 0.0f) {
        rx = f * vx; ry = f * vy; rz = f * vz;
      } else {
        const float invr2 = 1.0f / r2;
        const float gmfv = (g - f) * dot3(rij, real3f{vx, vy, vz}) * invr2;
        rx = fmaf(gmfv, rij.x, f * vx);
        ry = fmaf(gmfv, rij.y, f * vy);
        rz = fmaf(gmfv, rij.z, f * vz);
      }
      tx += rx; ty += ry; tz += rz;
    }
    cnt = 0;
  };
  // The group's candidates as ONE flat sequence: the non-empty ranges are compacted, in visiting order, into LDS with their running
  // offsets (cell cc = sub + 8 q: q-major order); step t0 tests candidates t0 .. t0 + 7 of the sequence, each lane walking its own
  // range pointer forward.  Cell by cell a group spent ceil(n / 8) steps on every cell — 39 steps for 195 candidates in 27 cells of
  // ~7 — where the flat sequence needs 25.
  int nR = 0, total = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int len = last4[q] - first4[q];
    // exclusive prefix of the lengths inside the group (three steps over eight lanes)
    int incl = len;
#pragma unroll
    for (int o = 1; o < kNearGroup; o <<= 1) {
      const int u = __shfl_up(incl, o, kNearGroup);
      if (sub >= o) incl += u;
    }
    const uint mine = (uint)(__ballot(len > 0) >> gbase) & 0xffu;
    if (len > 0) ranges[grp][nR + __popc(mine & ((1u << sub) - 1u))] = make_int2(first4[q], total + incl - len);
    nR += __popc(mine);
    total += __shfl(incl, kNearGroup - 1, kNearGroup);
  }
  if (sub == 0) ranges[grp][nR] = make_int2(0, total);  // (sentinel: the walk below compares with the NEXT range's offset)
  int c = 0;
  constexpr int kRows = 2;  // rows of eight candidates per step, their loads in flight together (57.0 us cell by cell, 50.5 flat, 48.9 with two rows, 48.3 with four)
  for (int t0 = 0; __any(t0 < total); t0 += kRows * kNearGroup) {
    if (__any(cnt > kNearCap - kRows * kNearGroup)) drain();  // (wave-uniform; lists that still have room are drained early, harmless)
    int j[kRows];
    float4 pj[kRows];
    bool in[kRows];
#pragma unroll
    for (int u = 0; u < kRows; ++u) {
      const int t = t0 + u * kNearGroup + sub;
      in[u] = t < total;
      j[u] = 0;
      pj[u] = pi;
      if (in[u]) {
        while (t >= ranges[grp][c + 1].y) ++c;
        const int2 r = ranges[grp][c];
        j[u] = r.x + (t - r.y);
        pj[u] = sortPos[j[u]];
      }
    }
#pragma unroll
    for (int u = 0; u < kRows; ++u) {
      const bool hit = in[u] && scan_distance2<SHEAR>(pi, pj[u], L, invL, shear) < rcut2s;
      const unsigned long long m = __ballot(hit);
      const uint mine = (uint)(m >> gbase) & 0xffu;
      if (hit) hitList[grp][cnt + __popc(mine & ((1u << sub) - 1u))] = j[u];
      cnt += __popc(mine);
    }
  }
  drain();
#pragma unroll
  for (int o = 1; o < kNearGroup; o <<= 1) {
    tx += __shfl_xor(tx, o, 64);
    ty += __shfl_xor(ty, o, 64);
    tz += __shfl_xor(tz, o, 64);
  }
  if (active && sub == 0) {
    float *o = Mv + 3 * (size_t)(INDIRECT ? groupIndex[id] : id);
    if (ACCUM) { o[0] += tx; o[1] += ty; o[2] += tz; }
    else { o[0] = tx; o[1] = ty; o[2] = tz; }
  }
}

// ---- pair records ---------------------------------------------------------------------------------------------------------------------
// k_pse_near8's scan (same mapping, same superset test), but a hit is EVALUATED ONCE INTO A RECORD instead of being summed: the table
// values and the distance vector of a pair do not depend on the vector the mobility is applied to.  A workgroup reserves the records of
// its 32 particles with one atomic (particle p's run is contiguous: first[p], count[p]); a pair that fails the exact cut-off after
// passing the scan's keeps its slot with F = C = 0 (adds +0).  A particle with more hits than the list holds (kNearCap) raises
// status[1]: the host then stays on k_pse_near8 for this handle.
// MODE 0: as described.  The other two keep a particle's CANDIDATES over several steps (PSENear::skin):
// MODE 1: the same scan on a grid whose cells are rc + skin wide; the candidates within rc + skin (a second, wider test on the distance the
//   scan has anyway) are written to candList[id * candCap ...] (their count to candCount[id]; more than candCap: status[3]);
// MODE 2: no cells — the candidates are the kept list, the positions the refreshed sortPos (k_pse_refresh_sorted): as long as nobody has moved
//   more than skin / 2 since the list was made every pair inside rc is on it.  Workgroup 0 reduces k_pse_refresh_sorted's per-workgroup
//   maxima of |displacement|^2 into status[2] (float bits): the host reads it with the counters and repeats the build from a fresh list
//   if the bound was broken (the same path as a build that outgrew its arrays).
struct PairCand {
  int *list;             // [N][cap]
  int *count;            // [N]
  int cap;
  float rcand2;          // (rc + skin)^2, MODE 1
  const float *dispParts;  // MODE 2
  int nDispParts;
};
template <bool SHEAR, int MODE>
__global__ void __launch_bounds__(kNearBlock) k_pse_pairs_build(const float4 *__restrict__ sortPos, const uint *__restrict__ cellStart,
                                                                 const int *__restrict__ cellEnd, uint validCell, int N, GridT<float> grid,
                                                                 real3f L, float shear, float rcut2, TableView tab, float4 *__restrict__ recA,
                                                                 float2 *__restrict__ recB, int2 *__restrict__ pairRange,
                                                                 int *__restrict__ status, long long cap, int nreg, PairCand cand) {
  __shared__ int hitList[kNearBlock / kNearGroup][kNearCap];
  __shared__ int2 ranges[kNearBlock / kNearGroup][29];
  __shared__ int groupCount[kNearBlock / kNearGroup], blockBase;
  __shared__ float dispMax[kNearBlock / 64];
  const int lane = threadIdx.x & 63, sub = threadIdx.x & (kNearGroup - 1), gbase = lane & ~(kNearGroup - 1);
  const int grp = threadIdx.x / kNearGroup;
  const int id = (int)xcd_contiguous_block(blockIdx.x, gridDim.x) * (kNearBlock / kNearGroup) + grp;
  const bool active = id < N;
  const float4 pi = sortPos[active ? id : 0];
  const real3f invL{1.0f / L.x, 1.0f / L.y, 1.0f / L.z};
  const float rcut2s = rcut2 * 1.00001f + 1e-30f;
  int nR = 0, total = 0;
  if (MODE == 2) {
    total = active ? cand.count[id] : 0;
    if (blockIdx.x == 0) {   // (uniform branch: the barrier below is reached by the whole workgroup)
      float m = 0.0f;
      for (int k = threadIdx.x; k < cand.nDispParts; k += kNearBlock) m = fmaxf(m, cand.dispParts[k]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      if (lane == 0) dispMax[threadIdx.x / 64] = m;
      __syncthreads();
      if (threadIdx.x == 0) {
        for (int w = 1; w < kNearBlock / 64; ++w) m = fmaxf(m, dispMax[w]);
        status[2] = __float_as_int(m);
      }
    }
  } else {
    const int3 n = grid.cellDim;
    const int npx = n.x > 1 ? 3 : 1, npy = n.y > 1 ? 3 : 1, npz = n.z > 1 ? 3 : 1;
    const int numberNeighbourCells = npx * npy * npz;
    const int3 celli = grid.getCell(real3f{pi.x, pi.y, pi.z});
    int first4[4], last4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cc = sub + kNearGroup * q;
      first4[q] = 0; last4[q] = 0;
      if (active && cc < numberNeighbourCells) {
        int3 cellj = celli;
        if (npx > 1) cellj.x += cc % 3 - 1;
        if (npy > 1) cellj.y += (cc / npx) % 3 - 1;
        if (npz > 1) cellj.z += cc / (npx * npy) - 1;
        cellj.x = grid.pbc_x(cellj.x);
        cellj.y = grid.pbc_y(cellj.y);
        cellj.z = grid.pbc_z(cellj.z);
        if (!(cellj.x < 0 || cellj.x >= n.x || cellj.y < 0 || cellj.y >= n.y || cellj.z < 0 || cellj.z >= n.z)) {
          const int icellj = grid.getCellIndex(cellj);
          const uint cs = cellStart[icellj];
          if (cs >= validCell) { first4[q] = (int)(cs - validCell); last4[q] = cellEnd[icellj]; }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int len = last4[q] - first4[q];
      int incl = len;
#pragma unroll
      for (int o = 1; o < kNearGroup; o <<= 1) {
        const int u = __shfl_up(incl, o, kNearGroup);
        if (sub >= o) incl += u;
      }
      const uint mine = (uint)(__ballot(len > 0) >> gbase) & 0xffu;
      if (len > 0) ranges[grp][nR + __popc(mine & ((1u << sub) - 1u))] = make_int2(first4[q], total + incl - len);
      nR += __popc(mine);
      total += __shfl(incl, kNearGroup - 1, kNearGroup);
    }
    if (sub == 0) ranges[grp][nR] = make_int2(0, total);
  }
  int c = 0, cnt = 0, ccnt = 0;
  bool over = false;
  // (measured: 2, 3, 4 and 6 rows of the cell scan 65.8-66.6 us — it is bound by its instruction count; 8 rows of the candidate scan and
  // 4 rounds of records at a time 46.5 against 44.1 us: neither is waiting for its dependent loads)
  constexpr int kRows = MODE == 2 ? UAMMD_PSE_CAND_ROWS : UAMMD_PSE_BUILD_ROWS;
  const int *myCand = cand.list + (size_t)(active ? id : 0) * cand.cap;
  for (int t0 = 0; __any(t0 < total); t0 += kRows * kNearGroup) {
    int j[kRows];
    float4 pj[kRows];
    bool in[kRows];
#pragma unroll
    for (int u = 0; u < kRows; ++u) {
      const int t = t0 + u * kNearGroup + sub;
      in[u] = t < total;
      j[u] = 0;
      pj[u] = pi;
      if (in[u]) {
        if (MODE == 2) j[u] = myCand[t];
        else {
          while (t >= ranges[grp][c + 1].y) ++c;
          const int2 r = ranges[grp][c];
          j[u] = r.x + (t - r.y);
        }
        pj[u] = sortPos[j[u]];
      }
    }
#pragma unroll
    for (int u = 0; u < kRows; ++u) {
      const float d2 = scan_distance2<SHEAR>(pi, pj[u], L, invL, shear);
      const bool hit = in[u] && d2 < rcut2s;
      const unsigned long long m = __ballot(hit);
      const uint mine = (uint)(m >> gbase) & 0xffu;
      const int at = cnt + __popc(mine & ((1u << sub) - 1u));
      if (hit && at < kNearCap) hitList[grp][at] = j[u];
      cnt += __popc(mine);
      if (MODE == 1) {
        const bool chit = in[u] && d2 < cand.rcand2;
        const uint cmine = (uint)(__ballot(chit) >> gbase) & 0xffu;
        const int cat = ccnt + __popc(cmine & ((1u << sub) - 1u));
        if (chit && cat < cand.cap) cand.list[(size_t)id * cand.cap + cat] = j[u];
        ccnt += __popc(cmine);
      }
    }
  }
  if (MODE == 1) {
    if (active && sub == 0) cand.count[id] = min(ccnt, cand.cap);
    if (__any(active && ccnt > cand.cap) && lane == 0) status[3] = 1;
  }
  if (cnt > kNearCap) { over = true; cnt = kNearCap; }
  if (!active) cnt = 0;
  if (sub == 0) groupCount[grp] = cnt;
  __syncthreads();
  // the 32 groups' offsets inside the workgroup's reservation: one scan by the first wave (thread 0 adding 32 LDS words one after the
  // other, then every group re-adding its predecessors', was ~3 us of a workgroup's life)
  static_assert(kNearBlock / kNearGroup == 32, "one lane of the first wave per group");
  if (threadIdx.x < 64) {
    const int c = lane < 32 ? groupCount[lane] : 0;
    const int incl = (int)wave_inclusive_scan((uint)c);
    if (lane < 32) groupCount[lane] = incl - c;   // exclusive: the group's offset
    if (lane == 31) blockBase = incl ? atomicAdd(&status[16 + 16 * (blockIdx.x % nreg)], incl) : 0;
  }
  if (__any(over) && lane == 0) status[1] = 1;
  __syncthreads();
  const long long regionCap = cap / nreg;
  const long long inRegion = (long long)blockBase + groupCount[grp];
  const long long off = (long long)(blockIdx.x % nreg) * regionCap + inRegion;
  const bool fits = inRegion + cnt <= regionCap;   // (past the region's capacity: the host grows the arrays and builds again)
  if (active && sub == 0) pairRange[id] = make_int2((int)off, fits ? cnt : 0);
  if (!fits) return;
  // (UAMMD_PSE_REC_ROWS rounds of eight records at a time: hit, position and the two table entries of each are four dependent reads)
  constexpr int kRec = UAMMD_PSE_REC_ROWS;
  for (int k0 = sub; __any(k0 < cnt); k0 += kRec * kNearGroup) {
    int jj[kRec];
    float4 pjj[kRec];
#pragma unroll
    for (int u = 0; u < kRec; ++u) {
      const int k = k0 + u * kNearGroup;
      jj[u] = k < cnt ? hitList[grp][k] : 0;
    }
#pragma unroll
    for (int u = 0; u < kRec; ++u) pjj[u] = sortPos[jj[u]];
    real3f rij[kRec];
    float r2[kRec];
    float2 fg[kRec];
#pragma unroll
    for (int u = 0; u < kRec; ++u) {
      rij[u] = scan_rij<SHEAR>(pi, pjj[u], L, invL, shear);
      r2[u] = dot3(rij[u], rij[u]);
      fg[u] = table_get(tab, sqrtf(fminf(r2[u], rcut2)));
    }
#pragma unroll
    for (int u = 0; u < kRec; ++u) {
      const int k = k0 + u * kNearGroup;
      float f = 0.0f, cc = 0.0f;
      if (r2[u] < rcut2) {
        f = fg[u].x;
        cc = r2[u] == 0.0f ? 0.0f : (fg[u].y - fg[u].x) * (1.0f / r2[u]);
      }
      if (k < cnt) {
        recA[off + k] = make_float4(f, cc, rij[u].x, rij[u].y);
        recB[off + k] = make_float2(rij[u].z, __int_as_float(jj[u]));
      }
    }
  }
}

// sortPos[s] = pos[index[s]] for a list that is kept while the particles move (PSENear::skin), and what decides whether it may be: the
// largest |displacement|^2 (minimum image) since the list was made, per workgroup (no atomics: k_pse_pairs_build<., 2> reduces them)
__global__ void __launch_bounds__(256) k_pse_refresh_sorted(const float4 *__restrict__ pos, const int *__restrict__ index,
                                                            const float4 *__restrict__ refPos, float4 *__restrict__ sortPos, int N, real3f L,
                                                            float *__restrict__ dispParts, int *__restrict__ status, int nStatus) {
  __shared__ float part[4];
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (status && blockIdx.x == 0)   // (the counters of the build that follows: no memset launch)
    for (int t = threadIdx.x; t < nStatus; t += 256) status[t] = 0;
  float d2 = 0.0f;
  if (s < N) {
    const float4 p = pos[index[s]], r = refPos[s];
    sortPos[s] = p;
    const real3f invL{1.0f / L.x, 1.0f / L.y, 1.0f / L.z};
    d2 = scan_distance2<false>(r, p, L, invL, 0.0f);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) d2 = fmaxf(d2, __shfl_xor(d2, o, 64));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x / 64] = d2;
  __syncthreads();
  if (threadIdx.x == 0) dispParts[blockIdx.x] = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
}

// (Measured and not kept in k_pse_pairs_build: skipping the neighbour cells whose nearest point is beyond the cut-off — half of the corner
// cells, a quarter of the edge cells, 26 % of the candidates — 63.6 -> 62.9 us: the build is not its scan.)
// Mv_i (+)= sum over i's records of F v_j + C (r . v_j) r  (RPYNearTransverser::compute, NearField.cuh:154-196, with (G - F) / r^2 folded
// into C when the record was made: one rounding per pair apart from k_pse_near8's terms).  Eight lanes per particle; a particle's records
// are contiguous: the group's loads are one 128-byte and one 64-byte segment per eight records, all of a particle's records and then all
// of its v_j in flight together (up to four rounds: 32 records; longer runs loop).
// (Measured and not kept: the Lanczos recurrence's k_l_c of the previous iteration and k_l_a inside this kernel — the input scaled by
// 1 / |w| on the way, every workgroup re-summing the |w|^2 partials, the row's update and its partial of w . v_i in the epilogue: two
// launches per iteration instead of four, product 14.2 -> 17.2 us, k_l_b 4.8 -> 7.3 us on 3125 partials, near noise 328 -> 320 us: 2.5 %
// for an interface between the solver and the matrix.)
template <int VSTRIDE, bool INDIRECT, bool ACCUM>
__global__ void __launch_bounds__(kNearBlock) k_pse_near_pairs(const float4 *__restrict__ recA, const float2 *__restrict__ recB,
                                                                const int2 *__restrict__ pairRange, const float *__restrict__ v,
                                                                const int *__restrict__ groupIndex, int N, float *__restrict__ Mv) {
  const int sub = threadIdx.x & (kNearGroup - 1);
  const int id = (int)xcd_contiguous_block(blockIdx.x, gridDim.x) * (kNearBlock / kNearGroup) + threadIdx.x / kNearGroup;
  const bool active = id < N;
  const int2 rg = active ? pairRange[id] : make_int2(0, 0);
  float tx = 0.f, ty = 0.f, tz = 0.f;
  constexpr int U = 4;
  for (int k0 = sub; k0 < rg.y; k0 += U * kNearGroup) {
    float4 a[U];
    float2 b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + u * kNearGroup;
      const bool in = k < rg.y;
      a[u] = in ? recA[(size_t)rg.x + k] : make_float4(0.f, 0.f, 0.f, 0.f);
      b[u] = in ? recB[(size_t)rg.x + k] : make_float2(0.f, __int_as_float(id));
    }
    float vx[U], vy[U], vz[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = __float_as_int(b[u].y);
      // (one 12-byte request: global loads only need their dwords aligned)
      struct __attribute__((packed, aligned(4))) V3 { float x, y, z; };
      const V3 vj = *(const V3 *)(v + (size_t)VSTRIDE * (INDIRECT ? groupIndex[j] : j));
      vx[u] = vj.x; vy[u] = vj.y; vz[u] = vj.z;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const real3f rij{a[u].z, a[u].w, b[u].x};
      const float gm = a[u].y * dot3(rij, real3f{vx[u], vy[u], vz[u]});
      tx += fmaf(gm, rij.x, a[u].x * vx[u]);
      ty += fmaf(gm, rij.y, a[u].x * vy[u]);
      tz += fmaf(gm, rij.z, a[u].x * vz[u]);
    }
  }
  // the group's eight partial sums by three DPP additions each (pairs, quads, the half row) instead of three ds_bpermute round trips
  static_assert(kNearGroup == 8, "eight lanes per particle");
  auto group_sum = [](float x) {
    x += dpp_move<0xB1, 0xf, true>(x);    // quad_perm [1, 0, 3, 2]
    x += dpp_move<0x4E, 0xf, true>(x);    // quad_perm [2, 3, 0, 1]
    x += dpp_move<0x141, 0xf, true>(x);   // row_half_mirror
    return x;
  };
  tx = group_sum(tx);
  ty = group_sum(ty);
  tz = group_sum(tz);
  if (active && sub == 0) {
    float *o = Mv + 3 * (size_t)(INDIRECT ? groupIndex[id] : id);
    if (ACCUM) { o[0] += tx; o[1] += ty; o[2] += tz; }
    else { o[0] = tx; o[1] = ty; o[2] = tz; }
  }
}

// The same product as one iteration of the Lanczos recurrence (lanczos_fused.hpp).  v_i = wPrev / hsup_(i-1) is not formed before the
// product: the pair sums run on wPrev as it is and the ROW's sum is scaled by 1 / hsup_(i-1) afterwards (M (w / h) = (M w) / h to
// rounding) — so nothing in the kernel waits for the norm: the |wPrev|^2 partials are requested first and summed, in lanczos.hip's
// order (k_l_c: the breakdown guard, e1 when the norm vanishes — then the sums are redone on e1), after the pair loop.  The rows' v_i
// and, one workgroup, hsup_(i-1) are written on the way; then w = M v_i - hsup_(i-1) v_(i-1) for the workgroup's rows and ONE partial of
// w . v_i per workgroup (k_l_a).  Vectors in cell order, stride 3.  (First form, measured: norm first, every v_j scaled as it is read —
// the product waited a round trip and two barriers before its first record, 13.4 -> ~24 us, the step 0.547 -> 0.562 ms.)
template <bool RIDER>
__global__ void __launch_bounds__(kNearBlock) k_pse_near_pairs_lanczos(const float4 *__restrict__ recA, const float2 *__restrict__ recB,
                                                                        const int2 *__restrict__ pairRange, int N, LanczosFusedArgs a,
                                                                        const float *__restrict__ riderFs, float *__restrict__ riderMFs) {
  __shared__ float sh[16];
  const int sub = threadIdx.x & (kNearGroup - 1);
  const int block = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int id = block * (kNearBlock / kNearGroup) + threadIdx.x / kNearGroup;
  const bool active = id < N;
  const bool scaled = a.wPrev != nullptr;
  const float *vsrc = scaled ? a.wPrev : a.viDirect;
  struct __attribute__((packed, aligned(4))) V3 { float x, y, z; };
  // everything the tail needs is requested up front: the norm's partials (npB <= 256: one per thread), the row's own vectors
  const float pb = (scaled && (int)threadIdx.x < a.npB) ? a.partsB[threadIdx.x] : 0.f;
  const float hsGiven = (!scaled && a.hsupPrev) ? *a.hsupPrev : 0.f;   // (v_i came from k_l_c: its hsup is in memory)
  const float hd = (scaled && !a.first) ? *a.hdiagPrev : 0.f, nz = (scaled && !a.first) ? *a.normz : 1.f;   // (first: no guard, |z| >= 0 = 0)
  const int2 rg = active ? pairRange[id] : make_int2(0, 0);
  const V3 zero3{0.f, 0.f, 0.f};
  const V3 own = active ? *(const V3 *)(vsrc + 3 * (size_t)id) : zero3;
  const V3 vp = (active && a.vPrev) ? *(const V3 *)(a.vPrev + 3 * (size_t)id) : zero3;
  float fx = 0.f, fy = 0.f, fz = 0.f;   // RIDER: the same records applied to a second vector (M_near F, NearField::Mdot)
  auto pair_sums = [&](bool e1, float &tx, float &ty, float &tz) __attribute__((always_inline)) {
    tx = ty = tz = 0.f;
    constexpr int U = 4;
    for (int k0 = sub; k0 < rg.y; k0 += U * kNearGroup) {
      float4 ra[U];
      float2 rb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = k0 + u * kNearGroup;
        const bool in = k < rg.y;
        ra[u] = in ? recA[(size_t)rg.x + k] : make_float4(0.f, 0.f, 0.f, 0.f);
        rb[u] = in ? recB[(size_t)rg.x + k] : make_float2(0.f, __int_as_float(id));
      }
      V3 vj[U], fj[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = __float_as_int(rb[u].y);
        if (e1) vj[u] = V3{(j == 0 && a.ownsFirstElement) ? 1.f : 0.f, 0.f, 0.f};
        else vj[u] = *(const V3 *)(vsrc + 3 * (size_t)j);
        if (RIDER && !e1) fj[u] = *(const V3 *)(riderFs + 3 * (size_t)j);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const real3f rij{ra[u].z, ra[u].w, rb[u].x};
        const float gm = ra[u].y * dot3(rij, real3f{vj[u].x, vj[u].y, vj[u].z});
        tx += fmaf(gm, rij.x, ra[u].x * vj[u].x);
        ty += fmaf(gm, rij.y, ra[u].x * vj[u].y);
        tz += fmaf(gm, rij.z, ra[u].x * vj[u].z);
        if (RIDER && !e1) {
          const float gf = ra[u].y * dot3(rij, real3f{fj[u].x, fj[u].y, fj[u].z});
          fx += fmaf(gf, rij.x, ra[u].x * fj[u].x);
          fy += fmaf(gf, rij.y, ra[u].x * fj[u].y);
          fz += fmaf(gf, rij.z, ra[u].x * fj[u].z);
        }
      }
    }
    static_assert(kNearGroup == 8, "eight lanes per particle");
    auto group_sum = [](float x) {
      x += dpp_move<0xB1, 0xf, true>(x);    // quad_perm [1, 0, 3, 2]
      x += dpp_move<0x4E, 0xf, true>(x);    // quad_perm [2, 3, 0, 1]
      x += dpp_move<0x141, 0xf, true>(x);   // row_half_mirror
      return x;
    };
    tx = group_sum(tx);
    ty = group_sum(ty);
    tz = group_sum(tz);
    if (RIDER && !e1) {
      fx = group_sum(fx);
      fy = group_sum(fy);
      fz = group_sum(fz);
    }
  };
  float tx, ty, tz;
  pair_sums(false, tx, ty, tz);
  if (RIDER && active && sub == 0) { float *o = riderMFs + 3 * (size_t)id; o[0] = fx; o[1] = fy; o[2] = fz; }
  // hsup_(i-1) (k_l_c): the partials in lanczos.hip's order — thread t holds partial t, the wave sums, (s0 + s2) + (s1 + s3)
  float hs = hsGiven, inv = 1.f;
  if (scaled) {
    const float ws = wave_sum_to_last(pb);
    if ((threadIdx.x & 63) == 63) sh[threadIdx.x >> 6] = ws;
    __syncthreads();
    hs = sqrtf((sh[0] + sh[2]) + (sh[1] + sh[3]));
    __syncthreads();
    if (!a.first && hs < 1e-3f * hd / nz) hs = 0.f;
    if (block == 0 && threadIdx.x == 0) *a.hsupPrev = hs;
    inv = hs > 0.f ? 1.0f / hs : 0.f;
  }
  V3 vi = own;
  if (scaled) {
    if (hs > 0.f) { vi.x *= inv; vi.y *= inv; vi.z *= inv; tx *= inv; ty *= inv; tz *= inv; }
    else {   // breakdown: v_i = e1 (k_l_c), the sums again on it (wave-uniform, rare)
      vi = V3{(id == 0 && a.ownsFirstElement) ? 1.f : 0.f, 0.f, 0.f};
      pair_sums(true, tx, ty, tz);
    }
  }
  // the row's v_i (written out when this kernel made it), w = M v_i - hsup_(i-1) v_(i-1), the partial of w . v_i (k_l_a)
  float part = 0.f;
  if (active && sub == 0) {
    if (scaled && a.viOut) { float *o = a.viOut + 3 * (size_t)id; o[0] = vi.x; o[1] = vi.y; o[2] = vi.z; }
    float wx = tx, wy = ty, wz = tz;
    if (a.vPrev) { wx = fmaf(-hs, vp.x, wx); wy = fmaf(-hs, vp.y, wy); wz = fmaf(-hs, vp.z, wz); }
    float *o = a.wOut + 3 * (size_t)id;
    o[0] = wx; o[1] = wy; o[2] = wz;
    part = fmaf(wz, vi.z, fmaf(wy, vi.y, wx * vi.x));
  }
  part = wave_sum_to_last(part);
  if ((threadIdx.x & 63) == 63) sh[4 + (threadIdx.x >> 6)] = part;
  __syncthreads();
  if (threadIdx.x == 0) a.partsA[block] = (sh[4] + sh[6]) + (sh[5] + sh[7]);
}

// ---- AUTO where the table has its packed copy: one WAVE per cell, candidates staged in LDS ------------------------------------------------
// k_pse_near8 still issues ~150 global load instructions per wave (a candidate load per scan step, position + v + two table reads per
// hit) and every 64-lane load costs the CU's one address unit ~16 clocks whatever it fetches: 59 us at N = 1e5.  All particles of a cell
// see the SAME 27 cells, so here a wave owns a cell: it fetches the 27 ranges (lanes 0..26), builds their prefix sum with shuffles,
// stages the ~200 candidates' positions and v ONCE (four 16-byte requests per lane: flat candidate index -> cell by binary search in
// the LDS prefix table) and then works from LDS — groups of 8 lanes take one owner each (passes of 8 owners), scan with the cheap
// superset test, compact the hits per group (ballot), and evaluate them exactly as the reference does; the only global reads of the
// drain are the table's, one 16-byte request per hit from a copy that holds {F_i, G_i, F_i+1, G_i+1} per entry.  Chunks of kCellCap
// candidates for crowded neighbourhoods.  Same pairs, same per-pair arithmetic, another summation order.
constexpr int kCellCap = 256;   // candidates staged per chunk (a hit is stored as one byte)
struct CellWaveLds {
  float4 pos[kCellCap];
  float vel[3 * kCellCap];
  unsigned char hits[kNearGroup][kCellCap];
  int pre[28], first[28];
};

template <int VSTRIDE, bool INDIRECT, bool ACCUM, bool SHEAR>
__global__ void __launch_bounds__(256) k_pse_near_cell(const float4 *__restrict__ sortPos, const float *__restrict__ v,
                                                        const int *__restrict__ groupIndex, const uint *__restrict__ cellStart,
                                                        const int *__restrict__ cellEnd, uint validCell, int ncells, GridT<float> grid,
                                                        real3f L, float shear, float rcut2, TableView tab, const float4 *__restrict__ table4,
                                                        float *__restrict__ Mv, int ablate) {
  __shared__ CellWaveLds lds[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & (kNearGroup - 1), grp = lane / kNearGroup, gbase = lane & ~(kNearGroup - 1);
  CellWaveLds &sh = lds[wave];
  const int c = (int)xcd_contiguous_block(blockIdx.x, gridDim.x) * 4 + wave;
  if (c >= ncells) return;  // (waves of a workgroup are independent: no workgroup barrier anywhere)
  const uint cs0 = cellStart[c];
  if (cs0 < validCell) return;
  const int o0 = (int)(cs0 - validCell), o1 = cellEnd[c];
  const int3 n = grid.cellDim;
  const int npx = n.x > 1 ? 3 : 1, npy = n.y > 1 ? 3 : 1, npz = n.z > 1 ? 3 : 1;
  const int numberNeighbourCells = npx * npy * npz;
  const int3 celli = make_int3(c % n.x, (c / n.x) % n.y, c / (n.x * n.y));
  // the 27 ranges in the reference's visiting order (x fastest) and their prefix sum
  int first = 0, count = 0;
  if (lane < numberNeighbourCells) {
    int3 cellj = celli;
    if (npx > 1) cellj.x += lane % 3 - 1;
    if (npy > 1) cellj.y += (lane / npx) % 3 - 1;
    if (npz > 1) cellj.z += lane / (npx * npy) - 1;
    cellj.x = grid.pbc_x(cellj.x);
    cellj.y = grid.pbc_y(cellj.y);
    cellj.z = grid.pbc_z(cellj.z);
    if (!(cellj.x < 0 || cellj.x >= n.x || cellj.y < 0 || cellj.y >= n.y || cellj.z < 0 || cellj.z >= n.z)) {
      const int icellj = grid.getCellIndex(cellj);
      const uint cs = cellStart[icellj];
      if (cs >= validCell) { first = (int)(cs - validCell); count = cellEnd[icellj] - first; }
    }
  }
  int incl = count;
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane < 28) { sh.pre[lane] = incl - count; sh.first[lane] = first; }  // (lanes >= the cell count hold the total)
  const int total = __shfl(incl, 27, 64);
  const real3f invL{1.0f / L.x, 1.0f / L.y, 1.0f / L.z};
  const float rcut2s = rcut2 * 1.00001f + 1e-30f;
  for (int base = 0; base < total; base += kCellCap) {
    const int chunk = min(kCellCap, total - base);
    // ---- stage: flat candidate index -> (cell, particle) ----
#pragma unroll
    for (int k = 0; k < kCellCap / 64; ++k) {
      const int t = base + lane + 64 * k;
      if (t < base + chunk) {
        int lo = 0;  // largest cell index with pre[cell] <= t (pre is non decreasing, 28 entries)
#pragma unroll
        for (int step = 16; step > 0; step >>= 1)
          if (lo + step < 28 && sh.pre[lo + step] <= t) lo += step;
        const int j = sh.first[lo] + (t - sh.pre[lo]);
        float4 pj = sortPos[j];
        pj.w = __int_as_float(j);
        const float *vp = v + (size_t)VSTRIDE * (INDIRECT ? groupIndex[j] : j);
        sh.pos[t - base] = pj;
        sh.vel[3 * (t - base)] = vp[0]; sh.vel[3 * (t - base) + 1] = vp[1]; sh.vel[3 * (t - base) + 2] = vp[2];
      }
    }
    // ---- passes of eight owners ----
    if (ablate & 1) continue;
    for (int ob = o0; ob < o1; ob += kNearGroup) {
      const int id = ob + grp;
      const bool active = id < o1;
      const float4 pi = sortPos[active ? id : o0];
      int cnt = 0;
      for (int t0 = 0; t0 < chunk; t0 += kNearGroup) {
        const int t = t0 + sub;
        const float4 pj = sh.pos[t < chunk ? t : 0];
        const bool hit = active && t < chunk && scan_distance2<SHEAR>(pi, pj, L, invL, shear) < rcut2s;
        const unsigned long long m = __ballot(hit);
        const uint mine = (uint)(m >> gbase) & 0xffu;
        if (hit) sh.hits[grp][cnt + __popc(mine & ((1u << sub) - 1u))] = (unsigned char)t;
        cnt += __popc(mine);
      }
      float tx = 0.f, ty = 0.f, tz = 0.f;
      if (ablate & 2) cnt = 0;
      for (int k = sub; k < cnt; k += kNearGroup) {
        const int t = sh.hits[grp][k];
        const float4 pj = sh.pos[t];
        const real3f vj{sh.vel[3 * t], sh.vel[3 * t + 1], sh.vel[3 * t + 2]};
        const real3f rij = sheared_distance(pi, pj, L, shear);
        const float r2 = dot3(rij, rij);
        if (r2 >= rcut2) continue;  // (the scan's test is a superset)
        // TabulatedFunction::operator() as table_get above, both samples in one request
        const float rs = sqrtf(r2);
        float f = 0.f, g = 0.f;
        if (!(rs >= tab.rmax)) {
          const float r = rs * tab.interval;
          const int i = r <= 0.0f ? 0 : (int)(r * (float)tab.Ntable);
          const float4 q = table4[i];
          if (r <= 0.0f) { f = q.x; g = q.y; }
          else {
            const float w = (r - (float)i * tab.dr) * (float)tab.Ntable;
            f = fmaf(w, q.z, fmaf(-w, q.x, q.x));
            g = fmaf(w, q.w, fmaf(-w, q.y, q.y));
          }
        }
        float rx, ry, rz;
        if (r2 == 0.0f) {
          rx = f * vj.x; ry = f * vj.y; rz = f * vj.z;
        } else {
          const float invr2 = 1.0f / r2;
          const float gmfv = (g - f) * dot3(rij, real3f{vj.x, vj.y, vj.z}) * invr2;
          rx = fmaf(gmfv, rij.x, f * vj.x);
          ry = fmaf(gmfv, rij.y, f * vj.y);
          rz = fmaf(gmfv, rij.z, f * vj.z);
        }
        tx += rx; ty += ry; tz += rz;
      }
#pragma unroll
      for (int o = 1; o < kNearGroup; o <<= 1) {
        tx += __shfl_xor(tx, o, 64);
        ty += __shfl_xor(ty, o, 64);
        tz += __shfl_xor(tz, o, 64);
      }
      if (active && sub == 0) {
        float *o = Mv + 3 * (size_t)(INDIRECT ? groupIndex[id] : id);
        if (ACCUM || base > 0) { o[0] += tx; o[1] += ty; o[2] += tz; }  // (later chunks add to what the first one stored)
        else { o[0] = tx; o[1] = ty; o[2] = tz; }
      }
    }
  }
}

// the Lanczos iteration runs in CELL order (dot products and norms do not care; the product then reads v_j next to pos_j and needs
// neither a gather of v nor a memset of Mv): noise of particle index[k] at slot k, and the result scattered back at the end
__global__ void __launch_bounds__(256) k_pse_noise_sorted(float *__restrict__ out3, const int *__restrict__ groupIndex, int N,
                                                           float variance, uint seed1, uint seed2) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= N) return;
  Saru rng((uint)groupIndex[k], seed1, seed2);
  const float2 a = rng.gf(0.0f, 1.0f);
  const float2 b = rng.gf(0.0f, 1.0f);
  out3[3 * (size_t)k] = a.x * variance;
  out3[3 * (size_t)k + 1] = a.y * variance;
  out3[3 * (size_t)k + 2] = b.x * variance;
}
// the same vector with the partials of its |.|^2 (at most 256 workgroups: the Lanczos run takes them instead of a k_l_norm2 launch,
// lanczos_set_znorm_parts) and, workgroup 0, the pair build's counters copied to the host-mapped block the host reads after the solve
// (instead of a copy command between the build and this kernel)
__global__ void __launch_bounds__(256) k_pse_noise_sorted_norm(float *__restrict__ out3, const int *__restrict__ groupIndex, int N,
                                                                float variance, uint seed1, uint seed2, float *__restrict__ parts,
                                                                const int *__restrict__ status, int *__restrict__ statusHost, int nStatus,
                                                                const float4 *__restrict__ force4, float *__restrict__ forceSorted) {
  __shared__ float sh[4];
  float acc = 0.f;
  for (int k = blockIdx.x * 256 + threadIdx.x; k < N; k += gridDim.x * 256) {
    const int idx = groupIndex[k];
    if (force4) {   // (uammd_pse_near_set_mdot_rider: F in the list's order for the first product)
      const float4 f = force4[idx];
      forceSorted[3 * (size_t)k] = f.x; forceSorted[3 * (size_t)k + 1] = f.y; forceSorted[3 * (size_t)k + 2] = f.z;
    }
    Saru rng((uint)idx, seed1, seed2);
    const float2 a = rng.gf(0.0f, 1.0f);
    const float2 b = rng.gf(0.0f, 1.0f);
    const float x = a.x * variance, y = a.y * variance, z = b.x * variance;
    out3[3 * (size_t)k] = x;
    out3[3 * (size_t)k + 1] = y;
    out3[3 * (size_t)k + 2] = z;
    acc = fmaf(z, z, fmaf(y, y, fmaf(x, x, acc)));
  }
  acc = wave_sum_to_last(acc);
  if ((threadIdx.x & 63) == 63) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) parts[blockIdx.x] = (sh[0] + sh[2]) + (sh[1] + sh[3]);
  if (status && blockIdx.x == 0) {   // (<= 5 words per thread: all read before the first is written across the bus)
    int v[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) v[u] = (int)threadIdx.x + 256 * u < nStatus ? status[threadIdx.x + 256 * u] : 0;
#pragma unroll
    for (int u = 0; u < 5; ++u)
      if ((int)threadIdx.x + 256 * u < nStatus) statusHost[threadIdx.x + 256 * u] = v[u];
    for (int t = threadIdx.x + 256 * 5; t < nStatus; t += 256) statusHost[t] = status[t];
  }
}
// (add3 / addTo: the rider's M_near F, in the list's order, added to the caller's MF on the way)
__global__ void __launch_bounds__(256) k_pse_unsort3(const float *__restrict__ in3, const int *__restrict__ groupIndex, int N,
                                                      float *__restrict__ out3, const float *__restrict__ add3 = nullptr,
                                                      float *__restrict__ addTo = nullptr) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= N) return;
  const size_t idx = (size_t)groupIndex[k];
  float *o = out3 + 3 * idx;
  o[0] = in3[3 * (size_t)k]; o[1] = in3[3 * (size_t)k + 1]; o[2] = in3[3 * (size_t)k + 2];
  if (add3) {
    float *m = addTo + 3 * idx;
    m[0] += add3[3 * (size_t)k]; m[1] += add3[3 * (size_t)k + 1]; m[2] += add3[3 * (size_t)k + 2];
  }
}

// SaruTransform (NearField.cuh:218-228): make_real3(gf(0,1), gf(0,1).x) * variance
__global__ void __launch_bounds__(256) k_pse_noise(float *__restrict__ out3, int N, float variance, uint seed1, uint seed2) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  Saru rng((uint)i, seed1, seed2);
  const float2 a = rng.gf(0.0f, 1.0f);
  const float2 b = rng.gf(0.0f, 1.0f);
  out3[3 * (size_t)i] = a.x * variance;
  out3[3 * (size_t)i + 1] = a.y * variance;
  out3[3 * (size_t)i + 2] = b.x * variance;
}

static TableView make_view(const PSENear *p) {
  TableView t;
  t.table = (const float2 *)p->table.ptr;
  t.Ntable = p->nPointsTable - 1;
  t.rmax = p->rcut;
  t.interval = (float)(1.0 / (p->rcut - 0.0f));
  t.dr = (float)(1.0 / (float)t.Ntable);
  return t;
}

// cl->update(box, rcut * safetyFactor) (NearField.cuh:231-237)
static bool pse_cand_usable(const PSENear *p) {
  return p->candEnabled && p->skinPercent > 0 && p->pairList && p->lazyList && p->nearKernel == 1 && !p->exactOrder && !p->pairsUnfit &&
         p->shear == 0.0f;
}
static int pse_update_list(PSENear *p, const float *d_pos, int N, hipStream_t st, bool full = false) {
  if (!full && p->lazyList && p->listValid && p->listPos == d_pos && p->N == N && p->listStream == st) return 0;
  const bool cand = pse_cand_usable(p);
  const bool same = p->listPos == d_pos && p->N == N && p->listStream == st;
  // the positions moved, the order and the candidate lists may stay: refresh instead of a build when the bound is expected to hold
  if (!full && cand && p->candValid && same && !p->pairsPending && p->lastDisp + 2.0f * p->maxInc <= 0.5f * p->skin) {
    const int nb = (N + 255) / 256;
    if (int e = p->dispParts.reserve(sizeof(float) * (size_t)nb)) return e;
    const real3f Lb{p->boxL[0], p->boxL[1], p->boxL[2]};
    if (int e = p->pairCursor.reserve(kPairStatusInts * sizeof(int))) return e;
    hipLaunchKernelGGL(k_pse_refresh_sorted, dim3(nb), dim3(256), 0, st, (const float4 *)d_pos, (const int *)p->cl.index.ptr,
                       (const float4 *)p->refPos.ptr, (float4 *)p->cl.sortPos.ptr, N, Lb, (float *)p->dispParts.ptr,
                       (int *)p->pairCursor.ptr, kPairStatusInts);
    UH_CHECK(hipGetLastError());
    p->statusZeroed = true;
    p->listValid = true;
    p->pairsValid = false;
    p->candScan = true;
    p->candBuild = false;
    ++p->stepsSinceFull;
    return 0;
  }
  if (cand && p->candValid && same) {   // a kept list ends here: how long did it live?
    p->candShortLived = p->stepsSinceFull < 2 ? p->candShortLived + 1 : 0;
    if (p->candShortLived >= 3) p->candEnabled = false;
  }
  p->listValid = false;  // (valid again only when the build below went through: a failed build must not be reused)
  p->pairsValid = false;
  p->pairsPending = false;  // (a build in flight is of the old list; the next one is ordered after it on the stream)
  p->candValid = false;
  p->candScan = false;
  p->statusZeroed = false;
  bool keep = pse_cand_usable(p);
  p->stepsSinceFull = 0;
  p->lastDisp = 0.0f;
  const float g = p->shear;
  const float safety = (float)(1 + 0.5 * g * g + 0.5 * std::sqrt(g * g * (g * g + 4.0)));  // NearField.cuh:24-27
  const int per[3] = {1, 1, 1};
  int cd[3], gper[3];
  float gL[3];
  for (;;) {
    p->skin = keep ? 0.01f * (float)p->skinPercent * p->rcut : 0.0f;
    const float rc = (p->rcut + p->skin) * safety;
    const float rc3[3] = {rc, rc, rc};
    if (int e = uammd_celllist_create_grid(p->boxL, per, rc3, cd, gL, gper)) return e;
    // the scan visits the 27 cells around a particle's: with fewer than three cells along a periodic direction it would meet a cell twice.
    // A box that holds three cells of the cut-off but not three of cut-off + skin keeps the reference's grid (no kept lists).
    if (!keep || (cd[0] >= 3 && cd[1] >= 3 && cd[2] >= 3)) break;
    keep = false;
    p->candEnabled = false;
  }
  p->candBuild = keep;
  p->N = N;
  if (int e = p->cl.update((const float4 *)d_pos, N, gL, gper, cd, st)) return e;
  if (keep) {   // the positions the kept list's displacement bound is measured from
    if (int e = p->refPos.reserve(sizeof(float4) * (size_t)N)) return e;
    UH_CHECK(hipMemcpyAsync(p->refPos.ptr, p->cl.sortPos.ptr, sizeof(float4) * (size_t)N, hipMemcpyDeviceToDevice, st));
  }
  ++p->fullBuilds;
  p->listValid = true;
  p->listPos = d_pos;
  p->listStream = st;  // (the list is ordered after the build on THIS stream only)
  return 0;
}

static TableView make_view(const PSENear *p);
// the pair records of the current list (see PSENear::pairList).  One host read per build: the number of records, to grow the arrays
// when they do not fit (the build then runs again).
static int pse_launch_pairs(PSENear *p, hipStream_t st, bool deferCopy = false) {
  const int N = p->N;
  if (!p->pairTotalHost) {
    UH_CHECK(hipHostMalloc((void **)&p->pairTotalHost, kPairStatusInts * sizeof(int), hipHostMallocMapped));
    UH_CHECK(hipHostGetDevicePointer((void **)&p->pairTotalDev, (void *)p->pairTotalHost, 0));
  }
  if (!p->pairsEvent) UH_CHECK(hipEventCreateWithFlags(&p->pairsEvent, hipEventDisableTiming));
  if (int e = p->pairRange.reserve(sizeof(int2) * (size_t)N)) return e;
  if (int e = p->pairCursor.reserve(kPairStatusInts * sizeof(int))) return e;
  if (p->pairCap < (size_t)N) p->pairCap = p->pairCapFirst ? std::max(p->pairCapFirst, (size_t)N) : (size_t)48 * (size_t)N;
  if (p->pairCap > (size_t)0x7fffff00) { p->pairsUnfit = true; return 0; }   // (record indices are ints)
  if (int e = p->recA.reserve(sizeof(float4) * p->pairCap)) return e;
  if (int e = p->recB.reserve(sizeof(float2) * p->pairCap)) return e;
  const real3f Lb{p->boxL[0], p->boxL[1], p->boxL[2]};
  const dim3 gr((N + kNearBlock / kNearGroup - 1) / (kNearBlock / kNearGroup));
  p->pairRegions = (int)std::min<long long>(kPairRegions, std::max<long long>(1, (long long)gr.x / 8));   // (a small system: fewer, larger regions)
  if (!p->statusZeroed) UH_CHECK(hipMemsetAsync(p->pairCursor.ptr, 0, kPairStatusInts * sizeof(int), st));
  p->statusZeroed = false;
  PairCand cand{nullptr, nullptr, 0, 0.0f, nullptr, 0};
  if (p->candBuild) {   // candidates per particle: the mean number within rc + skin with room for the density's fluctuations
    if (p->candCap == 0) {
      const double rcs = (double)p->rcut + p->skin, vol = (double)p->boxL[0] * p->boxL[1] * p->boxL[2];
      const double mean = (double)N / vol * 4.18879 * rcs * rcs * rcs;
      p->candCap = std::max(32, ((int)(1.6 * mean + 24.0) + 7) & ~7);
    }
    if ((size_t)N * (size_t)p->candCap > (size_t)1 << 31) { p->candBuild = false; p->candEnabled = false; }
  }
  if (p->candBuild || p->candScan) {
    if (int e = p->candList.reserve(sizeof(int) * (size_t)N * (size_t)p->candCap)) return e;
    if (int e = p->candCount.reserve(sizeof(int) * (size_t)N)) return e;
    const float rcs = p->rcut + p->skin;
    cand = PairCand{(int *)p->candList.ptr, (int *)p->candCount.ptr, p->candCap, rcs * rcs, (const float *)p->dispParts.ptr, (N + 255) / 256};
  }
#define UH_PAIRS(SH, MODE)                                                                                                          \
  hipLaunchKernelGGL((k_pse_pairs_build<SH, MODE>), gr, dim3(kNearBlock), 0, st, (const float4 *)p->cl.sortPos.ptr,                 \
                     (const uint *)p->cl.cellStart.ptr, (const int *)p->cl.cellEnd.ptr, p->cl.validCell, N, p->cl.grid, Lb,         \
                     p->shear, p->rcut * p->rcut, make_view(p), (float4 *)p->recA.ptr, (float2 *)p->recB.ptr,                       \
                     (int2 *)p->pairRange.ptr, (int *)p->pairCursor.ptr, (long long)p->pairCap, p->pairRegions, cand)
  if (p->candScan) { UH_PAIRS(false, 2); ++p->candBuilds; }
  else if (p->candBuild) UH_PAIRS(false, 1);
  else if (p->shear != 0.0f) UH_PAIRS(true, 0);
  else UH_PAIRS(false, 0);
#undef UH_PAIRS
  p->statusCopyDeferred = deferCopy;   // (the caller's next kernel on the stream copies the block: pse_settle_status otherwise)
  if (!deferCopy) {
    UH_CHECK(hipMemcpyAsync(p->pairTotalHost, p->pairCursor.ptr, kPairStatusInts * sizeof(int), hipMemcpyDeviceToHost, st));
    UH_CHECK(hipEventRecord(p->pairsEvent, st));
  }
  p->pairsPending = true;
  return 0;
}
// a deferred copy nobody has taken: issue it now (before anything waits for pairsEvent)
static int pse_settle_status(PSENear *p, hipStream_t st) {
  if (!p->statusCopyDeferred) return 0;
  p->statusCopyDeferred = false;
  UH_CHECK(hipMemcpyAsync(p->pairTotalHost, p->pairCursor.ptr, kPairStatusInts * sizeof(int), hipMemcpyDeviceToHost, st));
  UH_CHECK(hipEventRecord(p->pairsEvent, st));
  return 0;
}
// what the build's counters say once its event has completed: the records asked for, and whether every region held its share; if not,
// the capacity that would have (the fullest region's demand for all of them, + 25 %)
static long long pse_pairs_total(const PSENear *p) {
  long long total = 0;
  for (int r = 0; r < p->pairRegions; ++r) total += p->pairTotalHost[16 + 16 * r];
  return total;
}
static bool pse_pairs_fit(const PSENear *p) {
  const long long regionCap = (long long)(p->pairCap / p->pairRegions);
  for (int r = 0; r < p->pairRegions; ++r)
    if (p->pairTotalHost[16 + 16 * r] > regionCap) return false;
  return true;
}
static void pse_pairs_grow(PSENear *p) {
  long long worst = 0;
  for (int r = 0; r < p->pairRegions; ++r) worst = std::max(worst, (long long)p->pairTotalHost[16 + 16 * r]);
  const long long need = std::max(pse_pairs_total(p), worst * p->pairRegions);
  p->pairCap = (size_t)(need + need / 4 + p->pairRegions);
}
// The verdict on a build whose event has completed.  true: the records are what the products may stream (pairsValid).  false: something
// was changed so that the next launch can succeed — arrays grown, the list rebuilt from scratch on the stream because the kept one's
// displacement bound was broken (or because the products go back to scanning the cells: pairsUnfit) — and the caller launches again
// (unless pairsUnfit).
static int pse_pairs_verdict(PSENear *p, hipStream_t st, bool *ok) {
  *ok = false;
  const bool scanned = p->candScan, built = p->candBuild;
  if (scanned) {   // how far the particles are from where the list was made
    float d2 = 0.0f;
    std::memcpy(&d2, &p->pairTotalHost[2], sizeof(float));   // (float bits in the int block)
    const float d = std::sqrt(std::max(0.0f, d2));
    if (d > p->lastDisp) p->maxInc = std::max(p->maxInc, d - p->lastDisp);
    p->lastDisp = d;
    if (!(d <= 0.5f * p->skin)) {   // (also a NaN reading)
      ++p->candRepeats;
      return pse_update_list(p, (const float *)p->listPos, p->N, st, true);
    }
  }
  if (p->pairTotalHost[1]) {   // a particle with more neighbours than a hit list holds: k_pse_near8 from here on — on a list of this step
    p->pairsUnfit = true;
    if (scanned) return pse_update_list(p, (const float *)p->listPos, p->N, st, true);
    return 0;
  }
  if (built) {
    if (p->pairTotalHost[3]) {   // a particle with more candidates than its row holds: wider rows next time (twice), then no kept lists
      p->candCap = (p->candCap * 3 / 2 + 7) & ~7;
      if (++p->candGrow > 2) p->candEnabled = false;
    } else
      p->candValid = true;
    p->maxInc *= 0.7f;   // (an old extreme does not shorten the lists' lives for ever)
  }
  if (pse_pairs_fit(p)) { p->pairsValid = true; *ok = true; return 0; }
  pse_pairs_grow(p);
  if (scanned) p->candScan = true;   // (the same candidates again, into larger arrays)
  return 0;
}
static int pse_build_pairs(PSENear *p, hipStream_t st) {
  for (int attempt = 0; attempt < 5; ++attempt) {
    if (!p->pairsPending) {
      if (int e = pse_launch_pairs(p, st)) return e;
      if (p->pairsUnfit) return 0;
    }
    if (int e = pse_settle_status(p, st)) return e;
    UH_CHECK(hipEventSynchronize(p->pairsEvent));
    p->pairsPending = false;
    bool ok = false;
    if (int e = pse_pairs_verdict(p, st, &ok)) return e;
    if (ok || p->pairsUnfit) return 0;
  }
  p->pairsUnfit = true;
  return 0;
}

static int g_ablate = getenv("UAMMD_PSE_ABLATE") ? atoi(getenv("UAMMD_PSE_ABLATE")) : 0;
#define UH_NEAR8(VS, IND, ACC)                                                                                                       \
  do {                                                                                                                               \
    if (p->pairList && p->lazyList && p->listValid && !p->pairsUnfit && p->nearKernel == 1) {                                        \
      const bool ahead_ = p->optimistic && p->pairsPending && !p->pairsValid;                                                        \
      if (!p->pairsValid && !ahead_) { if (int e_ = pse_build_pairs(p, st)) return e_; }                                             \
      if (p->pairsValid || ahead_) {                                                                                                 \
        const dim3 gp((N + kNearBlock / kNearGroup - 1) / (kNearBlock / kNearGroup));                                                \
        hipLaunchKernelGGL((k_pse_near_pairs<VS, IND, ACC>), gp, dim3(kNearBlock), 0, st, (const float4 *)p->recA.ptr,               \
                           (const float2 *)p->recB.ptr, (const int2 *)p->pairRange.ptr, d_v, (const int *)p->cl.index.ptr, N, d_Mv); \
        break;                                                                                                                       \
      }                                                                                                                              \
    }                                                                                                                                \
    if (p->nearKernel == 0) {                                                                                                        \
      const int ncells = p->cl.grid.cellDim.x * p->cl.grid.cellDim.y * p->cl.grid.cellDim.z;                                         \
      const dim3 gc((ncells + 3) / 4);                                                                                               \
      if (p->shear != 0.0f)                                                                                                          \
        hipLaunchKernelGGL((k_pse_near_cell<VS, IND, ACC, true>), gc, dim3(256), 0, st, (const float4 *)p->cl.sortPos.ptr, d_v,      \
                           (const int *)p->cl.index.ptr, (const uint *)p->cl.cellStart.ptr, (const int *)p->cl.cellEnd.ptr,          \
                           p->cl.validCell, ncells, p->cl.grid, Lb, p->shear, p->rcut * p->rcut, make_view(p),                       \
                           (const float4 *)p->table4.ptr, d_Mv, g_ablate);                                                                     \
      else                                                                                                                           \
        hipLaunchKernelGGL((k_pse_near_cell<VS, IND, ACC, false>), gc, dim3(256), 0, st, (const float4 *)p->cl.sortPos.ptr, d_v,     \
                           (const int *)p->cl.index.ptr, (const uint *)p->cl.cellStart.ptr, (const int *)p->cl.cellEnd.ptr,          \
                           p->cl.validCell, ncells, p->cl.grid, Lb, p->shear, p->rcut * p->rcut, make_view(p),                       \
                           (const float4 *)p->table4.ptr, d_Mv, g_ablate);                                                                     \
      break;                                                                                                                         \
    }                                                                                                                                \
    const dim3 gr((N + kNearBlock / kNearGroup - 1) / (kNearBlock / kNearGroup));                                                    \
    if (p->shear != 0.0f)                                                                                                            \
      hipLaunchKernelGGL((k_pse_near8<VS, IND, ACC, true>), gr, dim3(kNearBlock), 0, st, (const float4 *)p->cl.sortPos.ptr, d_v,     \
                         (const int *)p->cl.index.ptr, (const uint *)p->cl.cellStart.ptr, (const int *)p->cl.cellEnd.ptr,            \
                         p->cl.validCell, N, p->cl.grid, Lb, p->shear, p->rcut * p->rcut, make_view(p), d_Mv);                       \
    else                                                                                                                             \
      hipLaunchKernelGGL((k_pse_near8<VS, IND, ACC, false>), gr, dim3(kNearBlock), 0, st, (const float4 *)p->cl.sortPos.ptr, d_v,    \
                         (const int *)p->cl.index.ptr, (const uint *)p->cl.cellStart.ptr, (const int *)p->cl.cellEnd.ptr,            \
                         p->cl.validCell, N, p->cl.grid, Lb, p->shear, p->rcut * p->rcut, make_view(p), d_Mv);                       \
  } while (0)

// d_Mv (+)= M_near v with v and Mv in the caller's particle order (stride VSTRIDE / 3)
template <int VSTRIDE, bool ACCUM>
static int pse_dot(PSENear *p, const float *d_v, float *d_Mv, hipStream_t st) {
  const int N = p->N;
  const real3f Lb{p->boxL[0], p->boxL[1], p->boxL[2]};
  if (!p->exactOrder) {
    UH_NEAR8(VSTRIDE, true, ACCUM);
    UH_CHECK(hipGetLastError());
    return 0;
  }
  if (!ACCUM) UH_CHECK(hipMemsetAsync(d_Mv, 0, sizeof(float) * 3 * (size_t)N, st));
  if (int e = p->sortV.reserve(sizeof(float4) * (size_t)N)) return e;
  hipLaunchKernelGGL((k_pse_gather_v<VSTRIDE>), dim3((N + 255) / 256), dim3(256), 0, st, d_v, (const int *)p->cl.index.ptr,
                     (float4 *)p->sortV.ptr, N);
  hipLaunchKernelGGL(k_pse_near, dim3((N + 127) / 128), dim3(128), 0, st, (const float4 *)p->cl.sortPos.ptr,
                     (const float4 *)p->sortV.ptr, (const int *)p->cl.index.ptr, (const uint *)p->cl.cellStart.ptr,
                     (const int *)p->cl.cellEnd.ptr, p->cl.validCell, N, p->cl.grid, Lb, p->shear, p->rcut * p->rcut, make_view(p), d_Mv);
  UH_CHECK(hipGetLastError());
  return 0;
}

// pse_ns::Dotctor (NearField.cuh:201-216): Mv = 0; cl->transverseList(Mv_tr) — caller's particle order
static int pse_lanczos_dot(void *ctx, const float *d_v, float *d_Mv, int n, void *stream) {
  (void)n;
  return pse_dot<3, false>(static_cast<PSENear *>(ctx), d_v, d_Mv, (hipStream_t)stream);
}
// the same product on vectors held in CELL order (the default path of computeStochasticDisplacements)
static int pse_lanczos_dot_sorted(void *ctx, const float *d_v, float *d_Mv, int n, void *stream) {
  (void)n;
  PSENear *p = static_cast<PSENear *>(ctx);
  hipStream_t st = (hipStream_t)stream;
  const int N = p->N;
  const real3f Lb{p->boxL[0], p->boxL[1], p->boxL[2]};
  UH_NEAR8(3, false, false);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// the sorted product as a fused Lanczos iteration (lanczos_fused.hpp): only on the pair-record path, declines otherwise
static int pse_lanczos_fused(void *ctx, LanczosFusedArgs *a, int n, void *stream) {
  (void)n;
  PSENear *p = static_cast<PSENear *>(ctx);
  hipStream_t st = (hipStream_t)stream;
  const int N = p->N;
  if (!(p->pairList && p->lazyList && p->listValid && !p->pairsUnfit && p->nearKernel == 1)) return 1;
  const bool ahead = p->optimistic && p->pairsPending && !p->pairsValid;
  if (!p->pairsValid && !ahead) { if (int e = pse_build_pairs(p, st)) return e; }
  if (!(p->pairsValid || ahead)) return 1;
  const int nb = (N + kNearBlock / kNearGroup - 1) / (kNearBlock / kNearGroup);
  if (nb > a->partsACap || a->npB > kNearBlock) return 1;
  a->npA = nb;
  if (p->riderArmed && !p->riderLaunched && p->riderFs.ptr && p->riderMFs.ptr) {   // the solve's first product carries M_near F along
    hipLaunchKernelGGL(k_pse_near_pairs_lanczos<true>, dim3(nb), dim3(kNearBlock), 0, st, (const float4 *)p->recA.ptr,
                       (const float2 *)p->recB.ptr, (const int2 *)p->pairRange.ptr, N, *a, (const float *)p->riderFs.ptr,
                       (float *)p->riderMFs.ptr);
    p->riderLaunched = true;
  } else
    hipLaunchKernelGGL(k_pse_near_pairs_lanczos<false>, dim3(nb), dim3(kNearBlock), 0, st, (const float4 *)p->recA.ptr,
                       (const float2 *)p->recB.ptr, (const int2 *)p->pairRange.ptr, N, *a, (const float *)nullptr, (float *)nullptr);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- BDHI::Lanczos: dense open-boundary RPY mobility, matrix free (Integrator/BDHI/BDHI_Lanczos.cu:56-118, BDHI.cuh:27-96) ----
// (templates over `real`: the double-precision instantiation is the DOUBLE_PRECISION build of the module, which is how the reference's own
// acceptance test of it is compiled — test/BDHI/Lanczos_Cholesky/Makefile:3 — and where its bar of 1e-7 applies)
template <class T> UH_D T rpy_fma(T a, T b, T c);
template <> UH_D float rpy_fma<float>(float a, float b, float c) { return fmaf(a, b, c); }
template <> UH_D double rpy_fma<double>(double a, double b, double c) { return fma(a, b, c); }
template <class T> UH_D void rpy_different_sizes(T M0, T r, T ai, T aj, T &c1, T &c2) {
  const T asum = ai + aj;
  const T asub = ai > aj ? ai - aj : aj - ai;
  if (r > asum) {
    const T invr = T(1) / r;
    const T pref = M0 * T(3) * T(0.25) * invr;
    const T denom = rpy_fma(ai, ai, aj * aj) / (T(3) * r * r);
    c1 = pref * (T(1) + denom);
    c2 = pref * rpy_fma(T(-3), denom, T(1)) * invr * invr;
  } else if (r > asub) {
    const T pref = M0 / (ai * aj * T(32) * r * r * r);
    T num = rpy_fma(T(3) * r, r, asub * asub);
    c1 = pref * rpy_fma(T(16) * r * r * r, asum, -(num * num));
    num = rpy_fma(-r, r, asub * asub);
    c2 = pref * (T(3) * num * num) / (r * r);
  } else {
    c1 = M0 / (ai > aj ? ai : aj);
    c2 = T(0);
  }
}

// NBody::transverse with NbodyMatrixFreeMobilityDot: thread per particle i, all j in ascending order through LDS tiles
// (positions + radius in one real4, v in another); Mv[i] = total (overwrites).
template <int VSTRIDE, class T>
__global__ void __launch_bounds__(128) k_rpy_nbody(const T *__restrict__ pos, const T *__restrict__ v, const T *__restrict__ radius, T rh, T M0, int N,
                                                    T *__restrict__ Mv) {
  __shared__ T tp[128][4], tv[128][3];
  const int i = blockIdx.x * 128 + threadIdx.x;
  const bool active = i < N;
  const T pix = active ? pos[4 * (size_t)i] : T(0), piy = active ? pos[4 * (size_t)i + 1] : T(0), piz = active ? pos[4 * (size_t)i + 2] : T(0);
  const T ai = active ? (radius ? radius[i] : rh) : T(1);
  T tx = 0, ty = 0, tz = 0;
  for (int base = 0; base < N; base += 128) {
    const int j = base + threadIdx.x;
    if (j < N) {
      tp[threadIdx.x][0] = pos[4 * (size_t)j]; tp[threadIdx.x][1] = pos[4 * (size_t)j + 1]; tp[threadIdx.x][2] = pos[4 * (size_t)j + 2];
      tp[threadIdx.x][3] = radius ? radius[j] : rh;
      const T *vj = v + (size_t)VSTRIDE * j;
      tv[threadIdx.x][0] = vj[0]; tv[threadIdx.x][1] = vj[1]; tv[threadIdx.x][2] = vj[2];
    }
    __syncthreads();
    const int cnt = min(128, N - base);
    if (active) {
      for (int t = 0; t < cnt; ++t) {
        const T rx = pix - tp[t][0], ry = piy - tp[t][1], rz = piz - tp[t][2];
        const T vx = tv[t][0], vy = tv[t][1], vz = tv[t][2];
        const T r2 = rpy_fma(rz, rz, rpy_fma(ry, ry, rx * rx));   // dot3's order
        const T r = sqrt(r2);
        T f, g;
        rpy_different_sizes<T>(M0, r, ai, tp[t][3], f, g);
        if (r == T(0)) {
          tx += f * vx; ty += f * vy; tz += f * vz;
        } else {
          const T gv = g * rpy_fma(rz, vz, rpy_fma(ry, vy, rx * vx));
          tx += rpy_fma(gv, rx, f * vx);
          ty += rpy_fma(gv, ry, f * vy);
          tz += rpy_fma(gv, rz, f * vz);
        }
      }
    }
    __syncthreads();
  }
  if (active) { Mv[3 * (size_t)i] = tx; Mv[3 * (size_t)i + 1] = ty; Mv[3 * (size_t)i + 2] = tz; }
}

template <class T> struct RpyDotCtxT { const T *pos, *radius; T rh, M0; int N; };
using RpyDotCtx = RpyDotCtxT<float>;
template <class T> static int rpy_lanczos_dot_t(void *ctx, const T *d_v, T *d_Mv, int n, void *stream) {
  const RpyDotCtxT<T> *c = static_cast<const RpyDotCtxT<T> *>(ctx);
  hipLaunchKernelGGL((k_rpy_nbody<3, T>), dim3((c->N + 127) / 128), dim3(128), 0, (hipStream_t)stream, c->pos, d_v, c->radius, c->rh, c->M0, c->N, d_Mv);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
static int rpy_lanczos_dot(void *ctx, const float *d_v, float *d_Mv, int n, void *stream) { return rpy_lanczos_dot_t<float>(ctx, d_v, d_Mv, n, stream); }
static int rpy_lanczos_dot64(void *ctx, const double *d_v, double *d_Mv, int n, void *stream) { return rpy_lanczos_dot_t<double>(ctx, d_v, d_Mv, n, stream); }

}  // namespace uammd_hip

using namespace uammd_hip;

extern "C" {

int uammd_pse_near_create(const float boxSize[3], float viscosity, float hydrodynamicRadius, float tolerance, float psi,
                          float shearStrain, unsigned int seed, uammd_pse_near **out, float *rcut_out, int *nPointsTable_out) {
  if (!boxSize || !out) { set_last_error("uammd_pse_near_create: null argument"); return -1; }
  // NearField::initializeDeterministicPart, NearField.cuh:65-99
  const double split = psi;
  const float rcut = (float)(std::sqrt(-std::log(tolerance)) / split);
  if (0.5 * boxSize[0] < rcut) {
    set_last_error("[BDHI::PSE] Cut off is too large, try increasing psi");
    return -2;
  }
  const double a = hydrodynamicRadius;
  const float textureTolerance = (float)(a * tolerance);
  double np = rcut / textureTolerance + 0.5;
  if (np > 2e30) np = 2e30;
  unsigned nPointsTable = np >= 4294967295.0 ? 4294967295u : (unsigned)np;
  nPointsTable = std::min(1u << 22, std::max(1u << 14, nPointsTable));
  PSENear *p = new PSENear();
  for (int k = 0; k < 3; ++k) p->boxL[k] = boxSize[k];
  p->rcut = rcut;
  p->shear = shearStrain;
  p->tolerance = tolerance;
  p->seed = seed;
  p->nPointsTable = (int)nPointsTable;
  // TabulatedFunction<real2>(table, nPointsTable, 0, rcut, rpy): Ntable = nPointsTable - 1 intervals, Ntable + 1 samples
  const int Ntable = (int)nPointsTable - 1;
  const float normalization = (float)(6 * M_PI * a * viscosity);
  std::vector<float2> host((size_t)Ntable + 1);
  for (int i = 0; i <= Ntable; ++i) {
    const double x = (i / (double)Ntable) * (rcut - 0.0f) + 0.0f;
    double F, G;
    rpy_near_FandG(x, (double)hydrodynamicRadius, (double)psi, (double)rcut, &F, &G);
    host[i] = make_float2((float)(F / (double)normalization), (float)(G / (double)normalization));
  }
  if (p->table.reserve(sizeof(float2) * host.size()) ||
      hipMemcpy(p->table.ptr, host.data(), sizeof(float2) * host.size(), hipMemcpyHostToDevice) != hipSuccess) {
    delete p;
    set_last_error("uammd_pse_near_create: could not upload the RPY table");
    return -3;
  }
  {  // the same samples, two per entry: {F_i, G_i, F_i+1, G_i+1} (one 16-byte request per interpolation)
    std::vector<float4> h4((size_t)Ntable + 1);
    for (int i = 0; i <= Ntable; ++i) {
      const float2 a = host[i], b = host[i < Ntable ? i + 1 : i];
      h4[i] = make_float4(a.x, a.y, b.x, b.y);
    }
    if (p->table4.reserve(sizeof(float4) * h4.size()) ||
        hipMemcpy(p->table4.ptr, h4.data(), sizeof(float4) * h4.size(), hipMemcpyHostToDevice) != hipSuccess) {
      delete p;
      set_last_error("uammd_pse_near_create: could not upload the RPY table");
      return -3;
    }
  }
  if (int e = uammd_lanczos_create(&p->lanczos)) { delete p; return e; }
  *out = reinterpret_cast<uammd_pse_near *>(p);
  if (rcut_out) *rcut_out = rcut;
  if (nPointsTable_out) *nPointsTable_out = (int)nPointsTable;
  return 0;
}

int uammd_pse_near_destroy(uammd_pse_near *h) {
  delete reinterpret_cast<PSENear *>(h);
  return 0;
}

int uammd_pse_near_set_shear_strain(uammd_pse_near *h, float shearStrain) {
  if (!h) { set_last_error("uammd_pse_near_set_shear_strain: null handle"); return -1; }
  reinterpret_cast<PSENear *>(h)->shear = shearStrain;
  reinterpret_cast<PSENear *>(h)->listValid = false;  // (the list's cut-off carries the shear's safety factor)
  reinterpret_cast<PSENear *>(h)->candValid = false;
  return 0;
}

// "exact_order" (0): 1 = the thread-per-particle walk in the reference's summation order instead of the eight-lanes-per-particle kernel.
// "lazy_list" (0): 1 = rebuild the cell list only after uammd_pse_near_positions_changed (CellList::update's needsRebuild,
// NeighbourList/CellList.cuh:134-136,192-204) or when the position array / particle count of the call changes.
// "pair_list" (1): with "lazy_list", the pairs of a list are evaluated once into records and every product streams them (PSENear::pairList);
// 0 = every product scans the cells.  (A particle with more than 96 neighbours inside the cut-off sends the handle to the scanning product.)
int uammd_pse_near_set_option(uammd_pse_near *h, const char *name, int value) {
  if (!h || !name) { set_last_error("uammd_pse_near_set_option: null argument"); return -1; }
  PSENear *p = reinterpret_cast<PSENear *>(h);
  if (std::string(name) == "exact_order") { p->exactOrder = value != 0; p->listValid = false; p->candValid = false; return 0; }
  if (std::string(name) == "pair_list") { p->pairList = value != 0; return 0; }
  if (std::string(name) == "optimistic_records") { p->optimisticRecords = value != 0; return 0; }
  if (std::string(name) == "pair_capacity" && value >= 0) { p->pairCapFirst = (size_t)value; p->pairCap = 0; p->pairsValid = false; p->pairsPending = false; return 0; }
  if (std::string(name) == "defer_checks") return uammd_lanczos_set_option(p->lanczos, "defer_checks", value);
  if (std::string(name) == "fuse_recurrence") return uammd_lanczos_set_option(p->lanczos, "fuse_recurrence", value);
  if (std::string(name) == "lazy_list") { p->lazyList = value != 0; p->listValid = false; p->candValid = false; return 0; }
  if (std::string(name) == "near_kernel" && (value == 0 || value == 1)) { p->nearKernel = value; p->listValid = false; p->candValid = false; return 0; }
  if (std::string(name) == "list_skin_percent" && value >= 0 && value <= 100) {
    p->skinPercent = value; p->candEnabled = true; p->candShortLived = 0; p->candGrow = 0; p->candCap = 0; p->maxInc = 0.f;
    p->listValid = false; p->candValid = false;
    return 0;
  }
  set_last_error("uammd_pse_near_set_option: unknown option %s", name);
  return -1;
}
// One-shot: the NEXT uammd_pse_near_stochastic on this handle also does uammd_pse_near_mdot(h, its positions, d_force, N, d_MF) — as a
// second right-hand side of the solve's first product where that product streams the pair records (they are read once for both),
// by the plain product before it returns otherwise (also when T = 0).  d_MF += M_near F either way, after the interleaved work.
int uammd_pse_near_set_mdot_rider(uammd_pse_near *h, const float *d_force, float *d_MF) {
  if (!h || (d_force && !d_MF)) { set_last_error("uammd_pse_near_set_mdot_rider: null argument"); return -1; }
  PSENear *p = reinterpret_cast<PSENear *>(h);
  p->riderForce = d_force;
  p->riderMF = d_force ? d_MF : nullptr;
  return 0;
}
int uammd_pse_near_positions_changed(uammd_pse_near *h) {
  if (!h) { set_last_error("uammd_pse_near_positions_changed: null handle"); return -1; }
  reinterpret_cast<PSENear *>(h)->listValid = false;
  return 0;
}

// cl->update (NearField.cuh:231-237) ahead of the products, with nothing waited for: the list and — with "lazy_list" and "pair_list" —
// the launch half of the pair records' build.  The products of the same positions on the same stream find both done; a caller
// (BDHI::PSE::computeMF: far field, then near field, BDHI_PSE.cuh:92-120) that calls this BEFORE queueing the far field has the
// records' counters read while the GPU works on the far field instead of draining the stream for them.
int uammd_pse_near_prepare(uammd_pse_near *h, const float *d_pos, int N, void *stream) {
  if (!h || (N > 0 && !d_pos)) { set_last_error("uammd_pse_near_prepare: null argument"); return -1; }
  if (N <= 0) return 0;
  PSENear *p = reinterpret_cast<PSENear *>(h);
  hipStream_t st = (hipStream_t)stream;
  if (int e = pse_update_list(p, d_pos, N, st)) return e;
  if (p->pairList && p->lazyList && p->listValid && !p->pairsUnfit && p->nearKernel == 1 && !p->exactOrder && !p->pairsValid && !p->pairsPending)
    return pse_launch_pairs(p, st);
  return 0;
}

// diagnostics: the pair records the products stream (0 while there are none: before the first product, with "pair_list" off, or after a
// particle overflowed its hit list) and the records allocated
int uammd_pse_near_pair_records(uammd_pse_near *h, long long *records, long long *capacity) {
  if (!h) { set_last_error("uammd_pse_near_pair_records: null handle"); return -1; }
  PSENear *p = reinterpret_cast<PSENear *>(h);
  if (records) *records = (p->pairsValid && p->pairTotalHost) ? pse_pairs_total(p) : 0;
  if (capacity) *capacity = (long long)p->pairCap;
  return 0;
}

// diagnostics of the kept candidate lists (option "list_skin_percent"): {list builds from scratch, record builds from a kept list, builds
// repeated because the displacement bound was broken, 1 while the mechanism is on for the handle}
int uammd_pse_near_list_stats(uammd_pse_near *h, long long out[4]) {
  if (!h || !out) { set_last_error("uammd_pse_near_list_stats: null argument"); return -1; }
  const PSENear *p = reinterpret_cast<const PSENear *>(h);
  out[0] = p->fullBuilds; out[1] = p->candBuilds; out[2] = p->candRepeats; out[3] = pse_cand_usable(p) ? 1 : 0;
  return 0;
}

// fn(ctx, stream) is handed to the Lanczos solver of the NEXT uammd_pse_near_stochastic (uammd_lanczos_set_interleave): it is called once,
// behind the solve's first convergence check, for the caller's other work on the stream — BDHI::PSE queues its far field there, so that the
// one wait of a step has the far field running behind it.  fn must not call into THIS handle (the solve may be streaming pair records
// whose counters are still unread: a product from inside fn is refused).  One-shot.
int uammd_pse_near_set_interleave(uammd_pse_near *h, uammd_interleave_fn fn, void *ctx) {
  if (!h) { set_last_error("uammd_pse_near_set_interleave: null handle"); return -1; }
  reinterpret_cast<PSENear *>(h)->interleave = fn;
  reinterpret_cast<PSENear *>(h)->interleaveCtx = ctx;
  return 0;
}

// one stage earlier (uammd_lanczos_set_interleave_early): before the kernels that wait for the host's answer to the check
int uammd_pse_near_set_interleave_early(uammd_pse_near *h, uammd_interleave_fn fn, void *ctx) {
  if (!h) { set_last_error("uammd_pse_near_set_interleave_early: null handle"); return -1; }
  reinterpret_cast<PSENear *>(h)->interleaveEarly = fn;
  reinterpret_cast<PSENear *>(h)->interleaveEarlyCtx = ctx;
  return 0;
}

// NearField::Mdot (NearField.cuh:239-250): d_MF real3[N] += M_near F, forces real4[N] (NULL: nothing to do)
int uammd_pse_near_mdot(uammd_pse_near *h, const float *d_pos, const float *d_force, int N, float *d_MF, void *stream) {
  if (!h || (N > 0 && (!d_pos || !d_MF))) { set_last_error("uammd_pse_near_mdot: null argument"); return -1; }
  if (!d_force || N <= 0) return 0;
  PSENear *p = reinterpret_cast<PSENear *>(h);
  if (p->optimistic) { set_last_error("uammd_pse_near_mdot: called from inside the handle's own solve (uammd_pse_near_set_interleave)"); return -1; }
  if (int e = pse_update_list(p, d_pos, N, (hipStream_t)stream)) return e;
  return pse_dot<4, true>(p, d_force, d_MF, (hipStream_t)stream);
}

// NearField::computeStochasticDisplacements (NearField.cuh:252-285): d_BdW real3[N] = prefactor sqrt(2 T) M_near^(1/2) dW
// (the Lanczos result OVERWRITES d_BdW).  seed2 = the per-call draw of System::rng().  Nothing happens when T == 0.
int uammd_pse_near_stochastic(uammd_pse_near *h, const float *d_pos, int N, float temperature, float prefactor,
                              unsigned int seed2, float *d_BdW, void *stream, int *iterations) {
  if (!h) { set_last_error("uammd_pse_near_stochastic: null handle"); return -1; }
  if (iterations) *iterations = 0;
  PSENear *p = reinterpret_cast<PSENear *>(h);
  hipStream_t st = (hipStream_t)stream;
  // The caller's interleaved work (uammd_pse_near_set_interleave) is queued EXACTLY once by this call, whatever path it takes: by the
  // solver behind its first check, or here on the way out (nothing to solve, an error before the solve, a solve that never waited) —
  // and no registration outlives the call (its context may live on the caller's stack).
  struct Hook {
    uammd_interleave_fn fn; void *ctx; bool fired = false; int rc = 0;
    static int tramp(void *self, void *stream) {
      Hook *k = static_cast<Hook *>(self);
      k->fired = true;
      return k->rc = k->fn(k->ctx, stream);
    }
    void flush(void *stream) { if (fn && !fired) { fired = true; rc = fn(ctx, stream); } }
  };
  struct Hooks {   // (the early one first, whatever the path)
    PSENear *p; void *stream; Hook early, late;
    ~Hooks() {
      if (early.fn) (void)uammd_lanczos_set_interleave_early(p->lanczos, nullptr, nullptr);
      if (late.fn) (void)uammd_lanczos_set_interleave(p->lanczos, nullptr, nullptr);
      early.flush(stream);
      late.flush(stream);
    }
  } hooks{p, stream, {p->interleaveEarly, p->interleaveEarlyCtx}, {p->interleave, p->interleaveCtx}};
  Hook &hook = hooks.late;
  p->interleave = p->interleaveEarly = nullptr;
  p->interleaveCtx = p->interleaveEarlyCtx = nullptr;
  auto arm = [&]() -> int {   // hand both to the solver of the coming run
    if (hooks.early.fn && !hooks.early.fired) { if (int e = uammd_lanczos_set_interleave_early(p->lanczos, &Hook::tramp, &hooks.early)) return e; }
    if (hook.fn && !hook.fired) { if (int e = uammd_lanczos_set_interleave(p->lanczos, &Hook::tramp, &hook)) return e; }
    return 0;
  };
  // the rider (uammd_pse_near_set_mdot_rider): M_near F added to riderMF by this call — inside the solve's first product, or, where no
  // product took it (T = 0, the scanning products, an unfused solve), by the plain product on the way out
  const float *riderF = p->riderForce;
  float *riderMF = p->riderMF;
  p->riderForce = nullptr;
  p->riderMF = nullptr;
  p->riderArmed = p->riderLaunched = false;
  bool riderDone = riderF == nullptr;
  auto leave = [&](int rc) -> int {   // (a hook's own failure is reported when nothing else failed)
    hooks.early.flush(stream);
    hook.flush(stream);
    p->riderArmed = false;
    if (!rc && !riderDone && N > 0 && d_pos) {
      riderDone = true;
      rc = pse_update_list(p, d_pos, N, st);
      if (!rc) rc = pse_dot<4, true>(p, riderF, riderMF, st);
    }
    const int hrc = hooks.early.rc ? hooks.early.rc : hook.rc;
    if (!rc && hrc) {
      if (!uammd_hip_last_error()[0]) set_last_error("uammd_pse_near_stochastic: the interleaved callback failed (%d)", hrc);
      return hrc;
    }
    return rc;
  };
  if (N > 0 && (!d_pos || !d_BdW)) { set_last_error("uammd_pse_near_stochastic: null argument"); return leave(-1); }
  if (temperature == 0.0f || N <= 0) return leave(0);
  if (int e = pse_update_list(p, d_pos, N, st)) return leave(e);
  if (int e = p->noise.reserve(sizeof(float) * 3 * (size_t)N)) return leave(e);
  const float noise_prefactor = prefactor * sqrtf(2 * temperature);
  int it = 0;
  if (p->exactOrder) {
    if (int e = arm()) return leave(e);
    hipLaunchKernelGGL(k_pse_noise, dim3((N + 255) / 256), dim3(256), 0, st, (float *)p->noise.ptr, N, noise_prefactor, p->seed,
                       seed2);
    if (hipGetLastError() != hipSuccess) { set_last_error("uammd_pse_near_stochastic: kernel launch failed"); return leave(-1); }
    const int rc = uammd_lanczos_run(p->lanczos, &pse_lanczos_dot, p, d_BdW, (const float *)p->noise.ptr, p->tolerance, 3 * N,
                                     stream, &it);
    if (iterations) *iterations = it;
    return leave(rc);
  }
  if (int e = p->sortedOut.reserve(sizeof(float) * 3 * (size_t)N)) return leave(e);
  auto solve = [&]() -> int {
    // the noise, the partials of its norm for the solver (lanczos_set_znorm_parts) and — when a pair build has left it to this kernel —
    // the build's status block for the host
    const int nzb = std::min(256, (N + 255) / 256);
    if (int e = p->zparts.reserve(sizeof(float) * 256)) return e;
    const bool publish = p->statusCopyDeferred;
    const bool rider = !riderDone;
    if (rider) {
      if (int e = p->riderFs.reserve(sizeof(float) * 3 * (size_t)N)) return e;
      if (int e = p->riderMFs.reserve(sizeof(float) * 3 * (size_t)N)) return e;
    }
    p->riderArmed = rider;
    p->riderLaunched = false;
    hipLaunchKernelGGL(k_pse_noise_sorted_norm, dim3(nzb), dim3(256), 0, st, (float *)p->noise.ptr, (const int *)p->cl.index.ptr, N,
                       noise_prefactor, p->seed, seed2, (float *)p->zparts.ptr, publish ? (const int *)p->pairCursor.ptr : nullptr,
                       p->pairTotalDev, kPairStatusInts, rider ? (const float4 *)riderF : nullptr, rider ? (float *)p->riderFs.ptr : nullptr);
    UH_CHECK(hipGetLastError());
    if (publish) {
      p->statusCopyDeferred = false;
      UH_CHECK(hipEventRecord(p->pairsEvent, st));
    }
    if (int e = lanczos_set_znorm_parts(p->lanczos, (const float *)p->zparts.ptr, nzb)) return e;
    // (the pair-record product also runs the recurrence's neighbours: two launches per iteration instead of four, lanczos_fused.hpp)
    if (int e = lanczos_set_fused(p->lanczos, &pse_lanczos_fused, p)) return e;
    const int rcRun = uammd_lanczos_run(p->lanczos, &pse_lanczos_dot_sorted, p, (float *)p->sortedOut.ptr, (const float *)p->noise.ptr,
                                        p->tolerance, 3 * N, stream, &it);
    (void)lanczos_set_fused(p->lanczos, nullptr, nullptr);
    return rcRun;
  };
  // the records' counters are read after the solve instead of before it (PSENear::optimisticRecords)
  const bool records = p->pairList && p->lazyList && p->listValid && !p->pairsUnfit && p->nearKernel == 1;
  if (records && p->optimisticRecords && !p->pairsValid && !p->pairsPending) { if (int e = pse_launch_pairs(p, st, true)) return leave(e); }
  const bool ahead = records && p->optimisticRecords && p->pairsPending && !p->pairsValid;
  int schedule[2] = {0, 0};
  if (ahead) { if (int e = uammd_lanczos_get_schedule(p->lanczos, schedule)) return leave(e); }
  // (not ahead: the records are settled BEFORE the noise is drawn — a kept list whose displacement bound turns out broken is replaced by
  // a fresh one, in another particle order, and the solve's vectors live in the list's order)
  if (records && !ahead && !p->pairsValid) { if (int e = pse_build_pairs(p, st)) return leave(e); }
  if (int e = arm()) return leave(e);   // (one-shot: a repeated solve below runs without them)
  p->optimistic = ahead;
  int rc = solve();
  p->optimistic = false;
  if (ahead && p->pairsPending) {
    if (int e = pse_settle_status(p, st)) return leave(e);
    UH_CHECK(hipEventSynchronize(p->pairsEvent));
    p->pairsPending = false;
    bool ok = false;
    if (int e = pse_pairs_verdict(p, st, &ok)) return leave(e);
    if (!ok) {   // the products of this solve missed pairs: the build again (pse_build_pairs from the first product: larger arrays, a fresh list, or back to the scanning product), the solve again
      if (int e = uammd_lanczos_set_schedule(p->lanczos, schedule)) return leave(e);
      rc = solve();
    }
  }
  if (iterations) *iterations = it;
  if (rc) return leave(rc);
  const bool carried = !riderDone && p->riderArmed && p->riderLaunched;   // the last solve's first product made M_near F (list order)
  hipLaunchKernelGGL(k_pse_unsort3, dim3((N + 255) / 256), dim3(256), 0, st, (const float *)p->sortedOut.ptr, (const int *)p->cl.index.ptr, N,
                     d_BdW, carried ? (const float *)p->riderMFs.ptr : nullptr, carried ? riderMF : nullptr);
  UH_CHECK(hipGetLastError());
  if (carried) riderDone = true;
  return leave(0);
}

// test hook: the Saru noise vector of computeStochasticDisplacements (real3[N])
int uammd_pse_near_noise(uammd_pse_near *h, int N, float variance, unsigned int seed2, float *d_out3, void *stream) {
  if (!h || !d_out3) { set_last_error("uammd_pse_near_noise: null argument"); return -1; }
  PSENear *p = reinterpret_cast<PSENear *>(h);
  hipLaunchKernelGGL(k_pse_noise, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_out3, N, variance, p->seed, seed2);
  UH_CHECK(hipGetLastError());
  return 0;
}

// test hook: raw M_near v for a real3 vector (what the Lanczos callback computes), d_Mv overwritten
int uammd_pse_near_dot(uammd_pse_near *h, const float *d_pos, const float *d_v3, int N, float *d_Mv3, void *stream) {
  if (!h || !d_pos || !d_v3 || !d_Mv3) { set_last_error("uammd_pse_near_dot: null argument"); return -1; }
  PSENear *p = reinterpret_cast<PSENear *>(h);
  if (int e = pse_update_list(p, d_pos, N, (hipStream_t)stream)) return e;
  return pse_lanczos_dot(p, d_v3, d_Mv3, 3 * N, stream);
}

// BDHI::Lanczos::computeMF / Dotctor (BDHI_Lanczos.cu:120-160): d_Mv real3[N] = M_RPY v, v with stride 3 (real3) or 4 (real4
// forces); d_radius nullable (then every particle has hydrodynamicRadius).  Open boundaries: no box.
int uammd_rpy_nbody_mdot(const float *d_pos, const float *d_v, int vstride, const float *d_radius, float hydrodynamicRadius,
                         float viscosity, int N, float *d_Mv, void *stream) {
  if (N <= 0) return 0;
  if (!d_pos || !d_v || !d_Mv || (vstride != 3 && vstride != 4)) { set_last_error("uammd_rpy_nbody_mdot: bad arguments"); return -1; }
  if (!d_radius && !(hydrodynamicRadius > 0)) {
    set_last_error("[BDHI::Lanczos] You need to provide Lanczos with either an hydrodynamic radius or via the individual particle radius.");
    return -2;
  }
  const float M0 = (float)(1 / (6 * M_PI * viscosity));
  if (vstride == 3)
    hipLaunchKernelGGL((k_rpy_nbody<3, float>), dim3((N + 127) / 128), dim3(128), 0, (hipStream_t)stream, d_pos, d_v, d_radius, hydrodynamicRadius, M0, N, d_Mv);
  else
    hipLaunchKernelGGL((k_rpy_nbody<4, float>), dim3((N + 127) / 128), dim3(128), 0, (hipStream_t)stream, d_pos, d_v, d_radius, hydrodynamicRadius, M0, N, d_Mv);
  UH_CHECK(hipGetLastError());
  return 0;
}
// ... with real = double
int uammd_rpy_nbody_mdot_f64(const double *d_pos, const double *d_v, int vstride, const double *d_radius, double hydrodynamicRadius, double viscosity,
                             int N, double *d_Mv, void *stream) {
  if (N <= 0) return 0;
  if (!d_pos || !d_v || !d_Mv || (vstride != 3 && vstride != 4)) { set_last_error("uammd_rpy_nbody_mdot_f64: bad arguments"); return -1; }
  if (!d_radius && !(hydrodynamicRadius > 0)) {
    set_last_error("[BDHI::Lanczos] You need to provide Lanczos with either an hydrodynamic radius or via the individual particle radius.");
    return -2;
  }
  const double M0 = 1 / (6 * M_PI * viscosity);
  if (vstride == 3)
    hipLaunchKernelGGL((k_rpy_nbody<3, double>), dim3((N + 127) / 128), dim3(128), 0, (hipStream_t)stream, d_pos, d_v, d_radius, hydrodynamicRadius, M0, N, d_Mv);
  else
    hipLaunchKernelGGL((k_rpy_nbody<4, double>), dim3((N + 127) / 128), dim3(128), 0, (hipStream_t)stream, d_pos, d_v, d_radius, hydrodynamicRadius, M0, N, d_Mv);
  UH_CHECK(hipGetLastError());
  return 0;
}

// BDHI::Lanczos::computeBdW (BDHI_Lanczos.cu:162-188): d_BdW real3[N] = M_RPY^(1/2) d_noise by the Lanczos iteration.  The
// reference draws the noise with cuRAND (third party, stream unpinned): here the caller supplies N(0,1) numbers.
int uammd_rpy_lanczos_bdw(uammd_lanczos *solver, const float *d_pos, const float *d_radius, float hydrodynamicRadius,
                          float viscosity, int N, const float *d_noise, float tolerance, float *d_BdW, void *stream,
                          int *iterations) {
  if (iterations) *iterations = 0;
  if (N <= 0) return 0;
  if (!solver || !d_pos || !d_noise || !d_BdW) { set_last_error("uammd_rpy_lanczos_bdw: null argument"); return -1; }
  RpyDotCtx ctx{d_pos, d_radius, hydrodynamicRadius, (float)(1 / (6 * M_PI * viscosity)), N};
  int it = 0;
  const int rc = uammd_lanczos_run(solver, &rpy_lanczos_dot, &ctx, d_BdW, d_noise, tolerance, 3 * N, stream, &it);
  if (iterations) *iterations = it;
  return rc;
}
int uammd_rpy_lanczos_bdw_f64(uammd_lanczos_f64 *solver, const double *d_pos, const double *d_radius, double hydrodynamicRadius, double viscosity, int N,
                              const double *d_noise, double tolerance, double *d_BdW, void *stream, int *iterations) {
  if (iterations) *iterations = 0;
  if (N <= 0) return 0;
  if (!solver || !d_pos || !d_noise || !d_BdW) { set_last_error("uammd_rpy_lanczos_bdw_f64: null argument"); return -1; }
  RpyDotCtxT<double> ctx{d_pos, d_radius, hydrodynamicRadius, 1 / (6 * M_PI * viscosity), N};
  int it = 0;
  const int rc = uammd_lanczos_run_f64(solver, &rpy_lanczos_dot64, &ctx, d_BdW, d_noise, tolerance, 3 * N, stream, &it);
  if (iterations) *iterations = it;
  return rc;
}

}  // extern "C"
