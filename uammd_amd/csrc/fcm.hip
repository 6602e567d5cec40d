// Path B — Force Coupling Method (triply periodic Stokes solver) for gfx950.
//
// Reference pipeline (Integrator/BDHI/FCM/FCM_impl.cuh:652-693):
//   spreadForces (IBM, :245-262) -> 3 batched R2C FFTs (:293-304) -> forceFourier2Vel (:375-397)
//   -> fourierBrownianNoise (:437-512) -> 3 batched C2R FFTs (:544-557) -> interpolateVelocity (:559-581)
//
// MI355X design
//   * The three Cartesian components live in three PLANAR padded grids [nz][ny][2(nx/2+1)] instead of the
//     reference's xyz-interleaved real3 grid: every rocFFT pass then streams unit-stride rows, and the
//     R2C/C2R transforms run IN PLACE (the padded real layout is exactly the Hermitian layout), so the
//     solver touches 1 grid-size of HBM per transform direction instead of 2 and needs no second buffer.
//   * forceFourier2Vel and fourierBrownianNoise are ONE kernel in gather form: every Fourier node adds
//     its own draw and, on the kx = 0 / kx = nx/2 planes, the conjugate of its partner's draw, which it
//     regenerates from the same Saru(id, seed, seed2) stream.  No node is written by two threads (the
//     reference's scatter form races on the kx = nx/2 plane) and the result is deterministic.
//   * spread/gather: one wave per particle, weights by ds_bpermute, f32 hardware atomics into L2 for the
//     spread (see also the tile-owned variant below).
//   * all work buffers are owned by the handle; steady state allocates nothing.
#include "ibm.hpp"
#include "celllist.hpp"
#include "saru.hpp"

#include <rocfft/rocfft.h>

#include <cmath>
#include <mutex>
#include <string>

namespace uammd_hip {

#define UH_ROCFFT(expr)                                                                      \
  do {                                                                                       \
    rocfft_status s_ = (expr);                                                               \
    if (s_ != rocfft_status_success) {                                                       \
      set_last_error("%s failed with rocfft_status %d (%s:%d)", #expr, (int)s_, __FILE__, __LINE__); \
      return -10 - (int)s_;                                                                  \
    }                                                                                        \
  } while (0)

// PSE far field (FarField.cuh:85-119): hydrodynamic radius, Ewald splitting, kernel splitting, shear strain
struct PseGreens { float rh, split, eta, shear; bool on; };

// the tile-sorted stencil arrays of a solve (fcm_prepare_tiles)
struct FcmPrep {
  // all arrays below are in TILE-SORTED order (slot = tileStart[tile] + rank) except tileOf/rank
  int4 *origin;     // int4[N]: first stencil node per axis (celli - P), unwrapped; .w = original particle index
  float *weights;   // float[wstride*N]
  float4 *force;    // float4[N] (xyz)
  int *tileOf;      // int[N]   (original order)
  int *rank;        // int[N]   (original order)
  int *tileCount;   // int[ntiles]
  int *tileStart;   // int[ntiles+1]
  int wstride;
  int3 tdim;        // tile edge per axis (<= kTile: the kernels' layouts are sized for kTile, shorter tiles leave rows / columns unused)
  // SLOT layout of a step that was prepared by the previous step's update kernel (k_fcm_step_prep; cap > 0): origin / weights stay in
  // the order of the last sorted solve (entry s, .w = the particle), and the spread's candidate records live in FIXED-CAPACITY tile
  // slots — rec[tile * cap + rank] = entry | particle << 21 | origin relative to the tile (3 x 7 bits) << 42 — so that no scan separates the binning
  // from the spread.  A tile that receives more than cap particles appends the rest to the overflow records rec[ntiles * cap + k],
  // which every tile tests (rare; counted, and reported to the host through slotFlag when it is no longer rare).
  unsigned long long *rec;   // entry (21 bits) | particle << 21 | packed origin << 42
  int *ovfTile;     // int[N]: the tile of overflow record k
  int cap;          // 0: the compact layout above
  int *slotCount;   // int[ntiles + 2]: this solve's tile populations, [ntiles] = overflow records, [ntiles + 1] = tile changes along the entry sequence
  int *slotCountNext;  // the other parity's counters: the spread hands them to the next update kernel zeroed
  int *slotFlag;    // host-mapped: [0] set when the overflow list is long enough to cost time, [1] = the last solve's tile changes along its entries
  const float4 *forceById;  // the caller's force array (slot layout: forces are fetched for the LISTED particles only)
};

struct FCM {
  uammd_fcm_parameters par;
  GridT<float> grid;
  IBMKernelDev kern;
  IBMKernelDev kernT{};    // torque window (FCM_ns::Kernels::GaussianTorque), set by uammd_fcm_set_torque_kernel
  bool haveTorqueKernel = false;
  int nxpad = 0;           // 2*(nx/2+1)
  size_t planeReal = 0;    // floats per component plane
  size_t planeCplx = 0;    // complex per component plane
  DeviceBuffer gridBuf, gridBufT, interBuf, work, prepOrigin, prepWeights, prepTileOf, prepRank, prepTileCount, prepTileStart, prepSorted;
  int emVelN = 0;
  DeviceBuffer emVel;              // velocities of uammd_fcm_step_euler_maruyama when the caller keeps none
  bool emBin = true;               // uammd_fcm_step_euler_maruyama: the update kernel also bins the positions it writes for the next call
  bool binnedPending = false;      // tileCount / tileOf / rank hold the binning of binnedPos (written by k_fcm_update_bin)
  const void *binnedPos = nullptr;
  int binnedN = 0;
  bool useTiles = false;   // grid divisible by the tile and >= 3 tiles per dimension
  int3 ntiles{0, 0, 0};
  int3 tdim{8, 8, 8};      // tile edge per axis, 4..8 nodes: the largest divisor of the axis that holds the stencil's reach (fcm_tiles_usable)
  int prepCapN = 0;
  bool tileCountZero = false;       // prepTileCount holds zeros (k_fcm_tile_scan leaves it so)
  // slot layout (FcmPrep::cap > 0): uammd_fcm_step_euler_maruyama's update kernel prepares the NEXT solve completely (k_fcm_step_prep)
  DeviceBuffer prepRec, prepOvfTile, prepSlotCount;
  bool slotsEnabled = !(getenv("UAMMD_FCM_SLOTS") && atoi(getenv("UAMMD_FCM_SLOTS")) == 0);   // option "slots" (environment: A/B runs of programs that do not set options)
  int slotCap = 0;                  // records per tile
  int slotParity = 0;               // which half of prepSlotCount the pending preparation counted into
  bool slotPending = false;         // origin / weights / rec / counters hold the preparation of slotPos (k_fcm_step_prep)
  bool slotDirty = true;            // the counters may hold anything (first use, or a pending preparation was dropped)
  const void *slotPos = nullptr;
  int slotN = 0;
  int slotRefresh = 128;            // a sorted (compact) solve every so many steps: the entries' order is what keeps the gather's windows local (option "slot_refresh")
  bool lastSolveSlots = false;      // the solve that has just run read the slot layout
  int orderN = 0;                   // origin[0 .. orderN).w is a permutation of the particles (entry -> particle) left by an earlier solve of orderN particles
  int slotSteps = 0;                // slot-layout steps since the last sorted solve (the entries' order is refreshed every kSlotRefresh)
  int *slotFlagHost = nullptr, *slotFlagDev = nullptr;  // mapped: the spread reports an overflow list that is long enough to cost time
  bool binBySlot = true;            // option "bin_by_slot": the step's update + binning pass walks the particles in the solve's tile order (k_fcm_update_bin)
  hipStream_t prepStream = nullptr;  // ... an ordering that only holds within one stream: a call on another stream waits for this one first
  bool prepStreamSet = false;
  bool forceAtomicSpread = false;  // test hook
  bool interGather = true;         // gather from an interleaved float4 copy of the velocity grids (k_fcm_interleave)
  int gatherPerWave = 2;           // particles per wave of k_fcm_gather_inter (1, 2, 4)
  int spreadWaves = 0;             // waves per tile of k_fcm_spread_tile: 0 = by the tiles' population (spread_waves), 2, 4
  bool tileGather = false;         // LDS-staged gather (k_fcm_gather_tile): measured SLOWER than the global gather, off
  bool accumulate = false;         // gather adds into the output (IBM::gather semantics; PSE far field)
  int zTileLog2 = 0;               // test / tuning hook: log2 of the fused z pass's node tile (0 = default)
  bool customFFT = true;           // power-of-two grids: the five-pass LDS FFT pipeline of fcm_fft.hpp instead of rocFFT + k_fcm_kspace
  PseGreens pse{0.f, 0.f, 0.f, 0.f, false};  // PSE far field: Hasimoto-split RPY greens function instead of 1/(eta k^2)
  rocfft_plan fwd = nullptr, inv = nullptr;
  rocfft_execution_info info = nullptr;
  size_t workBytes = 0;
  unsigned int seed2 = 0;  // the reference's `static uint seed2` (FCM_impl.cuh:517): per-handle here
  FcmPrep halfPrep{};      // a solve queued in two halves (fcm_displacements_impl): the stencil view of the first for the second
  int halfN = 0;
  bool halfPending = false;
  ~FCM() {
    if (slotFlagHost) (void)hipHostFree(slotFlagHost);
    if (fwd) rocfft_plan_destroy(fwd);
    if (inv) rocfft_plan_destroy(inv);
    if (info) rocfft_execution_info_destroy(info);
  }
};

static std::once_flag g_rocfft_once;
int rocfft_setup_once() {  // shared with poisson.hip
  std::call_once(g_rocfft_once, []() { (void)rocfft_setup(); });
  return 0;
}

// ---- spread / gather on the planar grids ---------------------------------------------------------------
template <bool SPREAD>
__global__ void __launch_bounds__(256) k_fcm_ibm(const float4 *__restrict__ pos, const float4 *__restrict__ force,
                                                  float *__restrict__ vout, float *__restrict__ g0, int N,
                                                  GridT<float> grid, int nxpad, size_t plane, size_t zstride,
                                                  IBMKernelDev kern, FastDiv dsx, FastDiv dsxy, bool accumulate) {
  const int lane = threadIdx.x & 63;
  const int id = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (id >= N) return;
  const float4 p4 = pos[id];
  const real3f pi{p4.x, p4.y, p4.z};
  const Stencil s = make_stencil(grid, kern, pi, false, lane);
  const int sx = s.support.x, sy = s.support.y, sz = s.support.z;
  const int nn = sx * sy * sz;
  float fx = 0.f, fy = 0.f, fz = 0.f;
  if (SPREAD) { const float4 f4 = force[id]; fx = f4.x; fy = f4.y; fz = f4.z; }
  float ax = 0.f, ay = 0.f, az = 0.f;
  const float dV = grid.cellVolume;
  float *g1 = g0 + plane, *g2 = g0 + 2 * plane;
  for (int i0 = 0; i0 < nn; i0 += 64) {
    const int i = i0 + lane;
    const bool in = i < nn;
    const uint iu = in ? (uint)i : 0u;
    const uint kk = dsxy.div(iu);
    const uint rem = iu - kk * (uint)(sx * sy);
    const uint jj = dsx.div(rem);
    const uint ii = rem - jj * (uint)sx;
    const float wx = stencil_weight(s, (int)ii);
    const float wy = stencil_weight(s, sx + (int)jj);
    const float wz = stencil_weight(s, sx + sy + (int)kk);
    if (!in) continue;
    // triply periodic: a single wrap is enough because support < cellDim (checked at create)
    int cx = s.celli.x + (int)ii - s.P.x, cy = s.celli.y + (int)jj - s.P.y, cz = s.celli.z + (int)kk - s.P.z;
    cx = cx < 0 ? cx + grid.cellDim.x : (cx >= grid.cellDim.x ? cx - grid.cellDim.x : cx);
    cy = cy < 0 ? cy + grid.cellDim.y : (cy >= grid.cellDim.y ? cy - grid.cellDim.y : cy);
    cz = cz < 0 ? cz + grid.cellDim.z : (cz >= grid.cellDim.z ? cz - grid.cellDim.z : cz);
    const size_t node = (size_t)cx + (size_t)nxpad * (size_t)cy + zstride * (size_t)cz;
    if (SPREAD) {
      unsafeAtomicAdd(&g0[node], fx * wx * wy * wz);
      unsafeAtomicAdd(&g1[node], fy * wx * wy * wz);
      unsafeAtomicAdd(&g2[node], fz * wx * wy * wz);
    } else {
      ax = fmaf(dV, g0[node] * wx * wy * wz, ax);
      ay = fmaf(dV, g1[node] * wx * wy * wz, ay);
      az = fmaf(dV, g2[node] * wx * wy * wz, az);
    }
  }
  if (!SPREAD) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      ax += __shfl_xor(ax, o, 64);
      ay += __shfl_xor(ay, o, 64);
      az += __shfl_xor(az, o, 64);
    }
    if (lane == 0) {
      float *o = vout + 3 * (size_t)id;  // IBM::gather ADDS (IBM.cu:227-233); the FCM solver starts from zero
      if (accumulate) { o[0] += ax; o[1] += ay; o[2] += az; } else { o[0] = ax; o[1] = ay; o[2] = az; }
    }
  }
}


// ---- tile-owned spreading (no global atomics, no grid memset) -------------------------------------------
// The grid is cut into T^3-node tiles; a workgroup OWNS one tile: it accumulates the contributions of every
// particle whose stencil overlaps the tile into an LDS copy of the tile (LDS float atomics, conflicts stay
// inside the CU) and then stores the tile with plain coalesced writes.  Particles are binned by tile with a
// small counting sort; the per-particle stencil origin and the 3*support 1-D weights are computed ONCE
// (k_fcm_prepare) and reused by every tile that the particle touches and by the gather.
constexpr int kTile = 8;
// A particle belongs to the tile of its stencil's ORIGIN (the lowest node, wrapped into the grid): the stencil then reaches from inside
// that tile towards + only, so a tile's candidates sit in the tiles at offsets -k .. 0 per axis, k = ceil((support - 1) / edge) —
// eight tiles for the supports and edges of the bench's sizes, where binning by the particle's own cell needed all 27 (660 candidate
// records per tile at C4 for ~105 accepted; now ~195).  rel: the origin relative to its tile, biased by 8, 7-bit fields.
// INVARIANT (fcm_tiles_usable, checked where useTiles is set): support <= 2 (edge - 1) <= cells, so an origin lies in (-n, n) and ONE wrap
// brings it into the grid; uammd_fcm_create admits support >= cells only on handles whose useTiles is false (the atomic spread).
UH_D int stencil_tile(int ox, int oy, int oz, int3 n, int3 tdim, int3 ntiles, int *rel) {
  const int wx = ox < 0 ? ox + n.x : ox, wy = oy < 0 ? oy + n.y : oy, wz = oz < 0 ? oz + n.z : oz;
  const int tx = wx / tdim.x, ty = wy / tdim.y, tz = wz / tdim.z;
  *rel = (wx - tx * tdim.x + 8) | (wy - ty * tdim.y + 8) << 7 | (wz - tz * tdim.z + 8) << 14;
  return tx + ntiles.x * (ty + ntiles.y * tz);
}


__global__ void __launch_bounds__(256) k_fcm_bin_count(const float4 *__restrict__ pos, int N, GridT<float> grid,
                                                        int3 ntiles, FcmPrep pr, int3 support) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= N) return;
  const float4 p4 = pos[id];
  const real3f pi{p4.x, p4.y, p4.z};
  const int3 celli = grid.getCell(pi);
  const int3 P = compute_support_shift(grid, pi, celli, support);
  int rel;
  const int t = stencil_tile(celli.x - P.x, celli.y - P.y, celli.z - P.z, grid.cellDim, pr.tdim, ntiles, &rel);
  pr.tileOf[id] = t;
  pr.rank[id] = atomicAdd(&pr.tileCount[t], 1);
}

// (Measured and not kept: binning that only COUNTS (atomics nobody waits for) with the slot taken from a per-tile cursor when the stencil is
// written: the binning pass is bound by the memory side's atomic rate, not by the round trip — 9.4 -> 9.1 us — and the second 1e5 atomics
// took the stencil kernel from 9.7 to 14.3 us: C4 step 0.1775 -> 0.182 ms.)
// integrateEulerMaruyamaD (BDHI_FCM.cu:67-92), pos += v dt, and k_fcm_bin_count for the position just written, in one launch: the two
// are a load -> store and a load -> atomic chain of the same length, each 5-9 us of latency as a kernel of its own
// (uammd_fcm_step_euler_maruyama: 4.7 + 8.9 -> 9.4 us, C4 step 0.2035 -> 0.2010 ms).
// (Measured and not kept: the update inside the interpolation kernel — one lane per wave loading, storing and taking its rank: 50 000
// single-lane memory instructions instead of 1 600 full ones — took the gather from 32 to 34 us with the update alone and to 127 us with
// the atomics.)
// bySlot: thread s takes the particle of slot s of the solve that has just run (pr.origin[s].w): neighbours in a wave are then particles of
// the same tile — still, after one step — and the wave counts them itself: one atomic per tile and wave (three or four) instead of 64.
// The binning pass is bound by the memory side's atomic rate (1e5 of them: ~9 us), not by its round trip.
__global__ void __launch_bounds__(256) k_fcm_update_bin(float4 *__restrict__ pos, const float *__restrict__ linearV, int N, float dt,
                                                         GridT<float> grid, int3 ntiles, FcmPrep pr, bool bin, bool bySlot, int3 support) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= N) return;
  const int id = bySlot ? pr.origin[s].w : s;
  float4 p = pos[id];
  p.x = fmaf(linearV[3 * id], dt, p.x);
  p.y = fmaf(linearV[3 * id + 1], dt, p.y);
  p.z = fmaf(linearV[3 * id + 2], dt, p.z);
  pos[id] = p;
  if (bin) {
    const real3f pi{p.x, p.y, p.z};
    const int3 celli = grid.getCell(pi);
    const int3 P = compute_support_shift(grid, pi, celli, support);
    int rel;
    const int t = stencil_tile(celli.x - P.x, celli.y - P.y, celli.z - P.z, grid.cellDim, pr.tdim, ntiles, &rel);
    pr.tileOf[id] = t;
    if (!bySlot) { pr.rank[id] = atomicAdd(&pr.tileCount[t], 1); return; }
    // up to eight tiles per wave are counted by the wave (the lanes of a tile: their number, a lane's place among them, the lowest
    // as the leader); what is left after eight rounds — a wave of strangers — takes its ranks one by one
    const int lane = threadIdx.x & 63;
    unsigned long long rem = __ballot(1);
    int cnt = 1, before = 0, lead = lane;
    for (int round = 0; round < 8 && rem; ++round) {
      const int l = __ffsll((unsigned long long)rem) - 1;
      const int t0 = __builtin_amdgcn_readlane(t, l);
      const unsigned long long m = __ballot(t == t0) & rem;
      if (t == t0 && ((rem >> lane) & 1ull)) {
        cnt = __popcll(m);
        before = __popcll(m & ((1ull << lane) - 1ull));
        lead = l;
      }
      rem &= ~m;
    }
    int got = 0;
    if (lane == lead) got = atomicAdd(&pr.tileCount[t], cnt);
    pr.rank[id] = __shfl(got, lead, 64) + before;
  }
}

__global__ void __launch_bounds__(1024) k_fcm_tile_scan(int *__restrict__ count, int ntiles, int *__restrict__ start) {
  // (the counters are left at ZERO for the next step's binning: no memset launch per step)
  // exclusive scan of the tile populations by one workgroup, 4096 consecutive tiles per round: a thread takes four consecutive tiles
  // (one 16-byte load and store, coalesced), the threads' sums are scanned with wave shuffles + one 16-entry LDS pass, the rounds are
  // chained by a running carry.  (History: a Hillis-Steele loop needed 20 barriers per 1024 tiles, 8.3 us for the 4096 tiles of C4;
  // per-thread runs of ntiles / 1024 consecutive tiles made every load a strided one, 51 us for the 32768 tiles of C5; one tile per
  // thread and round 23 us: a round costs a barrier's worth of time whatever it moves.)
  __shared__ int waveTotal[2][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int carry = 0, buf = 0;
  for (int base = 0; base < ntiles; base += 4096, buf ^= 1) {
    const int i = base + 4 * (int)threadIdx.x;
    int4 c = make_int4(0, 0, 0, 0);
    if (i + 3 < ntiles) {
      c = *reinterpret_cast<const int4 *>(count + i);
      *reinterpret_cast<int4 *>(count + i) = make_int4(0, 0, 0, 0);
    } else {
      if (i < ntiles) { c.x = count[i]; count[i] = 0; }
      if (i + 1 < ntiles) { c.y = count[i + 1]; count[i + 1] = 0; }
      if (i + 2 < ntiles) { c.z = count[i + 2]; count[i + 2] = 0; }
    }
    const int v = c.x + c.y + c.z + c.w;
    const int incl = (int)wave_inclusive_scan((uint)v);   // (six DPP additions: no ds_bpermute ladder)
    if (lane == 63) waveTotal[buf][wave] = incl;
    __syncthreads();   // (two buffers: the next round's writes cannot overtake this round's reads)
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int x = waveTotal[buf][w];
      if (w < wave) before += x;
      total += x;
    }
    const int s0 = carry + before + incl - v;
    const int4 o4 = make_int4(s0, s0 + c.x, s0 + c.x + c.y, s0 + c.x + c.y + c.z);
    if (i + 3 < ntiles) *reinterpret_cast<int4 *>(start + i) = o4;
    else {
      if (i < ntiles) start[i] = o4.x;
      if (i + 1 < ntiles) start[i + 1] = o4.y;
      if (i + 2 < ntiles) start[i + 2] = o4.z;
    }
    carry += total;
  }
  if (threadIdx.x == 0) start[ntiles] = carry;
}

// Stencil origin + the 3*support 1-D weights of every particle, written at the particle's tile-sorted slot.
// The window kind is a template parameter: with a run-time kind the compiler if-converts phi_axis' switch and evaluates
// EVERY window (exp, three sqrt, a division) for each of the 3*support weights — 4656 VALU instructions per particle,
// 22.9 us at C4 on 1.5 waves per SIMD.
#ifndef UAMMD_PREP_LANES
#define UAMMD_PREP_LANES 4
#endif
constexpr int kPrepLanes = UAMMD_PREP_LANES;
constexpr int kPrepLanesUpTo = 150000;   // (at 2e5 particles the chip is full with one lane each: 19.7 against 23.0 us with four)
template <int KIND, int LANES>
__global__ void __launch_bounds__(256) k_fcm_prepare(const float4 *__restrict__ pos, const float4 *__restrict__ force,
                                                      int N, GridT<float> grid, IBMKernelDev kern, FcmPrep pr) {
  kern.kind = KIND;  // constant-folds the switch
  // LANES threads per particle, each with every LANES-th weight: at 1e5 particles one thread per particle is 1.5
  // waves per SIMD walking 18 dependent exp chains; four lanes per particle put 6 waves on a SIMD with a quarter of the chain
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int id = gid / LANES, sub = gid % LANES;
  if (id >= N) return;
  const float4 p4 = pos[id];
  const real3f pi{p4.x, p4.y, p4.z};
  const int3 celli = grid.getCell(pi);
  const int3 P = compute_support_shift(grid, pi, celli, kern.support);
  const int ox = celli.x - P.x, oy = celli.y - P.y, oz = celli.z - P.z;
  const int slot = pr.tileStart[pr.tileOf[id]] + pr.rank[id];
  float *w = pr.weights + (size_t)pr.wstride * slot;
  const int sx = kern.support.x, sy = kern.support.y, sz = kern.support.z;
  for (int k = sub; k < sx + sy + sz; k += LANES) {
    float v;
    if (k < sx) v = phi_axis(kern, 0, grid.distanceToCellCenter(pi, make_int3(grid.pbc_x(ox + k), celli.y, celli.z)).x);
    else if (k < sx + sy) v = phi_axis(kern, 1, grid.distanceToCellCenter(pi, make_int3(celli.x, grid.pbc_y(oy + k - sx), celli.z)).y);
    else v = phi_axis(kern, 2, grid.distanceToCellCenter(pi, make_int3(celli.x, celli.y, grid.pbc_z(oz + k - sx - sy))).z);
    w[k] = v;
  }
  if (sub) return;
  pr.origin[slot] = make_int4(ox, oy, oz, id);
  // the spreading kernel's record: the force and, in .w, the stencil origin relative to the particle's OWN tile (each in
  // [0, edge), biased by 8 (stencil_tile), in 7-bit fields: x | y << 7 | z << 14): one 16-byte load per candidate, no image arithmetic (a neighbour
  // tile's frame is +-kTile away whatever the wrap); the spread adds a neighbour's shift and tests the three fields at once
  float4 fr = force ? force[id] : make_float4(0.f, 0.f, 0.f, 0.f);
  int rel;
  (void)stencil_tile(ox, oy, oz, grid.cellDim, pr.tdim, make_int3(1, 1, 1), &rel);
  fr.w = __int_as_float(rel);
  pr.force[slot] = fr;
}

// ---- the whole preparation of the NEXT solve inside the step's update kernel (round 5) --------------------------------------------------
// uammd_fcm_step_euler_maruyama used to end with k_fcm_update_bin and the next call began with k_fcm_tile_scan + k_fcm_prepare: three
// kernels in a dependent chain, each at its latency floor (8.6 + 4.8 + 9.8 us at C4, 13 % of the step).  This kernel is all three:
// LANES threads per particle; the particle of entry s of the last sorted solve (origin[s].w: the entries keep their order between
// sorted solves, so neighbours in a wave are neighbours in space) is moved, its tile found, its 3 * support weights and its stencil
// origin written at entry s (they do not depend on the rank), and its 64-bit record (entry, particle, origin relative to its tile) goes to
// slot rank of its tile's FIXED-CAPACITY range, the rank from one returning atomic per tile and wave (k_fcm_update_bin's ballots) —
// the weights are computed while that atomic is in flight.  No scan: the spread reads the 27 tile populations instead of 27 range
// bounds.  linearV == nullptr: no update (the preparation alone).
template <int KIND, int LANES>
__global__ void __launch_bounds__(256) k_fcm_step_prep(float4 *__restrict__ pos, const float *__restrict__ linearV, int N, float dt,
                                                        GridT<float> grid, IBMKernelDev kern, int3 ntiles, FcmPrep pr, int ovfCap) {
  kern.kind = KIND;  // constant-folds phi_axis' switch
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int s = gid / LANES, sub = gid % LANES;
  const bool live = s < N;
  const bool head = live && sub == 0;
  const int id = live ? pr.origin[s].w : 0;
  float4 p = pos[id];
  if (linearV) {
    p.x = fmaf(linearV[3 * id], dt, p.x);
    p.y = fmaf(linearV[3 * id + 1], dt, p.y);
    p.z = fmaf(linearV[3 * id + 2], dt, p.z);
    if (head) pos[id] = p;
  }
  const real3f pi{p.x, p.y, p.z};
  const int3 celli = grid.getCell(pi);
  const int3 P = compute_support_shift(grid, pi, celli, kern.support);
  const int ox = celli.x - P.x, oy = celli.y - P.y, oz = celli.z - P.z;
  int rel;
  const int t = stencil_tile(ox, oy, oz, grid.cellDim, pr.tdim, ntiles, &rel);
  // the rank: the heads of a wave that share a tile are counted by the wave (up to eight tiles per wave; strangers one by one)
  const int lane = threadIdx.x & 63;
  unsigned long long rem = __ballot(head);
  int cnt = 1, before = 0, lead = lane;
  for (int round = 0; round < 8 && rem; ++round) {
    const int l = __ffsll((unsigned long long)rem) - 1;
    const int t0 = __builtin_amdgcn_readlane(t, l);
    const unsigned long long m = __ballot(head && t == t0) & rem;
    if (head && t == t0 && ((rem >> lane) & 1ull)) {
      cnt = __popcll(m);
      before = __popcll(m & ((1ull << lane) - 1ull));
      lead = l;
    }
    rem &= ~m;
  }
  int got = 0;
  if (head && lane == lead) got = atomicAdd(&pr.slotCount[t], cnt);
  // how well the entries' order still follows the tiles (it is only an index permutation: a caller that re-lays its arrays out — a
  // sort, a compaction after migration — scrambles it without anybody noticing, and the gather's windows stop sharing cache lines:
  // 31 -> 77 us at C4).  Tile changes between consecutive entries of a wave: ~ntiles for a sorted order, ~N for a scrambled one;
  // the host compares (fcm_prepare_best) and goes through the sorted layout again when it is the latter.
  // (SAMPLED: one wave in 64 counts — thousands of atomics on one address serialise in the L2: with every wave counting the kernel
  // went from 13 to ~60 us — and the host scales the count back)
  if ((blockIdx.x & 15) == 0 && threadIdx.x < 64) {
    const int tPrev = __shfl_up(t, LANES, 64);
    const unsigned long long changes = __ballot(head && lane >= LANES && t != tPrev);
    if (lane == 0 && changes) atomicAdd(&pr.slotCount[ntiles.x * ntiles.y * ntiles.z + 1], (int)__popcll(changes));
  }
  // ... and while it is on its way: the stencil's weights
  const int sx = kern.support.x, sy = kern.support.y, sz = kern.support.z;
  if (live) {
    float *w = pr.weights + (size_t)pr.wstride * s;
    for (int k = sub; k < sx + sy + sz; k += LANES) {
      float v;
      if (k < sx) v = phi_axis(kern, 0, grid.distanceToCellCenter(pi, make_int3(grid.pbc_x(ox + k), celli.y, celli.z)).x);
      else if (k < sx + sy) v = phi_axis(kern, 1, grid.distanceToCellCenter(pi, make_int3(celli.x, grid.pbc_y(oy + k - sx), celli.z)).y);
      else v = phi_axis(kern, 2, grid.distanceToCellCenter(pi, make_int3(celli.x, celli.y, grid.pbc_z(oz + k - sx - sy))).z);
      w[k] = v;
    }
  }
  if (head) pr.origin[s] = make_int4(ox, oy, oz, id);
  const int rank = __shfl(got, lead, 64) + before;
  if (!head) return;
  const int nt = ntiles.x * ntiles.y * ntiles.z;
  int slot;
  if (rank < pr.cap) slot = t * pr.cap + rank;
  else {
    const int k = min(atomicAdd(&pr.slotCount[nt], 1), ovfCap - 1);  // (k < N <= ovfCap: one record per particle at most)
    slot = nt * pr.cap + k;
    pr.ovfTile[k] = t;
  }
  pr.rec[slot] = (unsigned long long)(uint)s | (unsigned long long)(uint)id << 21 | (unsigned long long)(uint)rel << 42;
}

// One workgroup (4 waves) per tile, three phases per chunk of candidates so that a tile costs TWO global round trips instead
// of one per particle (round 1 pulled each particle's weights inside the spreading loop: 26 dependent ~0.7 us loads per wave,
// 104 us per call at C4):
//   A  all 256 threads test the particles of the 27 surrounding tiles (one candidate each per round, every origin load in
//      flight together) and the accepted ones are appended IN CANDIDATE ORDER to a list in LDS (ballot + wave offsets: the
//      summation order, hence the result, is the same on every run);
//   B  the listed particles' 1-D weights are copied to LDS, all loads in flight together;
//   C  the waves take listed particles round-robin and spread them into their PRIVATE copy of the tile, held in registers.
//      (LDS float atomics retire ~1 lane per clock on gfx950: a shared LDS tile kept the LDS pipe 90 % busy and the kernel at
//      400 us; private LDS copies with read-modify-write were VALU-issue bound at 100 us.)  The four private copies are summed
//      through LDS when the tile is stored.
// LDS budget of phase B in words, chosen per call (spread_weight_words): 24 words per listed particle (kSpWT), 6144 = 256 listed
// particles (a C4 tile lists ~105), 3072 = 128 where tiles are sparse (a C5 tile lists ~26); with the list either fits five
// workgroups per CU beside the 24.6 KB floor of the final tile sum.
#ifndef UAMMD_SP_ABLATE   // (timing builds only, tools/variants_fcm.sh: 1 no weights copy, 2 no matrix phase, 4 no tile sum / store, 8 nothing accepted)
#define UAMMD_SP_ABLATE 0
#endif
#ifndef UAMMD_SP_WORDS   // (A/B builds: tools/variants_fcm.sh — 6144 words / 4 candidates per thread: within 1 % of these at C4 and 108^3)
#define UAMMD_SP_WORDS 6144
#endif
#ifndef UAMMD_SP_PER_THREAD
#define UAMMD_SP_PER_THREAD 2   // (round 5, ~195 candidates per C4 tile since the tiles are the stencil origins': 1 / 2 / 3 -> solve 0.1588 / 0.1595 / 0.1609 ms)
#endif
constexpr int kSpWeightWordsMax = UAMMD_SP_WORDS;
constexpr int kSpWT = 3 * kTile;      // LDS words per listed particle: its weights at the tile's 8 nodes along x, y, z
// The kernel's layouts are sized by a compile-time tile edge E: 8 (every grid whose axes divide by 4..8) or 9 — round 6, for grids like the
// PSE far field's 108 = 12 x 9, which otherwise takes 6-node tiles: 5832 tiles listing 137 particles each (799 k particle-tile pairs, 32 %
// of each matrix product used) against 1728 tiles listing 268 (463 k pairs; 27 of 32 rows and 81 of 96 columns in THREE products per step).
constexpr int kTileMax = 9;
constexpr int kSpPerThread = UAMMD_SP_PER_THREAD;       // candidates per thread and round of phase A (768 per round; a C4 tile sees ~660)
struct SpEntry {
  int o;  // stencil origin in the tile's frame (may be negative), biased by 24 and packed in 7-bit fields: ox | oy << 7 | oz << 14
  int slot;
  float fx, fy, fz;
};
// dynamic LDS: float wts[weightWords + 32] | SpEntry list[64 waves + 1]   (a list never holds more entries than the workgroup has threads;
// +1: phase C reads 8 words per 5-word entry); the waves' private tiles of the final sum alias the same block.  Two-wave tiles with
// 129 entries instead of 257: 15.6 KB per workgroup, ten of them on a CU — as many as the registers allow — instead of nine.
static size_t spread_lds_bytes(int weightWords, int waves = 4, int edge = kTile) {
  const size_t a = sizeof(float) * (size_t)(weightWords + 32) + sizeof(SpEntry) * (64 * waves + 1), b = sizeof(float) * waves * 3 * edge * edge * edge;
  return a > b ? a : b;
}
// two waves per tile where a tile lists few particles (measured at C5, ~26 listed: see DESIGN 5.4), four otherwise
static int spread_waves(int N, int3 ntiles, int3 support, int3 tdim, int option) {
  if (option == 2 || option == 4) return option;
  const double perTile = (double)N / ((double)ntiles.x * ntiles.y * ntiles.z);
  const double listed = perTile * (1.0 + (support.x - 1) / (double)tdim.x) * (1.0 + (support.y - 1) / (double)tdim.y) * (1.0 + (support.z - 1) / (double)tdim.z);
  return listed > 48.0 ? 4 : 2;
}
static int spread_edge(int3 tdim) { return (tdim.x > kTile || tdim.y > kTile || tdim.z > kTile) ? kTileMax : kTile; }
static int spread_weight_words(int N, int3 ntiles, int3 support, int3 tdim) {
  const double perTile = (double)N / ((double)ntiles.x * ntiles.y * ntiles.z);
  const double listed = perTile * (1.0 + (support.x - 1) / (double)tdim.x) * (1.0 + (support.y - 1) / (double)tdim.y) * (1.0 + (support.z - 1) / (double)tdim.z);
  const int full = spread_edge(tdim) == kTileMax ? 256 * 3 * kTileMax : kSpWeightWordsMax;   // (256 listed particles either way)
  return listed > 64.0 ? full : full / 2;
}
// W = waves per workgroup: 4, or 2 where the tiles are sparse (spread_waves) — a tile then costs the same chain of round trips for a
// few matrix steps: twice the tiles in flight for the same waves.  Measured at C5 (256^3, 6 particles per tile, ~26 listed): 179 / 168 /
// 224 us with 4 / 2 / 1 waves per tile; at C4 (24 per tile, ~105 listed) 62 / 98 with 4 / 2.
#ifdef UAMMD_SPREAD_TIMELINE   // diagnostic build (tools/variants_fcm.sh, tools/spread_timeline.py): where a tile's lifetime goes, 100 MHz ticks
__device__ unsigned long long g_spread_tl[8];
#define SP_STAMP(k) do { if (threadIdx.x == 0) atomicAdd(&g_spread_tl[k], (unsigned long long)(__builtin_amdgcn_s_memrealtime() - tl0)); } while (0)
#else
#define SP_STAMP(k) do {} while (0)
#endif
// SLOTS: the records come from fixed-capacity tile slots written by k_fcm_step_prep (FcmPrep::cap > 0) instead of the compact,
// scanned layout: 27 populations instead of 27 range bounds, a 28th range for the overflow records (usually empty), the forces of the
// LISTED particles fetched by particle index in phase B, and the other parity's counters handed back zeroed.
// SPEC (slot layout, two waves per tile, eight source tiles: the sparse case, e.g. C5 with 6 particles per tile): the first kSpecSlots
// slots of each of the eight source tiles are requested TOGETHER with the tiles' populations — the slots sit at fixed addresses, so the
// records need not wait for the counts — and a tile's life has two dependent round trips in front of its matrix steps instead of three.
// Round 5 measured the same idea at C4 (47 slots of 27 tiles, 660 real records among 1269 requested) and dropped it: 45.6 -> 48.4 us; where
// a tile lists ~26 particles the requested 128 records cost nothing and the round trip is a quarter of the tile's life.
template <int W, bool SLOTS = false, int E = kTile, bool SPEC = false>
__global__ void __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(E == kTile ? 5 : 4, 8)))
k_fcm_spread_tile(float *__restrict__ g0, int3 n, int nxpad, size_t plane, size_t zstride, int3 support, int3 ntiles,
                  FcmPrep pr, int weightWords) {
  static_assert(!SPEC || SLOTS, "the speculative first round belongs to the slot layout");
  constexpr int kSpecSlots = 8 * W;   // 8 source tiles x 8 W slots = one record per thread: 16 slots of a sparse tile (W = 2), 32 of a dense one
  constexpr int T3 = E * E * E;
  constexpr int WT = 3 * E;                 // LDS words per listed particle: its weights at the tile's E nodes along x, y, z
  constexpr int NM = (E * E + 31) / 32;     // matrix products per step: the tile's E^2 (x, y) columns in blocks of 32
  // the list + weights of phases A-C and the four private tiles of the final sum are never live together: one LDS block
  extern __shared__ __attribute__((aligned(16))) char smem[];
  struct { float *wts; SpEntry *list; } sh{reinterpret_cast<float *>(smem), reinterpret_cast<SpEntry *>(smem + sizeof(float) * (size_t)(weightWords + 32))};
  float *acc = reinterpret_cast<float *>(smem);
  constexpr int kRanges = SLOTS ? 28 : 27;  // (slots: + the overflow records)
  constexpr int kThreads = 64 * W;
  // (Measured and not kept, round 5: a SPECULATIVE first round in the slot layout — the first 47 slots of each of the 27 tiles requested
  // together with the tiles' populations, since the slots sit at fixed addresses: one round trip instead of two.  1269 records
  // requested for ~660 real ones, five candidates per thread instead of three: the kernel went from 45.6 to 48.4 us at C4.)
  constexpr int kU = kSpPerThread;
  __shared__ int rPrefix[kRanges + 1];
  __shared__ int2 rInfo[kRanges];   // {first slot of the range minus its offset in the flat candidate sequence, the packed shift of its tile}
  __shared__ int waveCnt[4 * kU];
  __shared__ unsigned char owner[kThreads * kU];
  __builtin_amdgcn_s_setprio(3);  // phases that load go ahead of the phase that computes (five workgroups share a CU)
#ifdef UAMMD_SPREAD_TIMELINE
  const unsigned long long tl0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  f32x16 accs[NM];  // this wave's private copy of the tile: MFMA accumulators (layout at the store below)
#pragma unroll
  for (int m = 0; m < NM; ++m) accs[m] = f32x16{0.f};
  const int half = lane >> 5, l32 = lane & 31;
  const int aKz = l32 / 3, aC = l32 - 3 * (l32 / 3);  // A operand row n = l32 = 3 kz + c
  const bool aValid = l32 < 3 * E;
  // B operand: product m holds the columns xy = l32 + 32 m, x = xy mod E, y = xy div E (E = 8: x = l32 & 7, y = (l32 >> 3) + 4 m); a
  // column beyond E^2 (the tail of the last product when E = 9) carries a zero weight
  int bAtX[NM], bAtY[NM];
  bool bValid[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const int col = l32 + 32 * m;
    bValid[m] = col < E * E;
    bAtX[m] = bValid[m] ? col % E : 0;
    bAtY[m] = E + (bValid[m] ? col / E : 0);
  }
  const int tile = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int tx = tile % ntiles.x, ty = (tile / ntiles.x) % ntiles.y, tz = tile / (ntiles.x * ntiles.y);
  const int3 td = pr.tdim;
  const int x0 = tx * td.x, y0 = ty * td.y, z0 = tz * td.z;
  const int sx = support.x, sy = support.y, sz = support.z;
  const int wstride = pr.wstride;
  const int capEntries = min(kThreads, weightWords / WT);
  const int numTiles = ntiles.x * ntiles.y * ntiles.z;
  int myTile = numTiles, myShift = 0;  // threads < kRanges: their range's tile and packed shift
  __shared__ int rCount[SPEC ? 8 : 1];
  unsigned long long specRec = 0ull;
  int specShift = 0;
  if (SPEC) {   // thread (k, j) = (source tile, slot): its record is on its way while the populations are read
    const int k = threadIdx.x / kSpecSlots, j = threadIdx.x & (kSpecSlots - 1);   // 8 x 8 W = the workgroup's threads
    const int dx = -(k & 1), dy = -((k >> 1) & 1), dz = -(k >> 2);          // (ox1 = oy1 = oz1 = 2: the launch chose SPEC for that)
    int ux = tx + dx, uy = ty + dy, uz = tz + dz;
    if (ux < 0) ux += ntiles.x;
    if (uy < 0) uy += ntiles.y;
    if (uz < 0) uz += ntiles.z;
    specShift = (td.x * dx + 16) | (td.y * dy + 16) << 7 | (td.z * dz + 16) << 14;
    if (j < pr.cap) specRec = pr.rec[(size_t)(ux + ntiles.x * (uy + ntiles.y * uz)) * pr.cap + j];
  }
  if (threadIdx.x < kRanges) {
    const int nb = threadIdx.x;
    // the tiles whose particles can reach this one: offsets -k .. 0 per axis (stencil_tile)
    const int ox1 = (sx - 2 + td.x) / td.x + 1, oy1 = (sy - 2 + td.y) / td.y + 1, oz1 = (sz - 2 + td.z) / td.z + 1;   // k + 1 each, <= 3
    const bool used = nb < ox1 * oy1 * oz1;
    const int dx = used ? -(nb % ox1) : 0, dy = used ? -((nb / ox1) % oy1) : 0, dz = used ? -(nb / (ox1 * oy1)) : 0;
    int ux = tx + dx, uy = ty + dy, uz = tz + dz;
    if (ux < 0) ux += ntiles.x; else if (ux >= ntiles.x) ux -= ntiles.x;
    if (uy < 0) uy += ntiles.y; else if (uy >= ntiles.y) uy -= ntiles.y;
    if (uz < 0) uz += ntiles.z; else if (uz >= ntiles.z) uz -= ntiles.z;
    if (nb < 27) myTile = ux + ntiles.x * (uy + ntiles.y * uz);
    // a record's origin is relative to its own tile (in [0, edge), biased by 8): in this tile's frame that is + one tile edge per tile
    // step (steps 0, -1, -2); with + 16 more every field of the shift is >= 0 and every field of record + shift is (origin in this
    // tile's frame) + 24, in [8, 31]: no carry or borrow between the 7-bit fields
    myShift = (td.x * dx + 16) | (td.y * dy + 16) << 7 | (td.z * dz + 16) << 14;
    int s, e;
    if (SLOTS) {
      const int c = pr.slotCount[myTile];
      s = myTile * pr.cap;
      e = s + (nb < 27 ? min(c, pr.cap) : c);  // (range 27: every overflow record, whatever its tile)
      if (SPEC && nb < 8) {   // the speculative round takes the range's first slots: what is left of it starts behind them
        const int took = min(min(c, pr.cap), kSpecSlots);
        rCount[nb] = took;
        s += took;
      }
    } else {
      s = pr.tileStart[myTile];
      e = pr.tileStart[myTile + 1];
    }
    if (nb < 27 && !used) e = s;   // (fewer than 27 source tiles)
    // inclusive scan of the range lengths inside wave 0
    const int incl = (int)wave_inclusive_scan((uint)(e - s));   // (DPP additions: the active lanes sit in rows 0 and 1)
    rPrefix[nb + 1] = incl;
    if (nb == 0) rPrefix[0] = 0;
    if (SLOTS && nb == 27) {  // this parity's counters have been read by every tile that needs them once all tiles ran: the other parity's
      pr.slotCountNext[tile] = 0;  // are zeroed here for the update kernel that follows this solve
      if (tile == 0) {
        pr.slotCountNext[numTiles] = 0;
        pr.slotCountNext[numTiles + 1] = 0;
        if (e - s > 2048) pr.slotFlag[0] = 1;  // (tell the host: the overflow list is long enough to cost time)
        pr.slotFlag[1] = pr.slotCount[numTiles + 1];  // ... and how scrambled the entries' order is
      }
    }
    rInfo[nb] = make_int2(s - (incl - (e - s)), myShift);
  }
  __syncthreads();
  SP_STAMP(0);  // ranges known
  const int total = rPrefix[kRanges];
  int listCount = 0;  // uniform over the workgroup

  // (always_inline: out of line, the lambda's captures — the accumulators, pr — live in scratch memory and the kernel is 8x slower;
  // it was inlined by size alone until round 5 added to it;
  // sx, sy, sz by VALUE: by reference they are three adjacent pointers of the closure, `axis == 0 ? sx : (axis == 1 ? sy : sz)` becomes a
  // load at a variable offset into the closure, and the closure — with every captured variable behind it — stays in scratch memory)
  auto spread_list = [&, sx, sy, sz, wstride](int count) __attribute__((always_inline)) {
    // (every call site is workgroup-uniform.)  The list was appended to by all four waves, possibly in an earlier sub-round of the
    // loop below with no barrier since: phase B reads every entry's slot, so the writes must have landed first.
    __syncthreads();
    // phase B
    // (Measured and not kept: the listed particles' weights COMPUTED here from their positions — one 16-byte load per listed
    // particle instead of 13 loads per thread in two dependent rounds, which tools/spread_timeline.py shows as 35 % of a tile's
    // lifetime; one (particle, axis) per thread, prepare's own expression, identical results: 57 -> 69 us.  The kernel is not waiting
    // for those loads as much as it is short of issue slots: 1890 exponentials per tile cost more than the round trips they replace.)
    // The weights go to LDS in the TILE's frame: 3 x 8 words per listed particle, word (axis, t) = the particle's weight at the tile's
    // node t along that axis (w[t - o], or 0 where its stencil does not reach).  A lane of the matrix phase then reads its operand
    // parts at CONSTANT offsets (its column's x and y, its row's plane): no origin to unpack, no index masks, range tests or selects
    // per step — the kernel is short of vector issue slots (profiles/r04_pmc_fcm_spread.txt), and that phase was ~40 instructions
    // per matrix step.  Thread r = tid & 31 < 24 copies word r of particles (tid >> 5) + (kThreads / 32) j: axis, t and the word's
    // place once per thread; all of a thread's loads in flight together (staged).
    float4 myForce = make_float4(0.f, 0.f, 0.f, 0.f);
    if (SLOTS && (int)threadIdx.x < count) myForce = pr.forceById[__float_as_int(sh.list[threadIdx.x].fx)];  // (in flight with the weights)
    {
      const int r = threadIdx.x & 31, axis = r / E, t = r - E * (r / E);
      const int sa = axis == 0 ? sx : (axis == 1 ? sy : sz), aoff = axis == 0 ? 0 : (axis == 1 ? sx : sx + sy);
      const int pp0 = threadIdx.x >> 5;
#ifdef UAMMD_SP_BRANCHY   // (round 5's form, tools/variants_fcm.sh: every staged element behind two branches — in range? does the stencil reach?)
      if (r < WT && !(UAMMD_SP_ABLATE & 1))
        staged_copy<8, float>(0, (count - pp0 + kThreads / 32 - 1) / (kThreads / 32), 1,
            [&](int j) {
              const SpEntry &en = sh.list[pp0 + (kThreads / 32) * j];
              const int i = t - (((en.o >> (7 * axis)) & 127) - 24);
              return (unsigned)i < (unsigned)sa ? pr.weights[(size_t)wstride * en.slot + aoff + i] : 0.0f;
            },
            [&](int j, float v) { sh.wts[(pp0 + (kThreads / 32) * j) * WT + r] = v; });
#else
      // Eight listed particles per round, their words requested together, WITHOUT branches: an element past the thread's last one repeats
      // the last (the same word stored twice), a word the stencil does not reach loads the stencil's first weight and keeps 0 — the
      // compiler cannot speculate a load, so the guarded form was two exec-mask branches per element.  (tools/time_fcm.py, 1000 solves per
      // figure, three interleaved runs: C4 0.1554 / 0.1559 / 0.1556 -> 0.1548 / 0.1546 / 0.1546 ms, 108^3 0.1679 / 0.1673 / 0.1680 ->
      // 0.1658 / 0.1656 / 0.1658, C5 within its run-to-run spread.)
      const int mine = (count - pp0 + kThreads / 32 - 1) / (kThreads / 32);
      if (r < WT && mine > 0 && !(UAMMD_SP_ABLATE & 1)) {
        for (int j0 = 0; j0 < mine; j0 += 8) {
          float v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int j = min(j0 + u, mine - 1);
            const SpEntry &en = sh.list[pp0 + (kThreads / 32) * j];
            const int i = t - (((en.o >> (7 * axis)) & 127) - 24);
            const bool reach = (unsigned)i < (unsigned)sa;
            const float w = pr.weights[(size_t)wstride * en.slot + aoff + (reach ? i : 0)];
            v[u] = reach ? w : 0.0f;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) sh.wts[(pp0 + (kThreads / 32) * min(j0 + u, mine - 1)) * WT + r] = v[u];
        }
      }
#endif
    }
    if (SLOTS && (int)threadIdx.x < count) { SpEntry &en = sh.list[threadIdx.x]; en.fx = myForce.x; en.fy = myForce.y; en.fz = myForce.z; }
    __syncthreads();
    SP_STAMP(2);  // weights in LDS
    // phase C on the matrix pipe.  For one tile the spreading is a product: G[n][xy] += sum_p A[n][p] B[p][xy] with
    // n = 3 kz + c (8 planes x 3 components = 24 of 32 rows), xy the 64 columns of the tile, A[n][p] = wz_p[kz] f_p[c] and
    // B[p][xy] = wx_p[x] wy_p[y]: two v_mfma_f32_32x32x2_f32 (columns 0..31 and 32..63) take two particles per step, lanes 0..31
    // building the operands of one particle and lanes 32..63 those of the other.  fp32 in, fp32 accumulate.
    // (History: lanes over the nodes of the stencil-tile intersection with LDS read-modify-write, 104 us per call at C4; lane =
    // column with 24 register accumulators on the VALU, 83 us of which this phase was 41 — measured by switching the phases
    // off one at a time: A 19, B 12, C 41, prologue + reduction + store 12; with the MFMA form C is ~22 and the call 65 us.)
    const int mineCount = (count - wave + W - 1) / W;  // entries wave, wave + W, ... of the list
    __builtin_amdgcn_s_setprio(0);  // (the arithmetic phase yields to the workgroups that are issuing loads: see the kernel's top)
    const int zAt = 2 * E + (aValid ? aKz : 0);  // (rows 3 E..31 of the A operand carry no plane: force 0, any word)
    // (Measured and not kept, round 5: the list put in the order [y < 4 only | both halves | y >= 4 only] — 8 of 13 possible y origins
    // leave one half of the tile's columns, i.e. one of the two products, untouched — and a step issuing only the products its two
    // particles need: 31 % fewer matrix instructions, 45.4 -> 50.7 us.  Switching the phases off one at a time (UAMMD_SP_ABLATE) prices
    // this loop at 18.7 us, the weights' copy at 8.6, the tile sum and store at 5.3, ranges + candidate tests at 9 — but the matrix pipe
    // itself is busy for ~14 of the 18.7 only on paper: the partition's two barriers and the branches around the products cost more.)
    for (int j = 0; j < ((UAMMD_SP_ABLATE & 2) ? 0 : mineCount); j += 2) {
      const int idx = j + half;
      const bool real = idx < mineCount;  // an odd tail re-reads the wave's first entry with zero force
      const int e = wave + W * (real ? idx : 0);
      const float fc = (real && aValid) ? reinterpret_cast<const float *>(&sh.list[e].fx)[aC] : 0.0f;
      const float *wt = sh.wts + e * WT;
      const float av = wt[zAt] * fc;
      float b[NM];
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const float w = wt[bAtX[m]] * wt[bAtY[m]];
        b[m] = (E == kTile || bValid[m]) ? w : 0.0f;
      }
#pragma unroll
      for (int m = 0; m < NM; ++m) accs[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[m], accs[m], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(3);
    __syncthreads();
    SP_STAMP(3);  // matrix phase done
  };

  // a stencil overlaps the tile iff -support < origin < tile edge on every axis, i.e. 25 - support <= field <= 23 + edge: the three
  // fields tested at once through their guard bits (bit 6 of a field survives `(field | guard) - lo` iff field >= lo)
  constexpr int kGuard = 0x40 | 0x40 << 7 | 0x40 << 14;
  const int lo3 = (25 - sx) | (25 - sy) << 7 | (25 - sz) << 14;
  const int hi3 = ((23 + td.x) | (23 + td.y) << 7 | (23 + td.z) << 14) | kGuard;
  // accept / ballot / append for one round of kU candidates per thread: org = record origin + tile shift (or garbage where !live),
  // key = where the candidate's weights are, fx.. = its force (compact) / its particle in fx (slots).  Sub-rounds in candidate order: the
  // list order is the summation order.
  auto take_round = [&](const bool (&live)[kU], const int (&org)[kU], const int (&key)[kU], const float (&fx)[kU], const float (&fy)[kU],
                        const float (&fz)[kU]) __attribute__((always_inline)) {
    bool accept[kU];
    unsigned long long m[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      accept[u] = !(UAMMD_SP_ABLATE & 8) && live[u] && ((((org[u] | kGuard) - lo3) & (hi3 - org[u])) & kGuard) == kGuard;
      m[u] = __ballot(accept[u]);
      if (lane == 0) waveCnt[4 * u + wave] = __popcll(m[u]);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int c0w = waveCnt[4 * u], c1w = W > 1 ? waveCnt[4 * u + 1] : 0, c2w = W > 2 ? waveCnt[4 * u + 2] : 0, c3w = W > 2 ? waveCnt[4 * u + 3] : 0;
      const int roundCount = c0w + c1w + c2w + c3w;
      if (listCount + roundCount > capEntries) {  // uniform: spread what is listed, then start a new list
        spread_list(listCount);
        listCount = 0;
      }
      if (accept[u]) {
        const int before = (wave > 0 ? c0w : 0) + (wave > 1 ? c1w : 0) + (wave > 2 ? c2w : 0);
        SpEntry en;
        en.o = org[u];
        en.slot = key[u];
        en.fx = fx[u]; en.fy = fy[u]; en.fz = fz[u];
        sh.list[listCount + before + __popcll(m[u] & ((1ull << lane) - 1ull))] = en;
      }
      listCount += roundCount;
    }
    __syncthreads();
  };
  if (SPEC) {   // the records requested with the populations: slot j of source tile k is real iff j < the tile's population
    bool live[kU];
    int org[kU], key[kU];
    float fx[kU], fy[kU], fz[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) { live[u] = false; org[u] = 0; key[u] = 0; fx[u] = fy[u] = fz[u] = 0.f; }
    live[0] = (int)(threadIdx.x & (kSpecSlots - 1)) < rCount[threadIdx.x / kSpecSlots];
    org[0] = specShift + (int)(specRec >> 42);
    key[0] = (int)(specRec & 0x1fffffull);
    fx[0] = __int_as_float((int)((specRec >> 21) & 0x1fffffull));
    take_round(live, org, key, fx, fy, fz);
  }
  const int perRound = min(capEntries, kThreads) * kU;
  for (int c0 = 0; c0 < total; c0 += perRound) {
    // phase A: kU candidates per thread, their records in flight together.  owner[] maps a candidate of this round to its range
    // (written by eight threads per range) instead of a binary search per candidate.
    {
      constexpr int kPerRange = W == 4 ? 8 : 4;  // threads that fill one range's part of owner[]
      const int nb = threadIdx.x / kPerRange;
      if (nb < kRanges) {
        const int lo = max(rPrefix[nb], c0), hi = min(rPrefix[nb + 1], c0 + perRound);
        for (int c = lo + (int)(threadIdx.x % kPerRange); c < hi; c += kPerRange) owner[c - c0] = (unsigned char)nb;
      }
    }
    __syncthreads();
    bool live[kU];
    int org[kU], key[kU];   // org: record + shift = the stencil origin in this tile's frame, + 24, in 7-bit fields
    float fx[kU], fy[kU], fz[kU];
    if (SLOTS) {
      unsigned long long rec[kU];
      int tc[kU];
      bool ovf[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int c = c0 + u * capEntries + (int)threadIdx.x;
        live[u] = (int)threadIdx.x < capEntries && c < total;
        const int own = live[u] ? owner[c - c0] : 0;
        const int2 ri = rInfo[own];
        ovf[u] = live[u] && own == 27;
        key[u] = live[u] ? ri.x + c : 0;
        org[u] = ri.y;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        rec[u] = live[u] ? pr.rec[key[u]] : 0ull;
        tc[u] = ovf[u] ? pr.ovfTile[key[u] - numTiles * pr.cap] : 0;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (ovf[u]) {  // an overflow record: its tile comes with it, not with a range; the shift from the tiles' distance
          int ddx = tc[u] % ntiles.x - tx, ddy = (tc[u] / ntiles.x) % ntiles.y - ty, ddz = tc[u] / (ntiles.x * ntiles.y) - tz;
          // INVARIANT (fcm_tiles_usable): >= 3 tiles per axis, so a source tile appears at exactly ONE of the steps 0, -1, -2 and the
          // single wrap below finds it; with <= 2 tiles per axis a tile would sit at two steps and the ranged path's enumeration would be
          // needed — such grids never get useTiles
          ddx -= ddx > 0 ? ntiles.x : 0;   // (a source tile sits at steps 0, -1, -2 of this one, around the box)
          ddy -= ddy > 0 ? ntiles.y : 0;
          ddz -= ddz > 0 ? ntiles.z : 0;
          if (ddx < -2 || ddy < -2 || ddz < -2) { live[u] = false; ddx = ddy = ddz = 0; }
          org[u] = (td.x * ddx + 16) | (td.y * ddy + 16) << 7 | (td.z * ddz + 16) << 14;
        }
        org[u] += (int)(rec[u] >> 42);
        key[u] = (int)(rec[u] & 0x1fffffull);
        fx[u] = __int_as_float((int)((rec[u] >> 21) & 0x1fffffull));
        fy[u] = fz[u] = 0.f;
      }
    } else {
      float4 frc[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int c = c0 + u * capEntries + (int)threadIdx.x;
        live[u] = (int)threadIdx.x < capEntries && c < total;
        const int2 ri = rInfo[live[u] ? owner[c - c0] : 0];
        key[u] = live[u] ? ri.x + c : 0;
        org[u] = ri.y;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) frc[u] = live[u] ? pr.force[key[u]] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        org[u] += __float_as_int(frc[u].w);
        fx[u] = frc[u].x; fy[u] = frc[u].y; fz[u] = frc[u].z;
      }
    }
    take_round(live, org, key, fx, fy, fz);
  }
  SP_STAMP(1);  // candidates tested, list complete
  if (listCount > 0) spread_list(listCount);
  if (UAMMD_SP_ABLATE & 4) { if (accs[0][0] + accs[NM - 1][0] == 123.456f) g0[0] = 1.f; return; }
  {
    // accumulator v of lane l is row n = 8 (v / 4) + 4 (l / 32) + v % 4, column l % 32 of its product; rows 3 E..31 are unused
    float *mine = acc + wave * 3 * T3;  // node (x, y, z) of the tile = xy + E^2 z
#pragma unroll
    for (int v = 0; v < (E == kTile ? 12 : 16); ++v) {
      const int nrow = 8 * (v / 4) + 4 * half + (v % 4);
      if (E != kTile && nrow >= 3 * E) continue;
      const int kz = nrow / 3, c = nrow - 3 * kz;
#pragma unroll
      for (int m = 0; m < NM; ++m)
        if (E == kTile || bValid[m]) mine[c * T3 + E * E * kz + 32 * m + l32] = accs[m][v];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < T3; i += kThreads) {
    const int lx = i % E, ly = (i / E) % E, lz = i / (E * E);
    if (lx >= td.x || ly >= td.y || lz >= td.z) continue;  // (a tile edge shorter than the layout's E: those rows belong to a neighbour)
    const size_t node = (size_t)(x0 + lx) + (size_t)nxpad * (size_t)(y0 + ly) + zstride * (size_t)(z0 + lz);
    if (W == 4) {
      g0[node] = (acc[i] + acc[3 * T3 + i]) + (acc[6 * T3 + i] + acc[9 * T3 + i]);
      g0[plane + node] = (acc[T3 + i] + acc[4 * T3 + i]) + (acc[7 * T3 + i] + acc[10 * T3 + i]);
      g0[2 * plane + node] = (acc[2 * T3 + i] + acc[5 * T3 + i]) + (acc[8 * T3 + i] + acc[11 * T3 + i]);
    } else {
      g0[node] = acc[i] + acc[3 * T3 + i];
      g0[plane + node] = acc[T3 + i] + acc[4 * T3 + i];
      g0[2 * plane + node] = acc[2 * T3 + i] + acc[5 * T3 + i];
    }
  }
  SP_STAMP(4);  // stored
#ifdef UAMMD_SPREAD_TIMELINE
  if (threadIdx.x == 0) atomicAdd(&g_spread_tl[7], 1ull);
#endif
}

// Gather with the precomputed origin/weights: one wave per tile-sorted slot (neighbouring waves touch the same
// nodes), result written at the particle's original index.
__global__ void __launch_bounds__(256) k_fcm_gather_prep(float *__restrict__ vout, const float *__restrict__ g0, int N,
                                                          int3 n, int nxpad, size_t plane, size_t zstride, int3 support,
                                                          float dV, FastDiv dsx, FastDiv dsxy, FcmPrep pr, bool accumulate) {
  const int lane = threadIdx.x & 63;
  const int slot = (int)xcd_contiguous_block(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
  if (slot >= N) return;
  const int4 o = pr.origin[slot];
  const int sx = support.x, sy = support.y, sz = support.z;
  const float wl = lane < sx + sy + sz ? pr.weights[(size_t)pr.wstride * slot + lane] : 0.0f;
  const int nn = sx * sy * sz;
  float ax = 0.f, ay = 0.f, az = 0.f;
  for (int i0 = 0; i0 < nn; i0 += 64) {
    const int i = i0 + lane;
    const bool in = i < nn;
    const uint iu = in ? (uint)i : 0u;
    const uint kk = dsxy.div(iu);
    const uint rem = iu - kk * (uint)(sx * sy);
    const uint jj = dsx.div(rem);
    const uint ii = rem - jj * (uint)sx;
    const float wx = __shfl(wl, (int)ii, 64), wy = __shfl(wl, sx + (int)jj, 64), wz = __shfl(wl, sx + sy + (int)kk, 64);
    if (!in) continue;
    int cx = o.x + (int)ii, cy = o.y + (int)jj, cz = o.z + (int)kk;
    cx = cx < 0 ? cx + n.x : (cx >= n.x ? cx - n.x : cx);
    cy = cy < 0 ? cy + n.y : (cy >= n.y ? cy - n.y : cy);
    cz = cz < 0 ? cz + n.z : (cz >= n.z ? cz - n.z : cz);
    const size_t node = (size_t)cx + (size_t)nxpad * (size_t)cy + zstride * (size_t)cz;
    ax = fmaf(dV, g0[node] * wx * wy * wz, ax);
    ay = fmaf(dV, g0[plane + node] * wx * wy * wz, ay);
    az = fmaf(dV, g0[2 * plane + node] * wx * wy * wz, az);
  }
#pragma unroll
  for (int o2 = 32; o2 > 0; o2 >>= 1) {
    ax += __shfl_xor(ax, o2, 64);
    ay += __shfl_xor(ay, o2, 64);
    az += __shfl_xor(az, o2, 64);
  }
  if (lane == 0) {
    float *out = vout + 3 * (size_t)o.w;
    if (accumulate) { out[0] += ax; out[1] += ay; out[2] += az; } else { out[0] = ax; out[1] = ay; out[2] = az; }
  }
}

// Tile-staged gather (supports <= 6), OPTIONAL ("tile_gather" = 1).  A wave-per-particle gather from global memory pulls 36
// (y,z) rows x 3 components = 108 cache lines per particle for 216 x 12 useful bytes; here a workgroup owns a tile, copies the
// 16^3 window (tile + 4 nodes each side) of the three velocity grids into LDS once with 8-byte coalesced reads, and its
// particles (contiguous in the tile-sorted prep arrays) interpolate from LDS.  MEASURED at C4: 112 us against 78 us for the
// global gather — 49 KB of window per ~24 particles (3 workgroups per CU, a barrier between load and use) costs more than
// the redundant L1 traffic it saves; kept for dense suspensions (>> 24 particles per tile), off by default.
constexpr int kGW = 16;                     // window edge
constexpr int kGRS = 17;                    // row stride (floats): odd, spreads the rows over the banks
constexpr int kGPS = 16 * kGRS + 4;         // plane stride
constexpr int kGComp = kGW * kGPS;          // floats per component
__global__ void __launch_bounds__(256) k_fcm_gather_tile(float *__restrict__ vout, const float *__restrict__ g0, int3 n,
                                                          int nxpad, size_t plane, size_t zstride, int3 support, int3 ntiles,
                                                          float dV, FastDiv dsx, FastDiv dsxy, FcmPrep pr, bool accumulate) {
  __shared__ float win[3 * kGComp];
  const int tile = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int first = pr.tileStart[tile], last = pr.tileStart[tile + 1];
  if (first == last) return;  // the whole workgroup leaves: no particle interpolates from this window
  const int tx = tile % ntiles.x, ty = (tile / ntiles.x) % ntiles.y, tz = tile / (ntiles.x * ntiles.y);
  const int wx0 = tx * pr.tdim.x, wy0 = ty * pr.tdim.y, wz0 = tz * pr.tdim.z;  // window origin: the tile's own (its particles' stencils start inside it)
  for (int e = threadIdx.x; e < 3 * kGW * kGW * (kGW / 2); e += 256) {
    const int q = e & 7, row = e >> 3;
    const int y = row & 15, z = (row >> 4) & 15, c = row >> 8;
    int gx = wx0 + 2 * q, gy = wy0 + y, gz = wz0 + z;
    gx = gx < 0 ? gx + n.x : (gx >= n.x ? gx - n.x : gx);  // n.x is a multiple of the tile: a pair never straddles the seam
    gy = gy < 0 ? gy + n.y : (gy >= n.y ? gy - n.y : gy);
    gz = gz < 0 ? gz + n.z : (gz >= n.z ? gz - n.z : gz);
    const float2 v = *reinterpret_cast<const float2 *>(g0 + (size_t)c * plane + (size_t)gx + (size_t)nxpad * (size_t)gy +
                                                       zstride * (size_t)gz);
    float *w = win + c * kGComp + 2 * q + kGRS * y + kGPS * z;
    w[0] = v.x;
    w[1] = v.y;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sx = support.x, sy = support.y, sz = support.z;
  const int nn = sx * sy * sz;
  for (int slot = first + wave; slot < last; slot += 4) {
    const int4 o = pr.origin[slot];
    const float wl = lane < sx + sy + sz ? pr.weights[(size_t)pr.wstride * slot + lane] : 0.0f;
    const int lx = (o.x < 0 ? o.x + n.x : o.x) - wx0, ly = (o.y < 0 ? o.y + n.y : o.y) - wy0, lz = (o.z < 0 ? o.z + n.z : o.z) - wz0;  // 0..7: the stencil stays inside the window
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int i0 = 0; i0 < nn; i0 += 64) {
      const int i = i0 + lane;
      const bool in = i < nn;
      const uint iu = in ? (uint)i : 0u;
      const uint kk = dsxy.div(iu);
      const uint rem = iu - kk * (uint)(sx * sy);
      const uint jj = dsx.div(rem);
      const uint ii = rem - jj * (uint)sx;
      const float wx = __shfl(wl, (int)ii, 64), wy = __shfl(wl, sx + (int)jj, 64), wz = __shfl(wl, sx + sy + (int)kk, 64);
      if (!in) continue;
      const float *w = win + (lx + (int)ii) + kGRS * (ly + (int)jj) + kGPS * (lz + (int)kk);
      ax = fmaf(dV, w[0] * wx * wy * wz, ax);
      ay = fmaf(dV, w[kGComp] * wx * wy * wz, ay);
      az = fmaf(dV, w[2 * kGComp] * wx * wy * wz, az);
    }
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) {
      ax += __shfl_xor(ax, o2, 64);
      ay += __shfl_xor(ay, o2, 64);
      az += __shfl_xor(az, o2, 64);
    }
    if (lane == 0) {
      float *out = vout + 3 * (size_t)o.w;
      if (accumulate) { out[0] += ax; out[1] += ay; out[2] += az; } else { out[0] = ax; out[1] = ay; out[2] = az; }
    }
  }
}

// Interleaved gather.  The wave-per-particle gather is bound by cache-line requests, not bytes: a stencil row is 6 nodes =
// 24 contiguous bytes in each of the three planar component grids, so one particle touches ~190 lines for 2.6 KB of data.
// k_fcm_interleave copies the three velocity grids into ONE float4-per-node grid (a streaming pass); a stencil row is then
// 96 contiguous bytes and the gather needs a third of the line requests (measured at C4: 78 us -> 48 + 10 us).
__global__ void __launch_bounds__(256) k_fcm_interleave(const float *__restrict__ g0, int3 n, int nxpad, size_t plane,
                                                         size_t zstride, float4 *__restrict__ out) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)n.x * n.y * n.z;
  if (t >= total) return;
  const int x = (int)(t % n.x), y = (int)((t / n.x) % n.y), z = (int)(t / ((size_t)n.x * n.y));
  const size_t node = (size_t)x + (size_t)nxpad * (size_t)y + zstride * (size_t)z;
  out[t] = make_float4(g0[node], g0[plane + node], g0[2 * plane + node], 0.0f);
}

// R = rounds of 64 stencil nodes whose loads are in flight together.  The first form of this kernel looped over the rounds with the
// load and its use in the same iteration: the compiler put s_waitcnt vmcnt(0) between them, and a wave's origin record, its weights
// and its four rounds were SIX dependent round trips — 48 us at C4, which is 12 rounds of waves x 6 x ~0.65 us and had been read as
// an L2 request limit.  Here the record and the weights are requested together, then every node of a chunk of R rounds, then the
// arithmetic: two round trips per wave.  The sums run in the same order as before (round by round, then the xor reduction).
constexpr int kGatherWaves = 4;  // (16 waves = 16 consecutive tile-sorted particles per workgroup, to share their lines in the CU's L1: 48.7 against 47.1 us)
// P = particles per wave: the kernel is bound by wave lifetimes (two round trips + the reduction, ~3.8 us, at 32 waves per CU), not by
// bytes; a wave that carries P particles keeps P R loads in flight for the same two round trips.
// PK: the grid holds 12 bytes per node (x, y, z) instead of the float4 — for grids too large for the 256 MB Infinity Cache, where the
// kernel waits for HBM and the unused w is a quarter of what it reads (C5: 268 -> 201 MB)
struct __attribute__((packed, aligned(4))) PackedNode { float x, y, z; };
template <int R, int P, bool PK = false>
__global__ void __launch_bounds__(64 * kGatherWaves) k_fcm_gather_inter(float *__restrict__ vout, const float4 *__restrict__ gi, int N, int3 n,
                                                           int3 support, float dV, FastDiv dsx, FastDiv dsxy, FcmPrep pr,
                                                           bool accumulate) {
  const int lane = threadIdx.x & 63;
  // (PK = a grid read from HBM: workgroups in launch order, so that the eight XCDs sweep the same region of the grid together and what
  // one of them brought into the Infinity Cache serves the others — C5 solve 0.697 -> 0.688 ms; a cache-resident grid keeps each XCD on
  // its own contiguous eighth of the entries, whose lines stay in that XCD's L2)
  const int blk = PK ? (int)blockIdx.x : (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int slot0 = (blk * kGatherWaves + (threadIdx.x >> 6)) * P;
  if (slot0 >= N) return;
  const int sx = support.x, sy = support.y, sz = support.z;
  int4 o[P];
  float wl[P];
#pragma unroll
  for (int q = 0; q < P; ++q) {
    const int slot = min(slot0 + q, N - 1);
    o[q] = pr.origin[slot];
    wl[q] = pr.weights[(size_t)pr.wstride * slot + min(lane, sx + sy + sz - 1)];  // (unconditional: no branch between the two loads)
  }
  const int nn = sx * sy * sz;
  float ax[P], ay[P], az[P];
#pragma unroll
  for (int q = 0; q < P; ++q) ax[q] = ay[q] = az[q] = 0.f;
  for (int base = 0; base < nn; base += 64 * R) {
    float4 v[P][R];
    int wsel[R];  // ii | jj << 8 | kk << 16, or -1
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = base + 64 * r + lane;
      const bool in = i < nn;
      const uint iu = in ? (uint)i : 0u;
      const uint kk = dsxy.div(iu);
      const uint rem = iu - kk * (uint)(sx * sy);
      const uint jj = dsx.div(rem);
      const uint ii = rem - jj * (uint)sx;
      wsel[r] = in ? (int)(ii | jj << 8 | kk << 16) : -1;
#pragma unroll
      for (int q = 0; q < P; ++q) {
        int cx = o[q].x + (int)ii, cy = o[q].y + (int)jj, cz = o[q].z + (int)kk;
        cx = cx < 0 ? cx + n.x : (cx >= n.x ? cx - n.x : cx);
        cy = cy < 0 ? cy + n.y : (cy >= n.y ? cy - n.y : cy);
        cz = cz < 0 ? cz + n.z : (cz >= n.z ? cz - n.z : cz);
        const size_t node = (size_t)cx + (size_t)n.x * ((size_t)cy + (size_t)n.y * (size_t)cz);  // (a lane past the stencil re-reads node 0 of it)
        if (PK) {
          const PackedNode pn = ((const PackedNode *)gi)[node];
          v[q][r] = make_float4(pn.x, pn.y, pn.z, 0.0f);
        } else
          v[q][r] = gi[node];
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int ws = wsel[r] < 0 ? 0 : wsel[r];
#pragma unroll
      for (int q = 0; q < P; ++q) {
        const float wx = __shfl(wl[q], ws & 255, 64), wy = __shfl(wl[q], sx + ((ws >> 8) & 255), 64), wz = __shfl(wl[q], sx + sy + (ws >> 16), 64);
        if (wsel[r] >= 0) {
          ax[q] = fmaf(dV, v[q][r].x * wx * wy * wz, ax[q]);
          ay[q] = fmaf(dV, v[q][r].y * wx * wy * wz, ay[q]);
          az[q] = fmaf(dV, v[q][r].z * wx * wy * wz, az[q]);
        }
      }
    }
  }
#pragma unroll
  for (int o2 = 32; o2 > 0; o2 >>= 1) {
#pragma unroll
    for (int q = 0; q < P; ++q) {
      ax[q] += __shfl_xor(ax[q], o2, 64);
      ay[q] += __shfl_xor(ay[q], o2, 64);
      az[q] += __shfl_xor(az[q], o2, 64);
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < P; ++q) {
      if (slot0 + q >= N) break;
      float *out = vout + 3 * (size_t)o[q].w;
      if (accumulate) { out[0] += ax[q]; out[1] += ay[q]; out[2] += az[q]; } else { out[0] = ax[q]; out[1] = ay[q]; out[2] = az[q]; }
    }
  }
}
// Column form of the same gather for supports <= 8: lane = one (ii, jj) column of the stencil (lane & 7, lane >> 3), a loop over the
// stencil's z planes.  With two particles per wave k_fcm_gather_inter is bound by its own instruction stream (~225 vector + LDS-pipe
// instructions per particle: three lane shuffles and a multiply-high division pair per node round, three wraps per node), not by
// memory; here the x / y wraps, the column's address and the product wx wy are computed once per particle, a plane costs an add, a load
// and four multiply-adds, and wz comes from a scalar lane read.  The terms are the same as k_fcm_gather_inter's (dV (g w), w = (wx wy)
// wz), summed in another order.
template <int P, int SZMAX>
__global__ void __launch_bounds__(64 * kGatherWaves) k_fcm_gather_col(float *__restrict__ vout, const float4 *__restrict__ gi, int N, int3 n,
                                                                      int3 support, float dV, FcmPrep pr, bool accumulate) {
  const int lane = threadIdx.x & 63;
  const int slot0 = ((int)xcd_contiguous_block(blockIdx.x, gridDim.x) * kGatherWaves + (threadIdx.x >> 6)) * P;
  if (slot0 >= N) return;
  const int sx = support.x, sy = support.y, sz = support.z;
  const int ii = lane & 7, jj = lane >> 3;
  const bool col = ii < sx && jj < sy;
  int4 o[P];
  float wl[P];
#pragma unroll
  for (int q = 0; q < P; ++q) {
    const int slot = min(slot0 + q, N - 1);
    o[q] = pr.origin[slot];
    wl[q] = pr.weights[(size_t)pr.wstride * slot + min(lane, sx + sy + sz - 1)];
  }
  float4 g[P][SZMAX];
  float wxy[P];
  const uint planeNodes = (uint)n.x * (uint)n.y;
#pragma unroll
  for (int q = 0; q < P; ++q) {
    int cx = o[q].x + ii, cy = o[q].y + jj;
    cx = cx < 0 ? cx + n.x : (cx >= n.x ? cx - n.x : cx);
    cy = cy < 0 ? cy + n.y : (cy >= n.y ? cy - n.y : cy);
    const uint base = col ? (uint)cx + (uint)n.x * (uint)cy : 0u;  // (a lane outside the stencil re-reads node 0 of the plane)
    const int oz = __builtin_amdgcn_readfirstlane(o[q].z);
#pragma unroll
    for (int kk = 0; kk < SZMAX; ++kk) {
      if (kk < sz) {
        int cz = oz + kk;
        cz = cz < 0 ? cz + n.z : (cz >= n.z ? cz - n.z : cz);
        // (a 32-bit byte offset from the grid's base: the launcher takes this kernel for grids below 128 MB, and the address is one shift
        // instead of a 64-bit multiply-add per load)
        g[q][kk] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(gi) + ((base + planeNodes * (uint)cz) << 4));
      }
    }
    wxy[q] = __shfl(wl[q], ii, 64) * __shfl(wl[q], sx + jj, 64);
  }
  float ax[P], ay[P], az[P];
#pragma unroll
  for (int q = 0; q < P; ++q) {
    ax[q] = ay[q] = az[q] = 0.f;
#pragma unroll
    for (int kk = 0; kk < SZMAX; ++kk) {
      if (kk < sz) {
        const float wz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wl[q]), sx + sy + kk));
        const float w = wxy[q] * wz;
        if (col) {
          // (the quadrature weight dV multiplies the finished sum, not every term: three instructions per node fewer in a kernel that is
          // short of issue slots; rounding-level difference from k_fcm_gather_inter's dV (g w) terms)
          ax[q] = fmaf(g[q][kk].x, w, ax[q]);
          ay[q] = fmaf(g[q][kk].y, w, ay[q]);
          az[q] = fmaf(g[q][kk].z, w, az[q]);
        }
      }
    }
  }
  // (six DPP additions per sum instead of six ds_bpermute round trips: the wave's tail was a fifth of its life)
#pragma unroll
  for (int q = 0; q < P; ++q) {
    ax[q] = wave_total(ax[q]);
    ay[q] = wave_total(ay[q]);
    az[q] = wave_total(az[q]);
  }
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < P; ++q) {
      if (slot0 + q >= N) break;
      ax[q] *= dV; ay[q] *= dV; az[q] *= dV;
      float *out = vout + 3 * (size_t)o[q].w;
      if (accumulate) { out[0] += ax[q]; out[1] += ay[q]; out[2] += az[q]; } else { out[0] = ax[q]; out[1] = ay[q]; out[2] = az[q]; }
    }
  }
}
// Support 6 x 6 in x, y — the FCM Gaussian at tolerance 1e-3, the bench's C4 / C5 — with every lane at work: k_fcm_gather_col keeps 36 of a
// wave's 64 lanes busy and issues sz loads per particle, and a 64-lane load costs the CU's address unit the same ~16 clocks whatever its
// lanes fetch.  Here a HALF wave takes a particle: lane l of the half is column (ii, jj) = (l % 6, l / 6) — all 32 lanes inside the stencil,
// rows jj = 0..4 and the first two columns of row 5 — for the loop over the z planes, and ONE more load fetches what is left of row 5
// (columns 2..5 of every plane: 4 sz <= 32 nodes, lane -> (2 + (l & 3), l >> 2)).  sz + 1 loads per PAIR of particles instead of 2 sz.
// Same terms as k_fcm_gather_col, summed in another order (the half wave's sum is the first five steps of wave_sum_to_last).
// S = 7 (the PSE far field's support at the bench's size): the first 32 of the 49 columns in the loop, the other 17 x sz nodes in four more
// loads — 11 per pair of particles where the column form issues 14 with 49 of 64 lanes at work.
template <int S, int SZMAX>
__global__ void __launch_bounds__(64 * kGatherWaves) k_fcm_gather_half(float *__restrict__ vout, const float4 *__restrict__ gi, int N, int3 n,
                                                                       int sz, float dV, FcmPrep pr, bool accumulate) {
  constexpr int R = S * S - 32;                          // columns the loop does not cover (row-major order: column c = (c % S, c / S))
  constexpr int ROUNDS = (R * SZMAX + 31) / 32;
  static_assert(R > 0 && 2 * S + SZMAX <= 32, "supports 6 and 7");
  const int lane = threadIdx.x & 63, h = lane >> 5, l = lane & 31;
  const int slot0 = ((int)xcd_contiguous_block(blockIdx.x, gridDim.x) * kGatherWaves + (threadIdx.x >> 6)) * 2;
  if (slot0 >= N) return;
  const bool valid = slot0 + h < N;
  const int slot = min(slot0 + h, N - 1);
  const int4 o = pr.origin[slot];
  const float wl = pr.weights[(size_t)pr.wstride * slot + min(l, 2 * S + sz - 1)];   // lanes 0..S-1 wx, S..2S-1 wy, 2S.. wz of the half's particle
  const int jj = l / S, ii = l - S * jj;
  const uint planeNodes = (uint)n.x * (uint)n.y;
  auto wrap = [](int c, int m) { return c < 0 ? c + m : (c >= m ? c - m : c); };
  const uint base = (uint)wrap(o.x + ii, n.x) + (uint)n.x * (uint)wrap(o.y + jj, n.y);
  float4 g[SZMAX];
#pragma unroll
  for (int kk = 0; kk < SZMAX; ++kk)
    if (kk < sz)
      g[kk] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(gi) + ((base + planeNodes * (uint)wrap(o.z + kk, n.z)) << 4));
  // the columns beyond the first 32, plane by plane: node m = l + 32 r -> column 32 + m % R of plane m / R
  float4 g2[ROUNDS];
  int ii2[ROUNDS], jj2[ROUNDS], kk2[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int m = l + 32 * r;
    kk2[r] = m / R;
    const int col = 32 + (m - R * kk2[r]);
    jj2[r] = col / S;
    ii2[r] = col - S * jj2[r];
    const bool rest = kk2[r] < sz;
    if (!rest) kk2[r] = 0;
    const uint b2 = (uint)wrap(o.x + ii2[r], n.x) + (uint)n.x * (uint)wrap(o.y + jj2[r], n.y) + planeNodes * (uint)wrap(o.z + kk2[r], n.z);
    g2[r] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(gi) + (b2 << 4));
    if (!rest) kk2[r] = -1;
  }
  const int hb = h << 5;
  const float wxy = __shfl(wl, hb + ii, 64) * __shfl(wl, hb + S + jj, 64);
  float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
  for (int kk = 0; kk < SZMAX; ++kk) {
    if (kk < sz) {
      const float wzA = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wl), 2 * S + kk));
      const float wzB = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wl), 32 + 2 * S + kk));
      const float w = wxy * (h ? wzB : wzA);
      ax = fmaf(g[kk].x, w, ax);
      ay = fmaf(g[kk].y, w, ay);
      az = fmaf(g[kk].z, w, az);
    }
  }
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const bool rest = kk2[r] >= 0;
    const float w = __shfl(wl, hb + ii2[r], 64) * __shfl(wl, hb + S + jj2[r], 64) * __shfl(wl, hb + 2 * S + (rest ? kk2[r] : 0), 64);
    if (rest) {
      ax = fmaf(g2[r].x, w, ax);
      ay = fmaf(g2[r].y, w, ay);
      az = fmaf(g2[r].z, w, az);
    }
  }
  auto half_sum = [](float x) {   // lanes 31 and 63 end with their half's sum
    x += dpp_move<0xB1, 0xf, true>(x);
    x += dpp_move<0x4E, 0xf, true>(x);
    x += dpp_move<0x141, 0xf, true>(x);
    x += dpp_move<0x140, 0xf, true>(x);
    x += dpp_move<0x142, 0xa, false>(x);
    return x;
  };
  ax = half_sum(ax);
  ay = half_sum(ay);
  az = half_sum(az);
  if (l == 31 && valid) {
    ax *= dV; ay *= dV; az *= dV;
    float *out = vout + 3 * (size_t)o.w;
    if (accumulate) { out[0] += ax; out[1] += ay; out[2] += az; } else { out[0] = ax; out[1] = ay; out[2] = az; }
  }
}
// (A window gather — a workgroup per tile stages the tile-edge + support window of the interleaved grid in LDS, 44 KB at C4, and its
// four waves interpolate the tile's ~24 particles from LDS — was written twice: round 1 on the planar grids, 112 us, and round 3 on the
// float4 grid with every load of a thread in flight together, 72 us against 47 us for the kernel above: three workgroups per CU do not
// hide the window's round trip and the per-particle shuffle chains.  Not kept.)
static void launch_gather_inter(hipStream_t st, float *vout, const float4 *gi, int N, int3 n, int3 support, float dV, FastDiv dsx,
                                FastDiv dsxy, const FcmPrep &pr, bool accumulate, int perWave = 2, bool packed = false) {
  // the column form where the float4 grid stays in the 256 MB Infinity Cache (C4: 39.5 -> 32.9 us; at C5, a 268 MB grid read from HBM,
  // its six partly filled loads per particle lose to the four full ones: 126 against 112 us)
  const size_t nodes = (size_t)n.x * n.y * n.z;
  static const int halfMode = getenv("UAMMD_FCM_GATHER_HALF") ? atoi(getenv("UAMMD_FCM_GATHER_HALF")) : 1;   // (A/B runs: 0 off, 2 also on grids read from HBM)
  if (packed) {  // (fcm_inter_packed: a grid read from HBM, 12 bytes per node; the float4 forms below do not apply)
    const int rounds = (support.x * support.y * support.z + 63) / 64;
    const int P = rounds <= 4 ? (perWave >= 4 ? 4 : 2) : 1;
    const dim3 g((N + kGatherWaves * P - 1) / (kGatherWaves * P)), b(64 * kGatherWaves);
#define UH_GIP(RR, PP) hipLaunchKernelGGL((k_fcm_gather_inter<RR, PP, true>), g, b, 0, st, vout, gi, N, n, support, dV, dsx, dsxy, pr, accumulate)
    if (rounds <= 1) UH_GIP(1, 2); else if (rounds <= 2) UH_GIP(2, 2); else if (rounds <= 4) { if (P == 4) UH_GIP(4, 4); else UH_GIP(4, 2); } else UH_GIP(8, 1);
#undef UH_GIP
    return;
  }
  if (halfMode == 2 && perWave >= 0 && support.x == 6 && support.y == 6 && support.z <= 8 && nodes * sizeof(float4) < ((size_t)1 << 32)) {
    const dim3 g((N + kGatherWaves * 2 - 1) / (kGatherWaves * 2)), b(64 * kGatherWaves);
    if (support.z <= 6) hipLaunchKernelGGL((k_fcm_gather_half<6, 6>), g, b, 0, st, vout, gi, N, n, support.z, dV, pr, accumulate);
    else hipLaunchKernelGGL((k_fcm_gather_half<6, 8>), g, b, 0, st, vout, gi, N, n, support.z, dV, pr, accumulate);
    return;
  }
  if (perWave >= 0 && support.x <= 8 && support.y <= 8 && support.z <= 8 && nodes * sizeof(float4) <= ((size_t)128 << 20)) {
    constexpr int P = 2;  // (32.9 / 33.5 / 36.4 us with 2 / 3 / 4 particles per wave at C4)
    const dim3 g((N + kGatherWaves * P - 1) / (kGatherWaves * P)), b(64 * kGatherWaves);
    if (support.x == 6 && support.y == 6 && halfMode != 0) {
      if (support.z <= 6) hipLaunchKernelGGL((k_fcm_gather_half<6, 6>), g, b, 0, st, vout, gi, N, n, support.z, dV, pr, accumulate);
      else hipLaunchKernelGGL((k_fcm_gather_half<6, 8>), g, b, 0, st, vout, gi, N, n, support.z, dV, pr, accumulate);
      return;
    }
    if (support.x == 7 && support.y == 7 && halfMode != 0) {
      if (support.z <= 7) hipLaunchKernelGGL((k_fcm_gather_half<7, 7>), g, b, 0, st, vout, gi, N, n, support.z, dV, pr, accumulate);
      else hipLaunchKernelGGL((k_fcm_gather_half<7, 8>), g, b, 0, st, vout, gi, N, n, support.z, dV, pr, accumulate);
      return;
    }
    if (support.z <= 6) hipLaunchKernelGGL((k_fcm_gather_col<2, 6>), g, b, 0, st, vout, gi, N, n, support, dV, pr, accumulate);
    else hipLaunchKernelGGL((k_fcm_gather_col<2, 8>), g, b, 0, st, vout, gi, N, n, support, dV, pr, accumulate);
    return;
  }
  if (perWave < 0) perWave = -perWave;  // (test hook: a negative value asks for k_fcm_gather_inter with that many particles per wave)
  const int rounds = (support.x * support.y * support.z + 63) / 64;
  // (measured at C4, support 6: 47.3 / 40.2 / 40.8 us with 1 / 2 / 4 particles per wave)
  const int P = (rounds <= 4 && perWave >= 2) ? (perWave >= 4 ? 4 : 2) : 1;
  const dim3 g((N + kGatherWaves * P - 1) / (kGatherWaves * P)), b(64 * kGatherWaves);
#define UH_GI(RR, PP) hipLaunchKernelGGL((k_fcm_gather_inter<RR, PP>), g, b, 0, st, vout, gi, N, n, support, dV, dsx, dsxy, pr, accumulate)
  if (rounds <= 1) { if (P == 4) UH_GI(1, 4); else if (P == 2) UH_GI(1, 2); else UH_GI(1, 1); }
  else if (rounds <= 2) { if (P == 4) UH_GI(2, 4); else if (P == 2) UH_GI(2, 2); else UH_GI(2, 1); }
  else if (rounds <= 4) { if (P == 4) UH_GI(4, 4); else if (P == 2) UH_GI(4, 2); else UH_GI(4, 1); }
  else UH_GI(8, 1);
#undef UH_GI
}

// ---- Fourier space ---------------------------------------------------------------------------------------
struct C3 { float xr, xi, yr, yi, zr, zi; };

UH_D int3 index_to_wavenumber(int i, int3 nk) {  // FCM/utils.cuh:27-35
  int ikx = i % (nk.x / 2 + 1);
  int iky = (i / (nk.x / 2 + 1)) % nk.y;
  int ikz = i / ((nk.x / 2 + 1) * nk.y);
  ikx -= nk.x * (ikx >= (nk.x / 2 + 1));
  iky -= nk.y * (iky >= (nk.y / 2 + 1));
  ikz -= nk.z * (ikz >= (nk.z / 2 + 1));
  return make_int3(ikx, iky, ikz);
}
UH_D real3f wavevector(int3 ik, real3f L) {  // FCM/utils.cuh:37-39
  const float twopi = 2.0f * 3.14159265358979323846f;
  return real3f{(twopi / L.x) * (float)ik.x, (twopi / L.y) * (float)ik.y, (twopi / L.z) * (float)ik.z};
}
UH_D real3f gradient_fourier(int3 ik, int3 nk, real3f k) {  // FCM/utils.cuh:41-51: unpaired (Nyquist) components -> 0
  return real3f{ik.x == (nk.x - ik.x) ? 0.0f : k.x, ik.y == (nk.y - ik.y) ? 0.0f : k.y, ik.z == (nk.z - ik.z) ? 0.0f : k.z};
}
// 1 / x for the Fourier-space operator: the hardware reciprocal refined by one Newton step (<= 1 ulp; host pass: the division).  The
// fused z pass is bound by its vector instruction count at C5 (SQ_INSTS_VALU x 4 cycles = 94 % of the SIMD cycles,
// profiles/r06_pmc_fft_z_fused.txt) and an IEEE division is ~12 of them, four per node; results within 1e-7 of the divided form.
UH_D float op_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const float r = __builtin_amdgcn_rcpf(x);
  return fmaf(fmaf(-x, r, 1.0f), r, r);
#else
  return 1.0f / x;
#endif
}
UH_D float op_sqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_sqrtf(x);   // (1 ulp; the IEEE-correct sqrtf is a ten-instruction sequence without fast math)
#else
  return sqrtf(x);
#endif
}
UH_D real3f project_inv(float invk2, real3f dk, real3f fr) {  // FCM/utils.cuh:70-74
  const float s = dot3(fr, real3f{dk.x * invk2, dk.y * invk2, dk.z * invk2});
  return real3f{fmaf(-dk.x, s, fr.x), fmaf(-dk.y, s, fr.y), fmaf(-dk.z, s, fr.z)};
}
UH_D real3f project(float k2, real3f dk, real3f fr) { return project_inv(op_rcp(k2), dk, fr); }
UH_D C3 project(float k2, real3f dk, C3 f) {
  const float invk2 = op_rcp(k2);
  const real3f re = project_inv(invk2, dk, real3f{f.xr, f.yr, f.zr});
  const real3f im = project_inv(invk2, dk, real3f{f.xi, f.yi, f.zi});
  return C3{re.x, im.x, re.y, im.y, re.z, im.z};
}
UH_D bool is_nyquist(int3 c, int3 n) {  // FCM/utils.cuh:133-167
  const bool X = (c.x == n.x - c.x) && (n.x % 2 == 0), Y = (c.y == n.y - c.y) && (n.y % 2 == 0),
             Z = (c.z == n.z - c.z) && (n.z % 2 == 0);
  return (X && c.y == 0 && c.z == 0) || (X && Y && c.z == 0) || (c.x == 0 && Y && c.z == 0) || (X && c.y == 0 && Z) ||
         (c.x == 0 && c.y == 0 && Z) || (c.x == 0 && Y && Z) || (X && Y && Z);
}
UH_D bool noise_skipped(int id, int3 c, int3 n) {  // FCM_impl.cuh:456-463: nodes that do not draw
  return id == 0 || (c.x == 0 && c.y == 0 && 2 * c.z >= n.z + 1) || (c.x == 0 && 2 * c.y >= n.y + 1);
}
UH_D C3 draw_noise(float prefactor, uint id, uint seed1, uint seed2, bool nyquist) {  // FCM/utils.cuh:117-131 + :466-476
  Saru rng(id, seed1, seed2);
  const float sc = 0.707106781186547f * prefactor;
  const float2 a = rng.gf_fast(0.0f, sc), b = rng.gf_fast(0.0f, sc), c = rng.gf_fast(0.0f, sc);
  C3 n{a.x, a.y, b.x, b.y, c.x, c.y};
  if (nyquist) {
    const float q = 1.41421356237310f;
    n.xr *= q; n.xi = 0.0f; n.yr *= q; n.yi = 0.0f; n.zr *= q; n.zi = 0.0f;
  }
  return n;
}

// One thread per (local) Fourier node, in place.  KLayout says where node (kx, y0 + yl, z) of component c lives:
// g0[kx + nkx*yl + zStride*z + compStride*c].  Single GPU: 3 planar grids (nyl = ny, zStride = nkx*ny, compStride = plane);
// slab-decomposed: the y-pencil layout [z][c][yl][kx] that the all-to-all transpose delivers.
struct KLayout { int nyl, y0; size_t compStride, zStride; FastDiv divNkx, divNyl; };
// PSE far field (FarField.cuh): B = sinc^2(k a) Hasimoto(k, xi, eta) / (eta_visc a^2 k^2 N) with the sheared wave vector,
// projection with the sheared k (no Nyquist zeroing), noise scaled by sqrt(B) AFTER the projection.
UH_D real3f pse_shear(real3f k, float shear) { k.y = fmaf(-shear, k.x, k.y); return k; }
UH_D float pse_greens(real3f k, const PseGreens &p, float viscosity, int3 n) {  // FarField.cuh:85-119
  const float k2 = dot3(k, k);
  if (k2 == 0.0f) return 0.0f;
  const real3f kE = pse_shear(k, p.shear);
  const float kE2 = dot3(kE, kE), kN2 = k2;
  const float kmod = op_sqrt(kE2);
  const float invk2 = op_rcp(kE2);
  const float sink = sinf(kmod * p.rh);
  const float inv4xi2 = 1.0f / (4.0f * p.split * p.split);   // (wave-uniform)
  const float kEw = kE2 * inv4xi2;
  const float kNU = kN2 * inv4xi2;
  const float tau = fmaf(p.eta, kNU, -kEw);
  const float hashimoto = (1.0f + kEw) * expf(tau) * invk2;
  const float scale = 1.0f / ((viscosity * p.rh * p.rh) * (float)(n.x * n.y * n.z));   // (wave-uniform: 1 / (eta a^2 N))
  return sink * sink * invk2 * hashimoto * scale;
}
UH_D C3 pse_project(real3f k, const C3 &f) {  // FarField.cuh:53-73
  const float invk2 = op_rcp(dot3(k, k));
  const float kfr = dot3(k, real3f{f.xr, f.yr, f.zr}) * invk2;
  const float kfi = dot3(k, real3f{f.xi, f.yi, f.zi}) * invk2;
  return C3{fmaf(-k.x, kfr, f.xr), fmaf(-k.x, kfi, f.xi), fmaf(-k.y, kfr, f.yr), fmaf(-k.y, kfi, f.yi), fmaf(-k.z, kfr, f.zr),
            fmaf(-k.z, kfi, f.zi)};
}

// The Fourier-space operator on ONE node: forceFourier2Vel (FCM_impl.cuh:375-397) + fourierBrownianNoise (:437-512) in gather form,
// or their PSE far-field counterparts (FarField.cuh:137-158, :235-308).  `in` = the transformed spread forces at the node (ignored
// unless haveForce), `id` = the reference's linear node index cell.x + nkx (cell.y + ny cell.z): it seeds the noise.
UH_D C3 fcm_kspace_node(int3 cell, int id, int3 nk, int nkx, real3f L, float viscosity, bool haveForce, float noisePrefactor, uint seed1,
                        uint seed2, const PseGreens &pse, const C3 &in) {
  C3 v{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float invCells = 1.0f / (float)(nk.x * nk.y * nk.z);   // (wave-uniform: the compiler keeps it out of the node loops)
  // indexToWaveNumber (FCM/utils.cuh:27-35) from the cell coordinates it would recompute by division
  const int3 ik = make_int3(cell.x - nk.x * (cell.x >= nkx), cell.y - nk.y * (cell.y >= nk.y / 2 + 1),
                            cell.z - nk.z * (cell.z >= nk.z / 2 + 1));
  const real3f k = wavevector(ik, L);
  const float k2 = dot3(k, k);
  const real3f dk = gradient_fourier(ik, nk, k);
  if (pse.on) {
    if (id != 0) {
      const float B = pse_greens(k, pse, viscosity, nk);
      const real3f ks = pse_shear(k, pse.shear);
      if (haveForce) {  // forceFourier2Vel, FarField.cuh:137-158: project(B * f)
        const float2 a = make_float2(in.xr, in.xi), b = make_float2(in.yr, in.yi), c = make_float2(in.zr, in.zi);
        v = pse_project(ks, C3{a.x * B, a.y * B, b.x * B, b.y * B, c.x * B, c.y * B});
      }
      if (noisePrefactor != 0.0f) {  // fourierBrownianNoise, FarField.cuh:235-308, in gather form
        const float Bsq = op_sqrt(B);
        const bool own = !noise_skipped(id, cell, nk);
        int idp = -1;
        if (cell.x == 0 || cell.x == nk.x - cell.x) {
          const int3 pc = make_int3(cell.x, (cell.y > 0) * (nk.y - cell.y), (cell.z > 0) * (nk.z - cell.z));
          const int cand = pc.x + nkx * (pc.y + pc.z * nk.y);
          if (cand != id && !noise_skipped(cand, pc, nk) && !is_nyquist(pc, nk)) idp = cand;
        }
        C3 mine{0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, theirs = mine;
        if (own) {
          const C3 z = pse_project(ks, draw_noise(noisePrefactor, (uint)id, seed1, seed2, is_nyquist(cell, nk)));
          mine = C3{z.xr * Bsq, z.xi * Bsq, z.yr * Bsq, z.yi * Bsq, z.zr * Bsq, z.zi * Bsq};
        }
        if (idp >= 0) {
          C3 f = draw_noise(noisePrefactor, (uint)idp, seed1, seed2, false);
          f.xi *= -1.0f; f.yi *= -1.0f; f.zi *= -1.0f;
          const C3 z = pse_project(ks, f);
          theirs = C3{z.xr * Bsq, z.xi * Bsq, z.yr * Bsq, z.yi * Bsq, z.zr * Bsq, z.zi * Bsq};
        }
        const C3 first = (idp >= 0 && idp < id) ? theirs : mine, second = (idp >= 0 && idp < id) ? mine : theirs;
        v.xr += first.xr; v.xi += first.xi; v.yr += first.yr; v.yi += first.yi; v.zr += first.zr; v.zi += first.zi;
        v.xr += second.xr; v.xi += second.xi; v.yr += second.yr; v.yi += second.yi; v.zr += second.zr; v.zi += second.zi;
      }
    }
    return v;
  }
  if (haveForce && id != 0) {  // forceFourier2Vel, FCM_impl.cuh:375-397
    const float2 a = make_float2(in.xr, in.xi), b = make_float2(in.yr, in.yi), c = make_float2(in.zr, in.zi);
    const float B = op_rcp(viscosity * k2);
    const float sc = B * invCells;
    const C3 pr = project(k2, dk, C3{a.x, a.y, b.x, b.y, c.x, c.y});
    v = C3{pr.xr * sc, pr.xi * sc, pr.yr * sc, pr.yi * sc, pr.zr * sc, pr.zi * sc};
  }
  if (noisePrefactor != 0.0f && id != 0) {  // fourierBrownianNoise, FCM_impl.cuh:437-512, in gather form
    const float Bsq = op_sqrt(op_rcp(k2 * viscosity));
    const bool own = !noise_skipped(id, cell, nk);
    // conjugate partner: only stored (and only written by the reference) on the kx == 0 / kx == nx - kx planes
    int idp = -1;
    if (cell.x == 0 || cell.x == nk.x - cell.x) {
      const int3 pc = make_int3(cell.x, (cell.y > 0) * (nk.y - cell.y), (cell.z > 0) * (nk.z - cell.z));
      const int cand = pc.x + nkx * (pc.y + pc.z * nk.y);
      if (cand != id && !noise_skipped(cand, pc, nk) && !is_nyquist(pc, nk)) idp = cand;
    }
    C3 mine{0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, theirs = mine;
    if (own) {
      const C3 n = draw_noise(noisePrefactor, (uint)id, seed1, seed2, is_nyquist(cell, nk));
      mine = project(k2, dk, C3{n.xr * Bsq, n.xi * Bsq, n.yr * Bsq, n.yi * Bsq, n.zr * Bsq, n.zi * Bsq});
    }
    if (idp >= 0) {
      const C3 n = draw_noise(noisePrefactor, (uint)idp, seed1, seed2, false);
      C3 f{n.xr * Bsq, n.xi * Bsq, n.yr * Bsq, n.yi * Bsq, n.zr * Bsq, n.zi * Bsq};
      f.xi *= -1.0f; f.yi *= -1.0f; f.zi *= -1.0f;
      theirs = project(k2, dk, f);
    }
    // same order of the two += as a sequential sweep over node ids
    const C3 first = (idp >= 0 && idp < id) ? theirs : mine, second = (idp >= 0 && idp < id) ? mine : theirs;
    v.xr += first.xr; v.xi += first.xi; v.yr += first.yr; v.yi += first.yi; v.zr += first.zr; v.zi += first.zi;
    v.xr += second.xr; v.xi += second.xi; v.yr += second.yr; v.yi += second.yi; v.zr += second.zr; v.zi += second.zi;
  }
  return v;
}


__global__ void __launch_bounds__(256) k_fcm_kspace(float2 *__restrict__ g0, KLayout lay, int3 nk, real3f L,
                                                     float viscosity, bool haveForce, float noisePrefactor,
                                                     uint seed1, uint seed2, PseGreens pse) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int nkx = nk.x / 2 + 1;
  const int total = nk.z * lay.nyl * nkx;
  if (t >= total) return;
  // (kx, yl, z) of this thread: three integer divisions by run-time constants would cost ~100 VALU slots; multiply-high instead
  const uint tq = lay.divNkx.div((uint)t);          // t / nkx
  const uint tz = lay.divNyl.div(tq);               // t / (nkx * nyl)
  const int3 cell = make_int3(t - (int)tq * nkx, lay.y0 + (int)(tq - tz * (uint)lay.nyl), (int)tz);
  const int id = cell.x + nkx * (cell.y + nk.y * cell.z);  // the reference's linear node index (it seeds the noise)
  const size_t a0 = (size_t)cell.x + (size_t)nkx * (size_t)(cell.y - lay.y0) + lay.zStride * (size_t)cell.z;
  float2 *g1 = g0 + lay.compStride, *g2 = g0 + 2 * lay.compStride;
  C3 in{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (haveForce && id != 0) {
    const float2 a = g0[a0], b = g1[a0], c = g2[a0];
    in = C3{a.x, a.y, b.x, b.y, c.x, c.y};
  }
  const C3 v = fcm_kspace_node(cell, id, nk, nkx, L, viscosity, haveForce, noisePrefactor, seed1, seed2, pse, in);
  g0[a0] = make_float2(v.xr, v.xi);
  g1[a0] = make_float2(v.yr, v.yi);
  g2[a0] = make_float2(v.zr, v.zi);
}

#include "fcm_fft.hpp"

// ---- z lines + Fourier-space operator, fused -------------------------------------------------------------------------------------------
// One workgroup holds the three components of `tl` consecutive (ky, kx) nodes (flat index q = ky nkx + kx, contiguous in memory) over
// all nz planes in LDS: forward z transform (skipped when there are no forces: the grid is then all noise), the Stokes / noise
// operator node by node, inverse z transform, store.  Replaces two strided rocFFT passes and k_fcm_kspace (three trips of the
// 25.6 MB grid through memory at C4) by one.
template <int LOG2TL, int NT, bool P2>
__global__ void __launch_bounds__(NT) k_fft_z_fused(float2 *__restrict__ g, size_t planeC, size_t zStride, int nyl, int y0, int nzArg,
                                                    int3 nk, real3f L, float viscosity, bool haveForce, float noisePrefactor,
                                                    uint seed1, uint seed2, PseGreens pse) {
  extern __shared__ float2 lds[];
  constexpr int TL = 1 << LOG2TL, JG = NT >> LOG2TL, MAXB = 6 * 256 / NT;
  const int nz = fft_len<P2>(nzArg);
  const int LS = nz + 1, nkx = nk.x / 2 + 1;
  float2 *tw = lds, *buf = lds + nz;
  const int tid = threadIdx.x, l = tid & (TL - 1), jg = tid >> LOG2TL;
  // element (component c, plane j, line q) at c planeC + j zStride + q; q = yl nkx + kx runs over this rank's y rows [y0, y0 + nyl)
  // (single GPU: the whole grid, zStride = ny nkx; slab decomposition: the y-pencil layout [z][c][yl][kx], zStride = 3 nyl nkx)
  const size_t slab = zStride;
  // (adjacent tiles share 128-byte lines — a tile of 8 nodes is 64 bytes of every z plane: same XCD, same L2)
  const int q0 = (int)xcd_contiguous_block(blockIdx.x, gridDim.x) * TL, nl = min(TL, nyl * nkx - q0);
  float2 *base = g + q0 + l;
  fft_twiddles<NT>(tw, nz, tid);
  // Power-of-two lines whose plan starts with a radix-4 pass and ends with a radix-8 one (128, 256, 32): the first forward pass is done on
  // the values as they arrive from memory and the last inverse pass on the values as they leave — two trips of the tile through LDS and
  // two barriers fewer of the kernel's eight.  Same butterflies on the same values: the same bits.
  constexpr int NLINES = 3 * TL, QF = 3 * 512 / NT, QL = 2 * 512 / NT;
  const int log2N = 31 - __builtin_clz((unsigned)nz);
  const bool edges = P2 && haveForce && nl == TL && log2N % 3 != 0 && log2N >= 5 && NLINES * (nz >> 2) <= QF * NT &&
                     NLINES * (nz >> 3) <= QL * NT;
  if (edges) {
    const int per = nz >> 2, total = NLINES * per;
    float2 v[QF][4];
#pragma unroll
    for (int q = 0; q < QF; ++q) {
      const int b = tid + q * NT;
      if (b < total) {
        const int j = b / NLINES, line = b - j * NLINES, c = line >> LOG2TL, ll = line & (TL - 1);
        const float2 *src = g + q0 + ll + (size_t)c * planeC + (size_t)j * slab;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[q][r] = src[(size_t)(r * per) * slab];
      }
    }
#pragma unroll
    for (int q = 0; q < QF; ++q) {
      const int b = tid + q * NT;
      if (b < total) {
        const int j = b / NLINES, line = b - j * NLINES;
        fft_butterfly<4, -1>(v[q]);
        float2 *dst = buf + line * LS + 4 * j;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[r] = v[q][r];
      }
    }
    __syncthreads();
    fft_lds_p2_inner<-1, MAXB, NT>(buf, LS, log2N, NLINES, tw, 1, tid, true, false);
  } else if (haveForce) {
    if (l < nl) {  // element e = c (nz / JG) + j / JG of this thread's 3 nz / JG
      const int perC = (nz + JG - 1) / JG;
      staged_copy<12, float2>(0, 3 * perC, 1,
          [&](int e) {
            const int c = e / perC, j = jg + (e - c * perC) * JG;
            return j < nz ? base[(size_t)c * planeC + (size_t)j * slab] : make_float2(0.f, 0.f);
          },
          [&](int e, float2 v) {
            const int c = e / perC, j = jg + (e - c * perC) * JG;
            if (j < nz) buf[(c * nl + l) * LS + j] = v;
          });
    }
    __syncthreads();
    fft_lds<-1, MAXB, NT, P2>(buf, LS, nz, 3 * nl, tw, 1, tid);
  } else
    __syncthreads();
  if (l < nl) {
    const int q = q0 + l, yl = q / nkx, kx = q - yl * nkx, ky = y0 + yl;
    for (int j = jg; j < nz; j += JG) {
      const int3 cell = make_int3(kx, ky, j);
      const int id = kx + nkx * (ky + nk.y * j);
      C3 in{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float2 *px = buf + l * LS + j, *py = buf + (nl + l) * LS + j, *pz = buf + (2 * nl + l) * LS + j;
      if (haveForce) in = C3{px->x, px->y, py->x, py->y, pz->x, pz->y};
      const C3 v = fcm_kspace_node(cell, id, nk, nkx, L, viscosity, haveForce, noisePrefactor, seed1, seed2, pse, in);
      *px = make_float2(v.xr, v.xi);
      *py = make_float2(v.yr, v.yi);
      *pz = make_float2(v.zr, v.zi);
    }
  }
  __syncthreads();
  if (edges) {
    fft_lds_p2_inner<1, MAXB, NT>(buf, LS, log2N, NLINES, tw, 1, tid, false, true);
    const int per = nz >> 3, total = NLINES * per;   // the last pass: radix 8, sub-transforms of nz / 8 points, outputs at j + r nz / 8
#pragma unroll
    for (int q = 0; q < QL; ++q) {
      const int b = tid + q * NT;
      if (b < total) {
        const int j = b / NLINES, line = b - j * NLINES, c = line >> LOG2TL, ll = line & (TL - 1);
        const float2 *p = buf + line * LS + j;
        float2 v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = p[r * per];
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = ctw<1>(v[r], tw[j * r]);
        fft_butterfly<8, 1>(v);
        float2 *dst = g + q0 + ll + (size_t)c * planeC + (size_t)j * slab;
#pragma unroll
        for (int r = 0; r < 8; ++r) dst[(size_t)(r * per) * slab] = v[r];
      }
    }
    return;
  }
  fft_lds<1, MAXB, NT, P2>(buf, LS, nz, 3 * nl, tw, 1, tid);
  if (l < nl)
    for (int c = 0; c < 3; ++c)
      for (int j = jg; j < nz; j += JG) base[(size_t)c * planeC + (size_t)j * slab] = buf[(c * nl + l) * LS + j];
}

// half * (i dk) x g on the planar complex grids: addTorqueCurl (ACC: out += ...) and computeVelocityCurlFourier (out = ...),
// FCM_impl.cuh:306-327, :590-617.  dk has its unpaired (Nyquist) components zeroed (getGradientFourier).
template <bool ACC>
__global__ void __launch_bounds__(256) k_fcm_half_curl(const float2 *__restrict__ in, float2 *__restrict__ out, size_t planeC,
                                                        int3 nk, real3f L, FastDiv divNkx, FastDiv divNy) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  const int nkx = nk.x / 2 + 1;
  if (id >= nk.z * nk.y * nkx) return;
  const uint tq = divNkx.div((uint)id), tz = divNy.div(tq);
  const int cx = id - (int)tq * nkx, cy = (int)(tq - tz * (uint)nk.y), cz = (int)tz;
  const int3 ik = make_int3(cx - nk.x * (cx >= nkx), cy - nk.y * (cy >= nk.y / 2 + 1), cz - nk.z * (cz >= nk.z / 2 + 1));
  const real3f dk = gradient_fourier(ik, nk, wavevector(ik, L));
  const float2 gx = in[id], gy = in[planeC + id], gz = in[2 * planeC + id];
  const float h = 0.5f;
  float2 cxv = make_float2(h * fmaf(-dk.y, gz.y, dk.z * gy.y), h * fmaf(dk.y, gz.x, -(dk.z * gy.x)));
  float2 cyv = make_float2(h * fmaf(-dk.z, gx.y, dk.x * gz.y), h * fmaf(dk.z, gx.x, -(dk.x * gz.x)));
  float2 czv = make_float2(h * fmaf(-dk.x, gy.y, dk.y * gx.y), h * fmaf(dk.x, gy.x, -(dk.y * gx.x)));
  if (ACC) {
    const float2 ox = out[id], oy = out[planeC + id], oz = out[2 * planeC + id];
    cxv.x += ox.x; cxv.y += ox.y; cyv.x += oy.x; cyv.y += oy.y; czv.x += oz.x; czv.y += oz.y;
  }
  out[id] = cxv;
  out[planeC + id] = cyv;
  out[2 * planeC + id] = czv;
}

// test hook: interleave the planar complex grids into complex3[Nk]
__global__ void k_fcm_export(const float2 *__restrict__ g0, size_t planeC, int total, float *__restrict__ out6) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= total) return;
  const float2 a = g0[id], b = g0[planeC + id], c = g0[2 * planeC + id];
  out6[6 * (size_t)id + 0] = a.x; out6[6 * (size_t)id + 1] = a.y;
  out6[6 * (size_t)id + 2] = b.x; out6[6 * (size_t)id + 3] = b.y;
  out6[6 * (size_t)id + 4] = c.x; out6[6 * (size_t)id + 5] = c.y;
}


// ---- the LDS FFT pipeline (fcm_fft.hpp) ---------------------------------------------------------------------------------------------------
// the y transform of `groups` planes of (n x nkx) complex: 16 lines of n points per workgroup = n / 64 radix-4 butterflies per thread
// (the shift-and-mask instantiations: powers of two from 4 up; a two-point axis takes the general passes)
static bool is_pow2(int n) { return n >= 4 && (n & (n - 1)) == 0; }
template <int SIGN> static void fft_launch_lines(float2 *g, int n, int nkx, int groups, hipStream_t st) {
  const int tiles = (nkx + 15) / 16;
  const dim3 gr(groups * tiles);
  const size_t ldsz = sizeof(float2) * (size_t)(n + 16 * (n + 1));
  // (512 threads per workgroup were measured slower here: 25 vs 22 us at C4; the fused z pass gains from them)
#define UH_LINES(MB, P) hipLaunchKernelGGL((k_fft_lines<SIGN, MB, 256, P>), gr, dim3(256), ldsz, st, g, n, nkx, tiles)
  if (is_pow2(n)) { if (n <= 128) UH_LINES(2, true); else if (n <= 256) UH_LINES(4, true); else UH_LINES(8, true); }
  else { if (n <= 128) UH_LINES(2, false); else if (n <= 256) UH_LINES(4, false); else UH_LINES(8, false); }
#undef UH_LINES
}
// Sizes the LDS passes serve without asking for more than the 64 KB of dynamic LDS a launch gets by default: the y pass holds 16 lines
// (8 (n + 16 (n + 1)) bytes: 35 KB at 256, 69.8 KB at 512), the fused z pass 12 lines of nz (53 KB at 512), the row passes <= 25 KB up
// to 1024.  Anything else takes rocFFT + k_fcm_kspace.
// (an axis the mixed-radix passes serve: 2^a 3^b 5^c 7^d 11^e within [lo, hi])
static bool fft_axis_ok(int n, int lo, int hi) { int e[5]; return n >= lo && n <= hi && fft_factors(n, e); }
static bool fcm_custom_fft_usable(const FCM *f) {
  return f->customFFT && f->grid.cellDim.x % 2 == 0 && fft_axis_ok(f->grid.cellDim.x, 16, 512) && fft_axis_ok(f->grid.cellDim.y, 2, 256) && fft_axis_ok(f->grid.cellDim.z, 2, 512) &&
         f->planeReal == (size_t)f->nxpad * f->grid.cellDim.y * f->grid.cellDim.z;
}
// the plane kernel's LDS: twiddles + ny rows of nx / 2 + 1 complex; its butterfly budget: rows ny nh / 4 <= 2 x 1024, columns
// (nh + 1) ny / 4 <= 3 x 1024
static size_t fcm_plane_fft_lds(int nx, int ny) { return sizeof(float2) * ((size_t)nx + (size_t)ny + (size_t)ny * (nx / 2 + 1)); }
static bool fcm_plane_fft_usable(int nx, int ny) {
  // (per device: a process may run solvers on several GPUs)
  static int ldsLimits[64];
  static bool ldsKnown[64] = {false};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  if (!ldsKnown[dev]) {
    int v = 0;
    ldsLimits[dev] = hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess ? v : 0;
    if (getenv("UAMMD_FCM_NO_PLANE_FFT")) ldsLimits[dev] = 0;
    ldsKnown[dev] = true;
  }
  const int ldsLimit = ldsLimits[dev];
  const int nh = nx / 2;
  return nx >= 32 && ny >= 16 && (size_t)ny * nh / 4 <= 2 * 1024 && (size_t)(nh + 1) * ny / 4 <= 3 * 1024 &&
         fcm_plane_fft_lds(nx, ny) <= (size_t)std::min(ldsLimit, 96 * 1024);
}
// forward x and y transforms of the three real component grids, in place
static int fcm_fft_forward_xy(FCM *f, float *g, hipStream_t st) {
  const int nx = f->grid.cellDim.x, ny = f->grid.cellDim.y, nz = f->grid.cellDim.z, nkx = nx / 2 + 1, nh = nx / 2;
  if (fcm_plane_fft_usable(nx, ny)) {  // one pass: a whole plane per workgroup
    const size_t lds = fcm_plane_fft_lds(nx, ny);
    static bool attrSet[64] = {false};  // (a function attribute belongs to the device's copy of the kernel)
    int dev = 0;
    UH_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attrSet[dev]) {
      UH_CHECK(hipFuncSetAttribute((const void *)k_fft_xy_r2c_plane<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      UH_CHECK(hipFuncSetAttribute((const void *)k_fft_xy_r2c_plane<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      UH_CHECK(hipFuncSetAttribute((const void *)k_fft_xy_r2c_plane<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      if (dev >= 0 && dev < 64) attrSet[dev] = true;
    }
    if (is_pow2(nx) && is_pow2(ny)) hipLaunchKernelGGL(k_fft_xy_r2c_plane<true>, dim3(3 * nz), dim3(kPlaneThreads), lds, st, g, nx, ny);
    else if (lds <= 80 * 1024) hipLaunchKernelGGL((k_fft_xy_r2c_plane<false, true>), dim3(3 * nz), dim3(kPlaneThreads), lds, st, g, nx, ny);   // two planes per CU
    else hipLaunchKernelGGL(k_fft_xy_r2c_plane<false>, dim3(3 * nz), dim3(kPlaneThreads), lds, st, g, nx, ny);
    return 0;
  }
  const int rows = std::max(1, std::min(16, 2048 / nh)), nrows = 3 * ny * nz;
  if (is_pow2(nx))
    hipLaunchKernelGGL((k_fft_x_r2c<false, true>), dim3((nrows + rows - 1) / rows), dim3(kFftThreads), sizeof(float2) * (size_t)(nx + rows * (nh + 1)), st, g,
                       nx, nrows, rows, (const float *)nullptr, (const float *)nullptr, 0);
  else
    hipLaunchKernelGGL((k_fft_x_r2c<false, false>), dim3((nrows + rows - 1) / rows), dim3(kFftThreads), sizeof(float2) * (size_t)(nx + rows * (nh + 1)), st, g,
                       nx, nrows, rows, (const float *)nullptr, (const float *)nullptr, 0);
  const int tiles = (nkx + 15) / 16;
  fft_launch_lines<-1>((float2 *)g, ny, nkx, 3 * nz, st);
  (void)tiles;
  return 0;
}
// z transform + Fourier-space operator + inverse z transform on lines of the layout (planeC, zStride, nyl, y0) — see k_fft_z_fused
#ifndef UAMMD_ZG_THREADS
#define UAMMD_ZG_THREADS 256
#endif
static int fcm_fft_z_fused_launch(FCM *f, float2 *g, size_t planeC, size_t zStride, int nyl, int y0, int3 cells, real3f L, bool haveForce,
                                  float noisePrefactor, uint seed2, hipStream_t st) {
  const int nz = cells.z, nkx = cells.x / 2 + 1, lines = nyl * nkx;
  // tile of (ky, kx) nodes per workgroup: 3 tl nz complex in LDS and <= 6 radix-4 butterflies per thread and pass
  int ltl = nz <= 256 ? 3 : 2;  // measured at C4: 8 nodes x 512 threads
  if (f->zTileLog2 > 0 && sizeof(float2) * (size_t)(nz + 3 * (1 << f->zTileLog2) * (nz + 1)) <= 64 * 1024) ltl = f->zTileLog2;  // (tuning option)
  const int tlz = 1 << ltl;
  const size_t ldsz = sizeof(float2) * (size_t)(nz + 3 * tlz * (nz + 1));
  const dim3 gz((lines + tlz - 1) / tlz), bz(512);
#ifndef UAMMD_ZP_THREADS
#define UAMMD_ZP_THREADS 512
#endif
#define UH_ZFUSED(LT, P) hipLaunchKernelGGL((k_fft_z_fused<LT, UAMMD_ZP_THREADS, P>), gz, dim3(UAMMD_ZP_THREADS), ldsz, st, g, planeC, zStride, nyl, y0, nz, cells, L, f->par.viscosity, \
                                         haveForce, noisePrefactor, f->par.seed, seed2, f->pse)
  if (is_pow2(nz)) { if (ltl == 4) UH_ZFUSED(4, true); else if (ltl == 3) UH_ZFUSED(3, true); else UH_ZFUSED(2, true); }
  else {
    // (mixed-radix lines: a radix-12 or radix-9 pass has 3 tl nz / 12 .. / 9 butterflies — 216 and 288 at 108 — for 512 threads; with 256
    // threads per workgroup the passes fill their waves and the LDS, not the thread count, sets how many tiles a CU holds)
#define UH_ZFUSED_G(LT) hipLaunchKernelGGL((k_fft_z_fused<LT, UAMMD_ZG_THREADS, false>), gz, dim3(UAMMD_ZG_THREADS), ldsz, st, g, planeC, zStride, nyl, y0, nz, \
                                           cells, L, f->par.viscosity, haveForce, noisePrefactor, f->par.seed, seed2, f->pse)
    if (ltl == 4) UH_ZFUSED_G(4); else if (ltl == 3) UH_ZFUSED_G(3); else UH_ZFUSED_G(2);
#undef UH_ZFUSED_G
  }
#undef UH_ZFUSED
  return 0;
}
static int fcm_fft_z_operator_y(FCM *f, float *g, bool haveForce, float noisePrefactor, hipStream_t st) {
  const int nx = f->grid.cellDim.x, ny = f->grid.cellDim.y, nz = f->grid.cellDim.z, nkx = nx / 2 + 1;
  const real3f L{f->par.boxSize[0], f->par.boxSize[1], f->par.boxSize[2]};
  if (int e = fcm_fft_z_fused_launch(f, (float2 *)g, f->planeCplx, (size_t)ny * nkx, ny, 0, f->grid.cellDim, L, haveForce, noisePrefactor, f->seed2, st)) return e;
  fft_launch_lines<1>((float2 *)g, ny, nkx, 3 * nz, st);
  return 0;
}
// inverse x transform of the three components: into the planar real grids in place, or into the gather's interleaved float4 grid
// The gather's grid with 12 bytes per node: where the float4 grid would not stay in the Infinity Cache between the inverse x pass and the
// gather (> 128 MB: the same bound launch_gather_inter uses for its cache-resident forms).  UAMMD_FCM_PACKED_INTER=0: float4 everywhere.
static bool fcm_inter_packed(const FCM *f) {
  static const bool on = !(getenv("UAMMD_FCM_PACKED_INTER") && atoi(getenv("UAMMD_FCM_PACKED_INTER")) == 0);
  const size_t nodes = (size_t)f->grid.cellDim.x * f->grid.cellDim.y * f->grid.cellDim.z;
  return on && nodes * sizeof(float4) > ((size_t)128 << 20) && f->gatherPerWave >= 0;
}
static int fcm_fft_inverse_x(FCM *f, float *g, float4 *inter, hipStream_t st, bool packed = false) {
  const int nx = f->grid.cellDim.x, ny = f->grid.cellDim.y, nz = f->grid.cellDim.z, nh = nx / 2;
  const int rows = std::max(1, std::min(8, 2048 / (3 * nh))), nrows = ny * nz;
  if (is_pow2(nx) && is_pow2(ny))
    hipLaunchKernelGGL(k_fft_x_c2r<true>, dim3((nrows + rows - 1) / rows), dim3(kFftThreads), sizeof(float2) * (size_t)(nx + 3 * rows * (nh + 1)), st,
                       g, f->planeReal, (size_t)ny * f->nxpad, ny, nx, nrows, rows, inter, packed ? -1 : 0, (size_t)0);
  else
    hipLaunchKernelGGL(k_fft_x_c2r<false>, dim3((nrows + rows - 1) / rows), dim3(kFftThreads), sizeof(float2) * (size_t)(nx + 3 * rows * (nh + 1)), st,
                       g, f->planeReal, (size_t)ny * f->nxpad, ny, nx, nrows, rows, inter, packed ? -1 : 0, (size_t)0);
  return 0;
}

static int fcm_make_plans(FCM *f) {
  std::call_once(g_rocfft_once, []() { (void)rocfft_setup(); });
  const size_t nx = (size_t)f->grid.cellDim.x, ny = (size_t)f->grid.cellDim.y, nz = (size_t)f->grid.cellDim.z;
  const size_t nkx = nx / 2 + 1;
  const size_t lengths[3] = {nx, ny, nz};
  const size_t rstr[3] = {1, (size_t)f->nxpad, (size_t)f->nxpad * ny};
  const size_t cstr[3] = {1, nkx, nkx * ny};
  rocfft_plan_description d = nullptr;
  UH_ROCFFT(rocfft_plan_description_create(&d));
  UH_ROCFFT(rocfft_plan_description_set_data_layout(d, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved,
                                                     nullptr, nullptr, 3, rstr, f->planeReal, 3, cstr, f->planeCplx));
  UH_ROCFFT(rocfft_plan_create(&f->fwd, rocfft_placement_inplace, rocfft_transform_type_real_forward,
                               rocfft_precision_single, 3, lengths, 3, d));
  UH_ROCFFT(rocfft_plan_description_destroy(d));
  UH_ROCFFT(rocfft_plan_description_create(&d));
  UH_ROCFFT(rocfft_plan_description_set_data_layout(d, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real,
                                                     nullptr, nullptr, 3, cstr, f->planeCplx, 3, rstr, f->planeReal));
  UH_ROCFFT(rocfft_plan_create(&f->inv, rocfft_placement_inplace, rocfft_transform_type_real_inverse,
                               rocfft_precision_single, 3, lengths, 3, d));
  UH_ROCFFT(rocfft_plan_description_destroy(d));
  size_t wf = 0, wi = 0;
  UH_ROCFFT(rocfft_plan_get_work_buffer_size(f->fwd, &wf));
  UH_ROCFFT(rocfft_plan_get_work_buffer_size(f->inv, &wi));
  f->workBytes = wf > wi ? wf : wi;
  if (f->workBytes) {
    if (int e = f->work.reserve(f->workBytes)) return e;
  }
  UH_ROCFFT(rocfft_execution_info_create(&f->info));
  if (f->workBytes) UH_ROCFFT(rocfft_execution_info_set_work_buffer(f->info, f->work.ptr, f->workBytes));
  return 0;
}

// bins the particles by tile and fills the tile-sorted stencil origins / weights / forces (reused by spread and gather)
static int fcm_prepare_tiles(FCM *f, const float *d_pos, const float *d_force, int N, hipStream_t st, FcmPrep *out, bool positionsKept = false) {
  const int nt = f->ntiles.x * f->ntiles.y * f->ntiles.z;
  const int wstride = f->kern.support.x + f->kern.support.y + f->kern.support.z;
  if (f->slotPending) {  // (a slot-layout preparation that nobody claimed: this call rewrites the arrays it points into)
    f->slotPending = false;
    f->slotDirty = true;
  }
  f->slotSteps = 0;  // a sorted solve: the entries' order is fresh
  if (f->prepCapN < N) {
    UH_CHECK(hipStreamSynchronize(st));
    f->orderN = 0;
    if (int e = f->prepOrigin.reserve(sizeof(int4) * (size_t)N)) return e;
    if (int e = f->prepWeights.reserve(sizeof(float) * (size_t)wstride * N)) return e;
    if (int e = f->prepSorted.reserve(sizeof(float4) * (size_t)N)) return e;
    if (int e = f->prepTileOf.reserve(sizeof(int) * (size_t)N)) return e;
    if (int e = f->prepRank.reserve(sizeof(int) * (size_t)N)) return e;
    if (int e = f->prepTileCount.reserve(sizeof(int) * (size_t)nt)) return e;
    f->tileCountZero = false;
    if (int e = f->prepTileStart.reserve(sizeof(int) * ((size_t)nt + 1))) return e;
    f->prepCapN = N;
  }
  FcmPrep pr{(int4 *)f->prepOrigin.ptr, (float *)f->prepWeights.ptr, (float4 *)f->prepSorted.ptr,
             (int *)f->prepTileOf.ptr, (int *)f->prepRank.ptr, (int *)f->prepTileCount.ptr,
             (int *)f->prepTileStart.ptr, wstride, f->tdim};
  if (f->prepStreamSet && f->prepStream != st) {  // the handle's buffers (and the zeroed counters) are ordered by the stream of the last call
    UH_CHECK(hipStreamSynchronize(f->prepStream));
    f->tileCountZero = false;
  }
  // the previous uammd_fcm_step_euler_maruyama binned the positions it wrote (k_fcm_update_bin): its counts are in tileCount, the tiles
  // and ranks in tileOf / rank.  Used when the caller vouches that the array is untouched and it is the same array; dropped otherwise.
  const bool binned = f->binnedPending && positionsKept && f->binnedPos == (const void *)d_pos && f->binnedN == N && f->prepStream == st;
  if (f->binnedPending && !binned) f->tileCountZero = false;
  f->binnedPending = false;
  f->prepStream = st;
  f->prepStreamSet = true;
  if (!f->tileCountZero) {  // first use of the buffer; afterwards k_fcm_tile_scan hands the counters back zeroed
    UH_CHECK(hipMemsetAsync(pr.tileCount, 0, sizeof(int) * (size_t)nt, st));
    f->tileCountZero = true;
  }
  if (!binned)
    hipLaunchKernelGGL(k_fcm_bin_count, dim3((N + 255) / 256), dim3(256), 0, st, (const float4 *)d_pos, N, f->grid,
                       f->ntiles, pr, f->kern.support);
  hipLaunchKernelGGL(k_fcm_tile_scan, dim3(1), dim3(1024), 0, st, pr.tileCount, nt, pr.tileStart);
#define UH_PREPARE(K)                                                                                          \
  case K:                                                                                                      \
    if (N <= kPrepLanesUpTo)                                                                                   \
      hipLaunchKernelGGL((k_fcm_prepare<K, kPrepLanes>), dim3((N * kPrepLanes + 255) / 256), dim3(256), 0, st, (const float4 *)d_pos, \
                         (const float4 *)d_force, N, f->grid, f->kern, pr);                                    \
    else                                                                                                       \
      hipLaunchKernelGGL((k_fcm_prepare<K, 1>), dim3((N + 255) / 256), dim3(256), 0, st, (const float4 *)d_pos, \
                         (const float4 *)d_force, N, f->grid, f->kern, pr);                                    \
    break;
  switch (f->kern.kind) {
    UH_PREPARE(kKernelGaussian) UH_PREPARE(kKernelPeskin3) UH_PREPARE(kKernelPeskin4) UH_PREPARE(kKernelConstant)
    UH_PREPARE(kKernelBarnettMagland) UH_PREPARE(kKernelSixPoint)
    default: set_last_error("fcm: window kind %d has no spreading kernel", f->kern.kind); return -3;
  }
#undef UH_PREPARE
  f->orderN = N;  // (origin[slot].w = the particle of compact slot `slot`: the entry order of the slot layout)
  *out = pr;
  return 0;
}

// tile-owned spreading is possible when the grid is a whole number (>= 3) of tiles per axis and a stencil only
// reaches the adjacent tiles
// (a particle is binned by ITS cell and its stencil reaches support / 2 (+ 1 with the shift of even supports) nodes either way: it stays
// within the adjacent tiles when support <= 2 (T - 1)).  The tile edge T is chosen per axis: the largest divisor of the axis in 4..8
// that satisfies this — 8 for power-of-two grids, 6 for the 108^3 grid of the PSE far field at psi = 0.5, 5 for 100, 7 for 84.  The
// kernels' LDS / matrix layouts are sized for 8; a shorter tile leaves rows and columns unused (and unwritten).
static bool fcm_tiles_usable(const int cells[3], const int support[3], int3 *tdim, bool only8 = false) {
  static const bool nine = getenv("UAMMD_FCM_NO_TILE9") == nullptr;   // (A/B runs: the 9-node edge of round 6 off)
  int T[3];
  for (int a = 0; a < 3; ++a) {
    T[a] = 0;
    // 8 where it divides the axis; then 9 (the spread's second instantiation: 108, 162, 180, 198, 270 ...) before the shorter edges
    const int order[6] = {8, 9, 7, 6, 5, 4};
    for (int k = 0; k < (only8 ? 1 : 6); ++k) {
      const int t = order[k];
      if (t == 9 && !nine) continue;
      if (cells[a] % t == 0 && cells[a] / t >= 3 && support[a] <= 2 * (t - 1)) { T[a] = t; break; }
    }
    if (!T[a]) return false;
  }
  *tdim = make_int3(T[0], T[1], T[2]);
  return true;
}

// ---- z-slab decomposed solver (SURVEY §8e path B; BASELINE configs[4]) ------------------------------------------
// One rank owns nzLocal xy-planes of the real grid (plus `halo` planes each side that receive the stencils of its own
// particles) and, after the transpose, nyLocal y-rows of the Fourier grid with all of z.  Layouts:
//   real     [zloc (nzLocal + 2 halo)][c (3)][y (ny)][x (nxpad)]   -> a halo slice is one contiguous block
//   cplx xy  [zl (nzLocal)][c][y (ny)][kx (nkx)]                     the owned planes after the in-place batched 2-D R2C
//   cplx z   [z (nz)][c][yl (nyLocal)][kx]                           what the all-to-all delivers; 1-D FFT along z, stride 3*nyl*nkx
// The exchanges themselves (halo add, transpose, halo fill) are NOT in here: they are RCCL calls made by the host layer.
struct FCMSlab {
  FCM loc;  // local window: grid (nx, ny, nzLocal + 2 halo), tile machinery, kernel
  int3 cells{0, 0, 0};
  real3f L{0, 0, 0};
  int nzl = 0, z0 = 0, halo = 0, nyl = 0, y0 = 0, nkx = 0;
  rocfft_plan fwdXY = nullptr, invXY = nullptr, fwdZ = nullptr, invZ = nullptr;
  ~FCMSlab() {
    for (rocfft_plan p : {fwdXY, invXY, fwdZ, invZ})
      if (p) rocfft_plan_destroy(p);
  }
};

static bool fcm_slab_custom_fft(const FCMSlab *s) {
  return s->loc.customFFT && s->cells.x % 2 == 0 && fft_axis_ok(s->cells.x, 16, 512) && fft_axis_ok(s->cells.y, 2, 256);
}
static int fcm_slab_make_plans(FCMSlab *s) {
  std::call_once(g_rocfft_once, []() { (void)rocfft_setup(); });
  FCM *f = &s->loc;
  const size_t nx = (size_t)s->cells.x, ny = (size_t)s->cells.y, nz = (size_t)s->cells.z, nkx = (size_t)s->nkx;
  const size_t len2[2] = {nx, ny};
  const size_t rstr[2] = {1, (size_t)f->nxpad}, cstr[2] = {1, nkx};
  const size_t rdist = (size_t)f->nxpad * ny, cdist = nkx * ny, batchXY = 3 * (size_t)s->nzl;
  rocfft_plan_description d = nullptr;
  UH_ROCFFT(rocfft_plan_description_create(&d));
  UH_ROCFFT(rocfft_plan_description_set_data_layout(d, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved,
                                                     nullptr, nullptr, 2, rstr, rdist, 2, cstr, cdist));
  UH_ROCFFT(rocfft_plan_create(&s->fwdXY, rocfft_placement_inplace, rocfft_transform_type_real_forward,
                               rocfft_precision_single, 2, len2, batchXY, d));
  UH_ROCFFT(rocfft_plan_description_destroy(d));
  UH_ROCFFT(rocfft_plan_description_create(&d));
  UH_ROCFFT(rocfft_plan_description_set_data_layout(d, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real,
                                                     nullptr, nullptr, 2, cstr, cdist, 2, rstr, rdist));
  UH_ROCFFT(rocfft_plan_create(&s->invXY, rocfft_placement_inplace, rocfft_transform_type_real_inverse,
                               rocfft_precision_single, 2, len2, batchXY, d));
  UH_ROCFFT(rocfft_plan_description_destroy(d));
  const size_t S = 3 * (size_t)s->nyl * nkx;
  const size_t len1[1] = {nz}, zstr[1] = {S};
  for (int inverse = 0; inverse < 2; ++inverse) {
    UH_ROCFFT(rocfft_plan_description_create(&d));
    UH_ROCFFT(rocfft_plan_description_set_data_layout(d, rocfft_array_type_complex_interleaved,
                                                       rocfft_array_type_complex_interleaved, nullptr, nullptr, 1, zstr, 1, 1,
                                                       zstr, 1));
    UH_ROCFFT(rocfft_plan_create(inverse ? &s->invZ : &s->fwdZ, rocfft_placement_inplace,
                                 inverse ? rocfft_transform_type_complex_inverse : rocfft_transform_type_complex_forward,
                                 rocfft_precision_single, 1, len1, S, d));
    UH_ROCFFT(rocfft_plan_description_destroy(d));
  }
  size_t w = 0;
  for (rocfft_plan p : {s->fwdXY, s->invXY, s->fwdZ, s->invZ}) {
    size_t wi = 0;
    UH_ROCFFT(rocfft_plan_get_work_buffer_size(p, &wi));
    w = wi > w ? wi : w;
  }
  f->workBytes = w;
  if (w) {
    if (int e = f->work.reserve(w)) return e;
  }
  UH_ROCFFT(rocfft_execution_info_create(&f->info));
  if (w) UH_ROCFFT(rocfft_execution_info_set_work_buffer(f->info, f->work.ptr, w));
  return 0;
}

}  // namespace uammd_hip

using namespace uammd_hip;

extern "C" {

int uammd_fcm_create(const uammd_fcm_parameters *par, uammd_fcm **out) {
  if (!par || !out) { set_last_error("uammd_fcm_create: null argument"); return -1; }
  // FCM_impl ctor checks, FCM_impl.cuh:56-92
  if (!(par->boxSize[0] > 0) || !(par->boxSize[1] > 0) || !(par->boxSize[2] > 0)) {
    set_last_error("FCM_impl requires a valid box");
    return -2;
  }
  if (par->cells[0] <= 0 || par->cells[1] <= 0 || par->cells[2] <= 0) {
    set_last_error("FCM_impl requires a valid grid dimension");
    return -2;
  }
  if (par->kernel.kind != UAMMD_IBM_KERNEL_GAUSSIAN && par->kernel.kind != UAMMD_IBM_KERNEL_PESKIN3 &&
      par->kernel.kind != UAMMD_IBM_KERNEL_PESKIN4 && par->kernel.kind != UAMMD_IBM_KERNEL_BARNETT_MAGLAND &&
      par->kernel.kind != UAMMD_IBM_KERNEL_SIXPOINT) {
    set_last_error("FCM_impl requires instances of the spreading kernels");
    return -2;
  }
  for (int a = 0; a < 3; ++a) {
    if (par->kernel.support[a] < 1 || par->kernel.support[a] > kMaxSupport) {
      set_last_error("uammd_fcm_create: kernel support %d outside [1, %d]", par->kernel.support[a], kMaxSupport);
      return -2;
    }
    // A support that is not smaller than the grid: the reference logs an ERROR and goes on (BDHI_FCM.cuh:58-64) — its own acceptance
    // program sweeps through such boxes (test/BDHI/FCM/FCM.cu:299-331, L from 2.1 distances) — with stencils that wrap around the box and
    // meet nodes more than once.  Followed (the host layer prints the reference's line) as far as ONE wrap per axis reaches, which is
    // as far as the reference's own Grid::pbc_cell goes (utils/Grid.cuh: one +- n): the grid must hold half a support.
    if (2 * par->cells[a] < par->kernel.support[a] + 1) {
      set_last_error("[BDHI::FCM] Kernel support is too big, try lowering the tolerance or increasing the box size!.");
      return -2;
    }
  }
  FCM *f = new FCM();
  f->par = *par;
  const int periodic[3] = {1, 1, 1};
  const BoxT<float> box = make_box<float>(par->boxSize, periodic);
  f->grid = make_grid<float>(box, make_int3(par->cells[0], par->cells[1], par->cells[2]));
  f->kern = to_dev(par->kernel);
  f->nxpad = 2 * (par->cells[0] / 2 + 1);
  f->planeReal = (size_t)f->nxpad * par->cells[1] * par->cells[2];
  f->planeCplx = f->planeReal / 2;
  if (int e = f->gridBuf.reserve(sizeof(float) * 3 * f->planeReal)) { delete f; return e; }
  f->useTiles = fcm_tiles_usable(par->cells, par->kernel.support, &f->tdim);
  f->ntiles = make_int3(par->cells[0] / f->tdim.x, par->cells[1] / f->tdim.y, par->cells[2] / f->tdim.z);
  if (int e = fcm_make_plans(f)) { delete f; return e; }
  *out = reinterpret_cast<uammd_fcm *>(f);
  return 0;
}

int uammd_fcm_destroy(uammd_fcm *h) {
  delete reinterpret_cast<FCM *>(h);
  return 0;
}

int uammd_fcm_get_seed2(uammd_fcm *h, unsigned int *seed2) {
  if (!h || !seed2) { set_last_error("uammd_fcm_get_seed2: null argument"); return -1; }
  *seed2 = reinterpret_cast<FCM *>(h)->seed2;
  return 0;
}
int uammd_fcm_set_seed2(uammd_fcm *h, unsigned int seed2) {
  if (!h) { set_last_error("uammd_fcm_set_seed2: null handle"); return -1; }
  reinterpret_cast<FCM *>(h)->seed2 = seed2;
  return 0;
}

int uammd_fcm_export_fourier(uammd_fcm *h, float *d_out6, void *stream) {
  if (!h || !d_out6) { set_last_error("uammd_fcm_export_fourier: null argument"); return -1; }
  FCM *f = reinterpret_cast<FCM *>(h);
  const int total = (int)f->planeCplx;
  hipLaunchKernelGGL(k_fcm_export, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const float2 *)f->gridBuf.ptr, f->planeCplx, total, d_out6);
  UH_CHECK(hipGetLastError());
  return 0;
}

// stage: 0 = full pipeline; 1 = stop after spread+FFT+k-space (the Fourier grid can then be exported)
// half: 0 = the whole solve; 1 = its first half (binning, stencils, spreading, the forward x / y transforms), 2 = the second (z transform,
// operator and noise, inverse transforms, gather) of a solve whose first half was queued on the same stream with the same arguments —
// for a caller with other work to queue in between (uammd_pse_far_displacements_half)
// The stencils of a solve, the cheapest way that applies: (1) the previous step's update kernel prepared it (k_fcm_step_prep) and the
// caller vouches that the array is untouched; (2) nobody prepared it but an earlier solve of these N particles left an entry order:
// the same kernel without the update, ONE launch instead of k_fcm_bin_count -> k_fcm_tile_scan -> k_fcm_prepare (every slotRefresh-th
// solve goes through the sorted layout again: the entries' order is what keeps the gather's windows local); (3) the sorted layout.
// *slots says which layout `pr` describes (the spread kernel's template argument).
static bool fcm_step_prep_launch(FCM *f, float *d_pos, const float *v, int N, float dt, hipStream_t st, int *rc);
// the spread of the last slot-layout solve reported how many tile changes its entry sequence had (FcmPrep::slotFlag[1]; read without
// a wait: a solve or two late is soon enough)
static bool fcm_entries_scrambled(FCM *f, int N) {
  if (!f->slotFlagHost) return false;
  const int nt = f->ntiles.x * f->ntiles.y * f->ntiles.z;
  const long changes = 64L * ((volatile int *)f->slotFlagHost)[1];   // (one wave in 64 counted)
  if (changes <= (long)std::max(4 * nt, N / 4)) return false;
  f->slotFlagHost[1] = 0;
  return true;
}
static int fcm_prepare_best(FCM *f, const float *d_pos, const float *d_force, int N, hipStream_t st, bool positionsKept, FcmPrep *out,
                            bool *slotsOut) {
  bool slots = f->slotPending && !f->tileGather && positionsKept && f->slotPos == (const void *)d_pos && f->slotN == N && f->prepStreamSet &&
               f->prepStream == st;
  // (a pending preparation is used whatever its order: it is correct, and the step's update kernel looks at the order itself)
  const bool scrambled = !slots && fcm_entries_scrambled(f, N);
  if (f->slotPending && !slots) f->slotDirty = true;  // dropped: its counters are garbage now
  f->slotPending = false;
  if (!slots && !scrambled && f->slotsEnabled && !f->tileGather && f->orderN == N && f->prepStreamSet && f->prepStream == st) {  // (k_fcm_gather_tile walks compact tile ranges)
    int rc = 0;
    if (fcm_step_prep_launch(f, const_cast<float *>(d_pos), nullptr, N, 0.0f, st, &rc)) {
      if (rc) return rc;
      slots = true;
      f->slotPending = false;
    }
  }
  *slotsOut = slots;
  if (!slots) return fcm_prepare_tiles(f, d_pos, d_force, N, st, out, positionsKept);
  const int nt = f->ntiles.x * f->ntiles.y * f->ntiles.z;
  int *counts = (int *)f->prepSlotCount.ptr;
  *out = FcmPrep{(int4 *)f->prepOrigin.ptr, (float *)f->prepWeights.ptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                 f->kern.support.x + f->kern.support.y + f->kern.support.z, f->tdim, (unsigned long long *)f->prepRec.ptr,
                 (int *)f->prepOvfTile.ptr, f->slotCap, counts + (size_t)f->slotParity * (nt + 2),
                 counts + (size_t)(f->slotParity ^ 1) * (nt + 2), f->slotFlagDev, (const float4 *)d_force};
  // (a compact binning that k_fcm_update_bin left pending — the step fell back to it, e.g. because the entries' order was reported scrambled —
  // is dropped here: its counts are still in tileCount, which the next sorted solve must not take for zeroed.  Found in round 6 as a memory
  // fault of test_fcm_step_bins_ahead run on its own: ranks from stale counts sent k_fcm_prepare's rows past the arrays)
  if (f->binnedPending) f->tileCountZero = false;
  f->binnedPending = false;
  // (no forces: no spread to hand the other parity's counters back zeroed)
  if (!d_force) UH_CHECK(hipMemsetAsync(out->slotCountNext, 0, sizeof(int) * (size_t)(nt + 2), st));
  return 0;
}

static int fcm_displacements_impl(uammd_fcm *h, const float *d_pos, const float *d_force, int N, float temperature,
                                  float prefactor, float *d_linearVelocity, int stage, void *stream, bool positionsKept, int half = 0) {
  if (!h) { set_last_error("uammd_fcm_displacements: null argument"); return -1; }
  if (N <= 0) return 0;  // nothing to move (an empty ParticleData has no arrays to point to)
  if (!d_pos || (!d_linearVelocity && stage == 0)) { set_last_error("uammd_fcm_displacements: null argument"); return -1; }
  FCM *f = reinterpret_cast<FCM *>(h);
  hipStream_t st = (hipStream_t)stream;
  float *g = (float *)f->gridBuf.ptr;
  const dim3 gp((N + 3) / 4), bp(256);
  const FastDiv dsx = make_fastdiv(f->kern.support.x), dsxy = make_fastdiv(f->kern.support.x * f->kern.support.y);
  UH_ROCFFT(rocfft_execution_info_set_stream(f->info, (void *)st));
  void *bufs[1] = {g};
  const size_t zs = (size_t)f->nxpad * f->grid.cellDim.y;  // planar component grids: z stride = one xy plane
  const bool tiles = f->useTiles && !f->forceAtomicSpread;
  const bool custom = stage == 0 && fcm_custom_fft_usable(f);  // (stage 1 exports the Fourier grid, which the fused z pass never stores)
  FcmPrep pr{};
  bool slots = false;
  if (half != 2) f->halfPending = false;   // (a first half whose second never came is forgotten by the next solve)
  if (half == 2) {
    if (!f->halfPending || f->halfN != N) { set_last_error("uammd_fcm: the second half of a solve without its first"); return -1; }
    pr = f->halfPrep;
    f->halfPending = false;
  } else if (tiles) {
    if (int e = fcm_prepare_best(f, d_pos, d_force, N, st, positionsKept, &pr, &slots)) return e;
  }
  if (half != 2) f->lastSolveSlots = slots;  // (the second half of a solve reads what the first half prepared)
  if (d_force && half != 2) {
    if (tiles) {
      const int nt = f->ntiles.x * f->ntiles.y * f->ntiles.z;
      const int ww = spread_weight_words(N, f->ntiles, f->kern.support, f->tdim);
      const int sw = spread_waves(N, f->ntiles, f->kern.support, f->tdim, f->spreadWaves);
      const int edge = spread_edge(f->tdim);
      const dim3 sg(nt), sb(64 * sw);
      const size_t lds = spread_lds_bytes(ww, sw, edge);
#define UH_SPREAD(W, S, E) hipLaunchKernelGGL((k_fcm_spread_tile<W, S, E>), sg, sb, lds, st, g, f->grid.cellDim, f->nxpad, f->planeReal, zs, f->kern.support, f->ntiles, pr, ww)
      // (SPEC: see the kernel — two waves per tile, exactly eight source tiles, slots deep enough)
      static const bool specOn = getenv("UAMMD_FCM_NO_SPEC") == nullptr;
      const bool eight = (f->kern.support.x - 2 + f->tdim.x) / f->tdim.x == 1 && (f->kern.support.y - 2 + f->tdim.y) / f->tdim.y == 1 &&
                         (f->kern.support.z - 2 + f->tdim.z) / f->tdim.z == 1;
      // (dense tiles, four waves, 8 x 32 slots: measured at C4 with no gain — 0.1578 / 0.1580 / 0.1593 against 0.1578 / 0.1589 / 0.1594 ms —
      // since the records' round trip hides behind the other tiles' matrix steps there; kept behind UAMMD_FCM_SPEC_DENSE=1 for A/B runs)
      static const bool specDense = getenv("UAMMD_FCM_SPEC_DENSE") != nullptr;
      // (the round lists up to one record per thread at once: the list and its weights must hold a workgroup's worth — capEntries >= 64 W)
      const bool spec = specOn && slots && eight && pr.cap >= 8 * sw && ww / (3 * edge) >= 64 * sw && (sw == 2 || specDense);
#define UH_SPREAD_SPEC(W, E) hipLaunchKernelGGL((k_fcm_spread_tile<W, true, E, true>), sg, sb, lds, st, g, f->grid.cellDim, f->nxpad, f->planeReal, zs, f->kern.support, f->ntiles, pr, ww)
      if (spec) {
        if (sw == 2) { if (edge == kTileMax) UH_SPREAD_SPEC(2, kTileMax); else UH_SPREAD_SPEC(2, kTile); }
        else { if (edge == kTileMax) UH_SPREAD_SPEC(4, kTileMax); else UH_SPREAD_SPEC(4, kTile); }
      }
      else if (edge == kTileMax) {
        if (sw == 2) { if (slots) UH_SPREAD(2, true, kTileMax); else UH_SPREAD(2, false, kTileMax); }
        else { if (slots) UH_SPREAD(4, true, kTileMax); else UH_SPREAD(4, false, kTileMax); }
      } else {
        if (sw == 2) { if (slots) UH_SPREAD(2, true, kTile); else UH_SPREAD(2, false, kTile); }
        else { if (slots) UH_SPREAD(4, true, kTile); else UH_SPREAD(4, false, kTile); }
      }
#undef UH_SPREAD
#undef UH_SPREAD_SPEC
    } else {
      UH_CHECK(hipMemsetAsync(g, 0, sizeof(float) * 3 * f->planeReal, st));
      hipLaunchKernelGGL((k_fcm_ibm<true>), gp, bp, 0, st, (const float4 *)d_pos, (const float4 *)d_force,
                         (float *)nullptr, g, N, f->grid, f->nxpad, f->planeReal, zs, f->kern, dsx, dsxy, false);
    }
    if (custom) { if (int e = fcm_fft_forward_xy(f, g, st)) return e; }
    else UH_ROCFFT(rocfft_execute(f->fwd, bufs, nullptr, f->info));
  }
  if (half == 1) {
    f->halfPrep = pr;
    f->halfN = N;
    f->halfPending = true;
    UH_CHECK(hipGetLastError());
    return 0;
  }
  float noisePrefactor = 0.0f;
  if (temperature > 0.0f) {  // addBrownianNoise, FCM_impl.cuh:514-542
    if (!f->pse.on) f->seed2++;  // PSE: the caller supplies seed2 (System::rng().next32() per call, FarField.cuh:499)
    const float dV = f->grid.cellVolume;
    const float fourierNormalization =
        (float)(1.0 / ((double)f->grid.cellDim.x * f->grid.cellDim.y * f->grid.cellDim.z));
    noisePrefactor = prefactor * sqrtf(fourierNormalization * 2 * temperature / dV);
    if (f->pse.on) noisePrefactor = prefactor * sqrtf(2 * temperature / dV);  // FarField.cuh:503: the 1/N lives in B
  }
  const int total = (int)f->planeCplx;
  const KLayout lay{f->grid.cellDim.y, 0, f->planeCplx, (size_t)(f->grid.cellDim.x / 2 + 1) * f->grid.cellDim.y,
                    make_fastdiv(f->grid.cellDim.x / 2 + 1), make_fastdiv(f->grid.cellDim.y)};
  if (custom) {
    if (int e = fcm_fft_z_operator_y(f, g, d_force != nullptr, noisePrefactor, st)) return e;
  } else {
    hipLaunchKernelGGL(k_fcm_kspace, dim3((total + 255) / 256), dim3(256), 0, st, (float2 *)g, lay,
                       f->grid.cellDim, real3f{f->par.boxSize[0], f->par.boxSize[1], f->par.boxSize[2]}, f->par.viscosity,
                       d_force != nullptr, noisePrefactor, f->par.seed, f->seed2, f->pse);
    if (stage == 1) { UH_CHECK(hipGetLastError()); return 0; }
    UH_ROCFFT(rocfft_execute(f->inv, bufs, nullptr, f->info));
  }
  const bool interPath = tiles && f->interGather && !(tiles && (f->kern.support.x <= 6 && f->kern.support.y <= 6 && f->kern.support.z <= 6) && f->tileGather);
  if (custom && !interPath) { if (int e = fcm_fft_inverse_x(f, g, nullptr, st)) return e; }
  const bool smallSupport = f->kern.support.x <= 6 && f->kern.support.y <= 6 && f->kern.support.z <= 6;
  if (tiles && smallSupport && f->tileGather)
    hipLaunchKernelGGL(k_fcm_gather_tile, dim3(f->ntiles.x * f->ntiles.y * f->ntiles.z), bp, 0, st, d_linearVelocity,
                       (const float *)g, f->grid.cellDim, f->nxpad, f->planeReal, zs, f->kern.support, f->ntiles,
                       f->grid.cellVolume, dsx, dsxy, pr, f->accumulate);
  else if (tiles && f->interGather) {
    const size_t nodes = (size_t)f->grid.cellDim.x * f->grid.cellDim.y * f->grid.cellDim.z;
    if (int e = f->interBuf.reserve(sizeof(float4) * nodes)) return e;
    const bool packed = custom && fcm_inter_packed(f);
    if (custom) {  // the inverse x transform writes the interleaved grid itself
      if (int e = fcm_fft_inverse_x(f, g, (float4 *)f->interBuf.ptr, st, packed)) return e;
    } else
      hipLaunchKernelGGL(k_fcm_interleave, dim3((unsigned)((nodes + 255) / 256)), bp, 0, st, (const float *)g, f->grid.cellDim, f->nxpad,
                         f->planeReal, zs, (float4 *)f->interBuf.ptr);
    launch_gather_inter(st, d_linearVelocity, (const float4 *)f->interBuf.ptr, N, f->grid.cellDim, f->kern.support, f->grid.cellVolume,
                        dsx, dsxy, pr, f->accumulate, f->gatherPerWave, packed);
  } else if (tiles)
    hipLaunchKernelGGL(k_fcm_gather_prep, gp, bp, 0, st, d_linearVelocity, (const float *)g, N, f->grid.cellDim, f->nxpad,
                       f->planeReal, zs, f->kern.support, f->grid.cellVolume, dsx, dsxy, pr, f->accumulate);
  else
    hipLaunchKernelGGL((k_fcm_ibm<false>), gp, bp, 0, st, (const float4 *)d_pos, (const float4 *)nullptr,
                       d_linearVelocity, g, N, f->grid, f->nxpad, f->planeReal, zs, f->kern, dsx, dsxy, f->accumulate);
  UH_CHECK(hipGetLastError());
  return 0;
}

int uammd_fcm_displacements_staged(uammd_fcm *h, const float *d_pos, const float *d_force, int N, float temperature,
                                   float prefactor, float *d_linearVelocity, int stage, void *stream) {
  return fcm_displacements_impl(h, d_pos, d_force, N, temperature, prefactor, d_linearVelocity, stage, stream, false);
}

// The slot layout's preparation as the step's update kernel (k_fcm_step_prep).  Returns false when the slot layout is not to be used for
// the next solve (then the caller runs the compact layout's update kernel); true with *rc set otherwise.
static bool fcm_step_prep_launch(FCM *f, float *d_pos, const float *v, int N, float dt, hipStream_t st, int *rc) {
  *rc = 0;
  if (f->slotFlagHost && f->slotFlagHost[0]) {  // the spread met a long overflow list: this system is too clustered for fixed capacities
    f->slotsEnabled = false;
    return false;
  }
  if (f->slotSteps >= f->slotRefresh) { f->slotSteps = 0; return false; }
  if (v && fcm_entries_scrambled(f, N)) return false;   // (the step's update kernel: the compact layout's, so that the next solve sorts)
  const int nt = f->ntiles.x * f->ntiles.y * f->ntiles.z;
  const double mean = (double)N / nt;
  const int cap = std::max(32, ((int)(3.0 * mean) + 16 + 7) & ~7);
  const size_t recBytes = sizeof(unsigned long long) * ((size_t)nt * cap + (size_t)N);
  if (recBytes > ((size_t)1 << 30) || N >= (1 << 21)) return false;  // (a record holds entry and particle in 21 bits each)
  auto fail = [&](int e) { *rc = e; return true; };
  if (!f->slotFlagHost) {
    if (hipHostMalloc((void **)&f->slotFlagHost, 64, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); f->slotsEnabled = false; return false; }
    f->slotFlagHost[0] = 0;
    f->slotFlagHost[1] = 0;
    if (hipHostGetDevicePointer((void **)&f->slotFlagDev, f->slotFlagHost, 0) != hipSuccess) { (void)hipGetLastError(); f->slotsEnabled = false; return false; }
  }
  if (f->prepRec.cap < recBytes || f->prepOvfTile.cap < sizeof(int) * (size_t)N || f->slotCap != cap || f->prepSlotCount.cap < sizeof(int) * 2 * ((size_t)nt + 2)) {
    if (hipStreamSynchronize(st) != hipSuccess) return fail(-1);
    if (int e = f->prepRec.reserve(recBytes)) return fail(e);
    if (int e = f->prepOvfTile.reserve(sizeof(int) * (size_t)N)) return fail(e);
    if (int e = f->prepSlotCount.reserve(sizeof(int) * 2 * ((size_t)nt + 2))) return fail(e);
    f->slotCap = cap;
    f->slotDirty = true;
  }
  int *counts = (int *)f->prepSlotCount.ptr;
  int parity;
  if (f->lastSolveSlots && !f->slotDirty) parity = f->slotParity ^ 1;  // (the spread of the solve above zeroed these)
  else {
    if (hipMemsetAsync(counts, 0, sizeof(int) * 2 * ((size_t)nt + 2), st) != hipSuccess) return fail(-1);
    f->slotDirty = false;
    parity = 0;
  }
  FcmPrep pr{};
  pr.origin = (int4 *)f->prepOrigin.ptr;   // entry -> particle from the solve above; rewritten entry by entry
  pr.weights = (float *)f->prepWeights.ptr;
  pr.wstride = f->kern.support.x + f->kern.support.y + f->kern.support.z;
  pr.tdim = f->tdim;
  pr.rec = (unsigned long long *)f->prepRec.ptr;
  pr.ovfTile = (int *)f->prepOvfTile.ptr;
  pr.cap = cap;
  pr.slotCount = counts + (size_t)parity * (nt + 2);
#define UH_STEP_PREP(K)                                                                                                                        \
  case K:                                                                                                                                      \
    if (N <= kPrepLanesUpTo)                                                                                                                   \
      hipLaunchKernelGGL((k_fcm_step_prep<K, kPrepLanes>), dim3((unsigned)(((size_t)N * kPrepLanes + 255) / 256)), dim3(256), 0, st, (float4 *)d_pos, v, N, \
                         dt, f->grid, f->kern, f->ntiles, pr, N);                                                                              \
    else                                                                                                                                       \
      hipLaunchKernelGGL((k_fcm_step_prep<K, 1>), dim3((N + 255) / 256), dim3(256), 0, st, (float4 *)d_pos, v, N, dt, f->grid, f->kern,         \
                         f->ntiles, pr, N);                                                                                                    \
    break;
  switch (f->kern.kind) {
    UH_STEP_PREP(kKernelGaussian) UH_STEP_PREP(kKernelPeskin3) UH_STEP_PREP(kKernelPeskin4) UH_STEP_PREP(kKernelConstant)
    UH_STEP_PREP(kKernelBarnettMagland) UH_STEP_PREP(kKernelSixPoint)
    default: return false;
  }
#undef UH_STEP_PREP
  if (hipGetLastError() != hipSuccess) return fail(-1);
  f->slotParity = parity;
  f->slotPending = true;
  f->slotPos = (const void *)d_pos;
  f->slotN = N;
  f->slotSteps++;
  if (f->binnedPending) f->tileCountZero = false;   // (see fcm_prepare_best)
  f->binnedPending = false;
  return true;
}

// BDHI::FCMIntegrator::forwardTime without torques (BDHI_FCM.cu:67-119): v = M F + sqrt(2 T / dt) dW as uammd_fcm_displacements, then
// integrateEulerMaruyamaD's pos += v dt — by a kernel that also bins the positions it writes, so that the next call (told that the array
// is untouched) starts at the tile scan: one launch per step less.  d_linearVelocity may be NULL; positions move in place.
int uammd_fcm_step_euler_maruyama(uammd_fcm *h, float *d_pos, const float *d_force, int N, float temperature, float prefactor,
                                  float dt, float *d_linearVelocity, int flags, void *stream) {
  if (!h || (N > 0 && !d_pos)) { set_last_error("uammd_fcm_step_euler_maruyama: null argument"); return -1; }
  if (N <= 0) return 0;
  FCM *f = reinterpret_cast<FCM *>(h);
  hipStream_t st = (hipStream_t)stream;
  float *v = d_linearVelocity;
  if (!v) {
    if (f->emVelN < N) {
      UH_CHECK(hipStreamSynchronize(st));
      if (int e = f->emVel.reserve(sizeof(float) * 3 * (size_t)N)) return e;
      f->emVelN = N;
    }
    v = (float *)f->emVel.ptr;
  }
  if (int e = fcm_displacements_impl(h, d_pos, d_force, N, temperature, prefactor, v, 0, stream, (flags & UAMMD_FCM_STEP_POSITIONS_KEPT) != 0))
    return e;
  const bool tiles = f->useTiles && !f->forceAtomicSpread;
  const bool bin = tiles && f->emBin && f->prepCapN >= N;  // (the tile scan of the solve above left the counters at zero)
  if (bin && f->slotsEnabled) {
    int rc = 0;
    if (fcm_step_prep_launch(f, d_pos, v, N, dt, st, &rc)) return rc;  // (launched, or failed: done either way)
  }
  FcmPrep pr{};
  if (bin) {
    pr.tileOf = (int *)f->prepTileOf.ptr; pr.rank = (int *)f->prepRank.ptr; pr.tileCount = (int *)f->prepTileCount.ptr;
    pr.origin = (int4 *)f->prepOrigin.ptr;   // (the stencils of the solve above: slot -> particle)
    pr.tdim = f->tdim;
    // the binning counts on top of tileCount: zero after a sorted solve's tile scan, but NOT after a slot-layout solve that dropped an
    // earlier unclaimed binning (fcm_prepare_best) — then they are zeroed here.  (Round 6, found by tests/test_gpu_ibm_fcm.py::
    // test_fcm_step_random_call_sequences: ranks on top of stale counts, then a sorted solve that claimed them: wrong stencil rows.)
    if (!f->tileCountZero) {
      UH_CHECK(hipMemsetAsync(pr.tileCount, 0, sizeof(int) * (size_t)(f->ntiles.x * f->ntiles.y * f->ntiles.z), st));
      f->tileCountZero = true;
    }
  }
  hipLaunchKernelGGL(k_fcm_update_bin, dim3((N + 255) / 256), dim3(256), 0, st, (float4 *)d_pos, (const float *)v, N, dt, f->grid, f->ntiles,
                     pr, bin, bin && f->binBySlot, f->kern.support);
  UH_CHECK(hipGetLastError());
  if (bin) { f->binnedPending = true; f->binnedPos = (const void *)d_pos; f->binnedN = N; }
  return 0;
}

int uammd_fcm_set_option(uammd_fcm *h, const char *name, int value) {
  if (!h || !name) { set_last_error("uammd_fcm_set_option: null argument"); return -1; }
  if (std::string(name) == "atomic_spread") { reinterpret_cast<FCM *>(h)->forceAtomicSpread = value != 0; return 0; }
  if (std::string(name) == "bin_ahead") { reinterpret_cast<FCM *>(h)->emBin = value != 0; return 0; }
  if (std::string(name) == "slot_refresh" && value >= 1) { reinterpret_cast<FCM *>(h)->slotRefresh = value; return 0; }
  if (std::string(name) == "slots") { reinterpret_cast<FCM *>(h)->slotsEnabled = value != 0; return 0; }
  if (std::string(name) == "bin_by_slot") { reinterpret_cast<FCM *>(h)->binBySlot = value != 0; return 0; }
  if (std::string(name) == "spread_waves") { reinterpret_cast<FCM *>(h)->spreadWaves = value; return 0; }
  if (std::string(name) == "gather_per_wave") { reinterpret_cast<FCM *>(h)->gatherPerWave = value; return 0; }
  if (std::string(name) == "tile_gather") {
    // (the staged window is loaded in aligned x pairs and is 16 nodes wide: an even tile edge along x, edge + support - 1 <= 16; elsewhere the
    // option is accepted and the global gather runs)
    FCM *f = reinterpret_cast<FCM *>(h);
    const bool fits = f->tdim.x % 2 == 0 && f->tdim.x + f->kern.support.x - 1 <= 16 && f->tdim.y + f->kern.support.y - 1 <= 16 &&
                      f->tdim.z + f->kern.support.z - 1 <= 16;
    f->tileGather = value != 0 && fits;
    return 0;
  }
  if (std::string(name) == "interleaved_gather") { reinterpret_cast<FCM *>(h)->interGather = value != 0; return 0; }
  if (std::string(name) == "z_tile_log2" && (value == 0 || value == 2 || value == 3 || value == 4)) { reinterpret_cast<FCM *>(h)->zTileLog2 = value; return 0; }
  if (std::string(name) == "custom_fft") { reinterpret_cast<FCM *>(h)->customFFT = value != 0; return 0; }
  set_last_error("uammd_fcm_set_option: unknown option %s", name);
  return -1;
}

int uammd_fcm_displacements(uammd_fcm *h, const float *d_pos, const float *d_force, int N, float temperature,
                            float prefactor, float *d_linearVelocity, void *stream) {
  return uammd_fcm_displacements_staged(h, d_pos, d_force, N, temperature, prefactor, d_linearVelocity, 0, stream);
}

// ---- slab-decomposed FCM building blocks --------------------------------------------------------------------------
int uammd_fcm_slab_create(const uammd_fcm_parameters *par, int nzLocal, int z0, int halo, int nyLocal, int y0,
                          uammd_fcm_slab **out) {
  if (!par || !out) { set_last_error("uammd_fcm_slab_create: null argument"); return -1; }
  if (par->cells[0] <= 0 || par->cells[1] <= 0 || par->cells[2] <= 0 || !(par->boxSize[0] > 0) || !(par->boxSize[1] > 0) ||
      !(par->boxSize[2] > 0)) {
    set_last_error("FCM_impl requires a valid box and grid dimension");
    return -2;
  }
  if (nzLocal <= 0 || z0 < 0 || z0 + nzLocal > par->cells[2] || nyLocal <= 0 || y0 < 0 || y0 + nyLocal > par->cells[1] ||
      halo < 0) {
    set_last_error("uammd_fcm_slab_create: slab [%d,%d) x rows [%d,%d) outside the %d x %d x %d grid", z0, z0 + nzLocal, y0,
                   y0 + nyLocal, par->cells[0], par->cells[1], par->cells[2]);
    return -2;
  }
  for (int a = 0; a < 3; ++a)
    if (par->kernel.support[a] < 1 || par->kernel.support[a] > kMaxSupport) {
      set_last_error("uammd_fcm_slab_create: kernel support %d outside [1, %d]", par->kernel.support[a], kMaxSupport);
      return -2;
    }
  if (halo < par->kernel.support[2] / 2 + 1) {
    set_last_error("uammd_fcm_slab_create: halo %d cannot hold a stencil of support %d", halo, par->kernel.support[2]);
    return -2;
  }
  FCMSlab *s = new FCMSlab();
  s->cells = make_int3(par->cells[0], par->cells[1], par->cells[2]);
  s->L = real3f{par->boxSize[0], par->boxSize[1], par->boxSize[2]};
  s->nzl = nzLocal; s->z0 = z0; s->halo = halo; s->nyl = nyLocal; s->y0 = y0;
  s->nkx = par->cells[0] / 2 + 1;
  FCM *f = &s->loc;
  f->par = *par;
  const int lc[3] = {par->cells[0], par->cells[1], nzLocal + 2 * halo};
  const float hz = par->boxSize[2] / (float)par->cells[2];
  const float lL[3] = {par->boxSize[0], par->boxSize[1], hz * (float)lc[2]};
  const int periodic[3] = {1, 1, 1};  // the window never wraps in z: the halo holds every stencil of an owned particle
  f->grid = make_grid<float>(make_box<float>(lL, periodic), make_int3(lc[0], lc[1], lc[2]));
  f->kern = to_dev(par->kernel);
  f->nxpad = 2 * s->nkx;
  f->planeReal = (size_t)f->nxpad * lc[1];  // component stride of the interleaved-plane layout
  f->useTiles = fcm_tiles_usable(lc, par->kernel.support, &f->tdim, true);  // (the slab window's halo is a whole number of 8-node tiles)
  f->ntiles = make_int3(lc[0] / f->tdim.x, lc[1] / f->tdim.y, lc[2] / f->tdim.z);
  if (int e = fcm_slab_make_plans(s)) { delete s; return e; }
  *out = reinterpret_cast<uammd_fcm_slab *>(s);
  return 0;
}

int uammd_fcm_slab_destroy(uammd_fcm_slab *h) {
  delete reinterpret_cast<FCMSlab *>(h);
  return 0;
}

int uammd_fcm_slab_set_option(uammd_fcm_slab *h, const char *name, int value) {
  if (!h || !name) { set_last_error("uammd_fcm_slab_set_option: null argument"); return -1; }
  if (std::string(name) == "atomic_spread") { reinterpret_cast<FCMSlab *>(h)->loc.forceAtomicSpread = value != 0; return 0; }
  set_last_error("uammd_fcm_slab_set_option: unknown option %s", name);
  return -1;
}

// d_posLocal: real4[N] in the window frame (z relative to the centre of the owned slab).  d_grid: the real window,
// 3*nxpad*ny*(nzLocal+2 halo) floats, OVERWRITTEN.  d_force may be NULL (the grid is zeroed, the stencils are still prepared
// for the gather).
int uammd_fcm_slab_spread(uammd_fcm_slab *h, const float *d_posLocal, const float *d_force, int N, float *d_grid,
                          void *stream) {
  if (!h || !d_grid || (N > 0 && !d_posLocal)) { set_last_error("uammd_fcm_slab_spread: null argument"); return -1; }
  FCMSlab *s = reinterpret_cast<FCMSlab *>(h);
  FCM *f = &s->loc;
  hipStream_t st = (hipStream_t)stream;
  const size_t zs = 3 * f->planeReal;
  const size_t total = zs * (size_t)f->grid.cellDim.z;
  const bool tiles = f->useTiles && !f->forceAtomicSpread;
  if (N <= 0 || !d_force || !tiles) UH_CHECK(hipMemsetAsync(d_grid, 0, sizeof(float) * total, st));
  if (N <= 0) return 0;
  FcmPrep pr{};
  if (tiles) {
    bool slots = false;
    if (int e = fcm_prepare_best(f, d_posLocal, d_force, N, st, false, &pr, &slots)) return e;
    f->lastSolveSlots = slots;
    if (d_force) {
      const int nt = f->ntiles.x * f->ntiles.y * f->ntiles.z;
      const int ww = spread_weight_words(N, f->ntiles, f->kern.support, f->tdim);
      if (slots)
        hipLaunchKernelGGL((k_fcm_spread_tile<4, true>), dim3(nt), dim3(256), spread_lds_bytes(ww), st, d_grid, f->grid.cellDim, f->nxpad,
                           f->planeReal, zs, f->kern.support, f->ntiles, pr, ww);
      else
        hipLaunchKernelGGL((k_fcm_spread_tile<4, false>), dim3(nt), dim3(256), spread_lds_bytes(ww), st, d_grid, f->grid.cellDim, f->nxpad,
                           f->planeReal, zs, f->kern.support, f->ntiles, pr, ww);
    }
  } else if (d_force) {
    const FastDiv dsx = make_fastdiv(f->kern.support.x), dsxy = make_fastdiv(f->kern.support.x * f->kern.support.y);
    hipLaunchKernelGGL((k_fcm_ibm<true>), dim3((N + 3) / 4), dim3(256), 0, st, (const float4 *)d_posLocal,
                       (const float4 *)d_force, (float *)nullptr, d_grid, N, f->grid, f->nxpad, f->planeReal, zs, f->kern, dsx,
                       dsxy, false);
  }
  UH_CHECK(hipGetLastError());
  return 0;
}

// Interpolates the window at the particles; in tile mode it reuses the stencils prepared by the last spread call, which
// must have been made with the same d_posLocal.
int uammd_fcm_slab_gather(uammd_fcm_slab *h, const float *d_posLocal, int N, const float *d_grid, float *d_vel,
                          void *stream) {
  if (!h || !d_grid || (N > 0 && (!d_posLocal || !d_vel))) { set_last_error("uammd_fcm_slab_gather: null argument"); return -1; }
  if (N <= 0) return 0;
  FCMSlab *s = reinterpret_cast<FCMSlab *>(h);
  FCM *f = &s->loc;
  hipStream_t st = (hipStream_t)stream;
  const size_t zs = 3 * f->planeReal;
  const FastDiv dsx = make_fastdiv(f->kern.support.x), dsxy = make_fastdiv(f->kern.support.x * f->kern.support.y);
  const bool tiles = f->useTiles && !f->forceAtomicSpread;
  if (tiles) {
    if (f->prepCapN < N) { set_last_error("uammd_fcm_slab_gather: call uammd_fcm_slab_spread with these positions first"); return -3; }
    FcmPrep pr{(int4 *)f->prepOrigin.ptr, (float *)f->prepWeights.ptr, (float4 *)f->prepSorted.ptr,
               (int *)f->prepTileOf.ptr, (int *)f->prepRank.ptr, (int *)f->prepTileCount.ptr,
               (int *)f->prepTileStart.ptr, f->kern.support.x + f->kern.support.y + f->kern.support.z, f->tdim};
    if (f->kern.support.x <= 6 && f->kern.support.y <= 6 && f->kern.support.z <= 6 && f->tileGather)
      hipLaunchKernelGGL(k_fcm_gather_tile, dim3(f->ntiles.x * f->ntiles.y * f->ntiles.z), dim3(256), 0, st, d_vel, d_grid,
                         f->grid.cellDim, f->nxpad, f->planeReal, zs, f->kern.support, f->ntiles, f->grid.cellVolume, dsx, dsxy,
                         pr, f->accumulate);
    else if (f->interGather) {
      const size_t nodes = (size_t)f->grid.cellDim.x * f->grid.cellDim.y * f->grid.cellDim.z;
      if (int e = f->interBuf.reserve(sizeof(float4) * nodes)) return e;
      hipLaunchKernelGGL(k_fcm_interleave, dim3((unsigned)((nodes + 255) / 256)), dim3(256), 0, st, d_grid, f->grid.cellDim,
                         f->nxpad, f->planeReal, zs, (float4 *)f->interBuf.ptr);
      launch_gather_inter(st, d_vel, (const float4 *)f->interBuf.ptr, N, f->grid.cellDim, f->kern.support, f->grid.cellVolume, dsx, dsxy,
                          pr, f->accumulate, f->gatherPerWave);
    } else
      hipLaunchKernelGGL(k_fcm_gather_prep, dim3((N + 3) / 4), dim3(256), 0, st, d_vel, d_grid, N, f->grid.cellDim, f->nxpad,
                         f->planeReal, zs, f->kern.support, f->grid.cellVolume, dsx, dsxy, pr, f->accumulate);
  } else {
    hipLaunchKernelGGL((k_fcm_ibm<false>), dim3((N + 3) / 4), dim3(256), 0, st, (const float4 *)d_posLocal,
                       (const float4 *)nullptr, d_vel, (float *)d_grid, N, f->grid, f->nxpad, f->planeReal, zs, f->kern, dsx, dsxy, false);
  }
  UH_CHECK(hipGetLastError());
  return 0;
}

// batched 2-D R2C of the nzLocal OWNED planes of the window, IN PLACE: afterwards the owned part of the window holds
// float2 [zl][c][y][kx] (a padded real row and its half spectrum occupy the same 2*(nx/2+1) floats)
// the same with the neighbours' halo contributions folded in on the way: d_fromDown / d_fromUp = `planes` planes each ([z][c][y][x] like
// the window) that are ADDED to the first / last `planes` owned planes while the x pass loads them.  Returns 1 when the solver's own
// FFT does not serve this grid (the caller adds the planes itself and calls uammd_fcm_slab_forward_xy).
int uammd_fcm_slab_forward_xy_fold(uammd_fcm_slab *h, float *d_grid, const float *d_fromDown, const float *d_fromUp, int planes, void *stream) {
  if (!h || !d_grid || !d_fromDown || !d_fromUp) { set_last_error("uammd_fcm_slab_forward_xy_fold: null argument"); return -1; }
  FCMSlab *s = reinterpret_cast<FCMSlab *>(h);
  if (planes < 0) { set_last_error("uammd_fcm_slab_forward_xy_fold: negative plane count"); return -1; }
  if (2 * planes > s->nzl) return 1;  // thin slab, the two folds overlap: not served (the caller adds the planes itself, in order)
  if (!fcm_slab_custom_fft(s)) return 1;
  float *owned = d_grid + (size_t)s->halo * 3 * s->loc.planeReal;
  const int nx = s->cells.x, ny = s->cells.y, nh = nx / 2, rows = std::max(1, std::min(16, 2048 / nh)), nrows = 3 * ny * s->nzl;
  if (is_pow2(nx)) {
  hipLaunchKernelGGL((k_fft_x_r2c<true, true>), dim3((nrows + rows - 1) / rows), dim3(kFftThreads), sizeof(float2) * (size_t)(nx + rows * (nh + 1)),
                     (hipStream_t)stream, owned, nx, nrows, rows, d_fromDown, d_fromUp, 3 * ny * planes);
  } else {
  hipLaunchKernelGGL((k_fft_x_r2c<true, false>), dim3((nrows + rows - 1) / rows), dim3(kFftThreads), sizeof(float2) * (size_t)(nx + rows * (nh + 1)),
                     (hipStream_t)stream, owned, nx, nrows, rows, d_fromDown, d_fromUp, 3 * ny * planes);
  }
  fft_launch_lines<-1>((float2 *)owned, ny, s->nkx, 3 * s->nzl, (hipStream_t)stream);
  UH_CHECK(hipGetLastError());
  return 0;
}

// The transposes of the slab FFT, device side of the all-to-all (uammd_comm_alltoall moves equal blocks, block p <-> rank p):
//   pack:   the owned planes' xy spectrum float2 [zl][c][y = (p, yl)][kx]  ->  send float2 [p][zl][c][yl][kx]      (block p: rank p's y rows)
//   (what arrives, recv [src][zl][c][yl][kx], IS the z buffer [z = (src, zl)][c][yl][kx]: no unpack on the way there)
//   unpack: what comes back, float2 [src (y block)][zl][c][yl][kx]          ->  the spectrum [zl][c][y = (src, yl)][kx]
__global__ void __launch_bounds__(256) k_slab_transpose(const float2 *__restrict__ in, float2 *__restrict__ out, int nzl, int P, int nyl, int nkx,
                                                         bool pack) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)nzl * 3 * P * nyl * nkx;
  if (t >= total) return;
  // t enumerates the SPECTRUM layout [zl][c][p][yl][kx]
  const int kx = (int)(t % nkx);
  size_t q = t / nkx;
  const int yl = (int)(q % nyl); q /= nyl;
  const int p = (int)(q % P); q /= P;
  const int c = (int)(q % 3);
  const int zl = (int)(q / 3);
  const size_t blk = ((((size_t)p * nzl + zl) * 3 + c) * nyl + yl) * nkx + kx;
  if (pack) out[blk] = in[t]; else out[t] = in[blk];
}
int uammd_fcm_slab_transpose_pack(uammd_fcm_slab *h, const float *d_grid, float *d_send, void *stream) {
  if (!h || !d_grid || !d_send) { set_last_error("uammd_fcm_slab_transpose_pack: null argument"); return -1; }
  FCMSlab *s = reinterpret_cast<FCMSlab *>(h);
  const float *owned = d_grid + (size_t)s->halo * 3 * s->loc.planeReal;
  const int P = s->cells.y / s->nyl;
  const size_t total = (size_t)s->nzl * 3 * s->cells.y * s->nkx;
  hipLaunchKernelGGL(k_slab_transpose, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float2 *)owned,
                     (float2 *)d_send, s->nzl, P, s->nyl, s->nkx, true);
  UH_CHECK(hipGetLastError());
  return 0;
}
int uammd_fcm_slab_transpose_unpack(uammd_fcm_slab *h, const float *d_recv, float *d_grid, void *stream) {
  if (!h || !d_grid || !d_recv) { set_last_error("uammd_fcm_slab_transpose_unpack: null argument"); return -1; }
  FCMSlab *s = reinterpret_cast<FCMSlab *>(h);
  float *owned = d_grid + (size_t)s->halo * 3 * s->loc.planeReal;
  const int P = s->cells.y / s->nyl;
  const size_t total = (size_t)s->nzl * 3 * s->cells.y * s->nkx;
  hipLaunchKernelGGL(k_slab_transpose, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float2 *)d_recv,
                     (float2 *)owned, s->nzl, P, s->nyl, s->nkx, false);
  UH_CHECK(hipGetLastError());
  return 0;
}

int uammd_fcm_slab_forward_xy(uammd_fcm_slab *h, float *d_grid, void *stream) {
  if (!h || !d_grid) { set_last_error("uammd_fcm_slab_forward_xy: null argument"); return -1; }
  FCMSlab *s = reinterpret_cast<FCMSlab *>(h);
  float *owned = d_grid + (size_t)s->halo * 3 * s->loc.planeReal;  // window layout [z][component][y][x]: the owned planes are one block
  if (fcm_slab_custom_fft(s)) {
    const int nx = s->cells.x, ny = s->cells.y, nh = nx / 2, rows = std::max(1, std::min(16, 2048 / nh)), nrows = 3 * ny * s->nzl;
    if (is_pow2(nx)) {
    hipLaunchKernelGGL((k_fft_x_r2c<false, true>), dim3((nrows + rows - 1) / rows), dim3(kFftThreads), sizeof(float2) * (size_t)(nx + rows * (nh + 1)),
                       (hipStream_t)stream, owned, nx, nrows, rows, (const float *)nullptr, (const float *)nullptr, 0);
    } else {
    hipLaunchKernelGGL((k_fft_x_r2c<false, false>), dim3((nrows + rows - 1) / rows), dim3(kFftThreads), sizeof(float2) * (size_t)(nx + rows * (nh + 1)),
                       (hipStream_t)stream, owned, nx, nrows, rows, (const float *)nullptr, (const float *)nullptr, 0);
    }
    fft_launch_lines<-1>((float2 *)owned, ny, s->nkx, 3 * s->nzl, (hipStream_t)stream);
    UH_CHECK(hipGetLastError());
    return 0;
  }
  UH_ROCFFT(rocfft_execution_info_set_stream(s->loc.info, stream));
  void *io[1] = {(void *)owned};
  UH_ROCFFT(rocfft_execute(s->fwdXY, io, nullptr, s->loc.info));
  return 0;
}

// batched 2-D C2R of the owned part of the window, in place; the halo planes are left untouched
int uammd_fcm_slab_inverse_xy(uammd_fcm_slab *h, float *d_grid, void *stream) {
  if (!h || !d_grid) { set_last_error("uammd_fcm_slab_inverse_xy: null argument"); return -1; }
  FCMSlab *s = reinterpret_cast<FCMSlab *>(h);
  float *owned = d_grid + (size_t)s->halo * 3 * s->loc.planeReal;
  if (fcm_slab_custom_fft(s)) {
    const int nx = s->cells.x, ny = s->cells.y, nh = nx / 2, rows = std::max(1, std::min(8, 2048 / (3 * nh))), nrows = ny * s->nzl;
    fft_launch_lines<1>((float2 *)owned, ny, s->nkx, 3 * s->nzl, (hipStream_t)stream);
    // rows (z, y) of the three components: component stride = one (ny x nxpad) plane, z stride = three of them
    if (is_pow2(nx) && is_pow2(ny)) {
    hipLaunchKernelGGL(k_fft_x_c2r<true>, dim3((nrows + rows - 1) / rows), dim3(kFftThreads), sizeof(float2) * (size_t)(nx + 3 * rows * (nh + 1)),
                       (hipStream_t)stream, owned, (size_t)ny * s->loc.nxpad, 3 * (size_t)ny * s->loc.nxpad, ny, nx,
                       nrows, rows, (float4 *)nullptr, 0, (size_t)0);
    } else {
    hipLaunchKernelGGL(k_fft_x_c2r<false>, dim3((nrows + rows - 1) / rows), dim3(kFftThreads), sizeof(float2) * (size_t)(nx + 3 * rows * (nh + 1)),
                       (hipStream_t)stream, owned, (size_t)ny * s->loc.nxpad, 3 * (size_t)ny * s->loc.nxpad, ny, nx,
                       nrows, rows, (float4 *)nullptr, 0, (size_t)0);
    }
    UH_CHECK(hipGetLastError());
    return 0;
  }
  UH_ROCFFT(rocfft_execution_info_set_stream(s->loc.info, stream));
  void *io[1] = {(void *)owned};
  UH_ROCFFT(rocfft_execute(s->invXY, io, nullptr, s->loc.info));
  return 0;
}

// The same inverse writing the OWNED planes of the gather's float4 window d_inter [z][y][x] = (vx, vy, vz, 0) (nz window planes of
// ny x nx nodes) instead of the planar real rows: the single-GPU path's fusion of the inverse row pass with the interleaving copy.
// The caller exchanges the halo planes of d_inter and hands it to uammd_fcm_slab_gather_inter.  Returns 1 (nothing done) when the
// grid does not take the custom FFT: the caller then uses uammd_fcm_slab_inverse_xy + uammd_fcm_slab_gather.
static int fcm_slab_inverse_xy_inter(uammd_fcm_slab *h, float *d_grid, float *d_inter, int wrapPlanes, void *stream);
int uammd_fcm_slab_inverse_xy_inter(uammd_fcm_slab *h, float *d_grid, float *d_inter, void *stream) {
  return fcm_slab_inverse_xy_inter(h, d_grid, d_inter, 0, stream);
}
// world size 1 (the rank is its own neighbour through the periodic z faces): the x pass also stores the first / last `wrapPlanes` owned
// planes into the halo planes above / below the owned block, so the window is ready for the gather without a halo exchange
int uammd_fcm_slab_inverse_xy_inter_wrap(uammd_fcm_slab *h, float *d_grid, float *d_inter, int wrapPlanes, void *stream) {
  return fcm_slab_inverse_xy_inter(h, d_grid, d_inter, wrapPlanes, stream);
}
static int fcm_slab_inverse_xy_inter(uammd_fcm_slab *h, float *d_grid, float *d_inter, int wrapPlanes, void *stream) {
  if (!h || !d_grid || !d_inter) { set_last_error("uammd_fcm_slab_inverse_xy_inter: null argument"); return -1; }
  FCMSlab *s = reinterpret_cast<FCMSlab *>(h);
  if (wrapPlanes < 0 || wrapPlanes > s->halo || wrapPlanes > s->nzl) { set_last_error("uammd_fcm_slab_inverse_xy_inter_wrap: more planes than the halo holds"); return -1; }
  if (!fcm_slab_custom_fft(s) || !(s->loc.useTiles && !s->loc.forceAtomicSpread)) return 1;  // (the float4 gather reads tile-prepared stencils)
  float *owned = d_grid + (size_t)s->halo * 3 * s->loc.planeReal;
  const int nx = s->cells.x, ny = s->cells.y, nh = nx / 2, rows = std::max(1, std::min(8, 2048 / (3 * nh))), nrows = ny * s->nzl;
  fft_launch_lines<1>((float2 *)owned, ny, s->nkx, 3 * s->nzl, (hipStream_t)stream);
  if (is_pow2(nx) && is_pow2(ny)) {
  hipLaunchKernelGGL(k_fft_x_c2r<true>, dim3((nrows + rows - 1) / rows), dim3(kFftThreads), sizeof(float2) * (size_t)(nx + 3 * rows * (nh + 1)),
                     (hipStream_t)stream, owned, (size_t)ny * s->loc.nxpad, 3 * (size_t)ny * s->loc.nxpad, ny, nx,
                     nrows, rows, (float4 *)d_inter + (size_t)s->halo * ny * nx, wrapPlanes * ny, (size_t)s->nzl * ny * nx);
  } else {
  hipLaunchKernelGGL(k_fft_x_c2r<false>, dim3((nrows + rows - 1) / rows), dim3(kFftThreads), sizeof(float2) * (size_t)(nx + 3 * rows * (nh + 1)),
                     (hipStream_t)stream, owned, (size_t)ny * s->loc.nxpad, 3 * (size_t)ny * s->loc.nxpad, ny, nx,
                     nrows, rows, (float4 *)d_inter + (size_t)s->halo * ny * nx, wrapPlanes * ny, (size_t)s->nzl * ny * nx);
  }
  UH_CHECK(hipGetLastError());
  return 0;
}

// gather from the float4 window filled by uammd_fcm_slab_inverse_xy_inter (+ the caller's halo exchange); needs the stencils prepared
// by uammd_fcm_slab_spread with the same positions
int uammd_fcm_slab_gather_inter(uammd_fcm_slab *h, const float *d_posLocal, int N, const float *d_inter, float *d_vel, void *stream) {
  if (!h || !d_inter || (N > 0 && (!d_posLocal || !d_vel))) { set_last_error("uammd_fcm_slab_gather_inter: null argument"); return -1; }
  if (N <= 0) return 0;
  FCMSlab *s = reinterpret_cast<FCMSlab *>(h);
  FCM *f = &s->loc;
  if (!(f->useTiles && !f->forceAtomicSpread) || f->prepCapN < N) {
    set_last_error("uammd_fcm_slab_gather_inter: needs the tile-prepared stencils of uammd_fcm_slab_spread");
    return -3;
  }
  const FastDiv dsx = make_fastdiv(f->kern.support.x), dsxy = make_fastdiv(f->kern.support.x * f->kern.support.y);
  FcmPrep pr{(int4 *)f->prepOrigin.ptr, (float *)f->prepWeights.ptr, (float4 *)f->prepSorted.ptr, (int *)f->prepTileOf.ptr,
             (int *)f->prepRank.ptr, (int *)f->prepTileCount.ptr, (int *)f->prepTileStart.ptr,
             f->kern.support.x + f->kern.support.y + f->kern.support.z, f->tdim};
  launch_gather_inter((hipStream_t)stream, d_vel, (const float4 *)d_inter, N, f->grid.cellDim, f->kern.support, f->grid.cellVolume, dsx,
                      dsxy, pr, f->accumulate);
  UH_CHECK(hipGetLastError());
  return 0;
}

// 1-D complex FFT along z of d_cplxZ [z][c][yl][kx], in place
int uammd_fcm_slab_fft_z(uammd_fcm_slab *h, float *d_cplxZ, int inverse, void *stream) {
  if (!h || !d_cplxZ) { set_last_error("uammd_fcm_slab_fft_z: null argument"); return -1; }
  FCMSlab *s = reinterpret_cast<FCMSlab *>(h);
  UH_ROCFFT(rocfft_execution_info_set_stream(s->loc.info, stream));
  void *io[1] = {d_cplxZ};
  UH_ROCFFT(rocfft_execute(inverse ? s->invZ : s->fwdZ, io, nullptr, s->loc.info));
  return 0;
}

// z transform + Fourier-space operator + inverse z transform of d_cplxZ [z][c][yl][kx] in one pass (power-of-two nz; returns 1 when
// the grid does not allow it: the caller then takes uammd_fcm_slab_fft_z / _kspace / _fft_z).  haveForce = 0: the buffer's content is
// ignored (noise only).
int uammd_fcm_slab_z_fused(uammd_fcm_slab *h, float *d_cplxZ, int haveForce, float temperature, float prefactor, unsigned int seed2,
                           void *stream) {
  if (!h || !d_cplxZ) { set_last_error("uammd_fcm_slab_z_fused: null argument"); return -1; }
  FCMSlab *s = reinterpret_cast<FCMSlab *>(h);
  if (!s->loc.customFFT || !fft_axis_ok(s->cells.z, 2, 512)) return 1;
  float noisePrefactor = 0.0f;
  if (temperature > 0.0f) {
    const float gL[3] = {s->L.x, s->L.y, s->L.z};
    const int per[3] = {1, 1, 1};
    const GridT<float> g = make_grid<float>(make_box<float>(gL, per), s->cells);
    const float fourierNormalization = (float)(1.0 / ((double)s->cells.x * s->cells.y * s->cells.z));
    noisePrefactor = prefactor * sqrtf(fourierNormalization * 2 * temperature / g.cellVolume);
  }
  const size_t comp = (size_t)s->nyl * s->nkx;
  if (int e = fcm_fft_z_fused_launch(&s->loc, (float2 *)d_cplxZ, comp, 3 * comp, s->nyl, s->y0, s->cells, s->L, haveForce != 0, noisePrefactor,
                                     seed2, (hipStream_t)stream)) return e;
  UH_CHECK(hipGetLastError());
  return 0;
}

// forceFourier2Vel + fourierBrownianNoise on this rank's y-rows of the Fourier grid.  seed2 is the value of the reference's
// call counter AFTER its increment for this call (all ranks pass the same one); ignored when temperature == 0.
int uammd_fcm_slab_kspace(uammd_fcm_slab *h, float *d_cplxZ, int haveForce, float temperature, float prefactor,
                          unsigned int seed2, void *stream) {
  if (!h || !d_cplxZ) { set_last_error("uammd_fcm_slab_kspace: null argument"); return -1; }
  FCMSlab *s = reinterpret_cast<FCMSlab *>(h);
  float noisePrefactor = 0.0f;
  if (temperature > 0.0f) {
    const float gL[3] = {s->L.x, s->L.y, s->L.z};
    const int per[3] = {1, 1, 1};
    const GridT<float> g = make_grid<float>(make_box<float>(gL, per), s->cells);
    const float fourierNormalization = (float)(1.0 / ((double)s->cells.x * s->cells.y * s->cells.z));
    noisePrefactor = prefactor * sqrtf(fourierNormalization * 2 * temperature / g.cellVolume);
  }
  const KLayout lay{s->nyl, s->y0, (size_t)s->nyl * s->nkx, 3 * (size_t)s->nyl * s->nkx, make_fastdiv(s->nkx), make_fastdiv(s->nyl)};
  const int total = s->cells.z * s->nyl * s->nkx;
  hipLaunchKernelGGL(k_fcm_kspace, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, (float2 *)d_cplxZ, lay,
                     s->cells, s->L, s->loc.par.viscosity, haveForce != 0, noisePrefactor, s->loc.par.seed, seed2, s->loc.pse);
  UH_CHECK(hipGetLastError());
  return 0;
}

// ---- PSE far field: the same pipeline with the Hasimoto-split RPY greens function -------------------------
// FarField::initializeGrid (FarField.cuh:646-654): cells before nextFFTWiseSize3D
int uammd_pse_far_raw_cells(const float boxSize[3], float psi, float tolerance, int cells_out[3]) {
  if (!boxSize || !cells_out) { set_last_error("uammd_pse_far_raw_cells: null argument"); return -1; }
  const float kcut = (float)(2 * psi * std::sqrt(-std::log(tolerance)));
  const double hgrid = 2 * M_PI / kcut;
  for (int k = 0; k < 3; ++k) cells_out[k] = (int)(2 * boxSize[k] / hgrid) + 1;
  return 0;
}

// FarField ctor (FarField.cuh:318-342): kernel from initializeKernel (:605-644) on the given (FFT-friendly) grid
int uammd_pse_far_create(const float boxSize[3], const int cells[3], float viscosity, float hydrodynamicRadius, float tolerance,
                         float psi, float shearStrain, unsigned int seed, uammd_fcm **out, int *support_out, float *eta_out) {
  if (!boxSize || !cells || !out) { set_last_error("uammd_pse_far_create: null argument"); return -1; }
  // Gaussian window: m standard deviations inside the support, support odd (sec. 4.1 of Lindbo & Tornberg)
  const double C = 0.976;
  double m = 1;
  while (std::erfc(m / std::sqrt(2.0)) > 0.1 * tolerance) m += 0.01;
  int support;
  while ((support = int(std::pow(m / C, 2) / M_PI + 0.5) + 1) % 2 == 0) m += tolerance;
  int P = support / 2;
  const int minCellDim = std::min(cells[0], std::min(cells[1], cells[2]));
  if (support > minCellDim) {
    support = minCellDim;
    if (support % 2 == 0) support--;
    P = support / 2;
    m = C * std::sqrt(M_PI * support);
  }
  const double pw = 2 * P + 1;
  const float cs[3] = {boxSize[0] / (float)cells[0], boxSize[1] / (float)cells[1], boxSize[2] / (float)cells[2]};
  const double h = std::min(cs[0], std::min(cs[1], cs[2]));
  const double w = pw * h / 2.0;
  const float eta = (float)std::pow(2.0 * psi * w / m, 2);
  const float width = (float)(std::sqrt(eta) / (2.0 * psi));  // pse_ns::Kernel(P, width), FarField.cuh:25-41
  uammd_fcm_parameters p{};
  for (int k = 0; k < 3; ++k) { p.boxSize[k] = boxSize[k]; p.cells[k] = cells[k]; }
  p.viscosity = viscosity;
  p.seed = seed;
  p.hydrodynamicRadius = hydrodynamicRadius;
  p.kernel.kind = UAMMD_IBM_KERNEL_GAUSSIAN;
  p.kernel.support[0] = p.kernel.support[1] = p.kernel.support[2] = 2 * P + 1;
  p.kernel.prefactor = (float)std::cbrt(1.0 / (width * width * width * std::pow(2.0 * M_PI, 1.5)));
  p.kernel.tau = (float)(-0.5 / (width * width));
  p.kernel.rmax = INFINITY;  // this window is not cut (FarField.cuh:37-39)
  if (int e = uammd_fcm_create(&p, out)) return e;
  FCM *f = reinterpret_cast<FCM *>(*out);
  f->pse = PseGreens{hydrodynamicRadius, psi, eta, shearStrain, true};
  f->accumulate = true;  // ibm.gather adds into MF (FarField.cuh:563-566)
  if (support_out) *support_out = 2 * P + 1;
  if (eta_out) *eta_out = eta;
  return 0;
}

int uammd_pse_far_set_shear_strain(uammd_fcm *h, float shearStrain) {
  if (!h) { set_last_error("uammd_pse_far_set_shear_strain: null handle"); return -1; }
  reinterpret_cast<FCM *>(h)->pse.shear = shearStrain;
  return 0;
}

// FarField::computeHydrodynamicDisplacements (FarField.cuh:569-589): d_MF real3[N] += M_far F + noise
int uammd_pse_far_displacements(uammd_fcm *h, const float *d_pos, const float *d_force, int N, float temperature,
                                float prefactor, unsigned int seed2, float *d_MF, void *stream) {
  if (!h) { set_last_error("uammd_pse_far_displacements: null handle"); return -1; }
  FCM *f = reinterpret_cast<FCM *>(h);
  if (!f->pse.on) { set_last_error("uammd_pse_far_displacements: not a PSE far-field handle"); return -1; }
  f->seed2 = seed2;
  return uammd_fcm_displacements_staged(h, d_pos, d_force, N, temperature, prefactor, d_MF, 0, stream);
}

// The same solve queued in two halves (1: binning, stencils, spreading, forward x / y transforms; 2: z transform + operator + noise,
// inverse transforms, gather): BDHI::PSE queues the near field's convergence check between them (uammd_pse_near_set_interleave_early /
// uammd_pse_near_set_interleave), so that the GPU has the first half to do while the host answers the check and the second while it
// reacts to the outcome.  Same arguments to both calls, same stream; the result is uammd_pse_far_displacements'.
int uammd_pse_far_displacements_half(uammd_fcm *h, const float *d_pos, const float *d_force, int N, float temperature, float prefactor,
                                     unsigned int seed2, float *d_MF, int half, void *stream) {
  if (!h) { set_last_error("uammd_pse_far_displacements_half: null handle"); return -1; }
  if (half != 1 && half != 2) { set_last_error("uammd_pse_far_displacements_half: half must be 1 or 2"); return -1; }
  FCM *f = reinterpret_cast<FCM *>(h);
  if (!f->pse.on) { set_last_error("uammd_pse_far_displacements_half: not a PSE far-field handle"); return -1; }
  f->seed2 = seed2;
  return fcm_displacements_impl(h, d_pos, d_force, N, temperature, prefactor, d_MF, 0, stream, false, half);
}

// ---- torques / rotation -------------------------------------------------------------------------------------
// FCM_ns::Kernels::GaussianTorque(width = a / (6 sqrt(pi))^(1/3), h, tolerance): FCM_kernels.cuh:60-80, BDHI_FCM.cuh:69-80
int uammd_fcm_torque_gaussian_kernel(float hydrodynamicRadius, float h, float tolerance, uammd_ibm_kernel *out) {
  if (!out || !(h > 0) || !(tolerance > 0) || !(hydrodynamicRadius > 0)) { set_last_error("uammd_fcm_torque_gaussian_kernel: bad arguments"); return -1; }
  const float width = (float)(hydrodynamicRadius / (std::pow(6 * std::sqrt(M_PI), 1 / 3.)));
  const float prefactor = (float)std::pow(2.0 * M_PI * (double)width * (double)width, -0.5);
  const float tau = (float)(-0.5 / ((double)width * (double)width));
  const float dr = (float)(0.5 * h);
  float r = dr;
  while (prefactor * expf(tau * r * r) > tolerance) r += dr;
  int support = (int)(2 * r / h + 0.5);
  if (support < 3) support = 3;
  out->kind = UAMMD_IBM_KERNEL_GAUSSIAN;
  out->support[0] = out->support[1] = out->support[2] = support;
  out->prefactor = prefactor;
  out->tau = tau;
  out->rmax = (float)support * h;
  out->invh[0] = out->invh[1] = out->invh[2] = 0.0f;
  return 0;
}

int uammd_fcm_set_torque_kernel(uammd_fcm *h, const uammd_ibm_kernel *kernelTorque) {
  if (!h || !kernelTorque) { set_last_error("uammd_fcm_set_torque_kernel: null argument"); return -1; }
  FCM *f = reinterpret_cast<FCM *>(h);
  for (int a = 0; a < 3; ++a)
    if (kernelTorque->support[a] < 1 || kernelTorque->support[a] > kMaxSupport || kernelTorque->support[a] >= f->par.cells[a]) {
      set_last_error("[BDHI::FCM] Kernel support is too big, try lowering the tolerance or increasing the box size!.");
      return -2;
    }
  f->kernT = to_dev(*kernelTorque);
  f->haveTorqueKernel = true;
  return f->gridBufT.reserve(sizeof(float) * 3 * f->planeReal);
}

// FCM_impl::computeHydrodynamicDisplacements with torques (FCM_impl.cuh:652-693): d_torque real4[N] (NULL = the plain call);
// d_angularVelocity real3[N] is OVERWRITTEN (only touched when d_torque is given).
int uammd_fcm_displacements_torque(uammd_fcm *h, const float *d_pos, const float *d_force, const float *d_torque, int N,
                                   float temperature, float prefactor, float *d_linearVelocity, float *d_angularVelocity,
                                   void *stream) {
  if (!d_torque) return uammd_fcm_displacements(h, d_pos, d_force, N, temperature, prefactor, d_linearVelocity, stream);
  if (!h || !d_pos || !d_linearVelocity || !d_angularVelocity) { set_last_error("uammd_fcm_displacements_torque: null argument"); return -1; }
  FCM *f = reinterpret_cast<FCM *>(h);
  if (!f->haveTorqueKernel) { set_last_error("uammd_fcm_displacements_torque: no torque kernel (uammd_fcm_set_torque_kernel)"); return -2; }
  if (f->pse.on) { set_last_error("uammd_fcm_displacements_torque: not available for the PSE far field"); return -2; }
  if (N <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  float *g = (float *)f->gridBuf.ptr, *gT = (float *)f->gridBufT.ptr;
  const size_t zs = (size_t)f->nxpad * f->grid.cellDim.y;
  const dim3 gp((N + 3) / 4), bp(256);
  UH_ROCFFT(rocfft_execution_info_set_stream(f->info, (void *)st));
  void *bufs[1] = {g}, *bufsT[1] = {gT};
  const int total = (int)f->planeCplx;
  const dim3 gk((total + 255) / 256);
  const real3f L{f->par.boxSize[0], f->par.boxSize[1], f->par.boxSize[2]};
  const FastDiv dNkx = make_fastdiv(f->grid.cellDim.x / 2 + 1), dNy = make_fastdiv(f->grid.cellDim.y);
  // forces (the generic one-wave-per-particle spread: the torque path is not the tuned one)
  {
    const FastDiv dsx = make_fastdiv(f->kern.support.x), dsxy = make_fastdiv(f->kern.support.x * f->kern.support.y);
    UH_CHECK(hipMemsetAsync(g, 0, sizeof(float) * 3 * f->planeReal, st));
    if (d_force) {
      hipLaunchKernelGGL((k_fcm_ibm<true>), gp, bp, 0, st, (const float4 *)d_pos, (const float4 *)d_force, (float *)nullptr, g, N,
                         f->grid, f->nxpad, f->planeReal, zs, f->kern, dsx, dsxy, false);
      UH_ROCFFT(rocfft_execute(f->fwd, bufs, nullptr, f->info));
    }
  }
  // torques: spread with the torque window, transform, add half the curl to the Fourier forces
  const FastDiv tsx = make_fastdiv(f->kernT.support.x), tsxy = make_fastdiv(f->kernT.support.x * f->kernT.support.y);
  UH_CHECK(hipMemsetAsync(gT, 0, sizeof(float) * 3 * f->planeReal, st));
  hipLaunchKernelGGL((k_fcm_ibm<true>), gp, bp, 0, st, (const float4 *)d_pos, (const float4 *)d_torque, (float *)nullptr, gT, N,
                     f->grid, f->nxpad, f->planeReal, zs, f->kernT, tsx, tsxy, false);
  UH_ROCFFT(rocfft_execute(f->fwd, bufsT, nullptr, f->info));
  hipLaunchKernelGGL((k_fcm_half_curl<true>), gk, bp, 0, st, (const float2 *)gT, (float2 *)g, f->planeCplx, f->grid.cellDim, L, dNkx, dNy);
  // Stokes + noise (the grid holds Fourier forces even when d_force is NULL: it was zeroed above)
  float noisePrefactor = 0.0f;
  if (temperature > 0.0f) {
    f->seed2++;
    const float fourierNormalization = (float)(1.0 / ((double)f->grid.cellDim.x * f->grid.cellDim.y * f->grid.cellDim.z));
    noisePrefactor = prefactor * sqrtf(fourierNormalization * 2 * temperature / f->grid.cellVolume);
  }
  const KLayout lay{f->grid.cellDim.y, 0, f->planeCplx, (size_t)(f->grid.cellDim.x / 2 + 1) * f->grid.cellDim.y, dNkx, dNy};
  hipLaunchKernelGGL(k_fcm_kspace, gk, bp, 0, st, (float2 *)g, lay, f->grid.cellDim, L, f->par.viscosity, true, noisePrefactor,
                     f->par.seed, f->seed2, f->pse);
  // angular velocity = half the curl of the velocity, interpolated with the torque window
  hipLaunchKernelGGL((k_fcm_half_curl<false>), gk, bp, 0, st, (const float2 *)g, (float2 *)gT, f->planeCplx, f->grid.cellDim, L, dNkx, dNy);
  UH_ROCFFT(rocfft_execute(f->inv, bufsT, nullptr, f->info));
  hipLaunchKernelGGL((k_fcm_ibm<false>), gp, bp, 0, st, (const float4 *)d_pos, (const float4 *)nullptr, d_angularVelocity, gT, N,
                     f->grid, f->nxpad, f->planeReal, zs, f->kernT, tsx, tsxy, false);
  UH_ROCFFT(rocfft_execute(f->inv, bufs, nullptr, f->info));
  {
    const FastDiv dsx = make_fastdiv(f->kern.support.x), dsxy = make_fastdiv(f->kern.support.x * f->kern.support.y);
    hipLaunchKernelGGL((k_fcm_ibm<false>), gp, bp, 0, st, (const float4 *)d_pos, (const float4 *)nullptr, d_linearVelocity, g, N,
                       f->grid, f->nxpad, f->planeReal, zs, f->kern, dsx, dsxy, false);
  }
  UH_CHECK(hipGetLastError());
  return 0;
}

double uammd_fcm_self_mobility(double hydrodynamicRadius, double viscosity, double Lx) {
  // FCM_impl::getSelfMobility, FCM_impl.cuh:102-119 (Hasimoto 1959, O(a^8))
  const long double rh = hydrodynamicRadius, L = Lx;
  const long double a = rh / L, a2 = a * a, a3 = a2 * a;
  const long double c = 2.83729747948061947666591710460773907l, b = 0.19457l;
  const long double pi = 3.141592653589793238462643383279502884L;
  const long double a6pref = 16.0l * pi * pi / 45.0l + 630.0L * b * b;
  return (double)(1.0l / (6.0l * pi * viscosity * rh) * (1.0l - c * a + (4.0l / 3.0l) * pi * a3 - a6pref * a3 * a3));
}

}  // extern "C"
#ifdef UAMMD_SPREAD_TIMELINE
extern "C" int uammd_debug_spread_timeline(unsigned long long out[8]) {
  unsigned long long zero[8] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(uammd_hip::g_spread_tl), sizeof(zero)) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(uammd_hip::g_spread_tl), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif
