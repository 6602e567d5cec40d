// Multi-GPU drivers in C++14 over the C ABI (include/uammd_hip.h): one process per GPU, RCCL over xGMI behind uammd::Comm.
//
// The reference is single GPU (no NCCL / MPI anywhere in its sources), so these classes have no reference counterpart: they are the host
// side of SURVEY 8(e), the same schedules as the Python harness (uammd_amd/parallel.py, parallel_fcm.py) on the same library kernels.
//   uammd::DistributedLJ   VerletNVT::GronbechJensen + PairForces<Potential::LJ, CellList> on z slabs of the periodic box: halo positions
//                          and migrating particles to the two z neighbours (point to point: one xGMI link each), no force reduction
//                          (the reference computes full per-particle forces, NeighbourList/common.cuh:10-34)
//   uammd::DistributedFCM  FCM_impl::computeHydrodynamicDisplacements (BDHI/FCM/FCM_impl.cuh:652-693) on z slabs of the grid: halo planes
//                          to the neighbours, two all-to-all transposes per solve, Fourier noise without communication
// Start one process per GPU, give all of them the same 128-byte id (Comm::uniqueId() on rank 0, handed over by a file, MPI, a socket:
// examples/lj_slab.cpp and examples/fcm_slab.cpp use a file), then every rank builds its Comm and its driver.
#ifndef UAMMD_DISTRIBUTED_H
#define UAMMD_DISTRIBUTED_H
#if defined(DOUBLE_PRECISION)
#error "Distributed.h: the slab drivers have a single-precision backend only on MI355X (uammd.h, PRECISION): build without -DDOUBLE_PRECISION"
#endif
#include "uammd.h"

#include <chrono>
#include <fstream>
#include <thread>

namespace uammd {

// the unique id through a file: rank 0 writes it (atomically, by rename), the others wait for it
inline std::vector<char> exchangeUniqueIdThroughFile(const std::string &path, int rank) {
  if (rank == 0) {
    const std::vector<char> id = Comm::uniqueId();
    { std::ofstream f(path + ".tmp", std::ios::binary); f.write(id.data(), (std::streamsize)id.size()); }
    if (std::rename((path + ".tmp").c_str(), path.c_str()) != 0) throw std::runtime_error("cannot publish the communicator id at " + path);
    return id;
  }
  for (int tries = 0; tries < 6000; ++tries) {
    std::ifstream f(path, std::ios::binary);
    std::vector<char> id(128);
    if (f && f.read(id.data(), 128) && f.gcount() == 128) return id;
    std::this_thread::sleep_for(std::chrono::milliseconds(10));
  }
  throw std::runtime_error("no communicator id appeared at " + path);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Path A.  Rank r owns the particles with z in its slab of the periodic box and works in a local frame (z relative to the slab centre)
// with a local box (Lx, Ly, width + 2.02 (rc + 4 skin)) that is not periodic in z.  Persistent arrays with spare rows: the owned particles
// at the head, the ghosts received from the two neighbours behind them.  With a skin the ownership and the halo MEMBERSHIP lists are
// refreshed every `exchangeEvery` steps (two host reads of message sizes); in between the listed particles' current positions are re-sent
// (fixed sizes, nothing synchronises).  Valid while nobody moves more than the skin between refreshes: checkSkin() verifies it.
class DistributedLJ {
  std::shared_ptr<Comm> comm;
  int rank, world;
  real3 L;
  real rc, skin, width, dt, friction, temperature, noiseAmplitude;
  int exchangeEvery;
  uint seed;
  int steps = 0, nOwned = 0, nAll = 0, cap = 0, nUpH = 0, nDownH = 0, gFromDown = 0, gFromUp = 0, refN = -1;
  detail::DeviceArray<real4> pos, force, ref;
  detail::DeviceArray<real> vel, rows, arrivals, sendUp, sendDown, maxd;
  detail::DeviceArray<int> ids, idx, holes, counts;
  detail::DeviceArray<char> tiles;
  detail::DeviceArray<unsigned char> listed;   // one byte per owned row: is it in a halo list (uammd_slab_refresh_lj writes it)
  bool fuseFirstHalfStep = false;              // a slab wider than two reaches: the up and down lists are disjoint
  uammd_celllist *cl = nullptr;
  Potential::LJ pot;
  hipStream_t st = 0;
  float boxL[3], updL[3];
  int boxPer[3], updPer[3], cellDim[3];
  bool haveDrift = false;

  int *idxRow(int k) { return idx.d + (size_t)k * cap; }
  void refill() {  // the listed particles' current positions to the neighbours, straight into the ghost tail
    detail::check(uammd_halo_pack((const float *)pos.d, idxRow(2), nUpH, idxRow(3), nDownH, -width, width, sendUp.d, sendDown.d, (void *)st));
    comm->haloExchange(sendUp.d, nUpH, sendDown.d, nDownH, (real *)(pos.d + nOwned), gFromDown, (real *)(pos.d + nOwned + gFromDown), gFromUp, 4, st);
  }
  // the membership refresh as ONE library call (uammd_slab_refresh_lj: displacement since the last refresh, leavers selected and migrated
  // with their velocities and ids, halo members selected, their positions exchanged into the ghost tail; two host reads of message sizes)
  void refresh() {
    int out[10];
    const bool useRef = skin > 0;
    if (useRef && refN == nOwned) haveDrift = true;
    detail::check(uammd_slab_refresh_lj(comm->handle(), (float *)pos.d, vel.d, ids.d, (float *)force.d, nOwned, cap, width, rc + real(3.0) * skin,
                                        idx.d, holes.d, counts.d, tiles.d, rows.d, arrivals.d, sendUp.d, useRef ? (float *)ref.d : nullptr, refN,
                                        useRef ? maxd.d : nullptr, fuseFirstHalfStep ? listed.d : nullptr, out, (void *)st));
    nOwned = out[0]; nAll = out[1]; nUpH = out[2]; nDownH = out[3]; gFromDown = out[4]; gFromUp = out[5];
    if (useRef) refN = nOwned;
  }
  void forces() {  // owned + ghost positions -> forces of the owned rows, accumulated
    detail::check(uammd_celllist_update(cl, (const float *)pos.d, nAll, updL, updPer, cellDim, (void *)st));
    detail::check(uammd_celllist_set_option(cl, "num_owned", nOwned));
    detail::check(uammd_lj_transverse_celllist(cl, pot.deviceTable(), pot.getNumberTypes(), boxL, boxPer, (float *)force.d, nullptr, nullptr, nullptr,
                                               UAMMD_LJ_ALGO_AUTO, (void *)st));
  }
  void forcesAndSecondHalfStep() {  // the same, GronbechJensen's second half step of the owned rows riding in the traversal's store
    detail::check(uammd_celllist_update(cl, (const float *)pos.d, nAll, updL, updPer, cellDim, (void *)st));
    detail::check(uammd_celllist_set_option(cl, "num_owned", nOwned));
    detail::check(uammd_lj_transverse_celllist_gj2(cl, pot.deviceTable(), pot.getNumberTypes(), boxL, boxPer, (float *)force.d, vel.d, nullptr, real(1.0),
                                                   dt, 0, UAMMD_LJ_ALGO_AUTO, (void *)st));
  }
  // a step between refreshes as four launches: GronbechJensen's first half step of the LISTED rows inside the halo pack, the exchange,
  // the half step of everybody else inside the list build's hash kernel (the listed rows masked), forces + second half step
  void fusedStep() {
    detail::check(uammd_halo_pack_gj1((float *)pos.d, vel.d, (float *)force.d, nullptr, real(1.0), ids.d, idxRow(2), nUpH, idxRow(3), nDownH, -width,
                                      width, sendUp.d, sendDown.d, dt, friction, 0, noiseAmplitude, (uint)steps, seed, (void *)st));
    comm->haloExchange(sendUp.d, nUpH, sendDown.d, nDownH, (real *)(pos.d + nOwned), gFromDown, (real *)(pos.d + nOwned + gFromDown), gFromUp, 4, st);
    detail::check(uammd_celllist_update_gj1(cl, (float *)pos.d, nAll, updL, updPer, cellDim, vel.d, (float *)force.d, nullptr, real(1.0), ids.d,
                                            listed.d, nOwned, dt, friction, 0, noiseAmplitude, (uint)steps, seed, (void *)st));
    detail::check(uammd_celllist_set_option(cl, "num_owned", nOwned));
    detail::check(uammd_lj_transverse_celllist_gj2(cl, pot.deviceTable(), pot.getNumberTypes(), boxL, boxPer, (float *)force.d, vel.d, nullptr, real(1.0),
                                                   dt, 0, UAMMD_LJ_ALGO_AUTO, (void *)st));
  }
  void integrate(int step) {
    // (the thermostat's stream is keyed by the particle's GLOBAL id, same seed on every rank: a particle draws the same kicks whichever
    // rank and row holds it — no correlation between slabs through equal row numbers, no dependence on the decomposition)
    detail::check(uammd_verletnvt_gj_keyed(step, (float *)pos.d, vel.d, (float *)force.d, nullptr, real(1.0), nullptr, ids.d, nOwned, dt, friction,
                                           0, noiseAmplitude, (uint)steps, seed, (void *)st));
  }
public:
  struct Parameters {
    real3 boxSize;          // the GLOBAL periodic box
    real cutOff = 2.5, sigma = 1, epsilon = 1;
    real temperature = 0, dt = 0, friction = 1;
    real skin = 0;          // 0: exchange every step
    int exchangeEvery = 1;
    uint seed = 1234;
    real capacityFactor = 1.25;
  };
  // localPos / localVel / localIds: the particles this rank owns, z already in the local frame (relative to the slab centre)
  DistributedLJ(std::shared_ptr<Comm> comm_, Parameters par, const std::vector<real4> &localPos, const std::vector<real3> &localVel,
                const std::vector<int> &localIds)
      : comm(comm_), rank(comm_->rank()), world(comm_->world()), L(par.boxSize), rc(par.cutOff), skin(par.skin), dt(par.dt),
        friction(par.friction), temperature(par.temperature), exchangeEvery(par.skin > 0 ? par.exchangeEvery : 1), seed(par.seed) {
    width = L.z / world;
    if (world > 1 && width < rc + 3 * skin) throw std::invalid_argument("slab thinner than the cut-off (+ skin): halo would need second neighbours");
    noiseAmplitude = std::sqrt(2 * dt * friction * temperature);
    nOwned = (int)localPos.size();
    const real reach = rc + 3 * skin;
    const int ghosts = (int)(2.2 * nOwned * reach / width) + 4096;
    cap = (int)(par.capacityFactor * nOwned) + ghosts;
    pos.resize(cap); force.resize(cap); ref.resize(cap);
    vel.resize((size_t)3 * cap); ids.resize(cap); idx.resize((size_t)4 * cap); holes.resize(cap); counts.resize(4);
    rows.resize((size_t)8 * cap); arrivals.resize((size_t)8 * cap); sendUp.resize((size_t)4 * cap); sendDown.resize((size_t)4 * cap); maxd.resize(1);
    size_t tb = 0;
    detail::check(uammd_slab_select_workspace(cap, &tb));
    tiles.resize(tb);
    fuseFirstHalfStep = width > 2 * reach;
    if (fuseFirstHalfStep) listed.resize(cap);
    detail::hipCheck(hipMemcpy(pos.d, localPos.data(), sizeof(real4) * nOwned, hipMemcpyHostToDevice), "hipMemcpy");
    detail::hipCheck(hipMemcpy(vel.d, localVel.data(), sizeof(real3) * nOwned, hipMemcpyHostToDevice), "hipMemcpy");
    detail::hipCheck(hipMemcpy(ids.d, localIds.data(), sizeof(int) * nOwned, hipMemcpyHostToDevice), "hipMemcpy");
    pot.setPotParameters(0, 0, Potential::LJ::InputPairParameters{par.cutOff, par.sigma, par.epsilon, false});
    detail::check(uammd_celllist_create(&cl));
    // 1 % slack so that a ghost sitting exactly on the halo face is still inside the local box
    boxL[0] = L.x; boxL[1] = L.y; boxL[2] = width + real(2.02) * (rc + 4 * skin);
    boxPer[0] = boxPer[1] = 1; boxPer[2] = 0;
    const float rc3[3] = {rc, rc, rc};
    detail::check(uammd_celllist_create_grid(boxL, boxPer, rc3, cellDim, updL, updPer));
  }
  DistributedLJ(const DistributedLJ &) = delete;
  ~DistributedLJ() { uammd_celllist_destroy(cl); }
  // VerletNVT::GronbechJensen::forwardTime (GronbechJensen.cu:88-115) on the slab
  void forwardTime() {
    steps++;
    if (steps == 1) { refresh(); forces(); }
    const bool refreshNow = (steps - 1) % exchangeEvery == 0;
    if (!refreshNow && fuseFirstHalfStep) { fusedStep(); return; }
    integrate(1);
    if (refreshNow) refresh(); else refill();
    forcesAndSecondHalfStep();
  }
  int numberOwned() const { return nOwned; }
  int numberGhosts() const { return nAll - nOwned; }
  real slabCentre() const { return -L.z / 2 + (rank + real(0.5)) * width; }
  // host copies of the owned particles (z in the local frame) — synchronises
  void download(std::vector<real4> &p, std::vector<real3> &v, std::vector<int> &i) {
    p.resize(nOwned); v.resize(nOwned); i.resize(nOwned);
    detail::hipCheck(hipStreamSynchronize(st), "hipStreamSynchronize");
    detail::hipCheck(hipMemcpy(p.data(), pos.d, sizeof(real4) * nOwned, hipMemcpyDeviceToHost), "hipMemcpy");
    detail::hipCheck(hipMemcpy(v.data(), vel.d, sizeof(real3) * nOwned, hipMemcpyDeviceToHost), "hipMemcpy");
    detail::hipCheck(hipMemcpy(i.data(), ids.d, sizeof(int) * nOwned, hipMemcpyDeviceToHost), "hipMemcpy");
  }
  // the cached exchange is exact only if nobody out-ran the skin between two refreshes (synchronises)
  void checkSkin() {
    if (!haveDrift) return;
    real m = 0;
    detail::hipCheck(hipMemcpy(&m, maxd.d, sizeof(real), hipMemcpyDeviceToHost), "hipMemcpy");
    if (m > skin) throw std::runtime_error("a particle moved " + std::to_string(m) + " between refreshes of the halo lists, more than the skin");
  }
  // total number of owned particles over the ranks (an all-reduce of one float per rank; synchronises)
  long totalParticles() {
    detail::DeviceArray<real> n(1);
    const real mine = (real)nOwned;
    detail::hipCheck(hipMemcpy(n.d, &mine, sizeof(real), hipMemcpyHostToDevice), "hipMemcpy");
    comm->allReduceSum(n.d, 1, st);
    detail::hipCheck(hipStreamSynchronize(st), "hipStreamSynchronize");
    real t = 0;
    detail::hipCheck(hipMemcpy(&t, n.d, sizeof(real), hipMemcpyDeviceToHost), "hipMemcpy");
    return (long)(t + real(0.5));
  }
};

// ---------------------------------------------------------------------------------------------------------------------------------------
// Path B.  Rank r owns nz / world xy planes of the grid inside a window with `halo` planes either side (whole 8-node tiles where the grid
// allows, so that the tile-owned spreading kernel runs unchanged) and, after the transpose, ny / world y rows of the Fourier grid.
// Per solve: spread -> halo planes to the neighbours, ADDED (2 messages) -> 2-D R2C of the owned planes -> all-to-all -> z transform +
// Stokes operator + noise + inverse z transform on the y rows -> all-to-all -> 2-D C2R -> halo planes from the neighbours, COPIED ->
// gather.  The Fourier noise needs no communication: node id draws from Saru(id, seed, seed2) with the GLOBAL node index and the
// operator regenerates a conjugate partner's draw locally, so the ranks only share (seed, seed2), advancing in lock step.
class DistributedFCM {
  std::shared_ptr<Comm> comm;
  int rank, world;
  int cells[3], nzl, nyl, nkx, nxpad, he, halo, nzw;
  real L[3];
  uammd_fcm_slab *h = nullptr;
  detail::DeviceArray<real> grid, inter, sendBuf, zbuf, haloDown, haloUp;
  uint seed2 = 0;
  hipStream_t st = 0;
  size_t planeFloats() const { return (size_t)3 * cells[1] * nxpad; }      // one z plane of the window [c][y][x]
  size_t interPlane() const { return (size_t)4 * cells[1] * cells[0]; }   // one z plane of the float4 window
public:
  struct Parameters {
    real3 boxSize;
    int3 cells;
    real viscosity = 1, tolerance = 1e-3;
    uint seed = 1234;
  };
  DistributedFCM(std::shared_ptr<Comm> comm_, Parameters par) : comm(comm_), rank(comm_->rank()), world(comm_->world()) {
    cells[0] = par.cells.x; cells[1] = par.cells.y; cells[2] = par.cells.z;
    L[0] = par.boxSize.x; L[1] = par.boxSize.y; L[2] = par.boxSize.z;
    if (cells[2] % world || cells[1] % world) throw std::invalid_argument("the grid does not split into this many z slabs and y row blocks");
    nzl = cells[2] / world; nyl = cells[1] / world; nkx = cells[0] / 2 + 1; nxpad = 2 * nkx;
    const real hmin = std::min(L[0] / cells[0], std::min(L[1] / cells[1], L[2] / cells[2]));
    uammd_fcm_parameters p{};
    float aeff = 0;
    detail::check(uammd_fcm_gaussian_kernel(hmin, par.tolerance, &p.kernel, &aeff));
    he = p.kernel.support[2] / 2 + 2;  // planes a stencil of an owned particle can reach outside the slab (+1: the even-support shift, +1: rounding)
    const bool tiled = cells[0] % 8 == 0 && cells[1] % 8 == 0 && nzl % 8 == 0;
    halo = tiled ? ((he + 7) / 8) * 8 : he;
    he = halo;
    if (nzl < he) throw std::invalid_argument("slab thinner than the spreading stencil: halo would need second neighbours");
    nzw = nzl + 2 * halo;
    for (int k = 0; k < 3; ++k) { p.boxSize[k] = L[k]; p.cells[k] = cells[k]; }
    p.viscosity = par.viscosity;
    p.seed = par.seed;
    detail::check(uammd_fcm_slab_create(&p, nzl, rank * nzl, halo, nyl, rank * nyl, &h));
    grid.resize((size_t)nzw * planeFloats());
    inter.resize((size_t)nzw * interPlane());
    const size_t spec = (size_t)nzl * 3 * cells[1] * nkx * 2;   // floats of the owned planes' spectrum = of this rank's z buffer
    sendBuf.resize(spec); zbuf.resize(spec);
    haloDown.resize((size_t)he * planeFloats()); haloUp.resize((size_t)he * planeFloats());
  }
  DistributedFCM(const DistributedFCM &) = delete;
  ~DistributedFCM() { uammd_fcm_slab_destroy(h); }
  int haloPlanes() const { return halo; }
  real slabWidth() const { return L[2] / world; }
  // d_posLocal real4[N] (z relative to the centre of the owned slab), d_force real4[N] -> d_vel real3[N] = M F + prefactor sqrt(2 T) M^(1/2) dW
  void computeHydrodynamicDisplacements(const real4 *d_posLocal, const real4 *d_force, int N, real temperature, real prefactor, real3 *d_vel) {
    if (temperature > 0) seed2++;
    void *s = (void *)st;
    detail::check(uammd_fcm_slab_spread(h, (const float *)d_posLocal, (const float *)d_force, N, grid.d, s));
    // the halo planes above the owned block go up, those below go down; what arrives is folded into the first / last owned planes
    real *g = grid.d;
    const size_t pf = planeFloats();
    comm->haloExchange(g + (size_t)(halo + nzl) * pf, 1, g + (size_t)(halo - he) * pf, 1, haloDown.d, 1, haloUp.d, 1, (int)((size_t)he * pf), st);
    int rc = uammd_fcm_slab_forward_xy_fold(h, g, haloDown.d, haloUp.d, he, s);
    if (rc == 1) {  // not served (rocFFT grid or overlapping folds): add the planes, then transform
      detail::check(uammd_slab_add2(g + (size_t)halo * pf, haloDown.d, g + (size_t)(halo + nzl - he) * pf, haloUp.d, (size_t)he * pf, s));
      rc = uammd_fcm_slab_forward_xy(h, g, s);
    }
    detail::check(rc);
    const size_t blockBytes = sizeof(real) * (size_t)nzl * 3 * nyl * nkx * 2;
    if (world > 1) {
      detail::check(uammd_fcm_slab_transpose_pack(h, g, sendBuf.d, s));
      comm->allToAll(sendBuf.d, zbuf.d, blockBytes, st);
    }
    real *z = world > 1 ? zbuf.d : g + (size_t)halo * pf;   // (one rank: the spectrum of the window IS the z buffer)
    rc = uammd_fcm_slab_z_fused(h, z, d_force != nullptr, temperature, prefactor, seed2, s);
    if (rc == 1) {
      detail::check(uammd_fcm_slab_fft_z(h, z, 0, s));
      detail::check(uammd_fcm_slab_kspace(h, z, d_force != nullptr, temperature, prefactor, seed2, s));
      rc = uammd_fcm_slab_fft_z(h, z, 1, s);
    }
    detail::check(rc);
    if (world > 1) {
      comm->allToAll(zbuf.d, sendBuf.d, blockBytes, st);
      detail::check(uammd_fcm_slab_transpose_unpack(h, sendBuf.d, g, s));
    }
    // back to real space: the float4 window when the library's own FFT serves the grid, the planar window otherwise
    rc = uammd_fcm_slab_inverse_xy_inter(h, g, inter.d, s);
    if (rc == 0) {
      const size_t ip = interPlane();
      real *w = inter.d;
      comm->haloExchange(w + (size_t)(halo + nzl - he) * ip, 1, w + (size_t)halo * ip, 1, w + (size_t)(halo - he) * ip, 1, w + (size_t)(halo + nzl) * ip, 1,
                         (int)((size_t)he * ip), st);
      detail::check(uammd_fcm_slab_gather_inter(h, (const float *)d_posLocal, N, w, (float *)d_vel, s));
    } else if (rc == 1) {
      detail::check(uammd_fcm_slab_inverse_xy(h, g, s));
      comm->haloExchange(g + (size_t)(halo + nzl - he) * pf, 1, g + (size_t)halo * pf, 1, g + (size_t)(halo - he) * pf, 1, g + (size_t)(halo + nzl) * pf, 1,
                         (int)((size_t)he * pf), st);
      detail::check(uammd_fcm_slab_gather(h, (const float *)d_posLocal, N, g, (float *)d_vel, s));
    } else
      detail::check(rc);
  }
  uint getSeed2() const { return seed2; }
};

}  // namespace uammd
#endif
