/* libuammd_hip — C-ABI of the MI355X (gfx950) implementation of UAMMD's data-parallel hot path.
 *
 * UAMMD is a header-only C++/CUDA template library: there is no reference FFI.  The entry points
 * below are what a C ABI for its two hot paths binds — one per reference call site that crosses
 * host->device — each citing the reference interface it replaces (paths relative to
 * /root/reference/src).  The UAMMD-shaped C++14 headers in include/uammd/ call these.
 *
 * Conventions
 *   - every pointer argument named d_* is a DEVICE pointer (HBM), everything else is host memory;
 *   - `real4` arrays are plain float[4*N] (x,y,z,w; w = particle type for positions), `real3`
 *     arrays are float[3*N], matching UAMMD's default `real=float` build (global/defines.h:33-44);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous
 *     with respect to the host unless stated;
 *   - return value: 0 = ok, non-zero = error; uammd_hip_last_error() returns the message
 *     (thread local).  The C++ headers turn non-zero into the exceptions UAMMD throws
 *     (utils/debugTools.h:20-64).
 */
#ifndef UAMMD_HIP_H
#define UAMMD_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UAMMD_HIP_ABI_VERSION 1

int uammd_hip_abi_version(void);
const char *uammd_hip_last_error(void);
/* number of visible devices / select one (System::System --device N, System/System.h:128-139) */
int uammd_hip_device_count(int *count);
int uammd_hip_set_device(int device);

/* ------------------------------------------------------------------------------------------------
 * Path A — neighbour search.  Replaces CellList / CellListBase / ParticleSorter:
 *   Interactor/NeighbourList/CellList.cuh:100-126,149-182        (createUpdateGrid, update, transverseList)
 *   Interactor/NeighbourList/CellList/CellListBase.cuh:124-172   (update, getCellList, CellListData)
 *   utils/ParticleSorter.cuh:156-164,178-187,243-274,303-321     (hash, stable radix sort, reorder)
 * ---------------------------------------------------------------------------------------------- */
typedef struct uammd_celllist uammd_celllist; /* opaque; owns index/hash/sortPos/cellStart/cellEnd */

/* POD view of the list handed to kernels — field-for-field CellListBase::CellListData
 * (CellListBase.cuh:145-160) plus the sorted Morton keys. */
typedef struct {
  const unsigned int *d_cellStart; /* first sorted index of the cell + VALID_CELL; < VALID_CELL = empty */
  const int *d_cellEnd;            /* one past the last sorted index of the cell */
  const float *d_sortPos;          /* real4[N], positions in sorted (Morton cell) order */
  const int *d_groupIndex;         /* sorted index -> index in the input array */
  const unsigned int *d_sortHash;  /* Morton key of each sorted particle */
  int cellDim[3];
  float boxSize[3];
  int periodic[3];
  unsigned int VALID_CELL;
  int numberParticles;
} uammd_celllist_data;

int uammd_celllist_create(uammd_celllist **out);
int uammd_celllist_destroy(uammd_celllist *h);

/* Host logic of CellList::createUpdateGrid (CellList.cuh:100-126): cellDim = int3(L/cutOff) by C
 * truncation in float, any dimension with <= 3 cells collapses to 1, an infinite dimension becomes
 * 64*cutOff and non periodic. */
int uammd_celllist_create_grid(const float L[3], const int periodic[3], const float cutOff[3], int cellDim_out[3],
                               float L_out[3], int periodic_out[3]);

/* CellListBase::update(pos, N, grid, st): hash -> stable sort on [0,maxbit) -> reorder -> cell
 * tables.  d_pos is real4[N]. */
int uammd_celllist_update(uammd_celllist *h, const float *d_pos, int numberParticles, const float L[3],
                          const int periodic[3], const int cellDim[3], void *stream);
int uammd_celllist_get(uammd_celllist *h, uammd_celllist_data *out);
/* NaN positions or particles outside a non-periodic box: the reference raises a device flag (CellListBase.cuh:82-85) and, in
 * UAMMD_DEBUG builds only, synchronises and throws overflow_error("CellList encountered NaN positions") inside every update
 * (:258-264); its release builds carry on silently.  Here the build kernels raise a flag in host-mapped memory and the NEXT
 * uammd_celllist_update / _get on the handle fails with that message (no per-step synchronisation).
 * uammd_celllist_check_errors synchronises `stream` and reports at once; option "strict_errors" = 1 makes every update do so
 * (the reference's debug behaviour); option "report_errors" = 0 gives the reference's release behaviour. */
int uammd_celllist_check_errors(uammd_celllist *h, void *stream);
/* Measurement hook (no reference counterpart): while enabled, every LJ traversal of this list that runs as ONE kernel (the TILE
 * kernels) is launched through hipExtLaunchKernel with a start / stop event pair attached to its own dispatch, and
 * uammd_lj_profile_read returns the summed kernel time (ms) and the number of launches since the enable.  bench.py derives
 * roofline.achieved from it: every launch of the timed region, no events between kernels. */
/* Slab decomposition (new design, the reference is single GPU): the positions a rank sends to its two z neighbours —
 * d_outUp[k] = pos[idxUp[k]] + (0, 0, dzUp), d_outDown[k] = pos[idxDown[k]] + (0, 0, dzDown) — in one launch. */
int uammd_halo_pack(const float *d_pos, const int *d_idxUp, int nUp, const int *d_idxDown, int nDown, float dzUp, float dzDown,
                    float *d_outUp, float *d_outDown, void *stream);
/* The decomposed step with GronbechJensen's first half step (Integrator/VerletNVT/GronbechJensen.cu:86-116) folded into the two
 * kernels that read the positions anyway, instead of a launch of its own before the exchange:
 *   uammd_halo_pack_gj1: the half step of every LISTED row (noise keyed by d_keys[row], the global particle id; force zeroed), then
 *     uammd_halo_pack of the new positions.  The up and down lists must be disjoint (a slab wider than two reaches): a row in both
 *     would take the half step twice.
 *   uammd_celllist_update_gj1: uammd_celllist_update over rows [0, numberParticles) = owned + ghosts, the half step applied, as the
 *     rows are loaded, to the owned rows [0, numberOwned) with d_skip[row] == 0 (d_skip: one byte per owned row, the listed rows —
 *     uammd_slab_refresh_lj writes it; NULL: nobody is skipped).  The list is built from the new positions; forces of the
 *     integrated rows are zeroed.  Bit-identical to uammd_verletnvt_gj_keyed(1, ...) followed by the plain calls. */
int uammd_halo_pack_gj1(float *d_pos, float *d_vel, float *d_force, const float *d_mass, float defaultMass, const int *d_keys,
                        const int *d_idxUp, int nUp, const int *d_idxDown, int nDown, float dzUp, float dzDown, float *d_outUp,
                        float *d_outDown, float dt, float friction, int is2D, float noiseAmplitude, unsigned int stepNum, unsigned int seed,
                        void *stream);
int uammd_celllist_update_gj1(uammd_celllist *h, float *d_pos, int numberParticles, const float L[3], const int periodic[3],
                              const int cellDim[3], float *d_vel, float *d_force, const float *d_mass, float defaultMass,
                              const int *d_keys, const unsigned char *d_skip, int numberOwned, float dt, float friction, int is2D,
                              float noiseAmplitude, unsigned int stepNum, unsigned int seed, void *stream);
/* The bookkeeping of a membership refresh of the slab decomposition (uammd_amd/csrc/slab.hip):
 *   uammd_slab_select: ASCENDING indices of the rows with z >= zUp (d_idxUp) and with z < zDown (d_idxDown), their numbers in
 *     d_counts[0..1] (device memory; the caller reads them to size its messages); d_workspace: uammd_slab_select_workspace(n) bytes;
 *   uammd_slab_pack_rows: the listed rows as 8 floats {x, y, z + dz, w, vx, vy, vz, id} for the neighbour above / below;
 *   uammd_slab_unpack_rows: arrivals (8-float rows, those from below first) into the holes the leavers make, arrivals in excess
 *     appended at row n, and when fewer arrive than leave the staying rows of the tail moved into the open holes: afterwards the
 *     first n - nUp - nDown + nArrive rows are the owned particles.  d_holes: nUp + nDown ints of scratch;
 *   uammd_slab_max_displacement: atomicMax of |pos - ref| over the rows into *d_max (a float the caller zeroes). */
int uammd_slab_select_workspace(int n, size_t *bytes);
int uammd_slab_select(const float *d_pos, int n, float zUp, float zDown, int *d_idxUp, int *d_idxDown, int *d_counts, void *d_workspace,
                      void *stream);
int uammd_slab_pack_rows(const float *d_pos, const float *d_vel, const int *d_ids, const int *d_idxUp, int nUp, const int *d_idxDown,
                         int nDown, float dzUp, float dzDown, float *d_outUp, float *d_outDown, void *stream);
int uammd_slab_unpack_rows(float *d_pos, float *d_vel, int *d_ids, int n, const int *d_idxUp, int nUp, const int *d_idxDown, int nDown,
                           const float *d_arrivals, int nArrive, int *d_holes, void *stream);
int uammd_slab_max_displacement(const float *d_pos, const float *d_ref, int n, float *d_max, void *stream);
/* two segments of `count` floats in one launch: dst0 += src0, dst1 += src1 (the halo planes of the spread grid folded into the owned
 * planes, FCM slab decomposition) and dst0 = src0, dst1 = src1 (the planes received before the gather); a segment's dst and src
 * must not overlap */
int uammd_slab_add2(float *d_dst0, const float *d_src0, float *d_dst1, const float *d_src1, size_t count, void *stream);
int uammd_slab_copy2(float *d_dst0, const float *d_src0, float *d_dst1, const float *d_src1, size_t count, void *stream);
/* Measurement hook (no reference counterpart): counters of the tile traversal (AUTO / TILE) of this list.  The call first copies the
 * counters accumulated so far into out (nullable; zeros when the hook was off) — out[0] = workgroups (2 x 2 x 2-cell bricks) that took
 * the in-kernel dense-brick fallback, out[1] = bricks launched, out[2..3] reserved, out[4..15] = phase times of diagnostic builds
 * (-DUAMMD_TILE_TIMELINE, tools/variants_tile.sh; zeros otherwise) —, then zeroes them and turns the hook on (enable != 0) or off.
 * tests/test_gpu_bench_state.py asserts with it that the melted state bench.py times runs on the main path. */
int uammd_lj_tile_stats(uammd_celllist *h, int enable, unsigned int out[16], void *stream);
int uammd_lj_profile_enable(uammd_celllist *h, int enable);
int uammd_lj_profile_read(uammd_celllist *h, double *total_ms, long long *launches);
/* options: "force_radix" = 1 makes the build use the stable radix sort path (test hook); "num_owned" = n marks the
 * particles with input index >= n as ghosts of a domain decomposition: they are neighbours of the others but the LJ
 * traversal computes nothing for them (n < 0 turns it off) */
int uammd_celllist_set_option(uammd_celllist *h, const char *name, int value);
/* library-wide tunables (none at present; the call is kept for ABI stability and returns an error for unknown names) */
int uammd_hip_set_tunable(const char *name, int value);

/* ParticleSorter::updateOrderWithCustomHash + applyCurrentOrder building blocks
 * (ParticleSorter.cuh:118-135,178-187): stable sort of (key,value) on key bits [0,end_bit) and a
 * gather out[i] = in[index[i]] for 4/8/12/16/24/32-byte elements.  Used by ParticleData::sortParticles. */
int uammd_sort_pairs(unsigned int *d_keys, int *d_values, int n, int end_bit, void *stream);
int uammd_gather(const void *d_in, const int *d_index, void *d_out, int n, int elem_bytes, void *stream);
/* the inverse: out[index[i]] = in[i] (distinct indices).  What assigning through pg->getPropertyIterator(prop) does in the reference
 * (ParticleData/ParticleGroup.cuh:473-496): the modules that solve on the gathered rows of a proper subgroup write positions back with it. */
int uammd_scatter(const void *d_in, const int *d_index, void *d_out, int n, int elem_bytes, void *stream);
/* out[i] = (float) in[i]: the single-precision image of a DOUBLE_PRECISION ParticleData's positions, whose cell list orders
 * ParticleData::sortParticles (ParticleData.cuh:492-522: a memory-locality order) */
int uammd_convert_f64_to_f32(const double *d_in, float *d_out, size_t count, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Path A — traversal with the Lennard-Jones Transverser.  Replaces
 *   CellList::transverseList<Radial<LJFunctor>::Transverser>      Interactor/NeighbourList/CellList.cuh:165-182
 *   NeighbourList_ns::transverseWithNeighbourContainer            Interactor/NeighbourList/common.cuh:10-34
 *   Radial<LJFunctor>::Transverser::compute/set                   Interactor/Potential/RadialPotential.cuh:107-127
 *   NBody::transverse (small-box fallback, PairForces.cu:49-53)   Interactor/NBodyBase.cuh:46-159
 * ---------------------------------------------------------------------------------------------- */
/* LJFunctor::PairParameters (Potential/Potential.cuh:31-35) */
typedef struct { float cutOff2, sigma2, epsilonDivSigma2, shift; } uammd_lj_pair_parameters;
/* LJFunctor::processPairParameters (Potential/Potential.cuh:66-82), host */
int uammd_lj_process_pair_parameters(float cutOff, float sigma, float epsilon, int shift, uammd_lj_pair_parameters *out);
/* A list caches what it needs of the table (largest cut-off; whether sigma = epsilon = 1) keyed by the table's device address.  After
 * rewriting a table in place — or uploading a new one into memory a previous table occupied — call this before the next traversal: every
 * list reads its table again (one small device-to-host copy).  Potential::LJ::setPotParameters between steps (legal in the reference,
 * Potential.cuh:60-82) goes through here in both host layers. */
int uammd_lj_table_changed(void);

/* flags for `algo` */
#define UAMMD_LJ_ALGO_AUTO 0    /* the kernel measured fastest for the grid: TILE where the grid allows it (forces to rounding level, the
                                   north star's stated tolerance), else EXACT */
#define UAMMD_LJ_ALGO_TILE 8    /* wave per pair of x-adjacent cells, 32 x 32 distance tiles on the matrix pipe (f32 MFMA), hits re-evaluated
                                   with the reference's arithmetic: same pairs, sums in another order (rounding-level differences) */
#define UAMMD_LJ_ALGO_TILE1 10  /* TILE with one wave per pair of cells and a private 4 x 3 x 3-cell halo (TILE = a workgroup of four waves per
                                   2 x 2 x 2 brick of cells sharing one staged halo); same results as TILE up to summation order */
#define UAMMD_LJ_ALGO_EXACT 9   /* the fastest kernel that keeps the reference's summation order, bit-identical to GENERAL
                                   (RING_HALF where the grid allows it, else RING, else GENERAL) */
#define UAMMD_LJ_ALGO_GENERAL 1 /* thread-per-particle walk of the 27 cells (any grid) */
#define UAMMD_LJ_ALGO_RING 6    /* GENERAL with a ring FIFO: a drain takes a few pairs from every lane (bit-identical to GENERAL) */
#define UAMMD_LJ_ALGO_RING_HALF 7 /* RING + half-precision superset prefilter, two candidates per load (bit-identical to GENERAL) */

/* d_paramTable: ntypes*ntypes PairParameters indexed [ti + ntypes*tj] (ParameterHandler.cuh:17-37);
 * d_force real4[·], d_energy/d_virial real[·] are nullable and ACCUMULATED into at the particle's
 * index in the input array (Transverser::set does +=); d_globalIndex (group->global, nullable). */
int uammd_lj_transverse_celllist(uammd_celllist *h, const uammd_lj_pair_parameters *d_paramTable, int ntypes,
                                 const float boxL[3], const int boxPeriodic[3], float *d_force, float *d_energy,
                                 float *d_virial, const int *d_globalIndex, int algo, void *stream);
int uammd_lj_transverse_nbody(const float *d_pos, int numberParticles, const uammd_lj_pair_parameters *d_paramTable,
                              int ntypes, const float boxL[3], const int boxPeriodic[3], float *d_force,
                              float *d_energy, float *d_virial, const int *d_globalIndex, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Path A — Verlet list.  Replaces
 *   VerletListBase::{update,needsRebuild,getVerletList,setCutOffMultiplier,forceNextUpdate,getNumberOfStepsSinceLastUpdate}
 *                                                               Interactor/NeighbourList/VerletList/VerletListBase.cuh:73-199
 *   BasicNeighbourListBase::{update,getBasicNeighbourList}      Interactor/NeighbourList/BasicList/BasicListBase.cuh:76-215
 *   fillBasicNeighbourList (K7), checkMaximumDrift (K8)         BasicListBase.cuh:41-75, VerletListBase.cuh:55-69
 *   VerletList::transverseList<Radial<LJFunctor>::Transverser>  Interactor/NeighbourList/VerletList.cuh:138-155
 * The list is built on the cell list of the STORED positions with cut-off multiplier*cutOff (1.08), capacity 32 per
 * particle growing by 32; entry k of sorted particle i is d_neighbourList[k*particleStride + i]; the particle itself is
 * one of its neighbours (r2 = 0 <= cutOff2, as in the reference).
 * ---------------------------------------------------------------------------------------------- */
typedef struct uammd_verletlist uammd_verletlist;
/* BasicNeighbourListBase::BasicNeighbourListData (BasicListBase.cuh:144-152), with VerletListBase's current sortPos */
typedef struct {
  const int *d_neighbourList;
  const int *d_numberNeighbours;
  const float *d_sortPos;   /* real4[N]: CURRENT positions in the order of the last build */
  const int *d_groupIndex;  /* sorted index -> index in the input array */
  int particleStride;       /* = numberParticles */
  int maxNeighboursPerParticle;
  int numberParticles;
} uammd_verletlist_data;
int uammd_verletlist_create(uammd_verletlist **out);
int uammd_verletlist_destroy(uammd_verletlist *h);
/* VerletListBase::update(pos, N, box, cutOff, st): rebuilds if forced / box, cutOff or N changed / some particle drifted
 * >= (multiplier*cutOff - cutOff)/2 from its stored position (one 4-byte read-back, as in the reference), then refreshes
 * sortPos.  *rebuilt (nullable) tells whether the list was rebuilt. */
int uammd_verletlist_update(uammd_verletlist *h, const float *d_pos, int numberParticles, const float L[3],
                            const int periodic[3], float cutOff, void *stream, int *rebuilt);
int uammd_verletlist_force_next_update(uammd_verletlist *h);
int uammd_verletlist_set_cutoff_multiplier(uammd_verletlist *h, float newMultiplier);
int uammd_verletlist_get_steps_since_last_update(uammd_verletlist *h, int *steps);
int uammd_verletlist_get(uammd_verletlist *h, uammd_verletlist_data *out);
int uammd_lj_transverse_verletlist(uammd_verletlist *h, const uammd_lj_pair_parameters *d_paramTable, int ntypes,
                                   const float boxL[3], const int boxPeriodic[3], float *d_force, float *d_energy,
                                   float *d_virial, const int *d_globalIndex, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Integrator kernels.  Replace
 *   VerletNVT::GronbechJensen_ns::integrateGPU<step>   Integrator/VerletNVT/GronbechJensen.cu:28-62
 *   VerletNVT::Basic_ns::integrateGPU<step>            Integrator/VerletNVT/Basic.cu:86-114
 *   VerletNVT::Basic_ns::initialVelocities             Integrator/VerletNVT/Basic.cu:12-29
 *   BD::EulerMaruyama_ns::integrateGPU                 Integrator/BrownianDynamics.cu:119-144
 *   BDHI::FCM_ns::integrateEulerMaruyamaD              Integrator/BDHI/BDHI_FCM.cu:67-92
 * d_vel is real3[N]; d_mass nullable (defaultMass > 0 wins); d_index = ParticleGroup index iterator
 * (nullable = identity).
 * ---------------------------------------------------------------------------------------------- */
int uammd_verletnvt_gj(int step, float *d_pos, float *d_vel, float *d_force, const float *d_mass, float defaultMass,
                       const int *d_index, int numberParticles, float dt, float friction, int is2D,
                       float noiseAmplitude, unsigned int stepNum, unsigned int seed, void *stream);
/* The same step with the thermostat's random stream of thread t keyed by d_noiseKey[t] instead of t (nullable: then exactly
 * uammd_verletnvt_gj).  New design for the domain-decomposed drivers (the reference is single GPU): the key is the particle's GLOBAL id,
 * so a particle draws the same kicks whichever rank and row holds it — the trajectory of an N-rank run does not depend on the
 * decomposition or on the order migration leaves the rows in, and slabs are not correlated through equal row numbers. */
int uammd_verletnvt_gj_keyed(int step, float *d_pos, float *d_vel, float *d_force, const float *d_mass, float defaultMass,
                             const int *d_index, const int *d_noiseKey, int numberParticles, float dt, float friction, int is2D,
                             float noiseAmplitude, unsigned int stepNum, unsigned int seed, void *stream);
/* One whole VerletNVT::GronbechJensen::forwardTime (Integrator/VerletNVT/GronbechJensen.cu:88-115) whose only interactor is
 * PairForces<Potential::LJ, CellList> on every particle (Interactor/PairForces.cu:43-78), fused into five launches: the first half step
 * rides in the cell list's hash kernel, the second in the traversal's store.  Bit-identical to the sequence
 *   uammd_verletnvt_gj(1) -> uammd_celllist_update(updateL, updatePeriodic, cellDim) -> uammd_lj_transverse_celllist(algo) -> uammd_verletnvt_gj(2)
 * which is also what runs where the list does not take the aggregated counting build or the tile kernel.  d_force holds f(t) on entry
 * (the integrator's first step computes it the plain way) and f(t + dt) on return; updateL / updatePeriodic / cellDim as
 * uammd_celllist_create_grid returns them for (boxL, boxPeriodic, cut-off). */
int uammd_verletnvt_gj_lj_step(uammd_celllist *h, float *d_pos, float *d_vel, float *d_force, const float *d_mass, float defaultMass,
                               int numberParticles, const float boxL[3], const int boxPeriodic[3], const float updateL[3],
                               const int updatePeriodic[3], const int cellDim[3], const uammd_lj_pair_parameters *d_paramTable,
                               int ntypes, float dt, float friction, int is2D, float noiseAmplitude, unsigned int stepNum,
                               unsigned int seed, int algo, void *stream);

/* The second half of that fusion alone, for lists with ghosts (the domain-decomposed drivers exchange the halo between the first half
 * step and the list build, so only the traversal's store can carry a half step): PairForces<Potential::LJ, CellList>::sum on the owned
 * rows (option num_owned of the list) with GronbechJensen's second half step (GronbechJensen.cu:28-62) applied to them in the store.
 * d_force must be zero on the owned rows on entry.  Same bits as uammd_lj_transverse_celllist -> uammd_verletnvt_gj(2) on the owned rows. */
int uammd_lj_transverse_celllist_gj2(uammd_celllist *h, const uammd_lj_pair_parameters *d_paramTable, int ntypes, const float boxL[3],
                                     const int boxPeriodic[3], float *d_force, float *d_vel, const float *d_mass, float defaultMass,
                                     float dt, int is2D, int algo, void *stream);

int uammd_verletnvt_basic(int step, float *d_pos, float *d_vel, float *d_force, const float *d_mass,
                          float defaultMass, const int *d_index, int numberParticles, float dt, float friction,
                          int is2D, float noiseAmplitude, unsigned int stepNum, unsigned int seed, void *stream);
int uammd_verletnvt_initial_velocities(float *d_vel, const int *d_index, float velAmplitude, int is2D,
                                       int numberParticles, unsigned int seed, void *stream);
/* VerletNVT::Basic::sumKineticEnergy (VerletNVT/Basic.cu:173-207): energy[i] += 0.5 m |v|^2 (Integrator::sumEnergy) */
int uammd_sum_kinetic_energy(const float *d_vel, float *d_energy, const float *d_mass, float defaultMass,
                             const int *d_index, int numberParticles, void *stream);
int uammd_bd_euler_maruyama(float *d_pos, const int *d_index, const float *d_force, const float K[9],
                            float selfMobility, const float *d_radius, float dt, int is2D, float temperature,
                            int numberParticles, unsigned int stepNum, unsigned int seed, void *stream);
int uammd_fcm_euler_maruyama(float *d_pos, const int *d_index, const float *d_linearVelocity, int numberParticles,
                             float dt, void *stream);
/* The other schemes of Integrator/BrownianDynamics.cuh:121-183 / .cu:160-387 — BD::MidPoint (two calls per step: substep 0 after the
 * first force evaluation, 1 after the second; d_aux real4[N] keeps the step's starting positions), BD::AdamsBashforth (d_aux real4[N] = the
 * previous step's forces in group order) and BD::Leimkuhler (d_originalIndex = ParticleData::getIdOrderedIndices, nullable = identity).
 * Arguments as uammd_bd_euler_maruyama.  MidPoint and AdamsBashforth key their noise by the GROUP index, as the reference does. */
#define UAMMD_BD_MIDPOINT 1
#define UAMMD_BD_ADAMS_BASHFORTH 2
#define UAMMD_BD_LEIMKUHLER 3
int uammd_bd_scheme_step(int scheme, int substep, float *d_pos, float *d_aux, const int *d_index, const int *d_originalIndex, const float *d_force,
                         const float K[9], float selfMobility, const float *d_radius, float dt, int is2D, float temperature, int numberParticles,
                         unsigned int stepNum, unsigned int seed, void *stream);
/* ... and the four schemes with real = double (d_pos, d_aux, d_force double4[N], d_radius double[N]; UAMMD_BD_EULER_MARUYAMA is the scheme of
 * uammd_bd_euler_maruyama).  The draws are float Gaussians, as in the reference's double build (Saru::gf). */
#define UAMMD_BD_EULER_MARUYAMA 0
int uammd_bd_scheme_step_f64(int scheme, int substep, double *d_pos, double *d_aux, const int *d_index, const int *d_originalIndex,
                             const double *d_force, const double K[9], double selfMobility, const double *d_radius, double dt, int is2D,
                             double temperature, int numberParticles, unsigned int stepNum, unsigned int seed, void *stream);
/* BDHI::EulerMaruyama_ns::integrateGPUD (Integrator/BDHI/BDHI_EulerMaruyama.cu:82-113): pos += dt (K pos + MF) + sqrt2Tdt BdW;
 * d_MF, d_BdW real3[N] (d_BdW nullable), K row-major 3x3 (nullable) */
int uammd_bdhi_euler_maruyama(float *d_pos, const int *d_index, const float *d_MF, const float *d_BdW, const float K[9],
                              int numberParticles, float sqrt2Tdt, float dt, int is2D, void *stream);
int uammd_fill_zero(void *d_ptr, size_t bytes, void *stream);
/* thrust::fill(pg->getPropertyIterator(prop), ..., T()) — zero element index[i] (elem_bytes each, a multiple of 4) for i < n */
int uammd_fill_zero_indexed(void *d_ptr, const int *d_index, int n, int elem_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Path B — Immersed Boundary spreading / interpolation on a regular grid.  Replaces
 *   IBM<Kernel,Grid,LinearIndex3D>::spread / gather     misc/IBM.cuh:99-203
 *   IBM_ns::particles2GridD / grid2ParticlesDTPP        misc/IBM.cu:83-147, :164-235
 * A user-defined window functor cannot cross a C ABI; the windows UAMMD ships are selected by `kind`
 * (a user-defined window is device code: it is compiled by hipcc with the user's
 * translation unit, as in the reference; include/uammd/device/Transverser.hip.hpp is that path for Transversers).
 * ---------------------------------------------------------------------------------------------- */
#define UAMMD_IBM_KERNEL_GAUSSIAN 0 /* prefactor*exp(tau r^2), 0 for r >= rmax   misc/IBM_kernels.cuh:28-40, BDHI/FCM/FCM_kernels.cuh:22-58 */
#define UAMMD_IBM_KERNEL_PESKIN3 1  /* Peskin::threePoint                          misc/IBM_kernels.cuh:115-137 */
#define UAMMD_IBM_KERNEL_PESKIN4 2  /* Peskin::fourPoint                           misc/IBM_kernels.cuh:140-160 */
#define UAMMD_IBM_KERNEL_CONSTANT 3 /* phi = 1 (test/misc/ibm/test_ibm_regular.cu:11-14) */
#define UAMMD_IBM_KERNEL_BARNETT_MAGLAND 4 /* exp(beta(sqrt(1-(r/(a alpha))^2)-1))/(a norm); prefactor=1/norm, tau=beta, rmax=alpha,
                                             invh[0]=a (a LENGTH, not an inverse)  misc/IBM_kernels.cuh:82-112, FCM_kernels.cuh:82-155 */
#define UAMMD_IBM_KERNEL_SIXPOINT 5 /* GaussianFlexible::sixPoint, support 6, invh = 1/h        misc/IBM_kernels.cuh:162-236 */
#define UAMMD_IBM_KERNEL_GAUSS2D 6         /* BDHI2D_ns::Gaussian: prefactor*exp(tau r^2) in x and y, 1 in z   Integrator/Hydro/BDHI_quasi2D.cuh:112-132 */
#define UAMMD_IBM_KERNEL_GAUSS2D_DRIFT_X 7 /* GaussianThermalDrift<0>: the same times r along x                  BDHI_quasi2D.cuh:134-153 */
#define UAMMD_IBM_KERNEL_GAUSS2D_DRIFT_Y 8 /* GaussianThermalDrift<1> */
typedef struct {
  int kind;
  int support[3];
  float prefactor, tau, rmax; /* Gaussian */
  float invh[3];              /* Peskin: 1/h per axis */
} uammd_ibm_kernel;

/* FCM_ns::Kernels::Gaussian(h, tolerance) (BDHI/FCM/FCM_kernels.cuh:22-58), host: fills `out` and returns the
 * effective hydrodynamic radius a = h*u(tol)*sqrt(pi) in *a_eff. */
int uammd_fcm_gaussian_kernel(float h, float tolerance, uammd_ibm_kernel *out, float *a_eff);
/* IBM_kernels::BarnettMagland(alpha, beta) (misc/IBM_kernels.cuh:82-112), host: alpha = half support in length
 * units of `lengthUnit`, norm by the reference's 20000-interval Simpson rule.  `support` = nodes per axis (the
 * reference leaves it to the caller's getSupport).  lengthUnit = 1 is the plain window; the FCM wrapper
 * FCM_ns::Kernels::BarnettMagland(h, tol) is (alpha = w/2, beta = 3.6 w, support = w, lengthUnit = h),
 * BDHI/FCM/FCM_kernels.cuh:82-155. */
int uammd_ibm_barnett_magland_kernel(float alpha, float beta, int support, float lengthUnit, uammd_ibm_kernel *out);
/* Kernel::adviseGridSize (FCM_kernels.cuh:47-50), host */
float uammd_fcm_advise_grid_size(float hydrodynamicRadius, float tolerance);

/* d_pos: float[posStride*N] (xyz first; posStride 3 or 4).  d_quantity / d_out: float[ncomp*N].  d_grid:
 * float[ncomp * nxStride*ny*nz], components interleaved, node index i + nxStride*(j + ny*k)
 * (LinearIndex3D(nxStride, ny, nz)).  spread ADDS into the grid, gather ADDS into d_out (as the reference).
 * ncomp is 1 or 3.  A grid with cellDim[2] == 1 is treated as 2D (IBM.cuh:182-194). */
int uammd_ibm_spread(const float *d_pos, int posStride, const float *d_quantity, int ncomp, int numberParticles,
                     const float L[3], const int periodic[3], const int cellDim[3], int nxStride,
                     const uammd_ibm_kernel *kernel, float *d_grid, void *stream);
int uammd_ibm_gather(const float *d_pos, int posStride, float *d_out, int ncomp, int numberParticles,
                     const float L[3], const int periodic[3], const int cellDim[3], int nxStride,
                     const uammd_ibm_kernel *kernel, const float *d_grid, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Path B — Force Coupling Method (triply periodic Stokes).  Replaces
 *   FCM_impl<Gaussian,GaussianTorque>  ctor / computeHydrodynamicDisplacements / getSelfMobility
 *                                       Integrator/BDHI/FCM/FCM_impl.cuh:56-129, :652-693
 *   spreadForces, forwardTransform, convolveFourier, addBrownianNoise, inverseTransform, interpolateVelocity
 *                                       FCM_impl.cuh:245-262, :293-304, :399-411, :514-581
 *   BDHI::FCM::computeMF                Integrator/BDHI/BDHI_FCM.cuh:131-142
 * The solver owns its grids (3 planar padded real grids transformed in place by rocFFT).
 * ---------------------------------------------------------------------------------------------- */
typedef struct uammd_fcm uammd_fcm;
typedef struct {
  float boxSize[3];
  int cells[3];
  float viscosity;
  unsigned int seed;        /* FCM_impl::Parameters::seed (0 is NOT replaced here: the C++ header draws it) */
  uammd_ibm_kernel kernel;  /* spreading window (uammd_fcm_gaussian_kernel) */
  float hydrodynamicRadius; /* reported by getHydrodynamicRadius / used by getSelfMobility */
} uammd_fcm_parameters;
int uammd_fcm_create(const uammd_fcm_parameters *par, uammd_fcm **out);
int uammd_fcm_destroy(uammd_fcm *h);
/* FCM_impl::computeHydrodynamicDisplacements(pos, force, torque=nullptr, N, T, prefactor, st):
 * d_linearVelocity real3[N] is OVERWRITTEN with M F + prefactor*sqrt(2 T) M^{1/2} dW.  d_force may be NULL
 * (noise only).  Every call with temperature > 0 advances the noise stream (the reference's static seed2). */
int uammd_fcm_displacements(uammd_fcm *h, const float *d_pos, const float *d_force, int numberParticles,
                            float temperature, float prefactor, float *d_linearVelocity, void *stream);
/* BDHI::FCMIntegrator::forwardTime without torques (Integrator/BDHI/BDHI_FCM.cu:94-119 -> computeHydrodynamicDisplacements, then
 * integrateEulerMaruyamaD :67-92): the same velocities as uammd_fcm_displacements, then pos[i].xyz += v_i dt IN PLACE (the fma of
 * uammd_fcm_euler_maruyama: identical positions) by a kernel that also bins the positions it writes, so that the next call can
 * start at the tile scan.  d_linearVelocity real3[N] receives v; it may be NULL (a buffer of the handle is used).
 * flags: UAMMD_FCM_STEP_POSITIONS_KEPT = d_pos holds exactly what the previous uammd_fcm_step_euler_maruyama call on this handle left
 * there (same array, size and stream; nobody asked for write access in between: ParticleData::getPosWriteRequestedSignal is how the
 * host layers know) -> that call's binning is used instead of a binning pass.  Without the flag nothing is assumed.
 * option "bin_ahead" = 0 switches the binning ahead off. */
#define UAMMD_FCM_STEP_POSITIONS_KEPT 1
int uammd_fcm_step_euler_maruyama(uammd_fcm *h, float *d_pos, const float *d_force, int numberParticles, float temperature,
                                  float prefactor, float dt, float *d_linearVelocity, int flags, void *stream);
/* Hasimoto-corrected self mobility, FCM_impl::getSelfMobility (FCM_impl.cuh:102-119), host */
double uammd_fcm_self_mobility(double hydrodynamicRadius, double viscosity, double Lx);
/* noise call counter (the reference's process-global `static uint seed2`, FCM_impl.cuh:517; per handle here) */
int uammd_fcm_get_seed2(uammd_fcm *h, unsigned int *seed2);
int uammd_fcm_set_seed2(uammd_fcm *h, unsigned int seed2);
/* test hooks: run the pipeline up to the Fourier-space kernel (stage = 1; stage = 0 is the full solve) and copy the
 * Fourier grid out as complex3[nz*ny*(nx/2+1)] = {x.re,x.im,y.re,y.im,z.re,z.im} (the reference's layout) */
int uammd_fcm_displacements_staged(uammd_fcm *h, const float *d_pos, const float *d_force, int numberParticles,
                                   float temperature, float prefactor, float *d_linearVelocity, int stage,
                                   void *stream);
int uammd_fcm_export_fourier(uammd_fcm *h, float *d_out6, void *stream);
/* "atomic_spread" = 1 forces the one-wave-per-particle atomic spread/gather instead of the tile-owned kernels;
 * "tile_gather" = 1 selects the LDS-staged gather (supports <= 6; slower than the default at 24 particles per tile) */
int uammd_fcm_set_option(uammd_fcm *h, const char *name, int value);

/* Torques and rotation (FCM_impl.cuh:306-358, :590-649; FCM_kernels.cuh:60-80; BDHI_FCM.cu:67-92).  The torque window is
 * GaussianTorque(width = a/(6 sqrt(pi))^(1/3), h, tolerance); torques are spread with it, half their curl is added to the
 * Fourier forces, and the angular velocity is half the curl of the Fourier velocity interpolated with the same window.
 * d_dir: orientation quaternions real4 (n, vx, vy, vz), nullable. */
int uammd_fcm_torque_gaussian_kernel(float hydrodynamicRadius, float h, float tolerance, uammd_ibm_kernel *out);
int uammd_fcm_set_torque_kernel(uammd_fcm *h, const uammd_ibm_kernel *kernelTorque);
int uammd_fcm_displacements_torque(uammd_fcm *h, const float *d_pos, const float *d_force, const float *d_torque,
                                   int numberParticles, float temperature, float prefactor, float *d_linearVelocity,
                                   float *d_angularVelocity, void *stream);
int uammd_fcm_euler_maruyama_dir(float *d_pos, float *d_dir, const int *d_index, const float *d_linearVelocity,
                                 const float *d_angularVelocity, int numberParticles, float dt, void *stream);

/* ------------------------------------------------------------------------------------------------
 * BDHI::Cholesky — dense open-boundary RPY mobility with an explicit Cholesky factor (the other half of
 * test/BDHI/Lanczos_Cholesky).  Replaces Integrator/BDHI/BDHI_Cholesky.cu:
 *   setup_step (fillMobilityRPYD :34-80, :158-178)   computeMF (symv, :196-233)   computeBdW (potrf + trmv, :235-262)
 * d_index: group index iterator (nullable = all particles); d_radius nullable when hydrodynamicRadius > 0.
 * rocSOLVER / rocBLAS are loaded on first use (not load-time dependencies of this library).
 * ---------------------------------------------------------------------------------------------- */
typedef struct uammd_bdhi_cholesky uammd_bdhi_cholesky;
int uammd_bdhi_cholesky_create(int numberParticles, float viscosity, float hydrodynamicRadius, uammd_bdhi_cholesky **out);
int uammd_bdhi_cholesky_destroy(uammd_bdhi_cholesky *h);
int uammd_bdhi_cholesky_setup_step(uammd_bdhi_cholesky *h, const float *d_pos, const int *d_index, const float *d_radius, void *stream);
/* d_force real4[N]; d_MF real3[N] (overwritten).  Rebuilds M first if computeBdW consumed it, like the reference. */
int uammd_bdhi_cholesky_mf(uammd_bdhi_cholesky *h, const float *d_pos, const float *d_force, const int *d_index, const float *d_radius,
                           float *d_MF, void *stream);
/* d_BdW real3[N]: N(0,1) draws on entry (the reference's cuRAND stream is unpinned third party), B dW on exit. */
int uammd_bdhi_cholesky_bdw(uammd_bdhi_cholesky *h, const float *d_pos, const int *d_index, const float *d_radius, float *d_BdW,
                            void *stream);

/* ------------------------------------------------------------------------------------------------
 * BDHI::True2D / BDHI::Quasi2D — hydrodynamics of particles confined to a plane.  Replaces
 *   BDHI2D<HydroKernel>::BDHI2D / forwardTime      Integrator/Hydro/BDHI_quasi2D.cuh:155-257, .cu:18-205
 *   spreadThermalDrift / spreadParticleForces / forceFourier2Vel / fourierBrownianNoise / interpolateVelocities /
 *   updateParticlePositions                         .cu:234-541
 * cells[0] <= 0: grid from h = 0.8 a (the tolerance parameter is ignored by the reference, .cu:143-145).
 * ---------------------------------------------------------------------------------------------- */
#define UAMMD_BDHI2D_TRUE2D 0  /* BDHI2D_ns::True2D:  f_k = 0, g_k = 1/k^4                       BDHI_quasi2D.cuh:83-93 */
#define UAMMD_BDHI2D_QUASI2D 1 /* BDHI2D_ns::Quasi2D: 3D Stokes integrated over z, thermal drift  BDHI_quasi2D.cuh:95-110 */
typedef struct uammd_bdhi2d uammd_bdhi2d;
typedef struct {
  float boxSize[2];
  float hydrodynamicRadius, viscosity, temperature, dt;
  int cells[2];
  unsigned int seed; /* the reference draws it from System::rng().next32() */
  int kernel;        /* UAMMD_BDHI2D_* */
} uammd_bdhi2d_parameters;
/* Errors (-2) carry the reference's messages ("Invalid box", "Invalid hydrodynamic radius"). */
int uammd_bdhi2d_create(const uammd_bdhi2d_parameters *par, uammd_bdhi2d **out, int cells[2], int *support);
int uammd_bdhi2d_destroy(uammd_bdhi2d *h);
/* One step's particle velocities (through interpolateVelocities): d_pos real4[N]; d_force real4[N] or NULL when no
 * Interactor is attached (then only the thermal terms act); d_vel real2[N], overwritten. */
int uammd_bdhi2d_velocities(uammd_bdhi2d *h, const float *d_pos, const float *d_force, int numberParticles, float *d_vel,
                            void *stream);
/* pos += make_real4(vel * dt) */
int uammd_bdhi2d_update_positions(float *d_pos, const float *d_vel, int numberParticles, float dt, void *stream);
int uammd_bdhi2d_get_counter(uammd_bdhi2d *h, unsigned int *counter); /* number of stochastic steps taken (Saru seed) */

/* ------------------------------------------------------------------------------------------------
 * BDHI::FIB — Fluctuating Immersed Boundary on a staggered grid.  Replaces
 *   FIB::FIB / forwardTime / forwardMidpoint     Integrator/BDHI/FIB/FIB.cuh:131-236, FIB.cu:87-135, :965-1000, :1058-1079
 *   addRandomAdvection, spreadParticleForces, solveStokesFourier, midPointStep     FIB.cu:274-391, :528-597, :667-724, :726-823
 * As in the reference both Scheme values run the simple midpoint scheme (FIB.cu:1072-1079) and the RFD thermal drift
 * is disabled (:400).  Exactly one of hydrodynamicRadius > 0 / cells[0] > 0 must be given (the other negative).
 * ---------------------------------------------------------------------------------------------- */
#define UAMMD_FIB_MIDPOINT 0
#define UAMMD_FIB_IMPROVED_MIDPOINT 1
typedef struct uammd_fib uammd_fib;
typedef struct {
  float boxSize[3];
  float temperature, viscosity, hydrodynamicRadius, dt;
  int cells[3];
  int scheme;        /* UAMMD_FIB_*; both run forwardMidpoint, as in the reference */
  unsigned int seed; /* the reference seeds cuRAND from System::rng(); here Saru(cell, seed, step) */
} uammd_fib_parameters;
int uammd_fib_create(const uammd_fib_parameters *par, uammd_fib **out, int cells[3], float *hydrodynamicRadius);
int uammd_fib_destroy(uammd_fib *h);
/* One step after the Interactors have filled d_force (real4[N]; NULL = no forces): d_pos real4[N] is advanced in place. */
int uammd_fib_forward(uammd_fib *h, float *d_pos, const float *d_force, int numberParticles, void *stream);
/* Test hook: take the 6*ncells fluid random numbers (slot-major: XX, YY, ZZ, XY, XZ, YZ) from this device array. */
int uammd_fib_set_noise(uammd_fib *h, const float *d_random);
float uammd_fib_self_mobility(float hydrodynamicRadius, float viscosity, float L); /* FIB::getSelfMobility, FIB.cuh:152-163 */

/* ------------------------------------------------------------------------------------------------
 * Hydro::ICM — Inertial Coupling Method, zero excess mass.  Replaces
 *   ICM::ICM / forwardTime                         Integrator/Hydro/ICM.cuh:123-231, ICM.cu:825-888, :1191-1224
 *   midPointStep, updateCellVelocityUnperturbed, spreadParticleForces, addThermalDrift, solveStokesFourier
 *                                                  ICM.cu:413-500, :793-822, :86-159, :161-275, :349-411
 * The fluid velocity persists in the handle between steps.  Exactly one of hydrodynamicRadius > 0 / cells[0] > 0.
 * ---------------------------------------------------------------------------------------------- */
typedef struct uammd_icm uammd_icm;
typedef struct {
  float boxSize[3];
  float temperature, viscosity, density, hydrodynamicRadius, dt;
  int cells[3];
  int sumThermalDrift;     /* default 0 in the reference */
  int removeTotalMomentum; /* default 1 in the reference */
  unsigned int seed;
} uammd_icm_parameters;
int uammd_icm_create(const uammd_icm_parameters *par, uammd_icm **out, int cells[3], float *hydrodynamicRadius);
int uammd_icm_destroy(uammd_icm *h);
/* One step = predictor (d_pos real4[N]: q^n -> q^{n+1/2}), the caller's Interactors at q^{n+1/2}, then the fluid update and the
 * corrector (d_pos -> q^{n+1}); d_force real4[N] or NULL when no Interactor is attached. */
int uammd_icm_predictor(uammd_icm *h, float *d_pos, int numberParticles, void *stream);
int uammd_icm_fluid_and_corrector(uammd_icm *h, float *d_pos, const float *d_force, int numberParticles, void *stream);
/* real3[nz][ny][nx]: the face-centred field, or (collocated = 1) ICM::getFluidVelocities (ICM.cuh:96-119, :176-199) */
int uammd_icm_get_fluid_velocity(uammd_icm *h, float *d_out, int collocated, void *stream);
int uammd_icm_set_fluid_velocity(uammd_icm *h, const float *d_in, void *stream);
int uammd_icm_set_noise(uammd_icm *h, const float *d_random); /* test hook: 6*ncells fluid random numbers, slot-major */

/* ------------------------------------------------------------------------------------------------
 * Triply periodic electrostatics (another consumer of the spread / FFT / gather engine).  Replaces
 *   Poisson::Poisson / sum / computeFieldPotentialAtParticles    Interactor/SpectralEwaldPoisson.cuh:83-136, .cu:71-160
 *   farField (spread q, R2C, chargeFourier2FieldAndPotential, 4 x C2R, gather + UnZip2Real4)   .cu:332-360, :410-559
 *   nearFieldForce / nearFieldEnergy / nearFieldFieldPotential over a CellList                 .cu:222-329, :362-408
 * split <= 0 disables the Ewald splitting (far field only).  As in the reference, Parameters::cells and ::support are
 * not consulted (the grid comes from upsampling or the tolerance heuristic).
 * ---------------------------------------------------------------------------------------------- */
typedef struct uammd_poisson uammd_poisson;
typedef struct {
  float boxSize[3];
  float epsilon;    /* permittivity */
  float tolerance;
  float gw;         /* Gaussian source width */
  float split;      /* Ewald splitting parameter, <= 0: no splitting */
  float upsampling; /* > 0: h = 1/upsampling */
} uammd_poisson_parameters;
typedef struct {
  int cells[3];
  int support;
  float nearFieldCutOff;
  int nTable;
  float h;
} uammd_poisson_info;
/* Errors (-2) carry the reference's messages: kernel support too large (.cu:95-102), near field cut off too large (:111-116). */
int uammd_poisson_create(const uammd_poisson_parameters *par, uammd_poisson **out, uammd_poisson_info *info);
int uammd_poisson_destroy(uammd_poisson *h);
/* "atomic_spread" = 1: spread the charges with global atomics instead of the tile-owned LDS kernel (test hook) */
int uammd_poisson_set_option(uammd_poisson *h, const char *name, int value);
/* Poisson::sum.  d_pos real4[N], d_charge real[N] (pd->getCharge).  The far field ADDS q E to d_force (real4[N]) and q phi
 * to d_energy (real[N]) whenever the pointer is non-null — the reference's interpolateFields adds both regardless of the
 * Computables (.cu:561-579), so a faithful caller passes both; nearForce / nearEnergy select the near-field passes
 * (comp.force / comp.energy). */
int uammd_poisson_sum(uammd_poisson *h, const float *d_pos, const float *d_charge, int numberParticles, float *d_force,
                      float *d_energy, int nearForce, int nearEnergy, void *stream);
/* Poisson::computeFieldPotentialAtParticles: d_fieldPotential real4[N] (Ex, Ey, Ez, phi) is ADDED to (the reference
 * zero-fills it first).  d_force / d_energy (nullable) receive the far-field side effect the reference's call has. */
int uammd_poisson_field_potential(uammd_poisson *h, const float *d_pos, const float *d_charge, int numberParticles,
                                  float *d_fieldPotential, float *d_force, float *d_energy, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Path B on several GPUs — z-slab decomposition of the FCM grid (the reference is single GPU, so these
 * have no reference counterpart: they are the per-rank compute stages of uammd_fcm_displacements, cut where the
 * exchanges happen.  The exchanges themselves are RCCL calls of the host layer, uammd_amd/parallel_fcm.py).
 * Rank layouts (float2 = one complex):
 *   real window  float  [nzLocal + 2*halo][3][ny][2*(nx/2+1)]   owned planes are [halo, halo + nzLocal)
 *   xy spectrum  float2 [nzLocal][3][ny][nx/2+1]                 = the owned planes after the in-place 2-D R2C
 *   z buffer     float2 [nz][3][nyLocal][nx/2+1]                 rows y0 .. y0 + nyLocal of the Fourier grid
 * Sequence per step: slab_spread -> (halo planes ADDED into the neighbours) -> slab_forward_xy -> (all-to-all) ->
 * slab_fft_z(0) -> slab_kspace -> slab_fft_z(1) -> (all-to-all) -> slab_inverse_xy -> (halo planes COPIED from the
 * neighbours) -> slab_gather.  Positions are in the window frame: z relative to the centre of the owned slab.
 * ---------------------------------------------------------------------------------------------- */
typedef struct uammd_fcm_slab uammd_fcm_slab;
int uammd_fcm_slab_create(const uammd_fcm_parameters *par, int nzLocal, int z0, int halo, int nyLocal, int y0,
                          uammd_fcm_slab **out);
int uammd_fcm_slab_destroy(uammd_fcm_slab *h);
int uammd_fcm_slab_set_option(uammd_fcm_slab *h, const char *name, int value);
int uammd_fcm_slab_spread(uammd_fcm_slab *h, const float *d_posLocal, const float *d_force, int numberParticles,
                          float *d_grid, void *stream);
int uammd_fcm_slab_gather(uammd_fcm_slab *h, const float *d_posLocal, int numberParticles, const float *d_grid,
                          float *d_linearVelocity, void *stream);
int uammd_fcm_slab_forward_xy(uammd_fcm_slab *h, float *d_grid, void *stream);
/* device side of the all-to-all transposes (uammd_comm_alltoall moves block p of the send buffer to rank p): pack the owned planes'
 * spectrum [zl][c][y][kx] into [p][zl][c][yl][kx]; what arrives is the z buffer as it stands; on the way back unpack
 * [src][zl][c][yl][kx] into the spectrum of the window.  Buffers: 3 nzLocal ny (nx/2+1) float2 each. */
int uammd_fcm_slab_transpose_pack(uammd_fcm_slab *h, const float *d_grid, float *d_send, void *stream);
int uammd_fcm_slab_transpose_unpack(uammd_fcm_slab *h, const float *d_recv, float *d_grid, void *stream);
/* the same with the halo fold on the way: d_fromDown / d_fromUp (`planes` planes each, laid out like the window) are added to the first /
 * last `planes` owned planes as the x pass loads them (one launch less than adding them first).  Returns 1 when the solver's own FFT
 * does not serve this grid: add the planes and call uammd_fcm_slab_forward_xy */
int uammd_fcm_slab_forward_xy_fold(uammd_fcm_slab *h, float *d_grid, const float *d_fromDown, const float *d_fromUp, int planes, void *stream); /* in place on the owned planes */
int uammd_fcm_slab_inverse_xy(uammd_fcm_slab *h, float *d_grid, void *stream);
/* the inverse writing the owned planes of a float4 window d_inter[z][y][x] = (vx, vy, vz, 0) (returns 1 and does nothing when the grid
 * does not take the library's own FFT), and the gather that reads that window once the caller has exchanged its halo planes */
int uammd_fcm_slab_inverse_xy_inter(uammd_fcm_slab *h, float *d_grid, float *d_inter, void *stream);
/* world size 1 (the rank is its own neighbour through the periodic z faces): the same, and the first / last `wrapPlanes` owned planes are
 * also stored into the halo planes above / below the owned block — the window is ready for the gather without a halo exchange */
int uammd_fcm_slab_inverse_xy_inter_wrap(uammd_fcm_slab *h, float *d_grid, float *d_inter, int wrapPlanes, void *stream);
int uammd_fcm_slab_gather_inter(uammd_fcm_slab *h, const float *d_posLocal, int N, const float *d_inter, float *d_vel, void *stream);
int uammd_fcm_slab_fft_z(uammd_fcm_slab *h, float *d_cplxZ, int inverse, void *stream);
/* the three calls above (forward z, operator, inverse z) in one pass over d_cplxZ when nz is a power of two; returns 1 (and does
 * nothing) when the grid does not allow it */
int uammd_fcm_slab_z_fused(uammd_fcm_slab *h, float *d_cplxZ, int haveForce, float temperature, float prefactor, unsigned int seed2,
                           void *stream);
int uammd_fcm_slab_kspace(uammd_fcm_slab *h, float *d_cplxZ, int haveForce, float temperature, float prefactor,
                          unsigned int seed2, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Path B, spectral Ewald variant — Positively Split Ewald RPY (BDHI::PSE).  Replaces
 *   pse_ns::NearField  ctor / Mdot / computeStochasticDisplacements / setShearStrain
 *                                              Integrator/BDHI/PSE/NearField.cuh:29-99, :239-285
 *   RPYPSE_near::FandG + TabulatedFunction     Integrator/BDHI/PSE/RPY_PSE.cuh:45-128, misc/TabulatedFunction.cuh:63-157
 *   pse_ns::FarField   ctor / computeHydrodynamicDisplacements / setShearStrain
 *                                              Integrator/BDHI/PSE/FarField.cuh:25-41, :85-308, :318-342, :569-654
 * Seeds: the reference draws `seed` from System::rng() in each constructor (near first, then far) and a fresh `seed2`
 * from the same generator at every stochastic call; they are explicit here (the C++/Python layers draw them).
 * The far field is a uammd_fcm handle in PSE mode (same spread / rocFFT / gather machinery, other greens function).
 * ---------------------------------------------------------------------------------------------- */
typedef struct uammd_pse_near uammd_pse_near;
/* errors with the reference's text "[BDHI::PSE] Cut off is too large, try increasing psi" when rcut > L/2 */
int uammd_pse_near_create(const float boxSize[3], float viscosity, float hydrodynamicRadius, float tolerance, float psi,
                          float shearStrain, unsigned int seed, uammd_pse_near **out, float *rcut_out,
                          int *nPointsTable_out);
int uammd_pse_near_destroy(uammd_pse_near *h);
int uammd_pse_near_set_shear_strain(uammd_pse_near *h, float shearStrain);
/* "exact_order" (default 0): 1 = thread-per-particle product in the reference's summation order (NearField.cuh:154-185 over
 * NeighbourList/common.cuh:10-34), bit for bit; 0 = eight lanes per particle, same pairs and per-pair arithmetic, another order.
 * "lazy_list" (default 0): 1 = the near-field cell list is rebuilt only after uammd_pse_near_positions_changed() (the host layers wire
 * it to ParticleData's position write signal) or when the position array / N of a call changes — CellList::update's needsRebuild
 * (NeighbourList/CellList.cuh:134-136,192-204); 0 = every mdot / stochastic / dot call rebuilds.
 * "pair_list" (default 1; takes effect with "lazy_list"): the pairs inside the cut-off are evaluated once per list build into 24-byte
 * records (F, (G - F) / r^2, r_ij, j) and the ~8 products of a step stream them; 0 = every product scans the 27 cells. */
int uammd_pse_near_set_option(uammd_pse_near *h, const char *name, int value);
int uammd_pse_near_positions_changed(uammd_pse_near *h);
/* The list update of the calls below (cl->update, NearField.cuh:231-237) ahead of them, nothing waited for; with "lazy_list" and
 * "pair_list" also the launch of the pair records' build.  Optional: a caller that queues other work between this and the first
 * product (BDHI::PSE::computeMF's far field, BDHI_PSE.cuh:92-120) hides the build's one host read behind it. */
int uammd_pse_near_prepare(uammd_pse_near *h, const float *d_pos, int numberParticles, void *stream);
/* diagnostics: pair records in use (0 while the products scan the cells) and allocated */
int uammd_pse_near_pair_records(uammd_pse_near *h, long long *records, long long *capacity);
/* Option "list_skin_percent" (40; with "lazy_list" and "pair_list", no shear): the particle order of a list build and every particle's
 * candidates within rc + skin (skin = percent / 100 x rc) are KEPT over the following steps; a step then refreshes the sorted positions
 * and makes its pair records from the candidates alone — exact while nobody has moved more than skin / 2 since the build, which is
 * measured on the device every step and read with the records' counters (a broken bound repeats the build from scratch, and the solve
 * that streamed the records).  Same pairs as a build per step, summed in another order.  0 = a list build per step.
 * uammd_pse_near_list_stats: {builds from scratch, record builds from a kept list, repeated builds, 1 while the mechanism is on}. */
int uammd_pse_near_list_stats(uammd_pse_near *h, long long out[4]);
/* One-shot: the NEXT uammd_pse_near_stochastic on the handle also performs uammd_pse_near_mdot(h, pos, d_force, N, d_MF) with its own
 * positions — BDHI_PSE.cuh:92-126 calls the two back to back on the same list.  Where the solve streams pair records its first product
 * applies them to F as a second right-hand side (the records are read once for both vectors); otherwise (T = 0, scanning products) the
 * plain product runs before the call returns.  d_force NULL disarms. */
int uammd_pse_near_set_mdot_rider(uammd_pse_near *h, const float *d_force, float *d_MF);
/* d_MF real3[N] += M_near F (d_force real4[N]; NULL = nothing to do) */
int uammd_pse_near_mdot(uammd_pse_near *h, const float *d_pos, const float *d_force, int numberParticles, float *d_MF,
                        void *stream);
/* d_BdW real3[N] = prefactor sqrt(2 T) M_near^(1/2) dW by Lanczos (OVERWRITES d_BdW, as the reference); no-op at T == 0 */
int uammd_pse_near_stochastic(uammd_pse_near *h, const float *d_pos, int numberParticles, float temperature,
                              float prefactor, unsigned int seed2, float *d_BdW, void *stream, int *iterations);
/* test hooks: the Saru noise vector, and the raw product d_Mv3 = M_near d_v3 (what the Lanczos callback computes) */
int uammd_pse_near_noise(uammd_pse_near *h, int numberParticles, float variance, unsigned int seed2, float *d_out3,
                         void *stream);
int uammd_pse_near_dot(uammd_pse_near *h, const float *d_pos, const float *d_v3, int numberParticles, float *d_Mv3,
                       void *stream);
/* FarField::initializeGrid: grid cells BEFORE nextFFTWiseSize3D (utils/Grid.cuh:142-213, host logic of the caller) */
int uammd_pse_far_raw_cells(const float boxSize[3], float psi, float tolerance, int cells_out[3]);
int uammd_pse_far_create(const float boxSize[3], const int cells[3], float viscosity, float hydrodynamicRadius,
                         float tolerance, float psi, float shearStrain, unsigned int seed, uammd_fcm **out,
                         int *support_out, float *eta_out);
int uammd_pse_far_set_shear_strain(uammd_fcm *h, float shearStrain);
/* d_MF real3[N] += M_far F + prefactor sqrt(2T) M_far^(1/2) dW   (IBM::gather adds; d_force NULL = noise only) */
int uammd_pse_far_displacements(uammd_fcm *h, const float *d_pos, const float *d_force, int numberParticles,
                                float temperature, float prefactor, unsigned int seed2, float *d_MF, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Lanczos sqrt(M) v.  Replaces lanczos::Solver (misc/LanczosAlgorithm.cuh:32-83,
 * misc/LanczosAlgorithm/LanczosAlgorithm.cu:27-262) and the MatrixDot functor (LanczosAlgorithm/MatrixDot.h:7-25):
 * the matrix is a host callback that ENQUEUES d_Mv = M d_v (n floats, device pointers) on `stream` and returns 0.
 * uammd_lanczos_run returns 0 and the iteration count, or an error whose message is the reference's exception text
 * ("[Lanczos] Could not converge", "[Lanczos] Unknown error (found NaN in result guess) at iteration i").
 * The adaptive convergence-check schedule (starts at 3) is kept per handle, as per Solver instance.
 * ---------------------------------------------------------------------------------------------- */
typedef struct uammd_lanczos uammd_lanczos;
typedef int (*uammd_matvec_fn)(void *ctx, const float *d_v, float *d_Mv, int n, void *stream);
int uammd_lanczos_create(uammd_lanczos **out);
int uammd_lanczos_destroy(uammd_lanczos *h);
int uammd_lanczos_run(uammd_lanczos *h, uammd_matvec_fn dot, void *ctx, float *d_Bv, const float *d_v,
                      float tolerance, int n, void *stream, int *iterations);
/* Solver::runIterations (LanczosAlgorithm.cuh:49-67, LanczosAlgorithm.cu:183-200): exactly numberIterations Lanczos steps without a
 * convergence test; d_Bv = the estimate after the last step, *residual = |Bz_m - Bz_(m-1)| / |Bz_(m-1)| between the last two estimates. */
int uammd_lanczos_run_iterations(uammd_lanczos *h, uammd_matvec_fn dot, void *ctx, float *d_Bv, const float *d_v, int numberIterations, int n,
                                 void *stream, float *residual);
int uammd_lanczos_set_iteration_hard_limit(uammd_lanczos *h, int limit);
/* "defer_checks" (default 1): the reference checks convergence at every iteration from an adaptive first step on
 * (LanczosAlgorithm.cu:218-232); here the checks of the iterations before the one the PREVIOUS run stopped at are evaluated together
 * at that iteration — one host round trip and one pass over the Krylov basis instead of one per iteration; the run still stops at
 * the first iteration whose error passes, with that iteration's estimate (the reference's result and iteration count).  The checks are
 * evaluated in the reference's order: a tridiagonal block that cannot be diagonalised fails the run (-20) only if no EARLIER check
 * ends it.  Side effect to know about: a run that would have stopped at iteration k has by then called `dot` for the iterations up to
 * the previous run's stopping point (at most 7 more calls than the reference makes) — a callback that counts its calls sees them.
 * 0 = every check as its iteration completes, the reference's call count exactly. */
int uammd_lanczos_set_option(uammd_lanczos *h, const char *name, int value);
/* Vectors sharded over several ranks (Lanczos row): `n` is then the LOCAL length, the matvec callback computes the
 * local rows of M v, and every dot product / norm of the recurrence is completed by `reduce`, which must sum d_values[0..count)
 * in place over all ranks on `stream` (RCCL all-reduce of 1 float, 3 per iteration + 2 per convergence check).  ownsFirstElement:
 * non-zero on the rank that holds global element 0.  reduce == NULL restores the single-rank behaviour. */
typedef int (*uammd_allreduce_fn)(void *ctx, float *d_values, int count, void *stream);
int uammd_lanczos_set_allreduce(uammd_lanczos *h, uammd_allreduce_fn reduce, void *ctx, int ownsFirstElement);
/* {check_convergence_steps, lastRunRequiredSteps}: the adaptive schedule a run leaves behind (LanczosAlgorithm.cu:175, :245-251); a caller
 * that repeats a run restores it first */
int uammd_lanczos_get_schedule(uammd_lanczos *h, int state[2]);
int uammd_lanczos_set_schedule(uammd_lanczos *h, const int state[2]);
/* fn(ctx, stream) is called ONCE during the next uammd_lanczos_run: after the host has answered the run's first convergence check and
 * before it waits for the outcome (or when the run ends, if it never waited).  What fn queues on the stream runs behind the check's
 * kernels while the host is busy with the check: the one wait of a run is no longer a drained stream.  One-shot. */
typedef int (*uammd_interleave_fn)(void *ctx, void *stream);
int uammd_lanczos_set_interleave(uammd_lanczos *h, uammd_interleave_fn fn, void *ctx);
/* one stage earlier: called after the first check's scalars are on their way to the host and BEFORE the kernels that wait for the host's
 * answer; what fn queues runs while the host solves the check's small eigenproblems.  One-shot; called before the callback above. */
int uammd_lanczos_set_interleave_early(uammd_lanczos *h, uammd_interleave_fn fn, void *ctx);
/* the same for the solve inside the next uammd_pse_near_stochastic (BDHI::PSE queues its far field there); fn must not call into the
 * near-field handle.  One-shot. */
int uammd_pse_near_set_interleave(uammd_pse_near *h, uammd_interleave_fn fn, void *ctx);
int uammd_pse_near_set_interleave_early(uammd_pse_near *h, uammd_interleave_fn fn, void *ctx);
/* uammd_pse_far_displacements queued in two halves on the same stream with the same arguments (1: binning, spreading, forward x / y
 * transforms; 2: z transforms + operator + noise, inverse transforms, gather) — for the two callbacks above */
int uammd_pse_far_displacements_half(uammd_fcm *h, const float *d_pos, const float *d_force, int numberParticles, float temperature,
                                     float prefactor, unsigned int seed2, float *d_MF, int half, void *stream);
int uammd_lanczos_get_last_run_required_steps(uammd_lanczos *h, int *steps);

/* BDHI::Lanczos (open boundaries, dense RPY mobility, matrix free).  Replaces
 *   Lanczos_ns::NbodyMatrixFreeMobilityDot + NBody::transverse   Integrator/BDHI/BDHI_Lanczos.cu:56-118, Interactor/NBodyBase.cuh:46-159
 *   RotnePragerYamakawa                                           Integrator/BDHI/BDHI.cuh:27-96
 *   Lanczos::computeMF / computeBdW                               BDHI_Lanczos.cu:120-188
 * d_Mv real3[N] is OVERWRITTEN.  The reference draws the noise of computeBdW with cuRAND (parity unpinned):
 * the caller supplies 3N standard normal numbers. */
int uammd_rpy_nbody_mdot(const float *d_pos, const float *d_v, int vstride, const float *d_radius, float hydrodynamicRadius,
                         float viscosity, int numberParticles, float *d_Mv, void *stream);
int uammd_rpy_lanczos_bdw(uammd_lanczos *solver, const float *d_pos, const float *d_radius, float hydrodynamicRadius,
                          float viscosity, int numberParticles, const float *d_noise, float tolerance, float *d_BdW,
                          void *stream, int *iterations);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU (new design: the reference is single GPU).  One process per GPU, RCCL over xGMI, loaded on first use.
 * Set-up: rank 0 calls uammd_comm_unique_id and hands the 128 bytes to the other processes by any means (MPI, a file, a TCP
 * store: outside this library); every process then calls uammd_comm_init on ITS device.  All calls are asynchronous on `stream`
 * except uammd_comm_exchange_counts and uammd_comm_exchange_counts_device (the latter takes the two sizes from device memory and
 * returns {toUp, toDown, fromDown, fromUp} with one synchronisation).  The ranks form a periodic ring along z: "up" = rank + 1, "down" = rank - 1.
 *   path A  uammd_halo_pack -> uammd_comm_halo_exchange (4 floats per row), receive straight into the tail of the position array
 *   path B  uammd_fcm_slab_* + uammd_comm_halo_exchange (grid planes) + uammd_comm_alltoall (FFT transposes)
 *   Lanczos uammd_lanczos_set_allreduce with a callback that calls uammd_comm_allreduce_sum
 * ---------------------------------------------------------------------------------------------- */
typedef struct uammd_comm uammd_comm;
int uammd_comm_unique_id(char id[128]);
/* the version of the librccl this library loaded (ncclGetVersion's integer: major * 10000 + minor * 100 + patch) */
int uammd_comm_rccl_version(int *version);
int uammd_comm_init(uammd_comm **out, int rank, int world, const char id[128]);
int uammd_comm_destroy(uammd_comm *h);
int uammd_comm_rank(const uammd_comm *h);
int uammd_comm_world(const uammd_comm *h);
int uammd_comm_halo_exchange(uammd_comm *h, const float *d_sendUp, int nUp, const float *d_sendDown, int nDown, float *d_recvFromDown,
                             int nFromDown, float *d_recvFromUp, int nFromUp, int floatsPerRow, void *stream);
int uammd_comm_exchange_counts(uammd_comm *h, const int toUpDown[2], int fromDownUp[2], void *stream);
int uammd_comm_exchange_counts_device(uammd_comm *h, const int *d_toUpDown, int all4[4], void *stream);
/* The membership refresh of a slab step in ONE call — the sequence above as uammd_amd/parallel.py (DistributedLJ) and
 * include/uammd/Distributed.h run it: displacement since the last refresh (when refRows == n; d_ref / d_maxDisplacement nullable),
 * leavers selected, the sizes to and from the two neighbours (host read 1), migration rows packed, exchanged (8 floats per row) and
 * dropped into the holes, forces of the owned rows zeroed, halo members selected with `reach` = rc + 3 skin, their sizes (host read 2),
 * positions packed with the frame shift and received straight into rows [n, n + ghosts) of d_pos, d_ref <- d_pos.
 * d_idx: int[4][capRows] (leavers up / down, halo up / down: rows 2 and 3 are the cached membership lists of the steps until the next
 * refresh); d_rows, d_arrivals: float[capRows][8]; d_send: float[capRows][4]; d_counts: int[4]; d_holes: int[capRows];
 * d_selectWorkspace: uammd_slab_select_workspace(capRows) bytes.  comm == NULL: a world of one in process.
 * d_listedMask (nullable, capRows bytes): 1 for the owned rows that are in a halo list, 0 for the others.
 * out = {owned rows, owned + ghosts, halo up, halo down, ghosts from below, from above, left up, left down, arrived from below, above}. */
int uammd_slab_refresh_lj(uammd_comm *comm, float *d_pos, float *d_vel, int *d_ids, float *d_force, int n, int capRows, float width, float reach,
                          int *d_idx, int *d_holes, int *d_counts, void *d_selectWorkspace, float *d_rows, float *d_arrivals, float *d_send,
                          float *d_ref, int refRows, float *d_maxDisplacement, unsigned char *d_listedMask, int out[10], void *stream);
int uammd_comm_alltoall(uammd_comm *h, const void *d_send, void *d_recv, size_t bytesPerPeer, void *stream);
int uammd_comm_allreduce_sum(uammd_comm *h, float *d_buf, int n, void *stream);

/* ------------------------------------------------------------------------------------------------
 * DOUBLE_PRECISION build of path B (global/defines.h:9-11,33-44 makes `real` a build switch; every accuracy assertion the
 * reference ships is compiled with it: test/CMakeLists.txt:9, test/BDHI/FCM/Makefile:9).  The `_f64` entry points below are
 * that build of the layout-generic kernels: IBM spread / gather, FCM_impl (deterministic part), PSE near / far field and
 * lanczos::Solver with real = double, so that the reference's own known answers run on the GPU at their own tolerances:
 *   test/BDHI/FCM/fcm_test.cu:85-144 (Hasimoto, 1e-8)        test/misc/ibm/test_ibm_regular.cu:113-136,240-274 (1e-10)
 *   test/misc/lanczos/test_lanczos.cu:236-269 (1e-7)          test/BDHI/PSE/pse_test.cu:64-117
 * Positions / forces are real4 = double[4], velocities real3 = double[3].  The tuned hot path (tile-owned MFMA spreading,
 * the in-LDS FFT, the pair prefilter) is single precision by construction: UAMMD's default `real`, and what bench.py measures.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int kind;        /* UAMMD_IBM_KERNEL_GAUSSIAN, _PESKIN3, _PESKIN4 or _CONSTANT */
  int support[3];
  double prefactor, tau, rmax;
  double invh[3];
} uammd_ibm_kernel_f64;
typedef struct {
  double boxSize[3];
  int cells[3];
  double viscosity;
  uammd_ibm_kernel_f64 kernel;
} uammd_fcm_parameters_f64;
typedef struct uammd_fcm_f64 uammd_fcm_f64;
typedef struct uammd_pse_near_f64 uammd_pse_near_f64;
typedef struct uammd_lanczos_f64 uammd_lanczos_f64;
int uammd_fcm_gaussian_kernel_f64(double h, double tolerance, uammd_ibm_kernel_f64 *out, double *a_eff);
double uammd_fcm_advise_grid_size_f64(double hydrodynamicRadius, double tolerance);
int uammd_ibm_spread_f64(const double *d_pos, int posStride, const double *d_quantity, int ncomp, int numberParticles,
                         const double boxSize[3], const int periodic[3], const int cellDim[3], int nxStride,
                         const uammd_ibm_kernel_f64 *kernel, double *d_grid, void *stream);
int uammd_ibm_gather_f64(const double *d_pos, int posStride, double *d_out, int ncomp, int numberParticles, const double boxSize[3],
                         const int periodic[3], const int cellDim[3], int nxStride, const uammd_ibm_kernel_f64 *kernel,
                         const double *d_grid, void *stream);
int uammd_fcm_create_f64(const uammd_fcm_parameters_f64 *par, uammd_fcm_f64 **out);
int uammd_fcm_destroy_f64(uammd_fcm_f64 *h);
/* d_velocity real3[N] = M F (overwritten; a PSE far-field handle ADDS, as FarField.cuh:563-566) */
int uammd_fcm_displacements_f64(uammd_fcm_f64 *h, const double *d_pos, const double *d_force, int numberParticles,
                                double *d_velocity, void *stream);
/* the same with the thermal term: d_velocity = M F + prefactor sqrt(2 T M) dW, Fourier noise keyed Saru(node, seed1, seed2)
 * (FCM_impl.cuh:437-542, FarField.cuh:235-308,471-501); d_force may be NULL (noise only) */
int uammd_fcm_displacements_thermal_f64(uammd_fcm_f64 *h, const double *d_pos, const double *d_force, int numberParticles, double temperature,
                                        double prefactor, unsigned int seed1, unsigned int seed2, double *d_velocity, void *stream);
/* BDHI::EulerMaruyama_ns::integrateGPUD with real = double (see uammd_bdhi_euler_maruyama) */
int uammd_bdhi_euler_maruyama_f64(double *d_pos, const int *d_index, const double *d_MF, const double *d_BdW, const double K[9], int numberParticles,
                                  double sqrt2Tdt, double dt, int is2D, void *stream);
/* BDHI::True2D / BDHI::Quasi2D with real = double (see uammd_bdhi2d_create: Integrator/Hydro/BDHI_quasi2D.cuh:155-257, .cu:61-88, :179-541;
 * the reference's test/BDHI/quasi2D/quasi2d_test.cu is compiled with -DDOUBLE_PRECISION).  d_pos / d_force real4[N] of doubles, d_vel real2[N]. */
typedef struct uammd_bdhi2d_f64 uammd_bdhi2d_f64;
typedef struct {
  double boxSize[2];
  double hydrodynamicRadius, viscosity, temperature, dt;
  int cells[2];
  unsigned int seed;
  int kernel; /* UAMMD_BDHI2D_* */
} uammd_bdhi2d_parameters_f64;
int uammd_bdhi2d_create_f64(const uammd_bdhi2d_parameters_f64 *par, uammd_bdhi2d_f64 **out, int cells[2], int *support);
int uammd_bdhi2d_destroy_f64(uammd_bdhi2d_f64 *h);
int uammd_bdhi2d_velocities_f64(uammd_bdhi2d_f64 *h, const double *d_pos, const double *d_force, int numberParticles, double *d_vel, void *stream);
int uammd_bdhi2d_update_positions_f64(double *d_pos, const double *d_vel, int numberParticles, double dt, void *stream);
/* Poisson with real = double (see uammd_poisson_create: Interactor/SpectralEwaldPoisson.cuh:83-136; the reference's
 * test/Potentials/Poisson/TriplyPeriodic tests are compiled with -DDOUBLE_PRECISION, the quadrupole at tolerance 1e-14 / support 41).  Near
 * field over all pairs with the minimum image.  d_pos / d_force / d_fieldPotential real4[N] of doubles, d_charge / d_energy real[N]. */
typedef struct uammd_poisson_f64 uammd_poisson_f64;
typedef struct {
  double boxSize[3];
  double epsilon, tolerance, gw, split, upsampling;
} uammd_poisson_parameters_f64;
typedef struct {
  int cells[3];
  int support;
  double nearFieldCutOff;
  int nTable;
  double h;
} uammd_poisson_info_f64;
int uammd_poisson_create_f64(const uammd_poisson_parameters_f64 *par, uammd_poisson_f64 **out, uammd_poisson_info_f64 *info);
int uammd_poisson_destroy_f64(uammd_poisson_f64 *h);
int uammd_poisson_sum_f64(uammd_poisson_f64 *h, const double *d_pos, const double *d_charge, int numberParticles, double *d_force, double *d_energy,
                          int nearForce, int nearEnergy, void *stream);
int uammd_poisson_field_potential_f64(uammd_poisson_f64 *h, const double *d_pos, const double *d_charge, int numberParticles, double *d_fieldPotential,
                                      double *d_force, double *d_energy, void *stream);
int uammd_pse_far_raw_cells_f64(const double boxSize[3], double psi, double tolerance, int cells_out[3]);
int uammd_pse_far_create_f64(const double boxSize[3], const int cells[3], double viscosity, double hydrodynamicRadius,
                             double tolerance, double psi, double shearStrain, uammd_fcm_f64 **out, int *support_out,
                             double *eta_out);
int uammd_pse_near_create_f64(const double boxSize[3], double viscosity, double hydrodynamicRadius, double tolerance, double psi,
                              uammd_pse_near_f64 **out, double *rcut_out, int *nPointsTable_out);
int uammd_pse_near_destroy_f64(uammd_pse_near_f64 *h);
/* d_MF real3[N] += M_near v, v = d_v with stride 4 (real4 forces) or 3 */
int uammd_pse_near_mdot_f64(uammd_pse_near_f64 *h, const double *d_pos, const double *d_v, int vstride, int numberParticles,
                            double *d_MF, void *stream);
typedef int (*uammd_matvec_fn_f64)(void *ctx, const double *d_v, double *d_Mv, int n, void *stream);
int uammd_lanczos_create_f64(uammd_lanczos_f64 **out);
/* NearField::computeStochasticDisplacements (NearField.cuh:255-284): d_BdW real3[N] = prefactor sqrt(2 T) sqrt(M_near) dW (overwritten),
 * noise keyed Saru(particle, seed1, seed2); `solver` is the caller's handle (its check schedule adapts from call to call) */
int uammd_pse_near_stochastic_f64(uammd_pse_near_f64 *h, uammd_lanczos_f64 *solver, const double *d_pos, int numberParticles, double temperature,
                                  double prefactor, unsigned int seed1, unsigned int seed2, double tolerance, double *d_BdW, void *stream,
                                  int *iterations);
int uammd_lanczos_destroy_f64(uammd_lanczos_f64 *h);
int uammd_lanczos_run_f64(uammd_lanczos_f64 *h, uammd_matvec_fn_f64 dot, void *ctx, double *d_Bv, const double *d_v,
                          double tolerance, int n, void *stream, int *iterations);
/* Solver::runIterations with real = double (see uammd_lanczos_run_iterations) */
int uammd_lanczos_run_iterations_f64(uammd_lanczos_f64 *h, uammd_matvec_fn_f64 dot, void *ctx, double *d_Bv, const double *d_v,
                                     int numberIterations, int n, void *stream, double *residual);
int uammd_lanczos_set_iteration_hard_limit_f64(uammd_lanczos_f64 *h, int limit);
int uammd_lanczos_get_last_run_required_steps_f64(uammd_lanczos_f64 *h, int *steps);
/* BDHI::Lanczos and BDHI::Cholesky with real = double (see uammd_rpy_nbody_mdot, uammd_rpy_lanczos_bdw, uammd_bdhi_cholesky_*: Integrator/BDHI/
 * BDHI_Lanczos.cu:56-188, BDHI_Cholesky.cu:34-262).  The reference's acceptance test of the two (test/BDHI/Lanczos_Cholesky) is compiled with
 * -DDOUBLE_PRECISION (its Makefile:3) and its bar of 1e-7 is for that build.  rocSOLVER dpotrf, rocBLAS dsymv / dtrmv. */
int uammd_rpy_nbody_mdot_f64(const double *d_pos, const double *d_v, int vstride, const double *d_radius, double hydrodynamicRadius, double viscosity,
                             int numberParticles, double *d_Mv, void *stream);
int uammd_rpy_lanczos_bdw_f64(uammd_lanczos_f64 *solver, const double *d_pos, const double *d_radius, double hydrodynamicRadius, double viscosity,
                              int numberParticles, const double *d_noise, double tolerance, double *d_BdW, void *stream, int *iterations);
typedef struct uammd_bdhi_cholesky_f64 uammd_bdhi_cholesky_f64;
int uammd_bdhi_cholesky_create_f64(int numberParticles, double viscosity, double hydrodynamicRadius, uammd_bdhi_cholesky_f64 **out);
int uammd_bdhi_cholesky_destroy_f64(uammd_bdhi_cholesky_f64 *h);
int uammd_bdhi_cholesky_setup_step_f64(uammd_bdhi_cholesky_f64 *h, const double *d_pos, const int *d_index, const double *d_radius, void *stream);
int uammd_bdhi_cholesky_mf_f64(uammd_bdhi_cholesky_f64 *h, const double *d_pos, const double *d_force, const int *d_index, const double *d_radius,
                               double *d_MF, void *stream);
int uammd_bdhi_cholesky_bdw_f64(uammd_bdhi_cholesky_f64 *h, const double *d_pos, const int *d_index, const double *d_radius, double *d_BdW,
                                void *stream);

#ifdef __cplusplus
}
#endif
#endif
