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
/* test/tuning hooks: "force_radix" = 1 makes the build use the stable radix sort path */
int uammd_celllist_set_option(uammd_celllist *h, const char *name, int value);
/* library-wide tunables: "lj_brick_bits" in 3..6 (cells per LDS brick = 2^k) */
int uammd_hip_set_tunable(const char *name, int value);

/* ParticleSorter::updateOrderWithCustomHash + applyCurrentOrder building blocks
 * (ParticleSorter.cuh:118-135,178-187): stable sort of (key,value) on key bits [0,end_bit) and a
 * gather out[i] = in[index[i]] for 4/8/12/16-byte elements.  Used by ParticleData::sortParticles. */
int uammd_sort_pairs(unsigned int *d_keys, int *d_values, int n, int end_bit, void *stream);
int uammd_gather(const void *d_in, const int *d_index, void *d_out, int n, int elem_bytes, void *stream);

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

/* flags for `algo` */
#define UAMMD_LJ_ALGO_AUTO 0    /* the kernel measured fastest for the grid (currently GENERAL) */
#define UAMMD_LJ_ALGO_GENERAL 1 /* thread-per-particle walk of the 27 cells (any grid) */
#define UAMMD_LJ_ALGO_BRICK 2   /* force the LDS-tiled kernel (error if the grid does not allow it) */
#define UAMMD_LJ_ALGO_QUAD 3    /* force the uniform-j (scalar-streamed neighbours) kernel */

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
int uammd_verletnvt_basic(int step, float *d_pos, float *d_vel, float *d_force, const float *d_mass,
                          float defaultMass, const int *d_index, int numberParticles, float dt, float friction,
                          int is2D, float noiseAmplitude, unsigned int stepNum, unsigned int seed, void *stream);
int uammd_verletnvt_initial_velocities(float *d_vel, const int *d_index, float velAmplitude, int is2D,
                                       int numberParticles, unsigned int seed, void *stream);
int uammd_bd_euler_maruyama(float *d_pos, const int *d_index, const float *d_force, const float K[9],
                            float selfMobility, const float *d_radius, float dt, int is2D, float temperature,
                            int numberParticles, unsigned int stepNum, unsigned int seed, void *stream);
int uammd_fcm_euler_maruyama(float *d_pos, const int *d_index, const float *d_linearVelocity, int numberParticles,
                             float dt, void *stream);
int uammd_fill_zero(void *d_ptr, size_t bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif
