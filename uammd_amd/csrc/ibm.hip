// Path B — generic Immersed Boundary spread / gather on a regular grid, for gfx950.
//
// Reference behaviour: IBM_ns::particles2GridD (misc/IBM.cu:83-147) and grid2ParticlesDTPP (:164-235):
// a block per particle walks the support^3 nodes; spread adds v*phiX*phiY*phiZ with atomics, gather
// sums dV*q*phiX*phiY*phiZ with a block reduction and adds it to the particle's output.
//
// Here one 64-lane WAVE owns a particle (4 particles per 256-thread workgroup): the 3*support 1-D
// weights are evaluated by the first lanes and fetched with ds_bpermute (no LDS allocation, no
// barrier), node indices come from a multiply-high division, the spread uses the hardware f32 atomic
// add of the L2 (unsafeAtomicAdd -> global_atomic_add_f32) and the gather reduces with DPP/shuffles.
// These are the layout-generic entry points (interleaved components, user-owned grid); the FCM solver
// has its own planar-grid variants in fcm.hip.
#include "ibm.hpp"
#include "celllist.hpp"

#include <cmath>

namespace uammd_hip {

IBMKernelDev to_dev(const uammd_ibm_kernel &k) {
  IBMKernelDev d;
  d.kind = k.kind;
  d.support = make_int3(k.support[0], k.support[1], k.support[2]);
  d.prefactor = k.prefactor;
  d.tau = k.tau;
  d.rmax = k.rmax;
  d.invhx = k.invh[0];
  d.invhy = k.invh[1];
  d.invhz = k.invh[2];
  return d;
}

struct NodeWalk {
  FastDiv dsx, dsxy;  // divide by support.x and by support.x*support.y
};

// Visits every node of the particle's stencil: f(nodeLinearIndex, wx*..., ii, jj, kk) for the lanes' share.
template <int NCOMP, bool SPREAD>
__global__ void __launch_bounds__(256) k_ibm(const float *__restrict__ pos, int posStride, const float *__restrict__ qin,
                                              float *__restrict__ qout, float *__restrict__ gridRW,
                                              const float *__restrict__ gridR, int N, GridT<float> grid, int nxStride,
                                              IBMKernelDev kern, NodeWalk nw, bool is2D) {
  const int lane = threadIdx.x & 63;
  const int id = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (id >= N) return;  // whole wave
  const real3f pi{pos[(size_t)posStride * id], pos[(size_t)posStride * id + 1], pos[(size_t)posStride * id + 2]};
  const Stencil s = make_stencil(grid, kern, pi, is2D, lane);
  const int sx = s.support.x, sy = s.support.y, sz = s.support.z;
  const int nn = sx * sy * sz;
  float v[NCOMP];
  float acc[NCOMP];
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) {
    v[c] = SPREAD ? qin[(size_t)NCOMP * id + c] : 0.0f;
    acc[c] = 0.0f;
  }
  const float dV = grid.cellVolume;
  for (int i0 = 0; i0 < nn; i0 += 64) {  // wave-uniform trip count (shuffles inside)
    const int i = i0 + lane;
    const bool in = i < nn;
    const uint iu = in ? (uint)i : 0u;
    const uint kk = nw.dsxy.div(iu);
    const uint rem = iu - kk * (uint)(sx * sy);
    const uint jj = nw.dsx.div(rem);
    const uint ii = rem - jj * (uint)sx;
    const float wx = stencil_weight(s, (int)ii);
    const float wy = stencil_weight(s, sx + (int)jj);
    const float wz = stencil_weight(s, sx + sy + (int)kk);
    if (!in) continue;
    const int cx = grid.pbc_x(s.celli.x + (int)ii - s.P.x);
    const int cy = grid.pbc_y(s.celli.y + (int)jj - s.P.y);
    const int cz = is2D ? 0 : grid.pbc_z(s.celli.z + (int)kk - s.P.z);
    if (cx < 0 || cy < 0 || cz < 0 || cx >= grid.cellDim.x || cy >= grid.cellDim.y || cz >= grid.cellDim.z) continue;
    const size_t node = (size_t)cx + (size_t)nxStride * ((size_t)cy + (size_t)grid.cellDim.y * (size_t)cz);
    if (SPREAD) {
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) unsafeAtomicAdd(&gridRW[NCOMP * node + c], v[c] * wx * wy * wz);
    } else {
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) acc[c] = fmaf(dV, gridR[NCOMP * node + c] * wx * wy * wz, acc[c]);
    }
  }
  if (!SPREAD) {
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      float t = acc[c];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
      if (lane == 0) qout[(size_t)NCOMP * id + c] += t;
    }
  }
}

static int check_ibm_args(const char *fn, int posStride, int ncomp, const int cellDim[3], int nxStride,
                          const uammd_ibm_kernel *k) {
  if (posStride < 3 || (ncomp != 1 && ncomp != 3) || !k || cellDim[0] < 1 || cellDim[1] < 1 || cellDim[2] < 1 ||
      nxStride < cellDim[0]) {
    set_last_error("%s: bad arguments (posStride=%d ncomp=%d cellDim=%d %d %d nxStride=%d)", fn, posStride, ncomp,
                   cellDim[0], cellDim[1], cellDim[2], nxStride);
    return -1;
  }
  for (int a = 0; a < 3; ++a)
    if (k->support[a] < 1 || k->support[a] > kMaxSupport) {
      set_last_error("%s: kernel support %d outside [1, %d]", fn, k->support[a], kMaxSupport);
      return -1;
    }
  if (k->kind < 0 || k->kind > kKernelGauss2DDriftY) { set_last_error("%s: unknown kernel kind %d", fn, k->kind); return -1; }
  return 0;
}

}  // namespace uammd_hip

using namespace uammd_hip;

extern "C" {

int uammd_ibm_spread(const float *d_pos, int posStride, const float *d_quantity, int ncomp, int N, const float L[3],
                     const int periodic[3], const int cellDim[3], int nxStride, const uammd_ibm_kernel *kernel,
                     float *d_grid, void *stream) {
  if (int e = check_ibm_args("uammd_ibm_spread", posStride, ncomp, cellDim, nxStride, kernel)) return e;
  if (N <= 0) return 0;
  const BoxT<float> box = make_box<float>(L, periodic);
  const GridT<float> grid = make_grid<float>(box, make_int3(cellDim[0], cellDim[1], cellDim[2]));
  const bool is2D = grid.cellDim.z == 1;
  IBMKernelDev k = to_dev(*kernel);
  if (is2D) k.support.z = 1;
  NodeWalk nw{make_fastdiv(k.support.x), make_fastdiv(k.support.x * k.support.y)};
  const dim3 g((N + 3) / 4), b(256);
  hipStream_t st = (hipStream_t)stream;
  if (ncomp == 1)
    hipLaunchKernelGGL((k_ibm<1, true>), g, b, 0, st, d_pos, posStride, d_quantity, (float *)nullptr, d_grid,
                       (const float *)nullptr, N, grid, nxStride, k, nw, is2D);
  else
    hipLaunchKernelGGL((k_ibm<3, true>), g, b, 0, st, d_pos, posStride, d_quantity, (float *)nullptr, d_grid,
                       (const float *)nullptr, N, grid, nxStride, k, nw, is2D);
  UH_CHECK(hipGetLastError());
  return 0;
}

int uammd_ibm_gather(const float *d_pos, int posStride, float *d_out, int ncomp, int N, const float L[3],
                     const int periodic[3], const int cellDim[3], int nxStride, const uammd_ibm_kernel *kernel,
                     const float *d_grid, void *stream) {
  if (int e = check_ibm_args("uammd_ibm_gather", posStride, ncomp, cellDim, nxStride, kernel)) return e;
  if (N <= 0) return 0;
  const BoxT<float> box = make_box<float>(L, periodic);
  const GridT<float> grid = make_grid<float>(box, make_int3(cellDim[0], cellDim[1], cellDim[2]));
  const bool is2D = grid.cellDim.z == 1;
  IBMKernelDev k = to_dev(*kernel);
  if (is2D) k.support.z = 1;
  NodeWalk nw{make_fastdiv(k.support.x), make_fastdiv(k.support.x * k.support.y)};
  const dim3 g((N + 3) / 4), b(256);
  hipStream_t st = (hipStream_t)stream;
  if (ncomp == 1)
    hipLaunchKernelGGL((k_ibm<1, false>), g, b, 0, st, d_pos, posStride, (const float *)nullptr, d_out,
                       (float *)nullptr, d_grid, N, grid, nxStride, k, nw, is2D);
  else
    hipLaunchKernelGGL((k_ibm<3, false>), g, b, 0, st, d_pos, posStride, (const float *)nullptr, d_out,
                       (float *)nullptr, d_grid, N, grid, nxStride, k, nw, is2D);
  UH_CHECK(hipGetLastError());
  return 0;
}

// FCM_ns::Kernels::Gaussian(h, tolerance): BDHI/FCM/FCM_kernels.cuh:22-58 (host arithmetic in `real` = float,
// except where the reference itself promotes to double).
static float fcm_upsampling(float tolerance) {
  const float amin = 0.55f, amax = 1.65f;
  const float x = (float)(-(double)log10f(3 * tolerance) / 10.0);
  const float factor = amin + x * (amax - amin);
  return factor < amax ? factor : amax;
}

int uammd_fcm_gaussian_kernel(float h, float tolerance, uammd_ibm_kernel *out, float *a_eff) {
  if (!out || !(h > 0) || !(tolerance > 0)) { set_last_error("uammd_fcm_gaussian_kernel: bad arguments"); return -1; }
  const float ups = fcm_upsampling(tolerance);
  const float width = h * ups;
  const float prefactor = (float)pow(2.0 * M_PI * (double)width * (double)width, -0.5);
  const float tau = (float)(-0.5 / ((double)width * (double)width));
  const float dr = (float)(0.5 * (double)h);
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
  if (a_eff) *a_eff = (float)((double)(h * ups) * sqrt(M_PI));
  return 0;
}

int uammd_ibm_barnett_magland_kernel(float alpha, float beta, int support, float lengthUnit, uammd_ibm_kernel *out) {
  if (!out || !(alpha > 0) || !(lengthUnit > 0) || support < 1 || support > kMaxSupport) {
    set_last_error("uammd_ibm_barnett_magland_kernel: bad arguments");
    return -1;
  }
  // norm = 2 * Simpson(BM, 0, alpha, 20000 intervals), single-precision samples added with a compensated sum
  // (misc/IBM_kernels.cuh:44-79, :93-97)
  const int Nr = 20000;
  const float dx = (alpha - 0.0f) / (float)Nr;
  float sum = 0.0f, c = 0.0f;
  for (int i = 0; i <= Nr; i++) {
    const float weight = (i == 0 || i == Nr) ? 1.0f : ((i % 2 == 1) ? 4.0f : 2.0f);
    const float z = (0.0f + (float)i * dx) / alpha;
    const float dz2 = 1.0f - z * z;
    const float f = weight * ((dz2 < 0.0f) ? 0.0f : expf(beta * (sqrtf(dz2) - 1.0f)));
    const float y = f - c;
    const float t = sum + y;
    c = (t - sum) - y;
    sum = t;
  }
  const float norm = (float)(2.0 * ((double)dx / 3.0 * (double)sum));
  out->kind = UAMMD_IBM_KERNEL_BARNETT_MAGLAND;
  out->support[0] = out->support[1] = out->support[2] = support;
  out->prefactor = (float)(1.0 / (double)norm);
  out->tau = beta;
  out->rmax = alpha;
  out->invh[0] = out->invh[1] = out->invh[2] = lengthUnit;
  return 0;
}

float uammd_fcm_advise_grid_size(float hydrodynamicRadius, float tolerance) {
  return (float)((double)hydrodynamicRadius / (sqrt(M_PI) * (double)fcm_upsampling(tolerance)));
}

}  // extern "C"
