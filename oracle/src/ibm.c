/* ORACLE — TEST INFRASTRUCTURE ONLY (see common.h).
 *
 * Path B, Immersed Boundary spreading / interpolation on a regular grid: CPU restatement of
 *   IBM_ns::detail::computeSupportShift        misc/IBM.cu:10-31
 *   IBM_ns::detail::fillSharedWeights          misc/IBM.cu:33-65
 *   IBM_ns::particles2GridD   (spread, K13)    misc/IBM.cu:83-147
 *   IBM_ns::grid2ParticlesDTPP (gather, K14)   misc/IBM.cu:164-235
 *   LinearIndex3D / DefaultWeightCompute / DefaultQuadratureWeights   misc/IBM.cuh:65-97
 *   window functions:
 *     IBM_kernels::Gaussian                    misc/IBM_kernels.cuh:28-40
 *     FCM_ns::Kernels::Gaussian (support, rmax, a_eff, cut at rmax)   Integrator/BDHI/FCM/FCM_kernels.cuh:22-58
 *     IBM_kernels::Peskin::threePoint/fourPoint misc/IBM_kernels.cuh:115-160
 *     "constant" kernel of the reference test   test/misc/ibm/test_ibm_regular.cu:11-14
 * The device kernels add with atomics (spread) and a block tree reduction (gather): the order of the
 * sums is not defined by the reference, so parity is by tolerance; here the sums run in stencil
 * order (x fastest), particles in array order.
 */
#include "common.h"

enum { KERNEL_GAUSSIAN = 0, KERNEL_PESKIN3 = 1, KERNEL_PESKIN4 = 2, KERNEL_CONSTANT = 3, KERNEL_BARNETT_MAGLAND = 4,
       KERNEL_SIXPOINT = 5, KERNEL_GAUSS2D = 6, KERNEL_GAUSS2D_DRIFT_X = 7, KERNEL_GAUSS2D_DRIFT_Y = 8 };

/* Window description shared with the C ABI (include/uammd_hip.h: uammd_ibm_kernel) */
typedef struct {
  int kind;
  int support[3];
  real prefactor, tau, rmax; /* Gaussian: prefactor*exp(tau r^2), 0 for r >= rmax (FCM cut; +inf = no cut) */
  real invh[3];              /* Peskin: 1/h per axis */
} IBMKernel;

static inline real phi_gaussian(const IBMKernel *k, real r) {
  if (r >= k->rmax) return 0; /* FCM_kernels.cuh:55-57: only the positive side is cut */
  return k->prefactor * EXP(k->tau * r * r);
}
static inline real phi_peskin3(real invh, real rr) { /* IBM_kernels.cuh:120-136 */
  const real r = FABS(rr) * invh;
  if (r < (real)0.5) {
    const real onediv3 = (real)(1 / 3.0);
    return invh * onediv3 * ((real)1.0 + SQRT(FMA((real)(-3.0) * r, r, (real)1.0)));
  } else if (r < (real)1.5) {
    const real onediv6 = (real)(1 / 6.0);
    const real omr = (real)1.0 - r;
    return invh * onediv6 * (FMA(-(real)3.0, r, (real)5.0) - SQRT(FMA((real)(-3.0) * omr, omr, (real)1.0)));
  }
  return 0;
}
ORACLE_API real oracle_phi_peskin3(real invh, real r) { return phi_peskin3(invh, r); }
static inline real phi_peskin4(real invh, real rr) { /* IBM_kernels.cuh:145-158 */
  const real r = FABS(rr) * invh;
  const real onediv8 = (real)0.125;
  if (r < (real)1.0) {
    return invh * onediv8 * (FMA(-(real)2.0, r, (real)3.0) + SQRT(FMA((real)4.0 * r, ((real)1.0 - r), (real)1.0)));
  } else if (r < (real)2.0) {
    return invh * onediv8 *
           (FMA(-(real)2.0, r, (real)5.0) - SQRT(FMA(-((real)4.0 * r), r, FMA((real)12.0, r, (real)(-7.0)))));
  }
  return 0;
}
/* IBM_kernels.cuh:82-90 (BM) and :107-109; prefactor = 1/norm, tau = beta, rmax = alpha */
static inline real bm_window(real zz, real alpha, real beta) {
  const real z = zz / alpha;
  const real z2 = z * z;
  const real dz2 = (real)1.0 - z2;
  return (dz2 < (real)0.0) ? (real)0.0 : EXP(beta * (SQRT(dz2) - (real)1.0));
}
/* invh[0] = length unit a: phi(r) = bm.phi(r/a)/a (FCM_kernels.cuh:151-154); a = 1 is the plain window */
static inline real phi_barnett_magland(const IBMKernel *k, real r) {
  return bm_window(r / k->invh[0], k->rmax, k->tau) * k->prefactor / k->invh[0];
}
/* IBM_kernels.cuh:93-97 with detail::integrate (:56-77) and detail::kahanSum (:44-54) */
ORACLE_API real oracle_bm_norm(real alpha, real beta) {
  int Nr = 20000;
  const real rmin = 0, rmax = alpha;
  const real dx = (rmax - rmin) / Nr;
  real sum = 0, c = 0;
  for (int i = 0; i <= Nr; i++) {
    real weight;
    if (i == 0 || i == Nr) weight = 1;
    else if (i % 2 == 1) weight = 4;
    else weight = 2;
    const real f = weight * bm_window(rmin + i * dx, alpha, beta);
    const real y = f - c;
    const real t = sum + y;
    c = (t - sum) - y;
    sum = t;
  }
  const double integral = dx / 3.0 * sum;
  return (real)(2.0 * integral);
}
/* IBM_kernels.cuh:171-221 (phi_impl) and :230-232 (phi) */
static inline real phi_sixpoint(real invh, real rr) {
  const real r = FABS(rr) * invh;
  const real K = (real)0.714075092976608;
  if (r >= (real)3) return 0;
  const real R = r - CEIL(r) + (real)1.0;
  const real R2 = R * R;
  const real R3 = R2 * R;
  const real alpha = (real)28.;
  const real beta = (real)(9.0 / 4.0) - (real)1.5 * (K + R2) + ((real)(22. / 3) - (real)7.0 * K) * R - (real)(7. / 3.) * R3;
  const real gamma = (real)0.25 * ((real)0.5 * ((real)161. / (real)36 - (real)59. / (real)6 * K + (real)5 * K * K) * R2 +
                                   (real)1. / (real)3 * ((real)-109. / (real)24 + (real)5 * K) * R2 * R2 +
                                   (real)5. / (real)18 * R3 * R3);
  const real discr = beta * beta - (real)4.0 * alpha * gamma;
  const real prefactor = (real)1. / ((real)2 * alpha) * (-beta + SQRT(discr)); /* sign(3/2 - K) = +1 */
  real v = 0;
  if (r <= (real)0) {
    const real rp1 = r + (real)1.0;
    v = (real)2. * prefactor + (real)0.25 + (real)(1. / 6) * ((real)4 - (real)3 * K) * rp1 - (real)(1. / 6) * rp1 * rp1 * rp1;
  } else if (r <= (real)1) {
    v = (real)2.0 * prefactor + (real)(5. / 8) - (real)0.25 * (K + r * r);
  } else if (r <= (real)2) {
    const real rm1 = r + (real)-1.0;
    v = (real)-3.0 * prefactor + (real)0.25 - (real)(1. / 6.) * ((real)4 - (real)3 * K) * rm1 + (real)(1. / 6) * rm1 * rm1 * rm1;
  } else if (r <= (real)3) {
    const real rm2 = r + (real)-2.0;
    v = prefactor - (real)(1. / 16) + (real)(1. / 8) * (K + rm2 * rm2) - (real)(1. / 12) * ((real)3 * K - (real)1) * rm2 -
        (real)(1. / 12) * rm2 * rm2 * rm2;
  }
  return v * invh;
}
ORACLE_API real oracle_phi_sixpoint(real invh, real r) { return phi_sixpoint(invh, r); }
static inline real phi_axis(const IBMKernel *k, int axis, real r) {
  switch (k->kind) {
    case KERNEL_GAUSSIAN: return phi_gaussian(k, r);
    case KERNEL_PESKIN3: return phi_peskin3(k->invh[axis], r);
    case KERNEL_PESKIN4: return phi_peskin4(k->invh[axis], r);
    case KERNEL_BARNETT_MAGLAND: return phi_barnett_magland(k, r);
    case KERNEL_SIXPOINT: return phi_sixpoint(k->invh[axis], r);
    /* BDHI2D_ns::Gaussian / GaussianThermalDrift<dir> (Integrator/Hydro/BDHI_quasi2D.cuh:112-153): phiZ = 1 */
    case KERNEL_GAUSS2D: return k->prefactor * EXP(k->tau * r * r);
    case KERNEL_GAUSS2D_DRIFT_X: return k->prefactor * EXP(k->tau * r * r) * (axis == 0 ? r : (real)1.0);
    case KERNEL_GAUSS2D_DRIFT_Y: return k->prefactor * EXP(k->tau * r * r) * (axis == 1 ? r : (real)1.0);
    default: return (real)1.0;
  }
}

/* IBM.cu:10-31 */
static int3 compute_support_shift(const Grid *g, real3 pos, int3 celli, int3 support) {
  int3 P = mki3(support.x / 2, support.y / 2, support.z / 2);
  real3 d = grid_distance_to_cell_center(g, pos, mki3(celli.x - P.x, celli.y - P.y, celli.z - P.z));
  d.x = FABS(d.x); d.y = FABS(d.y); d.z = FABS(d.z);
  const real3 cs = g->cellSize;
  if (cs.x > 0 && d.x > (real)support.x * cs.x / (real)2.0) P.x -= 1;
  if (cs.y > 0 && d.y > (real)support.y * cs.y / (real)2.0) P.y -= 1;
  if (cs.z > 0 && d.z > (real)support.z * cs.z / (real)2.0) P.z -= 1;
  return P;
}

typedef struct {
  int3 celli, P, support;
  real wx[64], wy[64], wz[64];
} Stencil;

/* thread-0 prologue of both kernels + fillSharedWeights (IBM.cu:33-65, :111-123) */
static void make_stencil(Stencil *s, const Grid *g, const IBMKernel *k, real3 pi, int is2D) {
  s->celli = grid_get_cell(g, pi);
  s->support = mki3(k->support[0], k->support[1], k->support[2]);
  s->P = compute_support_shift(g, pi, s->celli, s->support);
  if (is2D) { s->P.z = 0; s->support.z = 1; }
  for (int i = 0; i < s->support.x; i++) {
    int3 cj = mki3(grid_pbc_coord(g, 0, s->celli.x + i - s->P.x), s->celli.y, s->celli.z);
    s->wx[i] = (cj.x >= 0) ? phi_axis(k, 0, grid_distance_to_cell_center(g, pi, cj).x) : 0;
  }
  for (int i = 0; i < s->support.y; i++) {
    int3 cj = mki3(s->celli.x, grid_pbc_coord(g, 1, s->celli.y + i - s->P.y), s->celli.z);
    s->wy[i] = (cj.y >= 0) ? phi_axis(k, 1, grid_distance_to_cell_center(g, pi, cj).y) : 0;
  }
  for (int i = 0; i < s->support.z; i++) {
    int3 cj = mki3(s->celli.x, s->celli.y, grid_pbc_coord(g, 2, s->celli.z + i - s->P.z));
    s->wz[i] = (cj.z >= 0) ? phi_axis(k, 2, grid_distance_to_cell_center(g, pi, cj).z) : 0;
    /* 2D: the Peskin windows of the reference tests return phiZ = 1 (test_ibm_regular.cu:83-85) */
    if (is2D && (k->kind == KERNEL_PESKIN3 || k->kind == KERNEL_PESKIN4 || k->kind >= KERNEL_GAUSS2D)) s->wz[i] = 1;
  }
}

/* Spread: grid[cell] += v * phiX * phiY * phiZ (IBM.cu:125-146; DefaultWeightCompute IBM.cuh:88-97).
 * pos: real[posStride*N] (xyz first); v: real[ncomp*N]; grid: real[ncomp * nxStride*ny*nz], component
 * interleaved, LinearIndex3D(nxStride, ny, nz). */
ORACLE_API void oracle_ibm_spread(const real *pos, int posStride, const real *v, int ncomp, int N, const real *L,
                                  const int *periodic, const int *cellDim, int nxStride, const IBMKernel *k,
                                  real *gridData) {
  Box box = box_from(L, periodic);
  Grid g = grid_make(box, mki3(cellDim[0], cellDim[1], cellDim[2]));
  const int is2D = g.cellDim.z == 1; /* IBM.cuh:189-194 */
  if (oracle_get_parallel()) { /* cpu_baseline only: particles over the cores, atomic adds as the reference's atomicAdd (IBM.cu:143) */
#pragma omp parallel for schedule(static)
    for (int id = 0; id < N; id++) {
      Stencil s;
      real3 pi = mk3(pos[posStride * id], pos[posStride * id + 1], pos[posStride * id + 2]);
      make_stencil(&s, &g, k, pi, is2D);
      int nn = s.support.x * s.support.y * (is2D ? 1 : s.support.z);
      for (int i = 0; i < nn; i++) {
        const int ii = i % s.support.x, jj = (i / s.support.x) % s.support.y, kk = is2D ? 0 : (i / (s.support.x * s.support.y));
        const int3 cj = grid_pbc_cell(&g, mki3(s.celli.x + ii - s.P.x, s.celli.y + jj - s.P.y, is2D ? 0 : (s.celli.z + kk - s.P.z)));
        if (cj.x < 0 || cj.y < 0 || cj.z < 0) continue;
        if (cj.x >= g.cellDim.x || cj.y >= g.cellDim.y || cj.z >= g.cellDim.z) continue;
        const size_t jcell = (size_t)cj.x + (size_t)nxStride * ((size_t)cj.y + (size_t)g.cellDim.y * (size_t)cj.z);
        for (int c = 0; c < ncomp; c++) {
          const real add = v[ncomp * id + c] * s.wx[ii] * s.wy[jj] * s.wz[kk];
#pragma omp atomic
          gridData[ncomp * jcell + c] += add;
        }
      }
    }
    return;
  }
  Stencil s;
  for (int id = 0; id < N; id++) {
    real3 pi = mk3(pos[posStride * id], pos[posStride * id + 1], pos[posStride * id + 2]);
    make_stencil(&s, &g, k, pi, is2D);
    int nn = s.support.x * s.support.y * (is2D ? 1 : s.support.z);
    for (int i = 0; i < nn; i++) {
      const int ii = i % s.support.x, jj = (i / s.support.x) % s.support.y, kk = is2D ? 0 : (i / (s.support.x * s.support.y));
      const int3 cj = grid_pbc_cell(&g, mki3(s.celli.x + ii - s.P.x, s.celli.y + jj - s.P.y, is2D ? 0 : (s.celli.z + kk - s.P.z)));
      if (cj.x < 0 || cj.y < 0 || cj.z < 0) continue;
      if (cj.x >= g.cellDim.x || cj.y >= g.cellDim.y || cj.z >= g.cellDim.z) continue;
      const size_t jcell = (size_t)cj.x + (size_t)nxStride * ((size_t)cj.y + (size_t)g.cellDim.y * (size_t)cj.z);
      for (int c = 0; c < ncomp; c++) gridData[ncomp * jcell + c] += v[ncomp * id + c] * s.wx[ii] * s.wy[jj] * s.wz[kk];
    }
  }
}

/* Gather: out[id] += sum_cells dV * (q[cell] * phiX * phiY * phiZ) (IBM.cu:201-234), dV = cell volume. */
ORACLE_API void oracle_ibm_gather(const real *pos, int posStride, real *out, int ncomp, int N, const real *L,
                                  const int *periodic, const int *cellDim, int nxStride, const IBMKernel *k,
                                  const real *gridData) {
  Box box = box_from(L, periodic);
  Grid g = grid_make(box, mki3(cellDim[0], cellDim[1], cellDim[2]));
  const int is2D = g.cellDim.z == 1;
#pragma omp parallel for schedule(static)
  for (int id = 0; id < N; id++) {
    Stencil s;
    real3 pi = mk3(pos[posStride * id], pos[posStride * id + 1], pos[posStride * id + 2]);
    make_stencil(&s, &g, k, pi, is2D);
    real result[4] = {0, 0, 0, 0};
    int nn = s.support.x * s.support.y * (is2D ? 1 : s.support.z);
    const real dV = g.cellVolume;
    for (int i = 0; i < nn; i++) {
      const int ii = i % s.support.x, jj = (i / s.support.x) % s.support.y, kk = is2D ? 0 : (i / (s.support.x * s.support.y));
      const int3 cj = grid_pbc_cell(&g, mki3(s.celli.x + ii - s.P.x, s.celli.y + jj - s.P.y, is2D ? 0 : (s.celli.z + kk - s.P.z)));
      if (cj.x < 0 || cj.y < 0 || cj.z < 0) continue;
      if (cj.x >= g.cellDim.x || cj.y >= g.cellDim.y || cj.z >= g.cellDim.z) continue;
      const size_t jcell = (size_t)cj.x + (size_t)nxStride * ((size_t)cj.y + (size_t)g.cellDim.y * (size_t)cj.z);
      for (int c = 0; c < ncomp; c++) result[c] = FMA(dV, gridData[ncomp * jcell + c] * s.wx[ii] * s.wy[jj] * s.wz[kk], result[c]);
    }
    for (int c = 0; c < ncomp; c++) out[ncomp * id + c] += result[c];
  }
}

/* The stencil of one particle (tests): celli, P, support and the three 1-D weight rows. */
ORACLE_API void oracle_ibm_stencil(const real *pos3, const real *L, const int *periodic, const int *cellDim,
                                   const IBMKernel *k, int *celli, int *P, int *support, real *wx, real *wy, real *wz) {
  Box box = box_from(L, periodic);
  Grid g = grid_make(box, mki3(cellDim[0], cellDim[1], cellDim[2]));
  Stencil s;
  make_stencil(&s, &g, k, mk3(pos3[0], pos3[1], pos3[2]), g.cellDim.z == 1);
  celli[0] = s.celli.x; celli[1] = s.celli.y; celli[2] = s.celli.z;
  P[0] = s.P.x; P[1] = s.P.y; P[2] = s.P.z;
  support[0] = s.support.x; support[1] = s.support.y; support[2] = s.support.z;
  for (int i = 0; i < s.support.x; i++) wx[i] = s.wx[i];
  for (int i = 0; i < s.support.y; i++) wy[i] = s.wy[i];
  for (int i = 0; i < s.support.z; i++) wz[i] = s.wz[i];
}

/* FCM_ns::Kernels::Gaussian(h, tolerance): FCM_kernels.cuh:22-58 on top of IBM_kernels::Gaussian(width)
 * (IBM_kernels.cuh:28-40).  out = {upsampling, width, prefactor, tau, rmax, a_eff}; returns support. */
static real fcm_upsampling(real tolerance) { /* FCM_kernels.cuh:24-30 */
  real amin = (real)0.55;
  real amax = (real)1.65;
#ifdef DOUBLE_PRECISION
  real x = -log10(3 * tolerance) / 10.0;
#else
  real x = (real)(-(double)log10f(3 * tolerance) / 10.0);
#endif
  real factor = amin + x * (amax - amin);
  return factor < amax ? factor : amax;
}
ORACLE_API int oracle_fcm_gaussian_init(real h, real tolerance, real *out6) {
  const real ups = fcm_upsampling(tolerance);
  const real width = h * ups;
  /* IBM_kernels::Gaussian: prefactor(pow(2.0*M_PI*width*width, -0.5)), tau(-0.5/(width*width)) in double, stored as real */
  const real prefactor = (real)pow(2.0 * M_PI * (double)width * (double)width, -0.5);
  const real tau = (real)(-0.5 / ((double)width * (double)width));
  const real dr = (real)(0.5 * (double)h);
  real r = dr;
  while (prefactor * EXP(tau * r * r) > tolerance) r += dr;
  int support = (int)(2 * r / h + 0.5);
  if (support < 3) support = 3;
  out6[0] = ups; out6[1] = width; out6[2] = prefactor; out6[3] = tau;
  out6[4] = (real)support * h;
  out6[5] = (real)((double)(h * ups) * sqrt(M_PI));
  return support;
}
ORACLE_API real oracle_fcm_advise_grid_size(real hydrodynamicRadius, real tolerance) { /* FCM_kernels.cuh:47-50 */
  real factor = fcm_upsampling(tolerance);
  return (real)((double)hydrodynamicRadius / (sqrt(M_PI) * (double)factor));
}
