// BDHI::Cholesky — dense open-boundary RPY mobility with an explicit Cholesky factor (the small-N sibling of BDHI::Lanczos).
//
// Reference behaviour (Integrator/BDHI/BDHI_Cholesky.cu):
//   setup_step   fillMobilityRPYD: upper triangle of the 3N x 3N matrix, column major, 3x3 blocks
//                M_ij = c1(r) I + c2(r) r r, self blocks (M0/a_i) I                                  :34-80, :158-178
//   computeMF    real4 forces -> real3, MF = symv(upper, M, F)                                      :196-233
//   computeBdW   potrf(upper) overwrites M with U (M = U^T U), BdW = trmv(upper, transposed) dW     :235-262
// The dense algebra is library work (rocSOLVER potrf, rocBLAS symv / trmv); both libraries are loaded on first use so
// that libuammd_hip.so does not carry them as load-time dependencies.  The reference draws dW with cuRAND (unpinned third
// party): here the caller supplies the N(0,1) vector, as for BDHI::Lanczos.
#include "celllist.hpp"

#include <rocblas/rocblas.h>

#include <dlfcn.h>

#include <cmath>
#include <mutex>

namespace uammd_hip {

namespace {
struct DenseLibs {
  void *blas = nullptr, *solver = nullptr;
  rocblas_status (*create_handle)(rocblas_handle *) = nullptr;
  rocblas_status (*destroy_handle)(rocblas_handle) = nullptr;
  rocblas_status (*set_stream)(rocblas_handle, hipStream_t) = nullptr;
  rocblas_status (*ssymv)(rocblas_handle, rocblas_fill, rocblas_int, const float *, const float *, rocblas_int, const float *,
                          rocblas_int, const float *, float *, rocblas_int) = nullptr;
  rocblas_status (*strmv)(rocblas_handle, rocblas_fill, rocblas_operation, rocblas_diagonal, rocblas_int, const float *, rocblas_int,
                          float *, rocblas_int) = nullptr;
  rocblas_status (*spotrf)(rocblas_handle, const rocblas_fill, const rocblas_int, float *, const rocblas_int, rocblas_int *) = nullptr;
  bool ok = false;
};
DenseLibs g_dense;
std::once_flag g_dense_once;

template <class F> bool sym(void *lib, const char *name, F &out) {
  out = reinterpret_cast<F>(dlsym(lib, name));
  return out != nullptr;
}
const DenseLibs &dense_libs() {
  std::call_once(g_dense_once, []() {
    DenseLibs &d = g_dense;
    d.blas = dlopen("librocblas.so", RTLD_NOW | RTLD_GLOBAL);
    if (!d.blas) d.blas = dlopen("/opt/rocm/lib/librocblas.so", RTLD_NOW | RTLD_GLOBAL);
    d.solver = dlopen("librocsolver.so", RTLD_NOW | RTLD_GLOBAL);
    if (!d.solver) d.solver = dlopen("/opt/rocm/lib/librocsolver.so", RTLD_NOW | RTLD_GLOBAL);
    if (!d.blas || !d.solver) return;
    d.ok = sym(d.blas, "rocblas_create_handle", d.create_handle) && sym(d.blas, "rocblas_destroy_handle", d.destroy_handle) &&
           sym(d.blas, "rocblas_set_stream", d.set_stream) && sym(d.blas, "rocblas_ssymv", d.ssymv) &&
           sym(d.blas, "rocblas_strmv", d.strmv) && sym(d.solver, "rocsolver_spotrf", d.spotrf);
  });
  return g_dense;
}
}  // namespace

struct CholeskyBDHI {
  int N = 0;
  float viscosity = 1.f, rh = -1.f;
  DeviceBuffer M, force3, info;
  rocblas_handle handle = nullptr;
  bool isMup2date = false;
  ~CholeskyBDHI() {
    if (handle) dense_libs().destroy_handle(handle);
  }
};

// BDHI::RotnePragerYamakawa::operator() for two radii (Integrator/BDHI/BDHI.cuh:27-96): c1 = f(r), c2 = g(r)/r^2
UH_D void chol_rpy(float M0, float r, float ai, float aj, float &c1, float &c2) {
  const float asum = ai + aj;
  const float asub = fabsf(ai - aj);
  if (r > asum) {
    const float invr = 1.0f / r;
    const float pref = M0 * 3.0f * 0.25f * invr;
    const float denom = fmaf(ai, ai, aj * aj) / (3.0f * r * r);
    c1 = pref * (1.0f + denom);
    c2 = pref * fmaf(-3.0f, denom, 1.0f) * invr * invr;
  } else if (r > asub) {
    const float pref = M0 / (ai * aj * 32.0f * r * r * r);
    float num = fmaf(3.0f * r, r, asub * asub);
    c1 = pref * fmaf(16.0f * r * r * r, asum, -(num * num));
    num = fmaf(-r, r, asub * asub);
    c2 = pref * (3.0f * num * num) / (r * r);
  } else {
    c1 = M0 / (ai > aj ? ai : aj);
    c2 = 0.0f;
  }
}

// One thread per 3x3 block (i <= j) of the upper triangle; x fastest over i so that a wave writes 64 consecutive rows of the
// same three columns (the matrix is column major).  The reference runs one thread per ROW with a serial loop over j.
__global__ void __launch_bounds__(256) k_chol_fill(float *__restrict__ M, const float4 *__restrict__ pos, const int *__restrict__ index,
                                                   const float *__restrict__ radius, float rh, float M0, int N) {
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= N || j >= N || i > j) return;
  const size_t n = 3 * (size_t)N;
  const int gi = index ? index[i] : i, gj = index ? index[j] : j;
  const float ai = radius ? radius[gi] : rh, aj = radius ? radius[gj] : rh;
  float b[3][3];
  if (i == j) {
    float c1, c2;
    chol_rpy(M0, 0.0f, ai, ai, c1, c2);
    for (int k = 0; k < 3; ++k)
      for (int l = 0; l < 3; ++l) b[k][l] = k == l ? c1 : 0.0f;
  } else {
    const float4 pi = pos[gi], pj = pos[gj];
    const float rij[3] = {pj.x - pi.x, pj.y - pi.y, pj.z - pi.z};
    const float r = sqrtf(fmaf(rij[2], rij[2], fmaf(rij[1], rij[1], rij[0] * rij[0])));
    float c1, c2;
    chol_rpy(M0, r, ai, aj, c1, c2);
    for (int k = 0; k < 3; ++k)
      for (int l = 0; l < 3; ++l) b[k][l] = c2 * rij[k] * rij[l];
    for (int k = 0; k < 3; ++k) b[k][k] += c1;
  }
  for (int l = 0; l < 3; ++l)
    for (int k = 0; k < 3; ++k) M[3 * (size_t)i + k + n * (3 * (size_t)j + l)] = b[k][l];
}

__global__ void __launch_bounds__(256) k_chol_force3(const float4 *__restrict__ force, const int *__restrict__ index, float *__restrict__ out,
                                                     int N) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= N) return;
  const float4 f = force[index ? index[id] : id];
  out[3 * (size_t)id] = f.x; out[3 * (size_t)id + 1] = f.y; out[3 * (size_t)id + 2] = f.z;
}

#define UH_ROCBLAS(expr)                                                                                     \
  do {                                                                                                       \
    rocblas_status s_ = (expr);                                                                              \
    if (s_ != rocblas_status_success) {                                                                      \
      set_last_error("%s failed with rocblas_status %d (%s:%d)", #expr, (int)s_, __FILE__, __LINE__);       \
      return -20 - (int)s_;                                                                                  \
    }                                                                                                        \
  } while (0)

}  // namespace uammd_hip

using namespace uammd_hip;

extern "C" {

int uammd_bdhi_cholesky_create(int numberParticles, float viscosity, float hydrodynamicRadius, uammd_bdhi_cholesky **out) {
  if (!out || numberParticles <= 0 || !(viscosity > 0)) { set_last_error("uammd_bdhi_cholesky_create: bad arguments"); return -1; }
  const DenseLibs &d = dense_libs();
  if (!d.ok) { set_last_error("uammd_bdhi_cholesky_create: librocblas.so / librocsolver.so could not be loaded (%s)", dlerror()); return -5; }
  CholeskyBDHI *c = new (std::nothrow) CholeskyBDHI();
  if (!c) { set_last_error("uammd_bdhi_cholesky_create: out of host memory"); return -3; }
  c->N = numberParticles;
  c->viscosity = viscosity;
  c->rh = hydrodynamicRadius;
  const size_t n = 3 * (size_t)numberParticles;
  int e = c->M.reserve(sizeof(float) * (n * n + 1));
  if (!e) e = c->force3.reserve(sizeof(float) * n);
  if (!e) e = c->info.reserve(sizeof(int));
  if (e) { delete c; return e; }
  if (hipMemset(c->M.ptr, 0, sizeof(float) * (n * n + 1)) != hipSuccess) { delete c; set_last_error("hipMemset failed"); return -4; }
  if (d.create_handle(&c->handle) != rocblas_status_success) { delete c; set_last_error("rocblas_create_handle failed"); return -5; }
  *out = reinterpret_cast<uammd_bdhi_cholesky *>(c);
  return 0;
}

int uammd_bdhi_cholesky_destroy(uammd_bdhi_cholesky *h) {
  delete reinterpret_cast<CholeskyBDHI *>(h);
  return 0;
}

int uammd_bdhi_cholesky_setup_step(uammd_bdhi_cholesky *h, const float *d_pos, const int *d_index, const float *d_radius, void *stream) {
  if (!h || !d_pos) { set_last_error("uammd_bdhi_cholesky_setup_step: null argument"); return -1; }
  CholeskyBDHI *c = reinterpret_cast<CholeskyBDHI *>(h);
  if (!d_radius && !(c->rh > 0)) {
    set_last_error("[BDHI::Cholesky] You need to provide Cholesky with either an hydrodynamic radius or via the individual particle radius.");
    return -2;
  }
  const float M0 = (float)(1 / (6 * M_PI * c->viscosity));
  // hydrodynamicRadius > 0 wins over the per-particle radii (BDHI_Cholesky.cu:102-104, :54)
  hipLaunchKernelGGL(k_chol_fill, dim3((c->N + 63) / 64, (c->N + 3) / 4), dim3(256), 0, (hipStream_t)stream, (float *)c->M.ptr,
                     (const float4 *)d_pos, d_index, c->rh > 0 ? nullptr : d_radius, c->rh, M0, c->N);
  UH_CHECK(hipGetLastError());
  c->isMup2date = true;
  return 0;
}

int uammd_bdhi_cholesky_mf(uammd_bdhi_cholesky *h, const float *d_pos, const float *d_force, const int *d_index, const float *d_radius,
                           float *d_MF, void *stream) {
  if (!h || !d_force || !d_MF) { set_last_error("uammd_bdhi_cholesky_mf: null argument"); return -1; }
  CholeskyBDHI *c = reinterpret_cast<CholeskyBDHI *>(h);
  const DenseLibs &d = dense_libs();
  if (!c->isMup2date)  // "You should call computeMF immediately after setup_step" (:200-209)
    if (int e = uammd_bdhi_cholesky_setup_step(h, d_pos, d_index, d_radius, stream)) return e;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_chol_force3, dim3((c->N + 255) / 256), dim3(256), 0, st, (const float4 *)d_force, d_index, (float *)c->force3.ptr, c->N);
  UH_CHECK(hipGetLastError());
  UH_ROCBLAS(d.set_stream(c->handle, st));
  const float alpha = 1.0f, beta = 0.0f;
  UH_ROCBLAS(d.ssymv(c->handle, rocblas_fill_upper, 3 * c->N, &alpha, (const float *)c->M.ptr, 3 * c->N, (const float *)c->force3.ptr, 1,
                     &beta, d_MF, 1));
  return 0;
}

// d_BdW holds the N(0,1) draws on entry and B dW on exit.  Returns -6 when the factorisation finds M not positive definite.
int uammd_bdhi_cholesky_bdw(uammd_bdhi_cholesky *h, const float *d_pos, const int *d_index, const float *d_radius, float *d_BdW,
                            void *stream) {
  if (!h || !d_BdW) { set_last_error("uammd_bdhi_cholesky_bdw: null argument"); return -1; }
  CholeskyBDHI *c = reinterpret_cast<CholeskyBDHI *>(h);
  const DenseLibs &d = dense_libs();
  if (!c->isMup2date)
    if (int e = uammd_bdhi_cholesky_setup_step(h, d_pos, d_index, d_radius, stream)) return e;
  c->isMup2date = false;  // the factor overwrites M (:240-241)
  hipStream_t st = (hipStream_t)stream;
  UH_ROCBLAS(d.set_stream(c->handle, st));
  UH_ROCBLAS(d.spotrf(c->handle, rocblas_fill_upper, 3 * c->N, (float *)c->M.ptr, 3 * c->N, (rocblas_int *)c->info.ptr));
  int info = 0;
  UH_CHECK(hipMemcpyAsync(&info, c->info.ptr, sizeof(int), hipMemcpyDeviceToHost, st));
  UH_CHECK(hipStreamSynchronize(st));
  if (info != 0) { set_last_error("[BDHI::Cholesky] potrf: the mobility matrix is not positive definite (leading minor %d)", info); return -6; }
  UH_ROCBLAS(d.strmv(c->handle, rocblas_fill_upper, rocblas_operation_transpose, rocblas_diagonal_non_unit, 3 * c->N,
                     (const float *)c->M.ptr, 3 * c->N, d_BdW, 1));
  return 0;
}

}  // extern "C"
