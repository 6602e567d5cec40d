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
  rocblas_status (*dsymv)(rocblas_handle, rocblas_fill, rocblas_int, const double *, const double *, rocblas_int, const double *,
                          rocblas_int, const double *, double *, rocblas_int) = nullptr;
  rocblas_status (*dtrmv)(rocblas_handle, rocblas_fill, rocblas_operation, rocblas_diagonal, rocblas_int, const double *, rocblas_int,
                          double *, rocblas_int) = nullptr;
  rocblas_status (*dpotrf)(rocblas_handle, const rocblas_fill, const rocblas_int, double *, const rocblas_int, rocblas_int *) = nullptr;
  bool ok = false;
  // the library call of either precision behind one name
  rocblas_status symv(rocblas_handle h, rocblas_int n, const float *a, const float *A, const float *x, const float *b, float *y) const {
    return ssymv(h, rocblas_fill_upper, n, a, A, n, x, 1, b, y, 1);
  }
  rocblas_status symv(rocblas_handle h, rocblas_int n, const double *a, const double *A, const double *x, const double *b, double *y) const {
    return dsymv(h, rocblas_fill_upper, n, a, A, n, x, 1, b, y, 1);
  }
  rocblas_status trmv(rocblas_handle h, rocblas_int n, const float *A, float *x) const {
    return strmv(h, rocblas_fill_upper, rocblas_operation_transpose, rocblas_diagonal_non_unit, n, A, n, x, 1);
  }
  rocblas_status trmv(rocblas_handle h, rocblas_int n, const double *A, double *x) const {
    return dtrmv(h, rocblas_fill_upper, rocblas_operation_transpose, rocblas_diagonal_non_unit, n, A, n, x, 1);
  }
  rocblas_status potrf(rocblas_handle h, rocblas_int n, float *A, rocblas_int *info) const { return spotrf(h, rocblas_fill_upper, n, A, n, info); }
  rocblas_status potrf(rocblas_handle h, rocblas_int n, double *A, rocblas_int *info) const { return dpotrf(h, rocblas_fill_upper, n, A, n, info); }
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
           sym(d.blas, "rocblas_strmv", d.strmv) && sym(d.solver, "rocsolver_spotrf", d.spotrf) && sym(d.blas, "rocblas_dsymv", d.dsymv) &&
           sym(d.blas, "rocblas_dtrmv", d.dtrmv) && sym(d.solver, "rocsolver_dpotrf", d.dpotrf);
  });
  return g_dense;
}
}  // namespace

template <class T> struct CholeskyBDHIT {
  int N = 0;
  T viscosity = 1, rh = -1;
  DeviceBuffer M, force3, info;
  rocblas_handle handle = nullptr;
  bool isMup2date = false;
  ~CholeskyBDHIT() {
    if (handle) dense_libs().destroy_handle(handle);
  }
};

template <class T> UH_D T chol_fma(T a, T b, T c);
template <> UH_D float chol_fma<float>(float a, float b, float c) { return fmaf(a, b, c); }
template <> UH_D double chol_fma<double>(double a, double b, double c) { return fma(a, b, c); }
// BDHI::RotnePragerYamakawa::operator() for two radii (Integrator/BDHI/BDHI.cuh:27-96): c1 = f(r), c2 = g(r)/r^2
template <class T> UH_D void chol_rpy(T M0, T r, T ai, T aj, T &c1, T &c2) {
  const T asum = ai + aj;
  const T asub = ai > aj ? ai - aj : aj - ai;
  if (r > asum) {
    const T invr = T(1) / r;
    const T pref = M0 * T(3) * T(0.25) * invr;
    const T denom = chol_fma(ai, ai, aj * aj) / (T(3) * r * r);
    c1 = pref * (T(1) + denom);
    c2 = pref * chol_fma(T(-3), denom, T(1)) * invr * invr;
  } else if (r > asub) {
    const T pref = M0 / (ai * aj * T(32) * r * r * r);
    T num = chol_fma(T(3) * r, r, asub * asub);
    c1 = pref * chol_fma(T(16) * r * r * r, asum, -(num * num));
    num = chol_fma(-r, r, asub * asub);
    c2 = pref * (T(3) * num * num) / (r * r);
  } else {
    c1 = M0 / (ai > aj ? ai : aj);
    c2 = T(0);
  }
}

// One thread per 3x3 block (i <= j) of the upper triangle; x fastest over i so that a wave writes 64 consecutive rows of the
// same three columns (the matrix is column major).  The reference runs one thread per ROW with a serial loop over j.
template <class T>
__global__ void __launch_bounds__(256) k_chol_fill(T *__restrict__ M, const T *__restrict__ pos, const int *__restrict__ index,
                                                   const T *__restrict__ radius, T rh, T M0, int N) {
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= N || j >= N || i > j) return;
  const size_t n = 3 * (size_t)N;
  const int gi = index ? index[i] : i, gj = index ? index[j] : j;
  const T ai = radius ? radius[gi] : rh, aj = radius ? radius[gj] : rh;
  T b[3][3];
  if (i == j) {
    T c1, c2;
    chol_rpy<T>(M0, T(0), ai, ai, c1, c2);
    for (int k = 0; k < 3; ++k)
      for (int l = 0; l < 3; ++l) b[k][l] = k == l ? c1 : T(0);
  } else {
    const T rij[3] = {pos[4 * (size_t)gj] - pos[4 * (size_t)gi], pos[4 * (size_t)gj + 1] - pos[4 * (size_t)gi + 1], pos[4 * (size_t)gj + 2] - pos[4 * (size_t)gi + 2]};
    const T r = sqrt(chol_fma(rij[2], rij[2], chol_fma(rij[1], rij[1], rij[0] * rij[0])));
    T c1, c2;
    chol_rpy<T>(M0, r, ai, aj, c1, c2);
    for (int k = 0; k < 3; ++k)
      for (int l = 0; l < 3; ++l) b[k][l] = c2 * rij[k] * rij[l];
    for (int k = 0; k < 3; ++k) b[k][k] += c1;
  }
  for (int l = 0; l < 3; ++l)
    for (int k = 0; k < 3; ++k) M[3 * (size_t)i + k + n * (3 * (size_t)j + l)] = b[k][l];
}

template <class T>
__global__ void __launch_bounds__(256) k_chol_force3(const T *__restrict__ force, const int *__restrict__ index, T *__restrict__ out, int N) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= N) return;
  const T *f = force + 4 * (size_t)(index ? index[id] : id);
  out[3 * (size_t)id] = f[0]; out[3 * (size_t)id + 1] = f[1]; out[3 * (size_t)id + 2] = f[2];
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

namespace uammd_hip {
template <class T> static int chol_create(const char *fn, int numberParticles, T viscosity, T hydrodynamicRadius, void **out) {
  if (!out || numberParticles <= 0 || !(viscosity > 0)) { set_last_error("%s: bad arguments", fn); return -1; }
  const DenseLibs &d = dense_libs();
  if (!d.ok) { set_last_error("%s: librocblas.so / librocsolver.so could not be loaded (%s)", fn, dlerror()); return -5; }
  CholeskyBDHIT<T> *c = new (std::nothrow) CholeskyBDHIT<T>();
  if (!c) { set_last_error("%s: out of host memory", fn); return -3; }
  c->N = numberParticles;
  c->viscosity = viscosity;
  c->rh = hydrodynamicRadius;
  const size_t n = 3 * (size_t)numberParticles;
  int e = c->M.reserve(sizeof(T) * (n * n + 1));
  if (!e) e = c->force3.reserve(sizeof(T) * n);
  if (!e) e = c->info.reserve(sizeof(int));
  if (e) { delete c; return e; }
  if (hipMemset(c->M.ptr, 0, sizeof(T) * (n * n + 1)) != hipSuccess) { delete c; set_last_error("hipMemset failed"); return -4; }
  if (d.create_handle(&c->handle) != rocblas_status_success) { delete c; set_last_error("rocblas_create_handle failed"); return -5; }
  *out = c;
  return 0;
}
template <class T> static int chol_setup(const char *fn, void *h, const T *d_pos, const int *d_index, const T *d_radius, void *stream) {
  if (!h || !d_pos) { set_last_error("%s: null argument", fn); return -1; }
  CholeskyBDHIT<T> *c = static_cast<CholeskyBDHIT<T> *>(h);
  if (!d_radius && !(c->rh > 0)) {
    set_last_error("[BDHI::Cholesky] You need to provide Cholesky with either an hydrodynamic radius or via the individual particle radius.");
    return -2;
  }
  const T M0 = (T)(1 / (6 * M_PI * c->viscosity));
  // hydrodynamicRadius > 0 wins over the per-particle radii (BDHI_Cholesky.cu:102-104, :54)
  hipLaunchKernelGGL((k_chol_fill<T>), dim3((c->N + 63) / 64, (c->N + 3) / 4), dim3(256), 0, (hipStream_t)stream, (T *)c->M.ptr, d_pos, d_index,
                     c->rh > 0 ? (const T *)nullptr : d_radius, c->rh, M0, c->N);
  UH_CHECK(hipGetLastError());
  c->isMup2date = true;
  return 0;
}
template <class T> static int chol_mf(const char *fn, void *h, const T *d_pos, const T *d_force, const int *d_index, const T *d_radius, T *d_MF, void *stream) {
  if (!h || !d_force || !d_MF) { set_last_error("%s: null argument", fn); return -1; }
  CholeskyBDHIT<T> *c = static_cast<CholeskyBDHIT<T> *>(h);
  const DenseLibs &d = dense_libs();
  if (!c->isMup2date)  // "You should call computeMF immediately after setup_step" (:200-209)
    if (int e = chol_setup<T>(fn, h, d_pos, d_index, d_radius, stream)) return e;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL((k_chol_force3<T>), dim3((c->N + 255) / 256), dim3(256), 0, st, d_force, d_index, (T *)c->force3.ptr, c->N);
  UH_CHECK(hipGetLastError());
  UH_ROCBLAS(d.set_stream(c->handle, st));
  const T alpha = 1, beta = 0;
  UH_ROCBLAS(d.symv(c->handle, 3 * c->N, &alpha, (const T *)c->M.ptr, (const T *)c->force3.ptr, &beta, d_MF));
  return 0;
}
// d_BdW holds the N(0,1) draws on entry and B dW on exit.  Returns -6 when the factorisation finds M not positive definite.
template <class T> static int chol_bdw(const char *fn, void *h, const T *d_pos, const int *d_index, const T *d_radius, T *d_BdW, void *stream) {
  if (!h || !d_BdW) { set_last_error("%s: null argument", fn); return -1; }
  CholeskyBDHIT<T> *c = static_cast<CholeskyBDHIT<T> *>(h);
  const DenseLibs &d = dense_libs();
  if (!c->isMup2date)
    if (int e = chol_setup<T>(fn, h, d_pos, d_index, d_radius, stream)) return e;
  c->isMup2date = false;  // the factor overwrites M (:240-241)
  hipStream_t st = (hipStream_t)stream;
  UH_ROCBLAS(d.set_stream(c->handle, st));
  UH_ROCBLAS(d.potrf(c->handle, 3 * c->N, (T *)c->M.ptr, (rocblas_int *)c->info.ptr));
  int info = 0;
  UH_CHECK(hipMemcpyAsync(&info, c->info.ptr, sizeof(int), hipMemcpyDeviceToHost, st));
  UH_CHECK(hipStreamSynchronize(st));
  if (info != 0) { set_last_error("[BDHI::Cholesky] potrf: the mobility matrix is not positive definite (leading minor %d)", info); return -6; }
  UH_ROCBLAS(d.trmv(c->handle, 3 * c->N, (const T *)c->M.ptr, d_BdW));
  return 0;
}
}  // namespace uammd_hip

using namespace uammd_hip;

extern "C" {

int uammd_bdhi_cholesky_create(int numberParticles, float viscosity, float hydrodynamicRadius, uammd_bdhi_cholesky **out) {
  return chol_create<float>("uammd_bdhi_cholesky_create", numberParticles, viscosity, hydrodynamicRadius, reinterpret_cast<void **>(out));
}
int uammd_bdhi_cholesky_destroy(uammd_bdhi_cholesky *h) {
  delete reinterpret_cast<CholeskyBDHIT<float> *>(h);
  return 0;
}
int uammd_bdhi_cholesky_setup_step(uammd_bdhi_cholesky *h, const float *d_pos, const int *d_index, const float *d_radius, void *stream) {
  return chol_setup<float>("uammd_bdhi_cholesky_setup_step", h, d_pos, d_index, d_radius, stream);
}
int uammd_bdhi_cholesky_mf(uammd_bdhi_cholesky *h, const float *d_pos, const float *d_force, const int *d_index, const float *d_radius,
                           float *d_MF, void *stream) {
  return chol_mf<float>("uammd_bdhi_cholesky_mf", h, d_pos, d_force, d_index, d_radius, d_MF, stream);
}
int uammd_bdhi_cholesky_bdw(uammd_bdhi_cholesky *h, const float *d_pos, const int *d_index, const float *d_radius, float *d_BdW,
                            void *stream) {
  return chol_bdw<float>("uammd_bdhi_cholesky_bdw", h, d_pos, d_index, d_radius, d_BdW, stream);
}
// ... with real = double (rocSOLVER dpotrf, rocBLAS dsymv / dtrmv)
int uammd_bdhi_cholesky_create_f64(int numberParticles, double viscosity, double hydrodynamicRadius, uammd_bdhi_cholesky_f64 **out) {
  return chol_create<double>("uammd_bdhi_cholesky_create_f64", numberParticles, viscosity, hydrodynamicRadius, reinterpret_cast<void **>(out));
}
int uammd_bdhi_cholesky_destroy_f64(uammd_bdhi_cholesky_f64 *h) {
  delete reinterpret_cast<CholeskyBDHIT<double> *>(h);
  return 0;
}
int uammd_bdhi_cholesky_setup_step_f64(uammd_bdhi_cholesky_f64 *h, const double *d_pos, const int *d_index, const double *d_radius, void *stream) {
  return chol_setup<double>("uammd_bdhi_cholesky_setup_step_f64", h, d_pos, d_index, d_radius, stream);
}
int uammd_bdhi_cholesky_mf_f64(uammd_bdhi_cholesky_f64 *h, const double *d_pos, const double *d_force, const int *d_index, const double *d_radius,
                               double *d_MF, void *stream) {
  return chol_mf<double>("uammd_bdhi_cholesky_mf_f64", h, d_pos, d_force, d_index, d_radius, d_MF, stream);
}
int uammd_bdhi_cholesky_bdw_f64(uammd_bdhi_cholesky_f64 *h, const double *d_pos, const int *d_index, const double *d_radius, double *d_BdW,
                                void *stream) {
  return chol_bdw<double>("uammd_bdhi_cholesky_bdw_f64", h, d_pos, d_index, d_radius, d_BdW, stream);
}

}  // extern "C"
