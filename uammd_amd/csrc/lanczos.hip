// Lanczos sqrt(M) v — Krylov approximation of the square root of a symmetric positive matrix applied to a
// vector, for the Brownian increments of BDHI methods (PSE near field, open-boundary RPY).
//
// Reference behaviour (misc/LanczosAlgorithm/LanczosAlgorithm.cu):
//   Krylov recurrence                 :102-157  w = M v_i - h_(i-1,i) v_(i-1); h_ii = w.v_i; w -= h_ii v_i;
//                                               h_(i,i+1) = |w|; breakdown guard h < 1e-3 h_ii/|z| -> 0, w = e1
//   estimate  y = |z| V_m H^(1/2) e1   :43-82, :162-172 (LAPACKE steqr + CBLAS gemv on the host)
//   convergence |Bz_i - Bz_(i-1)|/|Bz_(i-1)| <= tol, first check after an adaptive number of steps (starts at 3),
//   hard limit 200, "Could not converge"        :202-262
// The reference synchronises with the host for EVERY BLAS-1 scalar (cuBLAS host pointer mode: >= 6 syncs per
// iteration) and re-allocates V every iteration behind a cudaDeviceSynchronize (:86-99).
//
// MI355X design: the recurrence scalars live in HBM and never visit the host.  Each iteration is the user's
// M v plus three streaming kernels; every reduction is "partials + every consumer block re-sums the 1 KB of
// partials", so there is no finalize launch, no atomics, and the result is deterministic.  The host only sees
// hdiag/hsup when a convergence check is due (one small D2H copy), solves the m x m tridiagonal problem there
// (implicit QL, double precision) and sends back the m coefficients of H^(1/2) e1; the tall-skinny product
// V_m y is a bandwidth-bound streaming kernel (N x m floats read once) — MFMA would buy nothing for one RHS.
#include "celllist.hpp"

#include <chrono>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#include "lanczos_fused.hpp"

namespace uammd_hip {

constexpr int kLB = 256;       // threads per block
constexpr int kLParts = 256;   // reduction partials (one per block)
constexpr int kLDevM = 32;     // Krylov sizes whose convergence check goes through the host-mapped block (k_l_publish)
constexpr int kLPre = 6;       // elements a thread of the recurrence kernels fetches ahead (n <= kLPre x 65536)
constexpr int kLBatch = 8;     // convergence checks evaluated together (see lanczos_run: deferred checks)
// host-mapped status block (doubles): [0] seqA  [1] seqB  [3] status  [4] seqY  [16 .. 16 + kLBatch) errors  [64 .. 96) hdiag  [96 .. 128) hsup
//                                     [128 + 32 k .. ) y of batched check k
constexpr int kLStatDoubles = 128 + 32 * kLBatch;

template <class T> struct LanczosT {
  DeviceBuffer V, w, Bold, parts, scal, ycoef;
  DeviceBuffer estimates;   // the K results of a batched convergence check
  int capCols = 0, capN = 0;
  int check_convergence_steps = 3;  // Solver::Solver(), LanczosAlgorithm.cu:175
  int iterationHardLimit = 200;
  int lastRunRequiredSteps = 0;
  uammd_interleave_fn interleave = nullptr;   // uammd_lanczos_set_interleave (one-shot)
  void *interleaveCtx = nullptr;
  uammd_interleave_fn interleaveEarly = nullptr;   // uammd_lanczos_set_interleave_early (one-shot)
  void *interleaveEarlyCtx = nullptr;
  bool deferChecks = true;   // evaluate the convergence checks of several iterations together (lanczos_run)
  // a product that runs the recurrence's first and last kernels itself (lanczos_fused.hpp; single precision, unsharded vectors)
  lanczos_fused_fn fused = nullptr;
  void *fusedCtx = nullptr;
  bool fuseRecurrence = true;   // option "fuse_recurrence"
  const T *zparts = nullptr;    // lanczos_set_znorm_parts (one-shot)
  int znp = 0;
  DeviceBuffer w2, partsWide, partsB;   // the second w buffer, the product's partials of w . v_i, k_l_b's partials of |w|^2
  // vector sharded over several ranks: every dot product / norm is completed by the caller's all-reduce
  uammd_allreduce_fn reduce = nullptr;   // (single precision only)
  void *reduceCtx = nullptr;
  bool ownsFirstElement = true;  // the rank that holds global element 0 (the breakdown fallback w = e1)
  // convergence checks without stream synchronisation: the check kernels leave {sequence number, error} in host-mapped memory and the
  // host spins on the sequence number (a hipStreamSynchronize + two small copies cost ~25 us each, four checks per run at the PSE size)
  volatile double *hostStat = nullptr;
  double *devStat = nullptr;
  unsigned seq = 0;
  ~LanczosT() { if (hostStat) (void)hipHostFree((void *)hostStat); }
};

// the wave's sum in lane 63: six DPP additions in single precision (a __shfl_xor ladder is six ds_bpermute round trips, and these kernels
// are nothing but latency: ~4.5 us each for 1.2 MB vectors), the ladder in double
UH_D float wave_sum_last(float x) { return wave_sum_to_last(x); }
UH_D double wave_sum_last(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}
template <class T> UH_D T block_sum(T x, T *sh) {
  static_assert(kLB == 256, "four waves per block");
  x = wave_sum_last(x);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 63) sh[wv] = x;
  __syncthreads();
  const T t = (sh[0] + sh[2]) + (sh[1] + sh[3]);   // (the four waves' sums, every thread the same: no second ladder)
  __syncthreads();
  return t;
}
template <class T> UH_D T sum_parts(const T *__restrict__ parts, int nparts, T *sh) {  // every block: same order, same result
  // (eight partials in flight at a time, added in the loop's own order: the same bits as one load per turn — the fused product leaves one
  // partial per workgroup, 3125 at the PSE size: twelve dependent round trips per thread were most of k_l_b's 6 us)
  T x = 0;
  int k = threadIdx.x;
  for (; k + 7 * kLB < nparts; k += 8 * kLB) {
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = parts[k + u * kLB];
#pragma unroll
    for (int u = 0; u < 8; ++u) x += v[u];
  }
  {
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = k + u * kLB < nparts ? parts[k + u * kLB] : T(0);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k + u * kLB < nparts) x += v[u];
  }
  T t = block_sum(x, sh);
  if (threadIdx.x == 0) sh[8] = t;
  __syncthreads();
  t = sh[8];
  __syncthreads();
  return t;
}

// scal layout (device floats): [0] = |z|, [1..] hdiag[i] at 1+i, hsup[i] at 1+cap+i
// parts <- partial sums of x[i]^2
template <class T>
__global__ void __launch_bounds__(kLB) k_l_norm2(const T *__restrict__ x, int n, T *__restrict__ parts) {
  __shared__ T sh[16];
  T a = T(0);
  for (int i = blockIdx.x * kLB + threadIdx.x; i < n; i += gridDim.x * kLB) a = fma_(x[i], x[i], a);
  const T t = block_sum(a, sh);
  if (threadIdx.x == 0) parts[blockIdx.x] = t;
}
// v0 = z / |z| ; scal[0] = |z|
template <class T>
__global__ void __launch_bounds__(kLB) k_l_first(const T *__restrict__ z, int n, const T *__restrict__ parts,
                                                 int nparts, T *__restrict__ v0, T *__restrict__ scal) {
  __shared__ T sh[16];
  const T normz = sqrt_(sum_parts(parts, nparts, sh));
  if (blockIdx.x == 0 && threadIdx.x == 0) scal[0] = normz;
  const T inv = T(1) / normz;
  for (int i = blockIdx.x * kLB + threadIdx.x; i < n; i += gridDim.x * kLB) v0[i] = z[i] * inv;
}
// w -= hsup[i-1] * v_(i-1) (if i > 0); parts <- partial w . v_i
template <class T>
__global__ void __launch_bounds__(kLB) k_l_a(T *__restrict__ w, const T *__restrict__ vprev,
                                             const T *__restrict__ vi, int n, const T *__restrict__ hsupPrev,
                                             T *__restrict__ parts) {
  __shared__ T sh[16];
  const T hp = hsupPrev ? *hsupPrev : T(0);
  T a = T(0);
  const int i0 = blockIdx.x * kLB + threadIdx.x, stride = gridDim.x * kLB;
  if ((long)n <= (long)kLPre * stride) {
    // a thread's few elements (n / 65536 of them: 4.6 at the PSE size) all in flight together instead of one round trip per turn of
    // the loop; the sums take their terms in the loop's order
    T xw[kLPre], xp[kLPre], xv[kLPre];
#pragma unroll
    for (int u = 0; u < kLPre; ++u) {
      const int i = i0 + u * stride;
      const bool in = i < n;
      xw[u] = in ? w[i] : T(0);
      xp[u] = in && vprev ? vprev[i] : T(0);
      xv[u] = in ? vi[i] : T(0);
    }
#pragma unroll
    for (int u = 0; u < kLPre; ++u) {
      const int i = i0 + u * stride;
      if (i < n) {
        T x = xw[u];
        if (vprev) { x = fma_(-hp, xp[u], x); w[i] = x; }
        a = fma_(x, xv[u], a);
      }
    }
  } else {
    for (int i = i0; i < n; i += stride) {
      T x = w[i];
      if (vprev) { x = fma_(-hp, vprev[i], x); w[i] = x; }
      a = fma_(x, vi[i], a);
    }
  }
  const T t = block_sum(a, sh);
  if (threadIdx.x == 0) parts[blockIdx.x] = t;
}
// sharded vectors: parts[0] <- sum of the g partials of this rank (then all-reduced by the caller's callback)
template <class T>
__global__ void __launch_bounds__(kLB) k_l_collapse(T *__restrict__ parts, int nparts) {
  __shared__ T sh[16];
  const T t = sum_parts(parts, nparts, sh);
  if (threadIdx.x == 0) parts[0] = t;
}
// hdiag_i = sum(partsA); w -= hdiag_i * v_i; partsB <- partial |w|^2
template <class T>
__global__ void __launch_bounds__(kLB) k_l_b(T *__restrict__ w, const T *__restrict__ vi, int n,
                                             const T *__restrict__ partsA, int nparts, T *__restrict__ hdiag_i,
                                             T *__restrict__ partsB, T *__restrict__ zero = nullptr) {
  __shared__ T sh[16];
  const int i0 = blockIdx.x * kLB + threadIdx.x, stride = gridDim.x * kLB;
  const bool pre = (long)n <= (long)kLPre * stride;   // (the thread's elements are on their way while the partials are summed)
  if (zero)   // (the run's first k_l_b also clears Bold, the estimate "before the first one": no memset launch)
    for (int i = i0; i < n; i += stride) zero[i] = T(0);
  T xw[kLPre], xv[kLPre];
  if (pre) {
#pragma unroll
    for (int u = 0; u < kLPre; ++u) {
      const int i = i0 + u * stride;
      xw[u] = i < n ? w[i] : T(0);
      xv[u] = i < n ? vi[i] : T(0);
    }
  }
  const T h = sum_parts(partsA, nparts, sh);
  if (blockIdx.x == 0 && threadIdx.x == 0) *hdiag_i = h;
  T a = T(0);
  if (pre) {
#pragma unroll
    for (int u = 0; u < kLPre; ++u) {
      const int i = i0 + u * stride;
      if (i < n) {
        const T x = fma_(-h, xv[u], xw[u]);
        w[i] = x;
        a = fma_(x, x, a);
      }
    }
  } else {
    for (int i = i0; i < n; i += stride) {
      const T x = fma_(-h, vi[i], w[i]);
      w[i] = x;
      a = fma_(x, x, a);
    }
  }
  const T t = block_sum(a, sh);
  if (threadIdx.x == 0) partsB[blockIdx.x] = t;
}
// hsup_i = |w| with the breakdown guard; v_(i+1) = w / hsup_i  (or e1)
template <class T>
__global__ void __launch_bounds__(kLB) k_l_c(const T *__restrict__ w, int n, const T *__restrict__ partsB,
                                             int nparts, const T *__restrict__ hdiag_i,
                                             const T *__restrict__ normz, T *__restrict__ hsup_i,
                                             T *__restrict__ vnext, bool ownsFirstElement) {
  __shared__ T sh[16];
  const int i0 = blockIdx.x * kLB + threadIdx.x, stride = gridDim.x * kLB;
  const bool pre = (long)n <= (long)kLPre * stride;
  T xw[kLPre];
  if (pre) {
#pragma unroll
    for (int u = 0; u < kLPre; ++u) xw[u] = i0 + u * stride < n ? w[i0 + u * stride] : T(0);
  }
  T hs = sqrt_(sum_parts(partsB, nparts, sh));
  const T tol = T(1e-3) * (*hdiag_i) / (*normz);
  if (hs < tol) hs = T(0);
  if (blockIdx.x == 0 && threadIdx.x == 0) *hsup_i = hs;
  const T inv = hs > T(0) ? T(1) / hs : T(0);
  if (pre) {
#pragma unroll
    for (int u = 0; u < kLPre; ++u) {
      const int i = i0 + u * stride;
      if (i < n) vnext[i] = hs > T(0) ? xw[u] * inv : ((i == 0 && ownsFirstElement) ? T(1) : T(0));
    }
  } else {
    for (int i = i0; i < n; i += stride)
      vnext[i] = hs > T(0) ? w[i] * inv : ((i == 0 && ownsFirstElement) ? T(1) : T(0));
  }
}
// (Round 4, measured and removed: the three recurrence kernels as ONE launch — every thread keeping its elements of w in registers across
// the two reductions, a generation-tagged counter in global memory as the barrier over the <= 256 resident blocks, partials as relaxed
// agent-scope atomics.  Same bits as the three kernels; PSE near noise 0.547 against 0.514 ms with them (7 iterations): two trips of
// every block to the device-coherent level per barrier — the arrival, the poll, then 256 uncached partial loads — cost more than the
// two kernel boundaries they replace (~5 us per kernel all in on this GPU); with __threadfence() in the barrier, 1.04 ms.)
// Bz = |z| * V[:, :m] * y ; partials of |Bold|^2 and |Bz - Bold|^2 ; then Bold <- Bz
template <class T>
__global__ void __launch_bounds__(kLB) k_l_estimate(const T *__restrict__ V, int n, int m,
                                                    const T *__restrict__ y, const T *__restrict__ normz,
                                                    T *__restrict__ Bz, T *__restrict__ Bold,
                                                    T *__restrict__ parts) {
  __shared__ T sh[16];
  const T nz = *normz;
  T a = T(0), b = T(0);
  for (int i = blockIdx.x * kLB + threadIdx.x; i < n; i += gridDim.x * kLB) {
    T s = T(0);
    for (int c = 0; c < m; ++c) s = fma_(V[(size_t)c * n + i], y[c], s);
    s *= nz;
    const T o = Bold[i];
    a = fma_(o, o, a);
    const T d = s - o;
    b = fma_(d, d, b);
    Bz[i] = s;
    Bold[i] = s;
  }
  const T ta = block_sum(a, sh);
  if (threadIdx.x == 0) parts[blockIdx.x] = ta;
  const T tb = block_sum(b, sh);
  if (threadIdx.x == 0) parts[kLParts + blockIdx.x] = tb;
}

// Convergence checks without a stream synchronisation.  hdiag / hsup are a few floats: k_l_publish copies them into host-mapped memory
// behind a sequence number; the host (spinning on that number) solves the m x m tridiagonal problem in double — microseconds there, 30-60 us
// for one GPU thread (a QL step is ~100 dependent double operations with a square root and two divisions) — and writes the m coefficients
// of H^(1/2) e1 back into mapped memory behind a second sequence number, which an ALREADY QUEUED one-thread relay kernel polls before the estimate
// reads them: the GPU never waits for a launch.  (The poll is bounded: a host that never answers makes the kernel give up with an error flag.)
// mapped block (floats): [0] seqA (scalars published)  [1] seqB (error published)  [2] err  [3] status  [4] seqY (host: y ready)
//                        [8 .. 8+32) y   [64 .. 64+32) hdiag   [96 .. 96+32) hsup
template <class T>
__global__ void k_l_publish(const T *__restrict__ hdiag, const T *__restrict__ hsup, int m, volatile double *__restrict__ stat, double seq) {
  const int k = threadIdx.x;
  if (k < m) { stat[64 + k] = hdiag[k]; stat[96 + k] = hsup[k]; }
  __threadfence_system();
  __syncthreads();
  if (k == 0) stat[0] = seq;
}
// ONE thread waits for the host's coefficients and hands them to device memory (256 workgroups polling host memory over the bus at once
// delayed the very write they were waiting for: 146 us per check)
template <class T>
__global__ void k_l_relay(volatile double *__restrict__ stat, int m, double seq, T *__restrict__ ycoef) {
  __shared__ int ok;
  if (threadIdx.x == 0) {
    long spins = 0;
    while (stat[4] != seq && ++spins < 200000000L) __builtin_amdgcn_s_sleep(8);  // (~0.25 us per poll: gives up after ~a minute, as the host does)
    ok = stat[4] == seq;
    if (!ok) stat[3] = -1.0;
  }
  __syncthreads();
  if ((int)threadIdx.x < m) ycoef[threadIdx.x] = ok ? (T)stat[8 + threadIdx.x] : T(0);
}
// err = |Bz - Bold| / |Bold| from the estimate's partials, left with a sequence number where the host can see it
template <class T>
__global__ void __launch_bounds__(kLB) k_l_error(const T *__restrict__ parts, int nparts, double *__restrict__ stat, double seq) {
  __shared__ T sh[16];
  const T a = sum_parts(parts, nparts, sh);
  const T b = sum_parts(parts + kLParts, nparts, sh);
  if (threadIdx.x == 0) {
    stat[2] = fabs(sqrt_(b) / sqrt_(a));
    __threadfence_system();
    stat[1] = seq;
  }
}

// ---- batched checks: K consecutive estimates from one pass over V -----------------------------------------------------------------
// estimate k (k < K) is the Krylov result of size m0 + k: Bz_k = |z| V[:, :m0 + k] y_k; the check of that iteration compares it with the
// one before (Bz_(-1) = Bold, the last estimate of an earlier check or zero).  Every Bz_k is stored (est + k n), the last one also into
// Bz and Bold; partials of |Bz_(k-1)|^2 and |Bz_k - Bz_(k-1)|^2 at parts[(2 k) kLParts + block] and parts[(2 k + 1) kLParts + block].
// Column by column in ascending order, then the product by |z|: the arithmetic of k_l_estimate.
template <class T>
__global__ void __launch_bounds__(kLB) k_l_estimate_batch(const T *__restrict__ V, int n, int m0, int K, const T *__restrict__ y /* [K][kLDevM] */,
                                                          const T *__restrict__ normz, T *__restrict__ est, T *__restrict__ Bz,
                                                          T *__restrict__ Bold, T *__restrict__ parts) {
  __shared__ T sh[16];
  __shared__ T ys[kLBatch * kLDevM];
  // (zero where estimate k has no term — beyond its Krylov size m0 + k, and whole rows beyond K: the sums below run without a branch and
  // fma(v, 0, s) leaves s as it is)
  for (int e = threadIdx.x; e < kLBatch * kLDevM; e += kLB) {
    const int k = e / kLDevM, c = e - k * kLDevM;
    ys[e] = (k < K && c < m0 + k) ? y[e] : T(0);
  }
  __syncthreads();
  const T nz = *normz;
  T a[kLBatch], b[kLBatch];
#pragma unroll
  for (int k = 0; k < kLBatch; ++k) a[k] = b[k] = T(0);
  const int i00 = blockIdx.x * kLB + threadIdx.x, stride0 = gridDim.x * kLB;
  const bool pre = (long)n <= (long)kLPre * stride0 && m0 + K - 1 <= 8;
  if (pre) {
    // a thread's few elements (n / 65536: 4.6 at the PSE size), each with its <= 8 columns and its previous estimate, ALL in flight
    // together instead of one round trip per element; the same terms in the same order
    T v[kLPre][8], pv[kLPre];
    const int mLast = m0 + K - 1;
#pragma unroll
    for (int e = 0; e < kLPre; ++e) {
      const int i = i00 + e * stride0;
      pv[e] = i < n ? Bold[i] : T(0);
#pragma unroll
      for (int u = 0; u < 8; ++u) v[e][u] = (i < n && u < mLast) ? V[(size_t)u * n + i] : T(0);
    }
#pragma unroll
    for (int e = 0; e < kLPre; ++e) {
      const int i = i00 + e * stride0;
      if (i >= n) continue;
      T s[kLBatch];
#pragma unroll
      for (int k = 0; k < kLBatch; ++k) s[k] = T(0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
#pragma unroll
        for (int k = 0; k < kLBatch; ++k) s[k] = fma_(v[e][u], ys[k * kLDevM + u], s[k]);
      }
      T prev = pv[e];
#pragma unroll
      for (int k = 0; k < kLBatch; ++k)
        if (k < K) {
          const T r = s[k] * nz;
          a[k] = fma_(prev, prev, a[k]);
          const T d = r - prev;
          b[k] = fma_(d, d, b[k]);
          est[(size_t)k * n + i] = r;
          prev = r;
        }
      Bz[i] = prev;
      Bold[i] = prev;
    }
  }
  for (int i = pre ? n : i00; i < n; i += stride0) {
    T s[kLBatch];
#pragma unroll
    for (int k = 0; k < kLBatch; ++k) s[k] = T(0);
    const int mLast = m0 + K - 1;
    // eight columns' loads and the previous estimate in flight together (one at a time this pass was ~mLast dependent round trips per
    // element: 21 us for a batch at the PSE size); the sums still take their terms in ascending column order
    T prev = Bold[i];
    for (int c0 = 0; c0 < mLast; c0 += 8) {
      T v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = c0 + u < mLast ? V[(size_t)(c0 + u) * n + i] : T(0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + u;
#pragma unroll
        for (int k = 0; k < kLBatch; ++k) s[k] = fma_(v[u], ys[k * kLDevM + (c & (kLDevM - 1))], s[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < kLBatch; ++k)
      if (k < K) {
        const T r = s[k] * nz;
        a[k] = fma_(prev, prev, a[k]);
        const T d = r - prev;
        b[k] = fma_(d, d, b[k]);
        est[(size_t)k * n + i] = r;
        prev = r;
      }
    Bz[i] = prev;
    Bold[i] = prev;
  }
  // the 2 K partial sums of the workgroup: every thread's terms through LDS, sixteen lanes per sum (sixteen reads and four shuffles each;
  // 2 K block_sum calls were 4 K barriers, and shuffling all sixteen values down every wave 96 dependent ds_bpermute)
  static_assert(kLB / 16 == 2 * kLBatch, "sixteen lanes per partial sum");
  __shared__ T red[2 * kLBatch][kLB + 1];
#pragma unroll
  for (int k = 0; k < kLBatch; ++k) { red[2 * k][threadIdx.x] = a[k]; red[2 * k + 1][threadIdx.x] = b[k]; }
  __syncthreads();
  {
    const int t = threadIdx.x >> 4, sub = threadIdx.x & 15;   // kLB / 16 = 2 kLBatch sums
    T x = T(0);
#pragma unroll
    for (int j = 0; j < kLB / 16; ++j) x += red[t][sub + 16 * j];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    if (sub == 0 && t < 2 * K) parts[(size_t)t * kLParts + blockIdx.x] = x;
  }
  (void)sh;
}
// ONE thread waits for the host's K coefficient vectors and hands them to device memory
template <class T>
__global__ void k_l_relay_batch(volatile double *__restrict__ stat, int K, int mLast, double seq, T *__restrict__ ycoef) {
  __shared__ int ok;
  if (threadIdx.x == 0) {
    long spins = 0;
    while (stat[4] != seq && ++spins < 200000000L) __builtin_amdgcn_s_sleep(8);
    ok = stat[4] == seq;
    if (!ok) stat[3] = -1.0;
  }
  __syncthreads();
  // only the mLast coefficients per check that the estimate reads: every read here is a trip over the bus, and K * kLDevM of them
  // were three rounds for the 64 threads (~9 us of this kernel's 25)
  for (int e = threadIdx.x; e < K * mLast; e += blockDim.x) {
    const int k = e / mLast, r = e - k * mLast;
    ycoef[k * kLDevM + r] = ok ? (T)stat[128 + k * kLDevM + r] : T(0);
  }
}
// err_k = |Bz_k - Bz_(k-1)| / |Bz_(k-1)| for the K checks, left with a sequence number where the host can see them
// (one wave per sum — 2 K waves — and one barrier)
template <class T>
__global__ void __launch_bounds__(64 * 2 * kLBatch) k_l_error_batch(const T *__restrict__ parts, int nparts, int K, double *__restrict__ stat, double seq) {
  __shared__ T tot[2 * kLBatch];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wv < 2 * K) {
    T x = T(0);
    static_assert(kLParts == 256, "four partials per lane");
    T v[4];   // (in flight together, added in the loop's order)
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = lane + 64 * u < nparts ? parts[(size_t)wv * kLParts + lane + 64 * u] : T(0);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (lane + 64 * u < nparts) x += v[u];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    if (lane == 0) tot[wv] = x;
  }
  __syncthreads();
  if ((int)threadIdx.x < K) stat[16 + threadIdx.x] = fabs(sqrt_(tot[2 * threadIdx.x + 1]) / sqrt_(tot[2 * threadIdx.x]));
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    stat[1] = seq;
  }
}
template <class T>
__global__ void __launch_bounds__(kLB) k_l_copy(const T *__restrict__ src, int n, T *__restrict__ dst) {
  for (int i = blockIdx.x * kLB + threadIdx.x; i < n; i += gridDim.x * kLB) dst[i] = src[i];
}

// Symmetric tridiagonal eigenproblem by implicit QL with eigenvector accumulation; d (diag, size m) returns the
// eigenvalues, e (sub-diagonal, e[0..m-2]) is destroyed, z (m x m, row major z[i*m+j]) must be the identity on
// entry and returns the eigenvectors in its columns.  Returns 0, or i+1 if eigenvalue i failed to converge.
static int tridiag_ql(std::vector<double> &d, std::vector<double> &e, std::vector<double> &z, int m) {
  e.resize(m, 0.0);
  for (int l = 0; l < m; ++l) {
    int iter = 0, mm;
    do {
      for (mm = l; mm < m - 1; ++mm) {
        const double dd = std::fabs(d[mm]) + std::fabs(d[mm + 1]);
        if (std::fabs(e[mm]) <= 2.220446049250313e-16 * dd) break;
      }
      if (mm != l) {
        if (iter++ == 60) return l + 1;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = std::hypot(g, 1.0);
        g = d[mm] - d[l] + e[l] / (g + (g >= 0 ? std::fabs(r) : -std::fabs(r)));
        double s = 1.0, c = 1.0, p = 0.0;
        int i;
        for (i = mm - 1; i >= l; --i) {
          double f = s * e[i], b = c * e[i];
          e[i + 1] = (r = std::hypot(f, g));
          if (r == 0.0) { d[i + 1] -= p; e[mm] = 0.0; break; }
          s = f / r;
          c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * b;
          d[i + 1] = g + (p = s * r);
          g = c * r - b;
          for (int k = 0; k < m; ++k) {
            f = z[(size_t)k * m + i + 1];
            z[(size_t)k * m + i + 1] = s * z[(size_t)k * m + i] + c * f;
            z[(size_t)k * m + i] = c * z[(size_t)k * m + i] - s * f;
          }
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p;
        e[l] = g;
        e[mm] = 0.0;
      }
    } while (mm != l);
  }
  return 0;
}

using Lanczos = LanczosT<float>;
static inline int lgrid(int n) { return std::min(kLParts, (n + kLB - 1) / kLB); }
int lanczos_set_znorm_parts(::uammd_lanczos *h, const float *parts, int np) {
  if (!h || np < 0 || np > kLParts) { set_last_error("lanczos_set_znorm_parts: bad arguments"); return -1; }
  Lanczos *L = reinterpret_cast<Lanczos *>(h);
  L->zparts = parts;
  L->znp = parts ? np : 0;
  return 0;
}
int lanczos_set_fused(::uammd_lanczos *h, lanczos_fused_fn fn, void *ctx) {
  if (!h) { set_last_error("lanczos_set_fused: null handle"); return -1; }
  Lanczos *L = reinterpret_cast<Lanczos *>(h);
  L->fused = fn;
  L->fusedCtx = ctx;
  return 0;
}

}  // namespace uammd_hip

using namespace uammd_hip;

extern "C" {

int uammd_lanczos_create(uammd_lanczos **out) {
  if (!out) { set_last_error("uammd_lanczos_create: null output"); return -1; }
  *out = reinterpret_cast<uammd_lanczos *>(new Lanczos());
  return 0;
}
int uammd_lanczos_destroy(uammd_lanczos *h) {
  delete reinterpret_cast<Lanczos *>(h);
  return 0;
}
int uammd_lanczos_set_iteration_hard_limit(uammd_lanczos *h, int limit) {
  if (!h || limit < 1) { set_last_error("uammd_lanczos_set_iteration_hard_limit: bad arguments"); return -1; }
  reinterpret_cast<Lanczos *>(h)->iterationHardLimit = limit;
  return 0;
}
int uammd_lanczos_set_option(uammd_lanczos *h, const char *name, int value) {
  if (!h || !name) { set_last_error("uammd_lanczos_set_option: null argument"); return -1; }
  if (std::string(name) == "defer_checks") { reinterpret_cast<Lanczos *>(h)->deferChecks = value != 0; return 0; }
  if (std::string(name) == "fuse_recurrence") { reinterpret_cast<Lanczos *>(h)->fuseRecurrence = value != 0; return 0; }
  set_last_error("uammd_lanczos_set_option: unknown option %s", name);
  return -1;
}
// the adaptive schedule a run leaves behind {check_convergence_steps, lastRunRequiredSteps} (LanczosAlgorithm.cu:175, :245-251): a caller
// that has to REPEAT a run (the PSE near field, when the matrix of the first attempt turns out to have been incomplete) puts it back first
int uammd_lanczos_get_schedule(uammd_lanczos *h, int state[2]) {
  if (!h || !state) { set_last_error("uammd_lanczos_get_schedule: null argument"); return -1; }
  state[0] = reinterpret_cast<Lanczos *>(h)->check_convergence_steps;
  state[1] = reinterpret_cast<Lanczos *>(h)->lastRunRequiredSteps;
  return 0;
}
int uammd_lanczos_set_schedule(uammd_lanczos *h, const int state[2]) {
  if (!h || !state || state[0] < 1 || state[1] < 0) { set_last_error("uammd_lanczos_set_schedule: bad argument"); return -1; }
  reinterpret_cast<Lanczos *>(h)->check_convergence_steps = state[0];
  reinterpret_cast<Lanczos *>(h)->lastRunRequiredSteps = state[1];
  return 0;
}
// fn(ctx, stream) is called ONCE during the NEXT run: after the kernels of its first convergence check are queued and before the host
// waits for them — or when the run ends, if it never waited.  Whatever fn queues on the stream runs behind the check while the host is
// busy with it: the one wait of a run stops being a drained stream.  One-shot: cleared when called.
int uammd_lanczos_set_interleave(uammd_lanczos *h, uammd_interleave_fn fn, void *ctx) {
  if (!h) { set_last_error("uammd_lanczos_set_interleave: null handle"); return -1; }
  reinterpret_cast<Lanczos *>(h)->interleave = fn;
  reinterpret_cast<Lanczos *>(h)->interleaveCtx = ctx;
  return 0;
}
// the same, one stage earlier: fn is called after the first check's scalars have been queued for the host and BEFORE the kernels that
// wait for the host's answer — what it queues runs while the host solves the check's tridiagonal problems (the GPU idled ~20 us there)
int uammd_lanczos_set_interleave_early(uammd_lanczos *h, uammd_interleave_fn fn, void *ctx) {
  if (!h) { set_last_error("uammd_lanczos_set_interleave_early: null handle"); return -1; }
  reinterpret_cast<Lanczos *>(h)->interleaveEarly = fn;
  reinterpret_cast<Lanczos *>(h)->interleaveEarlyCtx = ctx;
  return 0;
}
int uammd_lanczos_get_last_run_required_steps(uammd_lanczos *h, int *steps) {
  if (!h || !steps) { set_last_error("uammd_lanczos_get_last_run_required_steps: null argument"); return -1; }
  *steps = reinterpret_cast<Lanczos *>(h)->lastRunRequiredSteps;
  return 0;
}

int uammd_lanczos_set_allreduce(uammd_lanczos *h, uammd_allreduce_fn reduce, void *ctx, int ownsFirstElement) {
  if (!h) { set_last_error("uammd_lanczos_set_allreduce: null handle"); return -1; }
  Lanczos *L = reinterpret_cast<Lanczos *>(h);
  L->reduce = reduce;
  L->reduceCtx = ctx;
  L->ownsFirstElement = reduce ? ownsFirstElement != 0 : true;
  return 0;
}

}  // extern "C"

// the fused product is a single-precision, single-rank affair
static bool fused_usable(LanczosT<float> *L) { return L->fused && L->fuseRecurrence && !L->reduce; }
static bool fused_usable(LanczosT<double> *) { return false; }
static int fused_call(LanczosT<float> *L, const float *wPrev, const float *vi, const float *partsB, int npB, const float *hdiagPrev,
                      const float *normz, float *hsupPrev, float *viOut, const float *vPrev, float *wOut, float *partsA, int partsACap,
                      int *npA, int n, void *stream, bool first) {
  LanczosFusedArgs a{wPrev, vi, partsB, npB, hdiagPrev, normz, hsupPrev, viOut, L->ownsFirstElement, vPrev, wOut, partsA, partsACap, 0, first};
  const int rc = L->fused(L->fusedCtx, &a, n, stream);
  *npA = a.npA;
  return rc;
}
static int fused_call(LanczosT<double> *, const double *, const double *, const double *, int, const double *, const double *, double *,
                      double *, const double *, double *, double *, int, int *, int, void *, bool) { return 1; }

template <class T, class MatVec>
static int lanczos_run(LanczosT<T> *L, MatVec dot, void *ctx, T *d_Bv, const T *d_v, T tolerance, int n, void *stream, int *iterations) {
  if (!L || !dot || !d_Bv || !d_v || n < 1) { set_last_error("uammd_lanczos_run: bad arguments"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  const int cap = L->iterationHardLimit + 2;
  if (L->capCols < cap || L->capN < n) {
    UH_CHECK(hipStreamSynchronize(st));
    if (int e = L->V.reserve(sizeof(T) * (size_t)n * cap)) return e;
    if (int e = L->w.reserve(sizeof(T) * (size_t)n)) return e;
    if (int e = L->Bold.reserve(sizeof(T) * (size_t)n)) return e;
    if (int e = L->parts.reserve(sizeof(T) * 2 * kLParts * kLBatch)) return e;
    if (int e = L->scal.reserve(sizeof(T) * (2 * cap + 2))) return e;
    if (int e = L->ycoef.reserve(sizeof(T) * std::max(cap, kLBatch * kLDevM))) return e;
    L->capCols = cap;
    L->capN = n;
  }
  T *V = (T *)L->V.ptr, *w = (T *)L->w.ptr, *Bold = (T *)L->Bold.ptr, *parts = (T *)L->parts.ptr;
  T *scal = (T *)L->scal.ptr, *ycoef = (T *)L->ycoef.ptr;
  T *hdiag = scal + 1, *hsup = scal + 1 + cap;
  const int g = lgrid(n);
  // with sharded vectors a finished set of partials is collapsed to one number and summed over the ranks; the consumers
  // then "re-sum" a single partial
  const int np = L->reduce ? 1 : g;
  auto complete = [&](T *p) -> int {
    if (!L->reduce) return 0;
    hipLaunchKernelGGL(k_l_collapse<T>, dim3(1), dim3(kLB), 0, st, p, g);
    return L->reduce(L->reduceCtx, (float *)p, 1, stream);   // (reduce is only ever set on the float solver)
  };
  // With a fused product the run's first three launches go into their neighbours: the product of iteration 0 makes v_0 = z / |z| in its
  // prologue (from the partials of |z|^2: the caller's — lanczos_set_znorm_parts — or k_l_norm2's), the first k_l_b zeroes Bold.
  if (!L->partsB.ptr) { if (int e = L->partsB.reserve(sizeof(T) * kLParts)) return e; }
  bool v0Pending = fused_usable(L), boldPending = v0Pending;
  // (caller-supplied partials of |z|^2 are LOCAL sums: on a sharded solver the norm must go through the all-reduce, so they are dropped)
  const T *zparts = L->reduce ? nullptr : L->zparts;
  int znp = L->reduce ? 0 : L->znp;
  L->zparts = nullptr;
  L->znp = 0;
  auto plain_start = [&]() -> int {
    if (boldPending) UH_CHECK(hipMemsetAsync(Bold, 0, sizeof(T) * (size_t)n, st));   // oldBz = 0, :205-206
    boldPending = false;
    if (!zparts) {
      hipLaunchKernelGGL(k_l_norm2<T>, dim3(g), dim3(kLB), 0, st, d_v, n, parts);
      if (int rc = complete(parts)) return rc;
      zparts = parts;
      znp = np;
    }
    hipLaunchKernelGGL(k_l_first<T>, dim3(g), dim3(kLB), 0, st, d_v, n, zparts, znp, V, scal);
    v0Pending = false;
    return 0;
  };
  if (v0Pending) {
    if (!zparts) {
      hipLaunchKernelGGL(k_l_norm2<T>, dim3(g), dim3(kLB), 0, st, d_v, n, (T *)L->partsB.ptr);
      zparts = (const T *)L->partsB.ptr;
      znp = g;
    }
  } else {
    boldPending = true;
    if (int rc = plain_start()) return rc;
  }
  const int checkConvergenceSteps = std::min(L->check_convergence_steps, L->iterationHardLimit - 2);
  hipStreamCaptureStatus captureStatus = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(st, &captureStatus) == hipSuccess && captureStatus != hipStreamCaptureStatusNone;
  std::vector<T> hbuf(2 * cap + 2);
  std::vector<double> dd, ee, zz;
  std::vector<T> yy;
  // deferred checks (see below): iterations [checkFrom, evalAt) wait for their checks until evalAt
  int checkFrom = checkConvergenceSteps;
  int evalAt = std::max(checkConvergenceSteps, std::min(L->deferChecks ? L->lastRunRequiredSteps : 0, checkConvergenceSteps + kLBatch - 1));
  evalAt = std::min(evalAt, std::min(kLDevM - 1, L->iterationHardLimit - 1));
  // the fused iteration (lanczos_fused.hpp): two launches instead of four.  vNextPending: v_(i+1) and hsup_i of the iteration that has
  // just run are still to be made (by the next fused product's prologue, or by k_l_c if that product declines)
  // (|w|^2 partials in a buffer of their own: the fused iteration reads them in the NEXT product's prologue, after a convergence check
  // has used `parts` as its scratch)
  T *partsB = (T *)L->partsB.ptr;
  bool vNextPending = false;
  T *wcur = w;
  for (int i = 0; i < L->iterationHardLimit; ++i) {
    T *vi = V + (size_t)i * n;
    bool fusedDone = false;
    if (fused_usable(L)) {
      if (!L->w2.ptr || L->w2.cap < sizeof(T) * (size_t)n) { if (int e = L->w2.reserve(sizeof(T) * (size_t)n)) return e; }
      const int wideCap = 1 << 16;
      if (!L->partsWide.ptr) { if (int e = L->partsWide.reserve(sizeof(T) * (size_t)wideCap)) return e; }
      T *wnext = wcur == w ? (T *)L->w2.ptr : w;
      int npA = 0;
      const bool first = v0Pending;   // (i == 0: z is "the previous w", |z| "the previous hsup")
      const int rc = fused_call(L, first ? d_v : (vNextPending ? wcur : nullptr), vi, first ? zparts : partsB, first ? znp : np,
                                i > 0 ? hdiag + i - 1 : nullptr, scal, first ? scal : (i > 0 ? hsup + i - 1 : nullptr),
                                (first || vNextPending) ? vi : nullptr, i > 0 ? V + (size_t)(i - 1) * n : nullptr, wnext,
                                (T *)L->partsWide.ptr, wideCap, &npA, n, stream, first);
      if (rc < 0) {
        if (!uammd_hip_last_error()[0]) set_last_error("uammd_lanczos_run: the fused matrix-vector product failed (%d)", rc);
        return rc;
      }
      if (rc == 0) {
        wcur = wnext;
        v0Pending = false;
        hipLaunchKernelGGL(k_l_b<T>, dim3(g), dim3(kLB), 0, st, wcur, (const T *)vi, n, (const T *)L->partsWide.ptr, npA, hdiag + i,
                           partsB, boldPending ? Bold : (T *)nullptr);
        boldPending = false;
        vNextPending = true;
        fusedDone = true;
      }
    }
    if (!fusedDone) {
      if (v0Pending) { if (int rc = plain_start()) return rc; }   // (the product declined at iteration 0)
      if (vNextPending) {   // the previous iteration ran fused and left its last kernel to a product that now declines
        hipLaunchKernelGGL(k_l_c<T>, dim3(g), dim3(kLB), 0, st, (const T *)wcur, n, (const T *)(partsB), np,
                           (const T *)(hdiag + i - 1), (const T *)scal, hsup + i - 1, vi, L->ownsFirstElement);
        vNextPending = false;
      }
      if (int rc = dot(ctx, vi, wcur, n, stream)) {
        if (!uammd_hip_last_error()[0]) set_last_error("uammd_lanczos_run: the matrix-vector callback failed (%d)", rc);
        return rc;
      }
      hipLaunchKernelGGL(k_l_a<T>, dim3(g), dim3(kLB), 0, st, wcur, i > 0 ? (const T *)(V + (size_t)(i - 1) * n) : nullptr,
                         (const T *)vi, n, i > 0 ? (const T *)(hsup + i - 1) : nullptr, parts);
      if (int rc = complete(parts)) return rc;
      hipLaunchKernelGGL(k_l_b<T>, dim3(g), dim3(kLB), 0, st, wcur, (const T *)vi, n, (const T *)parts, np, hdiag + i,
                         partsB);
      if (int rc = complete(partsB)) return rc;
      hipLaunchKernelGGL(k_l_c<T>, dim3(g), dim3(kLB), 0, st, (const T *)wcur, n, (const T *)(partsB), np,
                         (const T *)(hdiag + i), (const T *)scal, hsup + i, V + (size_t)(i + 1) * n, L->ownsFirstElement);
    }
    if (i >= checkConvergenceSteps && capturing) {
      set_last_error("[Lanczos] the solver cannot run inside a stream capture: its convergence checks need the host between launches");
      return -24;
    }
    if (i >= checkConvergenceSteps && !L->reduce && i + 1 <= kLDevM) {
      // Deferred checks.  The reference evaluates the estimate and its error at EVERY iteration from check_convergence_steps on
      // (LanczosAlgorithm.cu:218-232) and stops at the first whose change is within the tolerance; a check is a host round trip and a
      // pass over V — ~34 us here, against ~30 us for the iteration itself at the PSE size, four of them per run.  But nothing a check
      // needs is lost by waiting: H only grows (the m x m problem of iteration i is the leading block of any later one) and V keeps its
      // columns.  So the checks of iterations [checkFrom, i] are evaluated TOGETHER at the iteration the previous run stopped at
      // (lastRunRequiredSteps: consecutive calls almost always need the same number): one round trip, one pass over V for all of them
      // (k_l_estimate_batch), and the run still stops at the FIRST iteration whose error passes — with that iteration's estimate, the
      // reference's result and iteration count.  A run that would have stopped earlier than predicted has done a few iterations for
      // nothing; one that needs more goes on checking every iteration.
      static const bool dumpRecurrence = getenv("UAMMD_LANCZOS_DUMP") != nullptr;   // (read once: this is the per-iteration loop)
      if (dumpRecurrence) {
        std::vector<T> dbg(2 * cap + 2);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(dbg.data(), scal, sizeof(T) * (2 * cap + 1), hipMemcpyDeviceToHost);
        fprintf(stderr, "[lanczos dump] i=%d normz=%g hdiag:", i, (double)dbg[0]);
        for (int k = 0; k <= i; ++k) fprintf(stderr, " %g", (double)dbg[1 + k]);
        fprintf(stderr, " hsup:");
        for (int k = 0; k <= i; ++k) fprintf(stderr, " %g", (double)dbg[1 + cap + k]);
        fprintf(stderr, "\n");
      }
      if (i < evalAt) continue;
      const int K = i - checkFrom + 1, m = i + 1, m0 = checkFrom + 1;   // checks of iterations checkFrom .. i, Krylov sizes m0 .. m
      if (!L->hostStat) {
        UH_CHECK(hipHostMalloc((void **)&L->hostStat, sizeof(double) * kLStatDoubles, hipHostMallocMapped | hipHostMallocCoherent));
        for (int k = 0; k < kLStatDoubles; ++k) L->hostStat[k] = 0.0;
        UH_CHECK(hipHostGetDevicePointer((void **)&L->devStat, (void *)L->hostStat, 0));
      }
      if (int e = L->estimates.reserve(sizeof(T) * (size_t)n * kLBatch)) return e;
      T *est = (T *)L->estimates.ptr;
      L->seq = (L->seq % 1000000u) + 1u;
      const double seq = (double)L->seq;
      volatile double *hs = L->hostStat;
      hs[3] = 0.0;
      hipLaunchKernelGGL(k_l_publish<T>, dim3(1), dim3(64), 0, st, (const T *)hdiag, (const T *)hsup, m, L->devStat, seq);
      if (L->interleaveEarly) {   // the caller's work for the time the host needs to answer (uammd_lanczos_set_interleave_early)
        uammd_interleave_fn fn = L->interleaveEarly;
        L->interleaveEarly = nullptr;
        if (int rc = fn(L->interleaveEarlyCtx, stream)) {
          (void)hipStreamSynchronize(st);
          if (!uammd_hip_last_error()[0]) set_last_error("uammd_lanczos_run: the interleaved callback failed (%d)", rc);
          return rc;
        }
      }
      hipLaunchKernelGGL(k_l_relay_batch<T>, dim3(1), dim3(64), 0, st, L->devStat, K, m, seq, ycoef);
      hipLaunchKernelGGL(k_l_estimate_batch<T>, dim3(g), dim3(kLB), 0, st, (const T *)V, n, m0, K, (const T *)ycoef, (const T *)scal, est,
                         d_Bv, Bold, parts);
      hipLaunchKernelGGL(k_l_error_batch<T>, dim3(1), dim3(64 * 2 * K), 0, st, (const T *)parts, g, K, L->devStat, seq);
      // The host waits for the GPU's sequence number: it spins for the first 5 ms (the answer is usually microseconds away, and one
      // sleep costs more than a whole check), then polls between 100 us sleeps — no core burnt while a long product or other work
      // queued on the stream runs first — bounded by WALL-CLOCK time (a minute), not by a number of reads.  A wait that does expire releases the queued relay kernel and drains the stream before the error goes out, so that
      // nothing of this run is left polling or writing into the status block when the next run resets it.
      auto wait = [&](int slot) -> int {
        const auto t0 = std::chrono::steady_clock::now();
        long spins = 0;
        while (hs[slot] != seq) {
          if ((++spins & 1023) != 0) continue;                       // (the clock is read once per 1024 polls)
          const auto waited = std::chrono::steady_clock::now() - t0;
          if (waited < std::chrono::milliseconds(5)) continue;       // the common case: spin (a sleep is >= 50 us of timer slack per check)
          std::this_thread::sleep_for(std::chrono::microseconds(100));
          if (waited > std::chrono::seconds(60)) {
            hs[4] = seq;                          // let the relay go (it hands out whatever y holds; the run is abandoned)
            (void)hipStreamSynchronize(st);
            set_last_error("[Lanczos] the convergence check did not report within 60 s (slot %d)", slot);
            return -23;
          }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        return 0;
      };
      const auto tA = std::chrono::steady_clock::now();
      if (int e = wait(0)) return e;
      const auto tB = std::chrono::steady_clock::now();
      int infoOf[kLDevM];   // per check: a diagonalisation that fails on a LATER, larger block must not fail a run that an earlier check ends
      for (int k = 0; k < K; ++k) {   // H^(1/2) e1 of the leading mk x mk block, mk = m0 + k
        const int mk = m0 + k;
        dd.assign(mk, 0.0);
        ee.assign(mk, 0.0);
        zz.assign((size_t)mk * mk, 0.0);
        for (int r = 0; r < mk; ++r) { dd[r] = hs[64 + r]; zz[(size_t)r * mk + r] = 1.0; }
        for (int r = 0; r + 1 < mk; ++r) ee[r] = hs[96 + r];
        infoOf[k] = tridiag_ql(dd, ee, zz, mk);
        for (int r = 0; r < kLDevM; ++r) {
          double acc = 0.0;
          if (r < mk)
            for (int j = 0; j < mk; ++j) acc += zz[(size_t)r * mk + j] * std::sqrt(dd[j]) * zz[j];
          hs[128 + kLDevM * k + r] = (double)(T)acc;
        }
      }
      __atomic_thread_fence(__ATOMIC_RELEASE);
      hs[4] = seq;   // the queued estimate kernel goes ahead (also after a failed diagonalisation: it must not be left waiting)
      if (L->interleave) {   // the caller's other work for this stream goes in behind the check (uammd_lanczos_set_interleave)
        uammd_interleave_fn fn = L->interleave;
        L->interleave = nullptr;
        if (int rc = fn(L->interleaveCtx, stream)) {
          (void)wait(1);   // (nothing of this run may be left writing into the status block)
          if (!uammd_hip_last_error()[0]) set_last_error("uammd_lanczos_run: the interleaved callback failed (%d)", rc);
          return rc;
        }
      }
      const auto tC = std::chrono::steady_clock::now();
      if (int e = wait(1)) return e;
      if (getenv("UAMMD_LANCZOS_DEBUG")) {
        const auto tD = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        fprintf(stderr, "[lanczos] i=%d (%d checks) wait scalars %.1f us, host QL %.1f us, wait error %.1f us\n", i, K, us(tA, tB), us(tB, tC), us(tC, tD));
      }
      if (hs[3] != 0.0) { set_last_error("[Lanczos] the estimate kernel gave up waiting for the host's coefficients"); return -23; }
      for (int k = 0; k < K; ++k) {   // in the reference's order: iteration ik's diagonalisation, then its error
        const int ik = checkFrom + k;   // the iteration this check belongs to
        if (infoOf[k]) {
          set_last_error("[Lanczos] Could not diagonalize tridiagonal krylov matrix, steqr failed with code %d", infoOf[k]);
          return -20;
        }
        if (ik == 0) continue;          // (the reference computes no error at iteration 0)
        const T err = (T)hs[16 + k];
        if (std::isnan(err)) {
          set_last_error("[Lanczos] Unknown error (found NaN in result guess) at iteration %d", ik);
          return -21;
        }
        if (err <= tolerance) {
          if (k < K - 1)   // the run stops at an earlier iteration than the last one evaluated: that iteration's estimate is the result
            hipLaunchKernelGGL(k_l_copy<T>, dim3(g), dim3(kLB), 0, st, (const T *)(est + (size_t)k * n), n, d_Bv);
          L->lastRunRequiredSteps = ik;
          if (ik - 2 > L->check_convergence_steps) L->check_convergence_steps += 1;
          else L->check_convergence_steps = std::max(1, L->check_convergence_steps - 2);
          if (iterations) *iterations = ik;
          UH_CHECK(hipGetLastError());
          return 0;
        }
      }
      checkFrom = i + 1;   // from here on every iteration is checked as it comes (Bold holds the last estimate)
      evalAt = i + 1;
    } else if (i >= checkConvergenceSteps) {
      const int m = i + 1;
      UH_CHECK(hipMemcpyAsync(hbuf.data(), scal, sizeof(T) * (2 * cap + 1), hipMemcpyDeviceToHost, st));
      UH_CHECK(hipStreamSynchronize(st));
      dd.assign(m, 0.0);
      ee.assign(m, 0.0);
      zz.assign((size_t)m * m, 0.0);
      for (int k = 0; k < m; ++k) { dd[k] = hbuf[1 + k]; zz[(size_t)k * m + k] = 1.0; }
      for (int k = 0; k + 1 < m; ++k) ee[k] = hbuf[1 + cap + k];
      if (int info = tridiag_ql(dd, ee, zz, m)) {
        set_last_error("[Lanczos] Could not diagonalize tridiagonal krylov matrix, steqr failed with code %d", info);
        return -20;
      }
      yy.assign(m, T(0));
      // H^(1/2) e1 = P * (sqrt(lambda_j) * P[0][j])   (:64-82); a negative eigenvalue gives NaN as in the reference
      for (int r = 0; r < m; ++r) {
        double s = 0.0;
        for (int j = 0; j < m; ++j) s += zz[(size_t)r * m + j] * std::sqrt(dd[j]) * zz[j];
        yy[r] = (T)s;
      }
      UH_CHECK(hipMemcpyAsync(ycoef, yy.data(), sizeof(T) * m, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_l_estimate<T>, dim3(g), dim3(kLB), 0, st, (const T *)V, n, m, (const T *)ycoef,
                         (const T *)scal, d_Bv, Bold, parts);
      if (i > 0) {
        if (int rc = complete(parts)) return rc;
        if (int rc = complete(parts + kLParts)) return rc;
        T hp[2 * kLParts];
        UH_CHECK(hipMemcpyAsync(hp, parts, sizeof(T) * 2 * kLParts, hipMemcpyDeviceToHost, st));
        UH_CHECK(hipStreamSynchronize(st));
        double a = 0.0, b = 0.0;
        for (int k = 0; k < np; ++k) { a += hp[k]; b += hp[kLParts + k]; }
        const T err = (T)std::fabs(std::sqrt(b) / std::sqrt(a));
        if (std::isnan(err)) {
          set_last_error("[Lanczos] Unknown error (found NaN in result guess) at iteration %d", i);
          return -21;
        }
        if (err <= tolerance) {
          // registerRequiredStepsForConverge, :253-262
          L->lastRunRequiredSteps = i;
          if (i - 2 > L->check_convergence_steps) L->check_convergence_steps += 1;
          else L->check_convergence_steps = std::max(1, L->check_convergence_steps - 2);
          if (iterations) *iterations = i;
          UH_CHECK(hipGetLastError());
          return 0;
        }
      }
    }
  }
  UH_CHECK(hipGetLastError());
  set_last_error("[Lanczos] Could not converge");
  return -22;
}


// Solver::runIterations (misc/LanczosAlgorithm/LanczosAlgorithm.cu:183-200): exactly numberIterations Lanczos steps, no convergence test;
// the estimate after the last-but-one step becomes "the previous estimate", the estimate after the last step is the result, and the
// return value (in *residual) is eq. 27's |Bz_m - Bz_(m-1)| / |Bz_(m-1)| between the two (inf / NaN with a single iteration: there is
// no previous estimate — as in the reference).  The plain four-launch iteration; the two estimates through the host's QL.
template <class T, class MatVec>
static int lanczos_run_iterations(LanczosT<T> *L, MatVec dot, void *ctx, T *d_Bv, const T *d_v, int numberIterations, int n, void *stream,
                                  T *residual) {
  if (!L || !dot || !d_Bv || !d_v || n < 1 || numberIterations < 1) { set_last_error("uammd_lanczos_run_iterations: bad arguments"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  const int cap = std::max(L->iterationHardLimit, numberIterations) + 2;
  if (L->capCols < cap || L->capN < n) {
    UH_CHECK(hipStreamSynchronize(st));
    if (int e = L->V.reserve(sizeof(T) * (size_t)n * cap)) return e;
    if (int e = L->w.reserve(sizeof(T) * (size_t)n)) return e;
    if (int e = L->Bold.reserve(sizeof(T) * (size_t)n)) return e;
    if (int e = L->parts.reserve(sizeof(T) * 2 * kLParts * kLBatch)) return e;
    if (int e = L->scal.reserve(sizeof(T) * (2 * cap + 2))) return e;
    if (int e = L->ycoef.reserve(sizeof(T) * std::max(cap, kLBatch * kLDevM))) return e;
    L->capCols = cap;
    L->capN = n;
  }
  if (!L->partsB.ptr) { if (int e = L->partsB.reserve(sizeof(T) * kLParts)) return e; }
  T *V = (T *)L->V.ptr, *w = (T *)L->w.ptr, *Bold = (T *)L->Bold.ptr, *parts = (T *)L->parts.ptr, *partsB = (T *)L->partsB.ptr;
  T *scal = (T *)L->scal.ptr, *ycoef = (T *)L->ycoef.ptr;
  // (scal = [|z|, hdiag[capCols], hsup[capCols]] with the stride of the handle's capacity, which run() may have made larger than cap)
  const int stride = L->capCols;
  T *hdiag = scal + 1, *hsup = scal + 1 + stride;
  const int g = lgrid(n);
  const int np = L->reduce ? 1 : g;
  auto complete = [&](T *p) -> int {
    if (!L->reduce) return 0;
    hipLaunchKernelGGL(k_l_collapse<T>, dim3(1), dim3(kLB), 0, st, p, g);
    return L->reduce(L->reduceCtx, (float *)p, 1, stream);
  };
  L->zparts = nullptr;
  L->znp = 0;
  UH_CHECK(hipMemsetAsync(Bold, 0, sizeof(T) * (size_t)n, st));
  hipLaunchKernelGGL(k_l_norm2<T>, dim3(g), dim3(kLB), 0, st, d_v, n, parts);
  if (int rc = complete(parts)) return rc;
  hipLaunchKernelGGL(k_l_first<T>, dim3(g), dim3(kLB), 0, st, d_v, n, (const T *)parts, np, V, scal);
  std::vector<T> hbuf(2 * (size_t)stride + 2), yy;
  std::vector<double> dd, ee, zz;
  T err = T(0);
  for (int i = 0; i < numberIterations; ++i) {
    T *vi = V + (size_t)i * n;
    if (int rc = dot(ctx, vi, w, n, stream)) {
      if (!uammd_hip_last_error()[0]) set_last_error("uammd_lanczos_run_iterations: the matrix-vector callback failed (%d)", rc);
      return rc;
    }
    hipLaunchKernelGGL(k_l_a<T>, dim3(g), dim3(kLB), 0, st, w, i > 0 ? (const T *)(V + (size_t)(i - 1) * n) : nullptr, (const T *)vi, n,
                       i > 0 ? (const T *)(hsup + i - 1) : nullptr, parts);
    if (int rc = complete(parts)) return rc;
    hipLaunchKernelGGL(k_l_b<T>, dim3(g), dim3(kLB), 0, st, w, (const T *)vi, n, (const T *)parts, np, hdiag + i, partsB, (T *)nullptr);
    if (int rc = complete(partsB)) return rc;
    hipLaunchKernelGGL(k_l_c<T>, dim3(g), dim3(kLB), 0, st, (const T *)w, n, (const T *)partsB, np, (const T *)(hdiag + i), (const T *)scal,
                       hsup + i, V + (size_t)(i + 1) * n, L->ownsFirstElement);
    if (i < numberIterations - 2) continue;
    const int m = i + 1;   // computeCurrentResultEstimation with the Krylov space of this iteration
    UH_CHECK(hipMemcpyAsync(hbuf.data(), scal, sizeof(T) * (2 * (size_t)stride + 1), hipMemcpyDeviceToHost, st));
    UH_CHECK(hipStreamSynchronize(st));
    dd.assign(m, 0.0);
    ee.assign(m, 0.0);
    zz.assign((size_t)m * m, 0.0);
    for (int k = 0; k < m; ++k) { dd[k] = hbuf[1 + k]; zz[(size_t)k * m + k] = 1.0; }
    for (int k = 0; k + 1 < m; ++k) ee[k] = hbuf[1 + stride + k];
    if (int info = tridiag_ql(dd, ee, zz, m)) {
      set_last_error("[Lanczos] Could not diagonalize tridiagonal krylov matrix, steqr failed with code %d", info);
      return -20;
    }
    yy.assign(m, T(0));
    for (int r = 0; r < m; ++r) {
      double sacc = 0.0;
      for (int j = 0; j < m; ++j) sacc += zz[(size_t)r * m + j] * std::sqrt(dd[j]) * zz[j];
      yy[r] = (T)sacc;
    }
    UH_CHECK(hipMemcpyAsync(ycoef, yy.data(), sizeof(T) * m, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_l_estimate<T>, dim3(g), dim3(kLB), 0, st, (const T *)V, n, m, (const T *)ycoef, (const T *)scal, d_Bv, Bold, parts);
    if (i == numberIterations - 1) {   // computeError: against the estimate of the step before (zero when there was none)
      if (int rc = complete(parts)) return rc;
      if (int rc = complete(parts + kLParts)) return rc;
      std::vector<T> hp(2 * kLParts);
      UH_CHECK(hipMemcpyAsync(hp.data(), parts, sizeof(T) * 2 * kLParts, hipMemcpyDeviceToHost, st));
      UH_CHECK(hipStreamSynchronize(st));
      double a = 0.0, b = 0.0;
      for (int k = 0; k < np; ++k) { a += hp[k]; b += hp[kLParts + k]; }
      err = (T)std::fabs(std::sqrt(b) / std::sqrt(a));
      if (std::isnan(err)) {
        set_last_error("[Lanczos] Unknown error (found NaN in result guess) at iteration %d", numberIterations);
        return -21;
      }
    } else {
      UH_CHECK(hipStreamSynchronize(st));   // (yy is reused by the next estimate: its upload must have left the host)
    }
  }
  UH_CHECK(hipGetLastError());
  if (residual) *residual = err;
  return 0;
}


extern "C" {

int uammd_lanczos_run(uammd_lanczos *hh, uammd_matvec_fn dot, void *ctx, float *d_Bv, const float *d_v, float tolerance, int n, void *stream,
                      int *iterations) {
  Lanczos *L = reinterpret_cast<Lanczos *>(hh);
  const int rc = lanczos_run<float>(L, dot, ctx, d_Bv, d_v, tolerance, n, stream, iterations);
  if (L && L->interleaveEarly) {   // a run that never waited: the callbacks' work goes in behind it, in their order
    uammd_interleave_fn fn = L->interleaveEarly;
    L->interleaveEarly = nullptr;
    const int rf = fn(L->interleaveEarlyCtx, stream);
    if (!rc && rf) {
      if (!uammd_hip_last_error()[0]) set_last_error("uammd_lanczos_run: the interleaved callback failed (%d)", rf);
      L->interleave = nullptr;
      return rf;
    }
  }
  if (L && L->interleave) {
    uammd_interleave_fn fn = L->interleave;
    L->interleave = nullptr;
    const int rf = fn(L->interleaveCtx, stream);
    if (!rc && rf) {
      if (!uammd_hip_last_error()[0]) set_last_error("uammd_lanczos_run: the interleaved callback failed (%d)", rf);
      return rf;
    }
  }
  return rc;
}

int uammd_lanczos_run_iterations(uammd_lanczos *h, uammd_matvec_fn dot, void *ctx, float *d_Bv, const float *d_v, int numberIterations, int n,
                                 void *stream, float *residual) {
  return lanczos_run_iterations<float>(reinterpret_cast<Lanczos *>(h), dot, ctx, d_Bv, d_v, numberIterations, n, stream, residual);
}
int uammd_lanczos_run_iterations_f64(uammd_lanczos_f64 *h, uammd_matvec_fn_f64 dot, void *ctx, double *d_Bv, const double *d_v,
                                     int numberIterations, int n, void *stream, double *residual) {
  return lanczos_run_iterations<double>(reinterpret_cast<LanczosT<double> *>(h), dot, ctx, d_Bv, d_v, numberIterations, n, stream, residual);
}

// ---- DOUBLE_PRECISION build (global/defines.h:9-11): lanczos::Solver with real = double, e.g. the reference's own test
// (test/misc/lanczos/test_lanczos.cu, tolerance 1e-7) on the GPU ----
int uammd_lanczos_create_f64(uammd_lanczos_f64 **out) {
  if (!out) { set_last_error("uammd_lanczos_create_f64: null output"); return -1; }
  *out = reinterpret_cast<uammd_lanczos_f64 *>(new LanczosT<double>());
  return 0;
}
int uammd_lanczos_destroy_f64(uammd_lanczos_f64 *h) {
  delete reinterpret_cast<LanczosT<double> *>(h);
  return 0;
}
int uammd_lanczos_set_iteration_hard_limit_f64(uammd_lanczos_f64 *h, int limit) {
  if (!h || limit < 1) { set_last_error("uammd_lanczos_set_iteration_hard_limit_f64: bad arguments"); return -1; }
  reinterpret_cast<LanczosT<double> *>(h)->iterationHardLimit = limit;
  return 0;
}
int uammd_lanczos_get_last_run_required_steps_f64(uammd_lanczos_f64 *h, int *steps) {
  if (!h || !steps) { set_last_error("uammd_lanczos_get_last_run_required_steps_f64: null argument"); return -1; }
  *steps = reinterpret_cast<LanczosT<double> *>(h)->lastRunRequiredSteps;
  return 0;
}
int uammd_lanczos_run_f64(uammd_lanczos_f64 *hh, uammd_matvec_fn_f64 dot, void *ctx, double *d_Bv, const double *d_v, double tolerance, int n,
                          void *stream, int *iterations) {
  return lanczos_run<double>(reinterpret_cast<LanczosT<double> *>(hh), dot, ctx, d_Bv, d_v, tolerance, n, stream, iterations);
}

}  // extern "C"
