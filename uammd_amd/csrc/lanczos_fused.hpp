// Internal to the library (C++ linkage, not part of the C ABI): a matrix-vector product that also runs the two kernels of the Lanczos
// recurrence that sit next to it (LanczosAlgorithm.cu:128-172), for products the library launches itself (the PSE near field).
//   unfused iteration i:  w = M v_i | k_l_a: w -= hsup_(i-1) v_(i-1), partials of w . v_i | k_l_b: hdiag_i, w -= hdiag_i v_i, partials
//                         of |w|^2 | k_l_c: hsup_i = |w| (breakdown guard), v_(i+1) = w / hsup_i          — four launches
//   fused:                product_i { prologue = k_l_c of iteration i - 1 (every workgroup sums the |w|^2 partials in the same order,
//                         reads v_i[j] as wPrev[j] / hsup_(i-1), writes its own rows of v_i and, one workgroup, hsup_(i-1));
//                         the pair sums; epilogue = k_l_a (its rows of w, one partial of w . v_i per workgroup) } | k_l_b   — two launches
//   the start of a run:   memset(Bold) | k_l_norm2 | k_l_first | ...  becomes  [k_l_norm2 unless the caller has the partials] | product_0 with
//                         k_l_first as its prologue | k_l_b, which also zeroes Bold
// v_i, hsup, hdiag and w hold the SAME values as in the unfused run except that the partials of w . v_i are cut by the product's
// workgroups instead of k_l_a's blocks (another summation order: rounding level).  wPrev and wOut are two buffers (a workgroup reads
// other rows' wPrev while their owners write wOut).
#pragma once
#include "device_common.hpp"
#include "../../include/uammd_hip.h"

namespace uammd_hip {

struct LanczosFusedArgs {
  // prologue.  wPrev == nullptr: iteration 0, v_0 = viDirect
  const float *wPrev, *viDirect;
  const float *partsB;   // partials of |wPrev|^2
  int npB;
  const float *hdiagPrev, *normz;
  float *hsupPrev;       // out (one workgroup)
  float *viOut;          // out: v_i (own rows); nullptr at iteration 0
  bool ownsFirstElement;
  // epilogue
  const float *vPrev;    // v_(i-1), nullptr at iteration 0
  float *wOut;           // out
  float *partsA;         // out: one partial of w . v_i per workgroup
  int partsACap;
  int npA;               // out: how many partials were written
  // iteration 0 with wPrev = the right-hand side z itself and partsB = the partials of |z|^2: the prologue is then k_l_first (v_0 = z / |z|,
  // |z| written to hsupPrev = scal[0]; no breakdown guard)
  bool first;
};
// returns 0 (done), 1 (this product cannot run fused now: the caller falls back to the four-launch iteration), < 0 error
typedef int (*lanczos_fused_fn)(void *ctx, LanczosFusedArgs *a, int n, void *stream);
int lanczos_set_fused(::uammd_lanczos *h, lanczos_fused_fn fn, void *ctx);
// one-shot, for the NEXT run of a solver with a fused product: the partials of |z|^2 (np <= 256 of them, device memory) of the vector the
// run will be called with — made by the kernel that made z — instead of a k_l_norm2 launch
int lanczos_set_znorm_parts(::uammd_lanczos *h, const float *parts, int np);

// the sum of nparts partials, the same order (and bits) in every workgroup of 256 threads: lanczos.hip's sum_parts
UH_D float fused_sum_parts(const float *__restrict__ parts, int nparts, float *sh /* >= 9 floats */) {
  float x = 0.f;
  for (int k = threadIdx.x; k < nparts; k += 256) x += parts[k];
  x = wave_sum_to_last(x);
  if ((threadIdx.x & 63) == 63) sh[threadIdx.x >> 6] = x;
  __syncthreads();
  const float t = (sh[0] + sh[2]) + (sh[1] + sh[3]);
  __syncthreads();
  return t;
}

}  // namespace uammd_hip
