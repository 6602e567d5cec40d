// Probe (gfx950): semantics and issue cost of the f32 -> fp6 / fp4 packing conversions as SIGN-BIT collectors for the LJ tile
// scan (32 v_alignbit per 64 x 32 distance tests today).  Build: hipcc --offload-arch=gfx950 -O2 cvt_probe.hip -o _build/cvt_probe
//   1. where does the sign of source element n land in the six result dwords of v_cvt_scalef32_2xpk16_fp6_f32, and is it kept for
//      zeros, denormals, tiny, huge and infinite inputs (scale 1.0);
//   2. ns per wave64 instruction per SIMD of that conversion, of v_cvt_scalef32_pk_fp4_f32, v_bfi_b32, v_alignbit_b32,
//      v_permlane32_swap_b32, v_bcnt_u32_b32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));

__global__ void k_sem(const float *in, unsigned *out, int fmt) {  // one wave; lane l: 32 inputs at in[32 l ..]
  v16f a, b;
  for (int i = 0; i < 16; ++i) { a[i] = in[32 * threadIdx.x + i]; b[i] = in[32 * threadIdx.x + 16 + i]; }
  v6u r;
  if (fmt == 0) r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, 1.0f);
  else r = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(a, b, 1.0f);
  for (int i = 0; i < 6; ++i) out[6 * threadIdx.x + i] = r[i];
}

#define REP 2048
template <int OP> __global__ void __launch_bounds__(256) k_rate(unsigned *out, float fa, unsigned ia) {
  v16f a, b;
  for (int i = 0; i < 16; ++i) { a[i] = fa + i + threadIdx.x; b[i] = fa - i - threadIdx.x; }
  v6u r0 = {0, 0, 0, 0, 0, 0}, r1 = r0, r2 = r0, r3 = r0;
  unsigned i0 = threadIdx.x + ia, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
  for (int r = 0; r < REP; ++r) {
    if (OP == 0) {  // 4 conversions of 32 values each
      asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %4, %5, 1.0\n v_cvt_scalef32_2xpk16_fp6_f32 %1, %5, %4, 1.0\n"
                   "v_cvt_scalef32_2xpk16_fp6_f32 %2, %4, %5, 1.0\n v_cvt_scalef32_2xpk16_fp6_f32 %3, %5, %4, 1.0\n"
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b));
    } else if (OP == 1) {  // 8 fp4 pair conversions
      asm volatile("v_cvt_scalef32_pk_fp4_f32 %0, %8, %9, 1.0\n v_cvt_scalef32_pk_fp4_f32 %1, %8, %9, 1.0 op_sel:[0,0,1,0]\n"
                   "v_cvt_scalef32_pk_fp4_f32 %2, %8, %9, 1.0\n v_cvt_scalef32_pk_fp4_f32 %3, %8, %9, 1.0 op_sel:[0,0,1,0]\n"
                   "v_cvt_scalef32_pk_fp4_f32 %4, %8, %9, 1.0\n v_cvt_scalef32_pk_fp4_f32 %5, %8, %9, 1.0 op_sel:[0,0,1,0]\n"
                   "v_cvt_scalef32_pk_fp4_f32 %6, %8, %9, 1.0\n v_cvt_scalef32_pk_fp4_f32 %7, %8, %9, 1.0 op_sel:[0,0,1,0]\n"
                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(a[0]), "v"(b[0]));
    } else if (OP == 2) {  // v_bfi_b32
      asm volatile("v_bfi_b32 %0, %8, %0, %1\n v_bfi_b32 %1, %8, %1, %2\n v_bfi_b32 %2, %8, %2, %3\n v_bfi_b32 %3, %8, %3, %4\n"
                   "v_bfi_b32 %4, %8, %4, %5\n v_bfi_b32 %5, %8, %5, %6\n v_bfi_b32 %6, %8, %6, %7\n v_bfi_b32 %7, %8, %7, %0\n"
                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(ia));
    } else if (OP == 3) {  // v_alignbit_b32
      asm volatile("v_alignbit_b32 %0, %0, %8, 31\n v_alignbit_b32 %1, %1, %8, 31\n v_alignbit_b32 %2, %2, %8, 31\n v_alignbit_b32 %3, %3, %8, 31\n"
                   "v_alignbit_b32 %4, %4, %8, 31\n v_alignbit_b32 %5, %5, %8, 31\n v_alignbit_b32 %6, %6, %8, 31\n v_alignbit_b32 %7, %7, %8, 31\n"
                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(ia));
    } else if (OP == 4) {  // v_permlane32_swap_b32 (4 swaps of two registers each)
      asm volatile("s_nop 1\n v_permlane32_swap_b32 %0, %1\n s_nop 1\n v_permlane32_swap_b32 %2, %3\n s_nop 1\n v_permlane32_swap_b32 %4, %5\n s_nop 1\n v_permlane32_swap_b32 %6, %7\n"
                   "s_nop 1\n v_permlane32_swap_b32 %0, %1\n s_nop 1\n v_permlane32_swap_b32 %2, %3\n s_nop 1\n v_permlane32_swap_b32 %4, %5\n s_nop 1\n v_permlane32_swap_b32 %6, %7\n"
                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
    } else if (OP == 5) {  // v_bcnt_u32_b32
      asm volatile("v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %8, %1\n v_bcnt_u32_b32 %2, %8, %2\n v_bcnt_u32_b32 %3, %8, %3\n"
                   "v_bcnt_u32_b32 %4, %8, %4\n v_bcnt_u32_b32 %5, %8, %5\n v_bcnt_u32_b32 %6, %8, %6\n v_bcnt_u32_b32 %7, %8, %7\n"
                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(ia));
    } else if (OP == 6) {  // v_fma_f32 (the yardstick)
      asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                   "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(fa));
    }
  }
  unsigned s = i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7;
  for (int i = 0; i < 6; ++i) s += r0[i] + r1[i] + r2[i] + r3[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP> double rate(unsigned *d, int wavesPerSimd, int perIter) {
  const int blocks = 256 * wavesPerSimd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.5f, 3u);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.5f, 3u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / (double(wavesPerSimd) * REP * perIter);
}

int main() {
  // ---- semantics ----
  std::vector<float> in(64 * 32, 1.0f);
  // lanes 0..31: element n = lane negative (-1.0), everything else +1.0 -> the bit that moves is the sign of element n
  for (int l = 0; l < 32; ++l) in[32 * l + l] = -1.0f;
  // lane 32..: value classes in element 0 (negative) with the rest positive; lane 48..: the same values positive with the rest negative
  const float vals[] = {0.0f, 1e-45f, 1e-39f, 1e-30f, 1e-10f, 1e-3f, 0.06f, 0.3f, 1.0f, 7.0f, 100.0f, 1e10f, 3e38f, INFINITY};
  const int nv = sizeof(vals) / sizeof(vals[0]);
  for (int v = 0; v < nv && v < 16; ++v) {
    in[32 * (32 + v) + 0] = -vals[v];
    for (int i = 0; i < 32; ++i) in[32 * (48 + v) + i] = i == 0 ? vals[v] : -1.0f;
  }
  float *din; unsigned *dout;
  hipMalloc(&din, in.size() * 4); hipMalloc(&dout, 64 * 6 * 4);
  hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
  for (int fmt = 0; fmt < 2; ++fmt) {
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, din, dout, fmt);
    std::vector<unsigned> out(64 * 6);
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    printf("== %s: sign position of source element n (dword, bit); expected (6 n + 5) ==\n", fmt == 0 ? "fp6" : "bf6");
    // reference: all +1.0
    bool ok = true;
    for (int n = 0; n < 32; ++n) {
      // bits that are set in lane n and whose position is 6 k + 5 for some k
      int found = -1, cnt = 0;
      for (int k = 0; k < 32; ++k) {
        const int pos = 6 * k + 5;
        if ((out[6 * n + pos / 32] >> (pos % 32)) & 1u) { found = k; ++cnt; }
      }
      if (cnt != 1 || found != n) { ok = false; printf("  element %d: %d sign bits set, last at value index %d\n", n, cnt, found); }
    }
    printf("  layout %s\n", ok ? "as expected: sign of element n (a[0..15], b[0..15]) at bit 6 n + 5 of the 192-bit result" : "DIFFERENT");
    for (int v = 0; v < nv && v < 16; ++v) {
      const unsigned neg = (out[6 * (32 + v)] >> 5) & 1u, pos = (out[6 * (48 + v)] >> 5) & 1u;
      int others = 0;
      for (int k = 1; k < 32; ++k) { const int p = 6 * k + 5; others += (out[6 * (48 + v) + p / 32] >> (p % 32)) & 1u; }
      printf("  |x| = %-12g  sign(-x) = %u  sign(+x) = %u  (31 other negatives seen: %d)   code(-x) = 0x%02x\n", vals[v], neg, pos, others, out[6 * (32 + v)] & 63u);
    }
  }
  // ---- rates ----
  unsigned *d; hipMalloc(&d, 256 * 256 * 16 * 4);
  for (int w : {1, 4}) {
    printf("waves/SIMD = %d: ns per wave64 instruction per SIMD\n", w);
    printf("  v_cvt_scalef32_2xpk16_fp6_f32  %.3f\n", rate<0>(d, w, 4));
    printf("  v_cvt_scalef32_pk_fp4_f32      %.3f\n", rate<1>(d, w, 8));
    printf("  v_bfi_b32                      %.3f\n", rate<2>(d, w, 8));
    printf("  v_alignbit_b32                 %.3f\n", rate<3>(d, w, 8));
    printf("  v_permlane32_swap_b32 (+nop 1) %.3f\n", rate<4>(d, w, 8));
    printf("  v_bcnt_u32_b32                 %.3f\n", rate<5>(d, w, 8));
    printf("  v_fma_f32                      %.3f\n", rate<6>(d, w, 8));
  }
  return 0;
}
