// Micro-benchmark: VALU issue cost (cycles per wave64 instruction per SIMD) of the instruction
// classes the LJ traversal uses, on gfx950.  Build: hipcc --offload-arch=gfx950 -O2 valu_ubench.hip
// Each kernel runs N dependent-free instruction streams (8 independent chains) per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP 4096

template <int OP> __global__ void __launch_bounds__(256) k(float *out, float a, float b, int ia) {
  float x0 = threadIdx.x * 1e-3f + a, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
  int i0 = threadIdx.x + ia, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
  for (int r = 0; r < REP; ++r) {
    if (OP == 0) {  // v_fma_f32
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if (OP == 1) {  // v_pk_fma_f32 (4 instr => count 4)
      asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                   "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
    } else if (OP == 2) {  // v_mul_f32
      asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                   "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
    } else if (OP == 3) {  // v_add_u32
      asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                   "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(ia));
    } else if (OP == 4) {  // v_rcp_f32
      asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                   "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    } else if (OP == 5) {  // v_cmp_lt_f32 -> sgpr pair + v_cndmask
      asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n"
                   "v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
    } else if (OP == 6) {  // v_div_scale + v_div_fmas + v_div_fixup (3 per group, 2 groups + 2 mov)
      asm volatile("v_div_scale_f32 %0, vcc, %0, %8, %0\n v_div_fmas_f32 %1, %1, %8, %9\n v_div_fixup_f32 %2, %2, %8, %9\n v_div_scale_f32 %3, vcc, %3, %8, %3\n"
                   "v_div_fmas_f32 %4, %4, %8, %9\n v_div_fixup_f32 %5, %5, %8, %9\n v_div_scale_f32 %6, vcc, %6, %8, %6\n v_div_fmas_f32 %7, %7, %8, %9\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
    } else if (OP == 7) {  // v_sub_f32
      asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n"
                   "v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
    } else if (OP == 8) {  // v_pk_add_f32 / v_pk_mul_f32 mix
      asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %5\n v_pk_add_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %5\n"
                   "v_pk_add_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %5\n v_pk_add_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %5\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
    } else if (OP == 9) {  // v_floor_f32
      asm volatile("v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3\n"
                   "v_floor_f32 %4, %4\n v_floor_f32 %5, %5\n v_floor_f32 %6, %6\n v_floor_f32 %7, %7\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    } else if (OP == 10) {  // v_mov_b32
      asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                   "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    } else if (OP == 11) {  // v_cmp_lt_f32 to an SGPR pair (VOP3) only
      asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[22:23], %1, %8\n v_cmp_lt_f32 s[24:25], %2, %8\n v_cmp_lt_f32 s[26:27], %3, %8\n"
                   "v_cmp_lt_f32 s[20:21], %4, %8\n v_cmp_lt_f32 s[22:23], %5, %8\n v_cmp_lt_f32 s[24:25], %6, %8\n v_cmp_lt_f32 s[26:27], %7, %8\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a)
                   : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    } else if (OP == 13) {  // v_pk_fma_f16 / v_pk_add_f16 / v_pk_mul_f16 (the half-precision prefilter's arithmetic)
      asm volatile("v_pk_fma_f16 %0, %0, %4, %4\n v_pk_add_f16 %1, %1, %4\n v_pk_mul_f16 %2, %2, %4\n v_pk_fma_f16 %3, %3, %4, %4\n"
                   "v_pk_fma_f16 %0, %0, %4, %4\n v_pk_add_f16 %1, %1, %4\n v_pk_mul_f16 %2, %2, %4\n v_pk_fma_f16 %3, %3, %4, %4\n"
                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(ia));
    } else if (OP == 14) {  // v_cmp_nge_f16 to an SGPR pair, low and high half (SDWA)
      asm volatile("v_cmp_nge_f16 s[20:21], %0, %4\n v_cmp_nge_f16_sdwa s[22:23], %1, %4 src0_sel:WORD_1 src1_sel:DWORD\n"
                   "v_cmp_nge_f16 s[24:25], %2, %4\n v_cmp_nge_f16_sdwa s[26:27], %3, %4 src0_sel:WORD_1 src1_sel:DWORD\n"
                   "v_cmp_nge_f16 s[20:21], %0, %4\n v_cmp_nge_f16_sdwa s[22:23], %1, %4 src0_sel:WORD_1 src1_sel:DWORD\n"
                   "v_cmp_nge_f16 s[24:25], %2, %4\n v_cmp_nge_f16_sdwa s[26:27], %3, %4 src0_sel:WORD_1 src1_sel:DWORD\n"
                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(ia) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    } else if (OP == 15) {  // v_and_or_b32 / v_add_u32 (the ring append)
      asm volatile("v_and_or_b32 %0, %0, %4, %4\n v_add_u32 %1, %1, %4\n v_and_or_b32 %2, %2, %4, %4\n v_add_u32 %3, %3, %4\n"
                   "v_and_or_b32 %0, %0, %4, %4\n v_add_u32 %1, %1, %4\n v_and_or_b32 %2, %2, %4, %4\n v_add_u32 %3, %3, %4\n"
                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(ia));
    } else if (OP == 12) {  // v_lshl_add_u32 / v_min_i32
      asm volatile("v_lshl_add_u32 %0, %0, 4, %4\n v_min_i32 %1, %1, %4\n v_lshl_add_u32 %2, %2, 4, %4\n v_min_i32 %3, %3, %4\n"
                   "v_lshl_add_u32 %0, %0, 4, %4\n v_min_i32 %1, %1, %4\n v_lshl_add_u32 %2, %2, 4, %4\n v_min_i32 %3, %3, %4\n"
                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(ia));
    }
  }
  p0 += p1 + p2 + p3;
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + i0 + i1 + i2 + i3 + p0.x + p0.y;
}

template <int OP> double run(float *d, int wavesPerSimd) {
  int blocks = 256 * wavesPerSimd;  // 256 CUs x (wavesPerSimd) blocks of 4 waves => wavesPerSimd waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, 3);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, 3);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // instructions per SIMD = wavesPerSimd * REP * 8 ; ns per instr per SIMD
  return ms * 1e6 / (double(wavesPerSimd) * REP * 8);
}

int main() {
  float *d; hipMalloc(&d, 256 * 256 * 16 * 4);
  const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_mul_f32", "v_add_u32", "v_rcp_f32", "v_cmp+v_cndmask(avg)", "v_div_scale/fmas/fixup(avg)", "v_sub_f32", "v_pk_add/mul_f32", "v_floor_f32", "v_mov_b32", "v_cmp->sgpr", "v_lshl_add/v_min_i32", "v_pk_fma/add/mul_f16", "v_cmp_f16 lo / hi(sdwa)", "v_and_or_b32/v_add_u32"};
  for (int w : {1, 4}) {
    printf("waves/SIMD=%d  (ns per wave64 instruction per SIMD; x clock GHz = cycles)\n", w);
    double t[16];
    t[0] = run<0>(d, w); t[1] = run<1>(d, w); t[2] = run<2>(d, w); t[3] = run<3>(d, w); t[4] = run<4>(d, w); t[5] = run<5>(d, w);
    t[6] = run<6>(d, w); t[7] = run<7>(d, w); t[8] = run<8>(d, w); t[9] = run<9>(d, w); t[10] = run<10>(d, w); t[11] = run<11>(d, w); t[12] = run<12>(d, w); t[13] = run<13>(d, w); t[14] = run<14>(d, w); t[15] = run<15>(d, w);
    for (int i = 0; i < 16; ++i) printf("  %-28s %.3f ns  (~%.2f cyc @2.4GHz)\n", names[i], t[i], t[i] * 2.4);
  }
  return 0;
}
