// Micro-benchmark: cost of one wave-wide load instruction from L1-resident memory vs from LDS, by width and by address pattern
// (all lanes one address / 5 groups of lanes, one address each (the LJ walk's pattern) / 64 consecutive elements).
// Build: hipcc --offload-arch=gfx950 -O3 tools/vmem_ubench.hip -o /tmp/vmem_ubench ; prints clocks per instruction per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int W> struct Vec;
template <> struct Vec<1> { using T = float; };
template <> struct Vec<2> { using T = float2; };
template <> struct Vec<3> { using T = float3; };
template <> struct Vec<4> { using T = float4; };
__device__ float sum(float v) { return v; }
__device__ float sum(float2 v) { return v.x + v.y; }
__device__ float sum(float3 v) { return v.x + v.y + v.z; }
__device__ float sum(float4 v) { return v.x + v.y + v.z + v.w; }

template <int W, bool LDS> __global__ void __launch_bounds__(256) k(const float4 *g, int pattern, int iters, float *out) {
  __shared__ float4 sh[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) sh[i] = g[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  int idx = pattern == 0 ? 0 : pattern == 1 ? (lane / 13) * 17 : lane;
  idx += (threadIdx.x >> 6) * 100;
  float acc = 0.f;
  using T = typename Vec<W>::T;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = (idx + u * 3 + it * 5) & 1023;
      T v;
      if (LDS) v = *reinterpret_cast<const T *>(&sh[e]);
      else v = *reinterpret_cast<const T *>(&g[e]);
      acc += sum(v);
    }
  }
  if (acc == 12345.f) out[0] = acc;
}

template <int W, bool LDS> void run(const float4 *g, float *out, int pattern) {
  const int iters = 2000, blocks = 256 * 8;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<W, LDS><<<blocks, 256>>>(g, pattern, 10, out);
  hipEventRecord(a);
  k<W, LDS><<<blocks, 256>>>(g, pattern, iters, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double instr_per_cu = (double)blocks * 4 * iters * 8 / 256.0;
  printf("%s width %d pattern %d: %.3f ms  -> %.1f clk/instr/CU @2.4GHz\n", LDS ? "LDS " : "GLOB", W, pattern, ms,
         ms * 1e-3 * 2.4e9 / instr_per_cu);
}

int main() {
  float4 *g; float *out;
  hipMalloc(&g, 1024 * 16); hipMalloc(&out, 16);
  std::vector<float4> h(1024, make_float4(1, 2, 3, 4));
  hipMemcpy(g, h.data(), 1024 * 16, hipMemcpyHostToDevice);
  for (int p = 0; p < 3; ++p) {
    run<1, false>(g, out, p); run<2, false>(g, out, p); run<3, false>(g, out, p); run<4, false>(g, out, p);
    run<1, true>(g, out, p); run<2, true>(g, out, p); run<3, true>(g, out, p); run<4, true>(g, out, p);
  }
  return 0;
}
