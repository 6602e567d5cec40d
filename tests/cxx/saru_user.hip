// Saru from third_party/saruprng.cuh in a USER kernel and on the host: the integer streams of the three constructors for the seeds given on
// the command line (tests/test_cxx_interface.py compares them with tests/golden/saru_u32.npz), and the float / Gaussian draws against the
// library's own (uammd_pse_near_noise draws make_real3(gf(0, 1), gf(0, 1).x) * variance from Saru(i, seed, seed2): NearField.cuh:218-228).
#include "uammd.cuh"
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_streams(unsigned int a, unsigned int b, unsigned int c, unsigned int *out) {
  Saru r1(a), r2(a, b), r3(a, b, c);
  for (int k = 0; k < 16; ++k) { out[k] = r1.u32(); out[16 + k] = r2.u32(); out[32 + k] = r3.u32(); }
}
__global__ void k_gauss(int n, unsigned int s1, unsigned int s2, float *out3) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Saru rng(i, s1, s2);
  const float2 a = rng.gf(0.f, 1.f), b = rng.gf(0.f, 1.f);
  out3[3 * i] = a.x; out3[3 * i + 1] = a.y; out3[3 * i + 2] = b.x;
}
int main(int argc, char **argv) {
  const unsigned int a = argc > 1 ? strtoul(argv[1], 0, 0) : 1u, b = argc > 2 ? strtoul(argv[2], 0, 0) : 2u, c = argc > 3 ? strtoul(argv[3], 0, 0) : 3u;
  unsigned int *d;
  (void)hipMalloc(&d, 48 * sizeof(unsigned int));
  hipLaunchKernelGGL(k_streams, dim3(1), dim3(1), 0, 0, a, b, c, d);
  unsigned int h[48];
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  Saru r1(a), r2(a, b), r3(a, b, c);   // the same on the host
  for (int k = 0; k < 16; ++k) {
    if (h[k] != r1.u32() || h[16 + k] != r2.u32() || h[32 + k] != r3.u32()) { std::printf("host / device streams differ\n"); return 1; }
  }
  for (int s = 0; s < 3; ++s) { std::printf("u32_%d", s + 1); for (int k = 0; k < 16; ++k) std::printf(" %u", h[16 * s + k]); std::printf("\n"); }
  const int n = 1000;
  float *g;
  (void)hipMalloc(&g, 3 * n * sizeof(float));
  hipLaunchKernelGGL(k_gauss, dim3((n + 127) / 128), dim3(128), 0, 0, n, 77u, 4242u, g);
  std::vector<float> hg(3 * n);
  (void)hipMemcpy(hg.data(), g, sizeof(float) * 3 * n, hipMemcpyDeviceToHost);
  std::printf("gauss");
  for (int k = 0; k < 12; ++k) std::printf(" %.9g", hg[k]);
  std::printf("\n");
  double m = 0, v = 0;
  for (float x : hg) { m += x; v += (double)x * x; }
  std::printf("moments %.6f %.6f\n", m / (3 * n), v / (3 * n));
  return 0;
}
