// Triply periodic electrostatics by spectral Ewald splitting — the Poisson Interactor of the reference, a second consumer
// of the spread / FFT / gather engine of Path B.
//
// Reference behaviour (Interactor/SpectralEwaldPoisson.cu, .cuh):
//   ctor      grid from the Gaussian far-field width, window support, near cut-off by a 1e-3 gw march, two tables   .cu:71-160
//   farField  spread q -> R2C -> (E, phi)(k) = (-i k, 1) q(k) / (eps k^2 Ncells) -> 4 x C2R -> gather real4,
//             force += q E, energy += q phi (always both)                                                          .cu:332-360, :410-559
//   nearField CellList at the near cut-off, tabulated G(r^2) and G'(r), self pair included                        .cu:222-329, :362-408
// HIP design: the charge grid is transformed in place; the four outputs are written as component PLANES by the k-space
// kernel (one batched in-place C2R), then interleaved once into a float4 grid so that the gather issues one 16-byte
// request per node instead of four (the gather is bound by cache-line requests on gfx950).  The near field
// runs one thread per Morton-sorted particle over a packed (x, y, z, q) array in the reference's neighbour order.
#include "celllist.hpp"
#include "ibm.hpp"

#include <rocfft/rocfft.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

namespace uammd_hip {

int rocfft_setup_once();  // fcm.hip

#define UH_ROCFFT(expr)                                                                      \
  do {                                                                                       \
    rocfft_status s_ = (expr);                                                               \
    if (s_ != rocfft_status_success) {                                                       \
      set_last_error("%s failed with rocfft_status %d (%s:%d)", #expr, (int)s_, __FILE__, __LINE__); \
      return -10 - (int)s_;                                                                  \
    }                                                                                        \
  } while (0)

struct Poisson {
  uammd_poisson_parameters par{};
  float L[3] = {0, 0, 0};
  int cells[3] = {0, 0, 0};
  GridT<float> grid{};
  uammd_ibm_kernel kernel{};
  IBMKernelDev kern{};
  int nxpad = 0;
  size_t planeReal = 0, planeCplx = 0;
  float cutoff = 0.f;
  int ntable = 0;
  DeviceBuffer tableField, tablePotential, gridQ, planes, inter, packed, work;
  // tile-owned spread: tile dimensions (divisors of the grid, >= support-1), particles binned by the tile of their
  // stencil origin
  struct TileSet {
    bool on = false;
    int3 tile{0, 0, 0}, ntiles{0, 0, 0};
    int waves = 0, row = 0, plane = 0;  // waves per workgroup, padded LDS strides (in elements)
    DeviceBuffer count, start, rank, pos, idx;
    int numTiles() const { return ntiles.x * ntiles.y * ntiles.z; }
  };
  TileSet spreadTiles;
  bool forceAtomicSpread = false;
  CellList cl;
  rocfft_plan fwd = nullptr, inv = nullptr;
  rocfft_execution_info info = nullptr;
  ~Poisson() {
    if (fwd) rocfft_plan_destroy(fwd);
    if (inv) rocfft_plan_destroy(inv);
    if (info) rocfft_execution_info_destroy(info);
  }
};

// ---- host: closed forms and heuristics (double arithmetic on float arguments, as the reference's host code) ----------
static float greens(float r2, float gw, float split, float epsilon) {  // .cu:15-38
  double G = 0;
  if (r2 > gw * gw * gw * gw) {
    const double r = sqrtf(r2);
    const float farw = sqrtf(4 * gw * gw + 1 / (split * split));
    G = (1.0 / (4.0 * M_PI * epsilon * r) * (erf(r / (2 * gw)) - erf(r / farw)));
  } else {
    const double pi32 = pow(M_PI, 1.5);
    const double gw2 = gw * gw;
    const double invsp2 = 1.0 / (split * split);
    const double selfterm = 1.0 / (4 * pi32 * gw) - 1.0 / (2 * pi32 * sqrt(4 * gw2 + invsp2));
    const double r2term = 1.0 / (6.0 * pi32 * pow(4.0 * gw2 + invsp2, 1.5)) - 1.0 / (48.0 * pi32 * gw2 * gw);
    const double r4term = 1.0 / (640.0 * pi32 * gw2 * gw2 * gw) - 1.0 / (20.0 * pi32 * pow(4 * gw2 + invsp2, 2.5));
    G = 1.0 / epsilon * (selfterm + r2 * r2term + r2 * r2 * r4term);
  }
  return (float)G;
}
static float greens_field(float r, float gw, float split, float epsilon) {  // .cu:40-62
  const double r2 = r * r;
  const double gw2 = gw * gw;
  const double newgw = sqrt(gw2 + 1 / (4.0 * split * split));
  const double newgw2 = newgw * newgw;
  double fmod = 0;
  if (r2 > gw * gw * gw * gw) {
    const double invrterm = exp(-0.25 * r2 / newgw2) / sqrt(M_PI * newgw2) - exp(-0.25 * r2 / gw2) / sqrt(M_PI * gw2);
    const double invr2term = erf(0.5 * r / newgw) - erf(0.5 * r / gw);
    fmod += 1 / (4 * M_PI) * (invrterm / r - invr2term / r2);
  } else if (r2 > 0) {
    const double pi32 = pow(M_PI, 1.5);
    const double rterm = 1 / (24 * pi32) * (1.0 / (gw2 * gw) - 1 / (newgw2 * newgw));
    const double r3term = 1 / (160 * pi32) * (1.0 / (newgw2 * newgw2 * newgw) - 1.0 / (gw2 * gw2 * gw));
    fmod += r * rterm + r2 * r * r3term;
  }
  return (float)(fmod / epsilon);
}
static double far_width(float gw, float split) {
  double w = gw;
  if (split > 0) w = sqrt(gw * gw + 1.0 / (4.0 * split * split));
  return w;
}
// nextFFTWiseSize3D (utils/Grid.cuh:142-213): smallest even 2^a 3^b 5^c 7^d 11^e >= n with c<=5, d<=4, e<=3
static int next_fft_wise(int n) {
  static const int primes[5] = {2, 3, 5, 7, 11}, maxExp[5] = {64, 64, 5, 4, 3};
  for (int c = std::max(n, 1);; ++c) {
    if (c % 2) continue;
    int m = c;
    bool ok = true;
    for (int p = 0; p < 5; ++p) {
      int e = 0;
      while (m % primes[p] == 0) { m /= primes[p]; ++e; }
      ok = ok && e <= maxExp[p];
    }
    if (ok && m == 1) return c;
  }
}

// ---- device ------------------------------------------------------------------------------------------------------
struct Table1 {
  const float *table;
  int Nm1;
  float rmax, interval, dr;
};
// TabulatedFunction::operator() with LinearInterpolation (misc/TabulatedFunction.cuh:63-75, :148-157), rmin = 0
UH_D float table_get1(const Table1 &t, float rs) {
  const float r = rs * t.interval;
  if (rs >= t.rmax) return 0.0f;
  if (r <= 0.0f) return t.table[0];
  const int i = (int)(r * (float)t.Nm1);
  const float r0 = (float)i * t.dr;
  const float v0 = t.table[i], v1 = t.table[i + 1];
  const float w = (r - r0) * (float)t.Nm1;
  return fmaf(w, v1, fmaf(-w, v0, v0));
}

// chargeFourier2FieldAndPotential (.cu:433-476); planes: Ex, Ey, Ez, phi, each complex[nz][ny][nkx]
__global__ void __launch_bounds__(256) k_poisson_convolve(const float2 *__restrict__ qk, float2 *__restrict__ planes, size_t planeCplx,
                                                          int3 n, real3f L, float epsilon, FastDiv dkx, FastDiv dny) {
  const uint id = blockIdx.x * 256 + threadIdx.x;
  const int nkx = n.x / 2 + 1;
  if (id >= (uint)(nkx * n.y * n.z)) return;
  const uint row = dkx.div(id);
  const int cx = (int)(id - row * (uint)nkx);
  const int cz = (int)dny.div(row);
  const int cy = (int)(row - (uint)cz * (uint)n.y);
  float2 ex{0.f, 0.f}, ey{0.f, 0.f}, ez{0.f, 0.f}, ph{0.f, 0.f};
  const bool xn = (cx == n.x - cx) && (n.x % 2 == 0), yn = (cy == n.y - cy) && (n.y % 2 == 0), zn = (cz == n.z - cz) && (n.z % 2 == 0);
  const bool nyquist = (xn && cy == 0 && cz == 0) || (xn && yn && cz == 0) || (cx == 0 && yn && cz == 0) || (xn && cy == 0 && zn) ||
                       (cx == 0 && cy == 0 && zn) || (cx == 0 && yn && zn) || (xn && yn && zn);
  if (!(cx == 0 && cy == 0 && cz == 0) && !nyquist) {
    const float px = (2.0f * (float)M_PI) / L.x, py = (2.0f * (float)M_PI) / L.y, pz = (2.0f * (float)M_PI) / L.z;
    float kx = (float)cx * px, ky = (float)cy * py, kz = (float)cz * pz;
    if (cx >= n.x / 2 + 1) kx -= (float)n.x * px;
    if (cy >= n.y / 2 + 1) ky -= (float)n.y * py;
    if (cz >= n.z / 2 + 1) kz -= (float)n.z * pz;
    const float k2 = fmaf(kz, kz, fmaf(ky, ky, kx * kx));
    const float2 fk = qk[id];
    const float B = 1.0f / (k2 * epsilon * (float)(n.x * n.y * n.z));
    ex = make_float2(kx * fk.y * B, -kx * fk.x * B);
    ey = make_float2(ky * fk.y * B, -ky * fk.x * B);
    ez = make_float2(kz * fk.y * B, -kz * fk.x * B);
    ph = make_float2(fk.x * B, fk.y * B);
  }
  planes[id] = ex;
  planes[planeCplx + id] = ey;
  planes[2 * planeCplx + id] = ez;
  planes[3 * planeCplx + id] = ph;
}

__global__ void __launch_bounds__(256) k_poisson_interleave(const float *__restrict__ planes, size_t planeReal,
                                                            float4 *__restrict__ out, uint total) {
  const uint i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  out[i] = make_float4(planes[i], planes[planeReal + i], planes[2 * planeReal + i], planes[3 * planeReal + i]);
}

// IBM::gather of the real4 grid fused with UnZip2Real4 (.cu:529-559): one wave per particle, one float4 per node
__global__ void __launch_bounds__(256) k_poisson_gather(const float4 *__restrict__ packed, const int *__restrict__ groupIndex,
                                                        const float4 *__restrict__ grid4, float4 *__restrict__ force,
                                                        float *__restrict__ energy, float4 *__restrict__ fieldPotential, int N,
                                                        GridT<float> grid, int nxStride, IBMKernelDev kern, FastDiv dsx,
                                                        FastDiv dsxy) {
  kern.kind = kKernelGaussian;  // Poisson's only window; a constant kind folds phi_axis' switch (else every window is evaluated)
  const int lane = threadIdx.x & 63;
  const int sid = (int)xcd_contiguous_block(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
  if (sid >= N) return;
  const float4 p = packed[sid];  // Morton-sorted (x, y, z, q): neighbouring waves read the same grid lines
  const Stencil s = make_stencil(grid, kern, real3f{p.x, p.y, p.z}, false, lane);
  const int sx = s.support.x, sy = s.support.y, sz = s.support.z;
  const int nn = sx * sy * sz;
  const float dV = grid.cellVolume;
  float ax = 0.f, ay = 0.f, az = 0.f, aw = 0.f;
  // kR rounds of 64 nodes with all their loads in flight before the first is used (round by round — load, use, next load — a support-9
  // stencil was twelve dependent round trips per particle: 1.37 ms per call at 1e6 charges); same nodes, same order of the sums
  constexpr int kR = 6;
  for (int base = 0; base < nn; base += 64 * kR) {
    float4 g[kR];
    int sel[kR];  // ii | jj << 8 | kk << 16, or -1
#pragma unroll
    for (int r = 0; r < kR; ++r) {
      const int i = base + 64 * r + lane;
      const bool in = i < nn;
      const uint iu = in ? (uint)i : 0u;
      const uint kk = dsxy.div(iu);
      const uint rem = iu - kk * (uint)(sx * sy);
      const uint jj = dsx.div(rem);
      const uint ii = rem - jj * (uint)sx;
      sel[r] = in ? (int)(ii | jj << 8 | kk << 16) : -1;
      const int cx = grid.pbc_x(s.celli.x + (int)ii - s.P.x);
      const int cy = grid.pbc_y(s.celli.y + (int)jj - s.P.y);
      const int cz = grid.pbc_z(s.celli.z + (int)kk - s.P.z);
      g[r] = grid4[(size_t)cx + (size_t)nxStride * ((size_t)cy + (size_t)grid.cellDim.y * (size_t)cz)];  // (a lane past the stencil re-reads its node 0)
    }
#pragma unroll
    for (int r = 0; r < kR; ++r) {
      const int w = sel[r] < 0 ? 0 : sel[r];
      const float wx = stencil_weight(s, w & 255);
      const float wy = stencil_weight(s, sx + ((w >> 8) & 255));
      const float wz = stencil_weight(s, sx + sy + (w >> 16));
      if (sel[r] >= 0) {
        ax = fmaf(dV, g[r].x * wx * wy * wz, ax);
        ay = fmaf(dV, g[r].y * wx * wy * wz, ay);
        az = fmaf(dV, g[r].z * wx * wy * wz, az);
        aw = fmaf(dV, g[r].w * wx * wy * wz, aw);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ax += __shfl_xor(ax, o, 64);
    ay += __shfl_xor(ay, o, 64);
    az += __shfl_xor(az, o, 64);
    aw += __shfl_xor(aw, o, 64);
  }
  if (lane != 0) return;
  const float q = p.w;
  const int id = groupIndex[sid];
  if (force) {
    float4 f = force[id];
    f.x += q * ax; f.y += q * ay; f.z += q * az; f.w += q * 0.0f;
    force[id] = f;
  }
  if (energy) energy[id] += q * aw;
  if (fieldPotential) {
    float4 f = fieldPotential[id];
    f.x += ax; f.y += ay; f.z += az; f.w += aw;
    fieldPotential[id] = f;
  }
}

// Column form of the same gather (fcm.hip k_fcm_gather_col): lane = one (ii, jj) column of the stencil, G = ceil(sx sy / 64) columns per
// lane, a loop over the stencil's z planes.  The node-by-node form above is bound by its own instruction stream — two multiply-high
// divisions, three wraps, three lane shuffles and the address arithmetic per node, ~55 instructions x 729 nodes per particle at support 9:
// 1.37 ms per call at 1e6 charges — here the wraps, the column's address and wx wy are per column, a plane costs an add, a load and five
// multiply-adds, wz is a scalar lane read.  Same terms (dV (g w), w = (wx wy) wz), another order of the sums.
template <int G>
__global__ void __launch_bounds__(256) k_poisson_gather_col(const float4 *__restrict__ packed, const int *__restrict__ groupIndex,
                                                            const float4 *__restrict__ grid4, float4 *__restrict__ force,
                                                            float *__restrict__ energy, float4 *__restrict__ fieldPotential, int N,
                                                            GridT<float> grid, int nxStride, IBMKernelDev kern, FastDiv dsx) {
  kern.kind = kKernelGaussian;
  const int lane = threadIdx.x & 63;
  const int sid = (int)xcd_contiguous_block(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
  if (sid >= N) return;
  const float4 p = packed[sid];
  const Stencil s = make_stencil(grid, kern, real3f{p.x, p.y, p.z}, false, lane);
  const int sx = s.support.x, sy = s.support.y, sz = s.support.z;
  const float dV = grid.cellVolume;
  uint base[G];
  float wxy[G];
  bool col[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const uint c = (uint)(lane + 64 * g);
    col[g] = c < (uint)(sx * sy);
    const uint cc = col[g] ? c : 0u;
    const uint jj = dsx.div(cc), ii = cc - jj * (uint)sx;
    const int cx = grid.pbc_x(s.celli.x + (int)ii - s.P.x);
    const int cy = grid.pbc_y(s.celli.y + (int)jj - s.P.y);
    base[g] = (uint)cx + (uint)nxStride * (uint)cy;
    wxy[g] = stencil_weight(s, (int)ii) * stencil_weight(s, sx + (int)jj);
  }
  const uint planeNodes = (uint)nxStride * (uint)grid.cellDim.y;
  const int z0 = __builtin_amdgcn_readfirstlane(s.celli.z - s.P.z);
  float ax = 0.f, ay = 0.f, az = 0.f, aw = 0.f;
  constexpr int kPlanes = 3;  // planes whose loads are in flight together
  for (int k0 = 0; k0 < sz; k0 += kPlanes) {
    float4 v[kPlanes][G];
#pragma unroll
    for (int u = 0; u < kPlanes; ++u) {
      const int cz = grid.pbc_z(z0 + min(k0 + u, sz - 1));
#pragma unroll
      for (int g = 0; g < G; ++g) v[u][g] = grid4[(size_t)(base[g] + planeNodes * (uint)cz)];  // (idle columns re-read column 0)
    }
#pragma unroll
    for (int u = 0; u < kPlanes; ++u) {
      if (k0 + u < sz) {
        const float wz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s.w), sx + sy + k0 + u));
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (col[g]) {
            const float w = wxy[g] * wz;
            ax = fmaf(dV, v[u][g].x * w, ax);
            ay = fmaf(dV, v[u][g].y * w, ay);
            az = fmaf(dV, v[u][g].z * w, az);
            aw = fmaf(dV, v[u][g].w * w, aw);
          }
        }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ax += __shfl_xor(ax, o, 64);
    ay += __shfl_xor(ay, o, 64);
    az += __shfl_xor(az, o, 64);
    aw += __shfl_xor(aw, o, 64);
  }
  if (lane != 0) return;
  const float q = p.w;
  const int id = groupIndex[sid];
  if (force) {
    float4 f = force[id];
    f.x += q * ax; f.y += q * ay; f.z += q * az; f.w += q * 0.0f;
    force[id] = f;
  }
  if (energy) energy[id] += q * aw;
  if (fieldPotential) {
    float4 f = fieldPotential[id];
    f.x += ax; f.y += ay; f.z += az; f.w += aw;
    fieldPotential[id] = f;
  }
}

// IBM::spread of the charges (misc/IBM.cu:83-147), one wave per Morton-sorted particle: the atomics of neighbouring waves
// land on the same lines of one XCD's L2 (7.2 ms -> see DESIGN for the unsorted figure)
__global__ void __launch_bounds__(256) k_poisson_spread(const float4 *__restrict__ packed, float *__restrict__ gridQ, int N,
                                                        GridT<float> grid, int nxStride, IBMKernelDev kern, FastDiv dsx,
                                                        FastDiv dsxy) {
  kern.kind = kKernelGaussian;  // Poisson's only window; a constant kind folds phi_axis' switch (else every window is evaluated)
  const int lane = threadIdx.x & 63;
  const int sid = (int)xcd_contiguous_block(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
  if (sid >= N) return;
  const float4 p = packed[sid];
  const Stencil s = make_stencil(grid, kern, real3f{p.x, p.y, p.z}, false, lane);
  const int sx = s.support.x, sy = s.support.y, sz = s.support.z;
  const int nn = sx * sy * sz;
  for (int i0 = 0; i0 < nn; i0 += 64) {
    const int i = i0 + lane;
    const bool in = i < nn;
    const uint iu = in ? (uint)i : 0u;
    const uint kk = dsxy.div(iu);
    const uint rem = iu - kk * (uint)(sx * sy);
    const uint jj = dsx.div(rem);
    const uint ii = rem - jj * (uint)sx;
    const float wx = stencil_weight(s, (int)ii);
    const float wy = stencil_weight(s, sx + (int)jj);
    const float wz = stencil_weight(s, sx + sy + (int)kk);
    if (!in) continue;
    const int cx = grid.pbc_x(s.celli.x + (int)ii - s.P.x);
    const int cy = grid.pbc_y(s.celli.y + (int)jj - s.P.y);
    const int cz = grid.pbc_z(s.celli.z + (int)kk - s.P.z);
    unsafeAtomicAdd(&gridQ[(size_t)cx + (size_t)nxStride * ((size_t)cy + (size_t)grid.cellDim.y * (size_t)cz)], p.w * wx * wy * wz);
  }
}

// ---- tile-owned spread --------------------------------------------------------------------------------------------
// 1e6 charges x 10^3 nodes is 1e9 node updates: global f32 atomics manage ~1.4e11/s on gfx950 (7 ms), and LESS when the
// particles are Morton sorted (same-line conflicts).  Instead the grid is cut into tiles (tX x tY x tZ nodes, divisors of
// the grid), the particles are counting-sorted by the tile holding their stencil ORIGIN, and one workgroup per tile
// spreads ITS particles into a halo-extended copy of the tile in LDS ((T + support - 1)^3 nodes: every stencil fits, no
// clipping, each particle is visited once), every wave in a private copy (LDS float atomics retire ~1 lane/clk), lanes
// over the support^2 (y, z) rows with the x loop innermost: read-modify-write of consecutive words, rows padded to an odd
// stride.  The extended tile is then added to the grid with one global atomic per node.
struct TileGeom {
  int3 t, nt;  // tile dimensions, tiles per axis
  int txp;     // padded row stride of the halo-extended tile (odd)
  int plane;   // padded xy-plane stride
};

UH_D int3 stencil_origin(const GridT<float> &g, const IBMKernelDev &k, real3f pi) {  // wrapped into [0, n)
  const int3 c = g.getCell(pi);
  const int3 P = compute_support_shift(g, pi, c, k.support);
  int3 o = make_int3(c.x - P.x, c.y - P.y, c.z - P.z);
  if (o.x < 0) o.x += g.cellDim.x;
  if (o.y < 0) o.y += g.cellDim.y;
  if (o.z < 0) o.z += g.cellDim.z;
  return o;
}

__global__ void __launch_bounds__(256) k_poisson_tile_count(const float4 *__restrict__ packed, int N, GridT<float> grid,
                                                            IBMKernelDev kern, TileGeom tg, FastDiv dx, FastDiv dy, FastDiv dz,
                                                            int *__restrict__ count, int *__restrict__ rank) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool active = i < N;
  int t = -1 - (int)(threadIdx.x & 63);  // (idle lanes: distinct negative keys, never equal to a neighbour's)
  if (active) {
    const float4 p = packed[i];
    const int3 o = stencil_origin(grid, kern, real3f{p.x, p.y, p.z});
    t = (int)dx.div((uint)o.x) + tg.nt.x * ((int)dy.div((uint)o.y) + tg.nt.y * (int)dz.div((uint)o.z));
  }
  // The particles arrive Morton-sorted: a wave is a handful of RUNS of lanes with the same tile.  One atomic per run (its first lane
  // reserves the run's ranks) instead of one per particle: 1e6 atomics on 4096 counters serialise in L2 — 220 us per call.
  const int lane = threadIdx.x & 63;
  const int prev = __shfl_up(t, 1, 64);
  const bool head = lane == 0 || prev != t;
  const unsigned long long heads = __ballot(head);
  const unsigned long long below = heads & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
  const int h = 63 - __builtin_clzll(below);                         // first lane of this lane's run
  const unsigned long long after = heads & ~((2ull << h) - 1ull);    // (h < 63 whenever a later head exists)
  const int end = (h == 63 || after == 0ull) ? 64 : __builtin_ctzll(after);
  int base = 0;
  if (head && active) base = atomicAdd(&count[t], end - h);  // (a run of active lanes is all active: idle lanes have their own keys)
  base = __shfl(base, h, 64);
  if (active) {
    rank[i] = base + (lane - h);  // rank within the tile (order of arrival of the runs)
    rank[N + i] = t;
  }
}

// exclusive scan of the tile counts by ONE workgroup (the tile table is small)
__global__ void __launch_bounds__(1024) k_poisson_tile_scan(const int *__restrict__ count, int *__restrict__ start, int ntiles) {
  __shared__ int sh[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < ntiles; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < ntiles ? count[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < ntiles) start[i] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) start[ntiles] = carry;
}

__global__ void __launch_bounds__(256) k_poisson_tile_place(const float4 *__restrict__ packed, const int *__restrict__ groupIndex,
                                                            int N, const int *__restrict__ start, const int *__restrict__ rank,
                                                            float4 *__restrict__ tilePos, int *__restrict__ tileIdx) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int slot = start[rank[N + i]] + rank[i];
  tilePos[slot] = packed[i];
  tileIdx[slot] = groupIndex[i];
}

// 1-D weights of a particle (wave-uniform position): lane l < 3s evaluates node l of its axis — the expression of make_stencil
UH_D float tile_weights(const GridT<float> &grid, const IBMKernelDev &kern, float px, float py, float pz, int ox, int oy, int oz,
                        int s, int lane) {
  const real3f pi{px, py, pz};
  const int axis = lane < s ? 0 : (lane < 2 * s ? 1 : 2);
  const int li = lane - axis * s;
  float wl = 0.0f;
  if (lane < 3 * s) {
    if (axis == 0) wl = phi_axis(kern, 0, grid.distanceToCellCenter(pi, make_int3(grid.pbc_x(ox + li), 0, 0)).x);
    else if (axis == 1) wl = phi_axis(kern, 1, grid.distanceToCellCenter(pi, make_int3(0, grid.pbc_y(oy + li), 0)).y);
    else wl = phi_axis(kern, 2, grid.distanceToCellCenter(pi, make_int3(0, 0, grid.pbc_z(oz + li))).z);
  }
  return wl;
}

// S = support (compile time, rows fully unrolled: all the LDS reads of a row are in flight together) or 0 (any support)
template <int S>
__global__ void __launch_bounds__(256) k_poisson_spread_tile(const float4 *__restrict__ tilePos, const int *__restrict__ start,
                                                             float *__restrict__ gridQ, GridT<float> grid, int nxStride,
                                                             IBMKernelDev kern, TileGeom tg) {
  kern.kind = kKernelGaussian;  // see k_poisson_gather
  extern __shared__ float acc[];  // one private copy of the halo-extended tile per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = blockDim.x >> 6;
  const int s = S ? S : kern.support.x;  // cubic support (Poisson_ns::Gaussian has one)
  const int ex = tg.txp, ey = tg.t.y + s - 1, ez = tg.t.z + s - 1;
  const int plane = tg.plane;  // >= ex*ey, padded so that the lanes of a pass fall on distinct banks
  const int copyN = plane * ez;
  for (int i = threadIdx.x; i < NW * copyN; i += blockDim.x) acc[i] = 0.0f;
  __syncthreads();
  float *mine = acc + wave * copyN;
  const int tile = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int tx = tile % tg.nt.x, ty = (tile / tg.nt.x) % tg.nt.y, tz = tile / (tg.nt.x * tg.nt.y);
  const int x0 = tx * tg.t.x, y0 = ty * tg.t.y, z0 = tz * tg.t.z;
  const int3 n = grid.cellDim;
  const int b0 = start[tile], b1 = start[tile + 1];
  const int rows = s * s;
  const float rs = __builtin_amdgcn_rcpf((float)s);
  for (int base = b0; base < b1; base += 64) {
    const int k = base + lane;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    int rx = 0, ry = 0, rz = 0;
    if (k < b1) {
      p = tilePos[k];
      const int3 o = stencil_origin(grid, kern, real3f{p.x, p.y, p.z});
      rx = o.x - x0; ry = o.y - y0; rz = o.z - z0;  // in [0, T): the particles are binned by the tile of their origin
    }
    const int cnt = min(64, b1 - base);
#define PARTICLE(j, W, RX, RY, RZ, Q)                                                                                         \
    const float Q = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p.w), j));                                       \
    const int RX = __builtin_amdgcn_readlane(rx, j), RY = __builtin_amdgcn_readlane(ry, j), RZ = __builtin_amdgcn_readlane(rz, j); \
    const float W = tile_weights(grid, kern, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p.x), j)),             \
                                 __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p.y), j)),                          \
                                 __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p.z), j)), x0 + RX, y0 + RY, z0 + RZ, s, lane);
    // particles of the batch dealt to the waves; the weights of the NEXT particle are evaluated before the current one is
    // spread, so that the exp chain overlaps the LDS round trips
    int j = wave;
    if (j >= cnt) continue;
    float wl, q;
    int sRx, sRy, sRz;
    { PARTICLE(j, w0, a0, a1, a2, q0) wl = w0; q = q0; sRx = a0; sRy = a1; sRz = a2; }
    while (j < cnt) {
      const int jn = j + NW;
      float wlN = 0.f, qN = 0.f;
      int nRx = 0, nRy = 0, nRz = 0;
      if (jn < cnt) { PARTICLE(jn, w1, c0, c1, c2, q1) wlN = w1; qN = q1; nRx = c0; nRy = c1; nRz = c2; }
      if (S) {
        float qwx[S ? S : 1];
#pragma unroll
        for (int ii = 0; ii < S; ++ii) qwx[ii] = q * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wl), ii));
        for (int r0 = 0; r0 < rows; r0 += 64) {  // lanes over the s*s (y, z) rows of the stencil; wave-uniform trip count
          const int r = r0 + lane;
          const bool in = r < rows;
          const int ru = in ? r : 0;
          const int kk = (int)(((float)ru + 0.5f) * rs);  // exact for these small integers
          const int jj = ru - kk * s;
          const float wyz = __shfl(wl, s + jj, 64) * __shfl(wl, 2 * s + kk, 64);
          float *row = mine + sRx + ex * (sRy + jj) + plane * (sRz + kk);
          float v[S ? S : 1];
#pragma unroll
          for (int ii = 0; ii < S; ++ii) v[ii] = row[ii];
          if (in) {
#pragma unroll
            for (int ii = 0; ii < S; ++ii) row[ii] = fmaf(qwx[ii], wyz, v[ii]);
          }
        }
      } else {
        for (int r0 = 0; r0 < rows; r0 += 64) {
          const int r = r0 + lane;
          const bool in = r < rows;
          const int ru = in ? r : 0;
          const int kk = (int)(((float)ru + 0.5f) * rs);
          const int jj = ru - kk * s;
          const float wyz = __shfl(wl, s + jj, 64) * __shfl(wl, 2 * s + kk, 64);
          float *row = mine + sRx + ex * (sRy + jj) + plane * (sRz + kk);
          for (int i0 = 0; i0 < s; i0 += 4) {  // groups of four: reads back to back, then the FMAs, then the writes
            float v[4], qwx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int ii = min(i0 + u, s - 1);
              qwx[u] = q * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wl), ii));
              v[u] = row[ii];
            }
            if (in) {
#pragma unroll
              for (int u = 0; u < 4; ++u)
                if (i0 + u < s) row[i0 + u] = fmaf(qwx[u], wyz, v[u]);
            }
          }
        }
      }
      j = jn; wl = wlN; q = qN; sRx = nRx; sRy = nRy; sRz = nRz;
    }
#undef PARTICLE
  }
  __syncthreads();
  // flush: the halo-extended tile overlaps its neighbours, so the store is an atomic add (one per node and tile: ~2e7 for a
  // 128^3 grid against the 1e9 node updates done in LDS)
  const int fx = tg.t.x + s - 1;
  const int total = fx * ey * ez;
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const int lx = i % fx, ly = (i / fx) % ey, lz = i / (fx * ey);
    const int a = lx + ex * ly + plane * lz;
    float v = acc[a];
    for (int w = 1; w < NW; ++w) v += acc[w * copyN + a];
    if (v == 0.0f) continue;
    int gx = x0 + lx, gy = y0 + ly, gz = z0 + lz;
    if (gx >= n.x) gx -= n.x;
    if (gy >= n.y) gy -= n.y;
    if (gz >= n.z) gz -= n.z;
    unsafeAtomicAdd(&gridQ[(size_t)gx + (size_t)nxStride * ((size_t)gy + (size_t)n.y * (size_t)gz)], v);
  }
}

__global__ void __launch_bounds__(256) k_poisson_pack(const float4 *__restrict__ sortPos, const int *__restrict__ groupIndex,
                                                      const float *__restrict__ charge, float4 *__restrict__ packed, int N) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const float4 p = sortPos[i];
  packed[i] = make_float4(p.x, p.y, p.z, charge[groupIndex[i]]);
}

// transverseList with NearField{Force,Energy,FieldPotential}Transverser (.cu:222-329).  MODE 0: force4 += (total, 0);
// 1: energy += total; 2: fieldPotential4 += (E, phi).  Neighbour order = NeighbourList/common.cuh:10-34.
template <int MODE>
__global__ void __launch_bounds__(128) k_poisson_near(const float4 *__restrict__ packed, const int *__restrict__ groupIndex,
                                                      const uint *__restrict__ cellStart, const int *__restrict__ cellEnd,
                                                      uint validCell, int N, GridT<float> grid, BoxT<float> box, Table1 tabF,
                                                      Table1 tabP, float *__restrict__ out) {
  const int id = (int)xcd_contiguous_block(blockIdx.x, gridDim.x) * 128 + threadIdx.x;
  if (id >= N) return;
  const float4 pi = packed[id];
  const float qi = pi.w;
  const int3 n = grid.cellDim;
  const int npx = n.x > 1 ? 3 : 1, npy = n.y > 1 ? 3 : 1, npz = n.z > 1 ? 3 : 1;
  const int numberNeighbourCells = npx * npy * npz;
  const int3 celli = grid.getCell(real3f{pi.x, pi.y, pi.z});
  // every pair the exact tests keep has r2 below this: r2 < rc^2 for the potential table, sqrt(r2) < rc for the field (a square root
  // correctly rounded cannot report < rc for r2 >= rc^2 (1 + 1e-5))
  const float rcAny = MODE == 1 ? sqrtf(tabP.rmax) : (MODE == 0 ? tabF.rmax : fmaxf(tabF.rmax, sqrtf(tabP.rmax)));
  const float r2skip = rcAny * rcAny * 1.00001f + 1e-30f;
  float tx = 0.f, ty = 0.f, tz = 0.f, tw = 0.f;
  for (int cc = 0; cc < numberNeighbourCells; ++cc) {
    int3 cellj = celli;
    if (npx > 1) cellj.x += cc % 3 - 1;
    if (npy > 1) cellj.y += (cc / npx) % 3 - 1;
    if (npz > 1) cellj.z += cc / (npx * npy) - 1;
    cellj.x = grid.pbc_x(cellj.x);
    cellj.y = grid.pbc_y(cellj.y);
    cellj.z = grid.pbc_z(cellj.z);
    if (cellj.x < 0 || cellj.x >= n.x || cellj.y < 0 || cellj.y >= n.y || cellj.z < 0 || cellj.z >= n.z) continue;
    const int icellj = grid.getCellIndex(cellj);
    const uint cs = cellStart[icellj];
    if (cs < validCell) continue;
    const int first = (int)(cs - validCell), last = cellEnd[icellj];
    // (four candidates per trip with their loads in flight together; evaluated one after the other, in the walk's order)
    for (int j0 = first; j0 < last; j0 += 4) {
     float4 c4[4];
#pragma unroll
     for (int u = 0; u < 4; ++u) c4[u] = packed[min(j0 + u, last - 1)];
#pragma unroll
     for (int u = 0; u < 4; ++u) {
      if (j0 + u >= last) break;
      const float4 pj = c4[u];
      const real3f rij = box.apply_pbc(real3f{pj.x - pi.x, pj.y - pi.y, pj.z - pi.z});
      const float r2 = dot3(rij, rij);
      // both tables return 0 at and beyond the cut-off (TabulatedFunction.cuh:150-151): such pairs add exactly zero, so
      // skipping them (85 % of the 27-cell candidates) changes no bit of the result
      // (each mode tests the table(s) it reads: r2 against rc^2 for G, sqrt(r2) against rc for the field)
      if (r2 >= r2skip) continue;  // (clearly outside: the exact tests below, with their square root, only see the rest)
      if (MODE == 1 && r2 >= tabP.rmax) continue;
      if (MODE == 0 && sqrtf(r2) >= tabF.rmax) continue;
      if (MODE == 2 && r2 >= tabP.rmax && sqrtf(r2) >= tabF.rmax) continue;
      if (MODE == 1) {
        tw += qi * pj.w * table_get1(tabP, r2);
      } else if (MODE == 0) {
        const float r = sqrtf(r2);
        const float fmod = -qi * pj.w * table_get1(tabF, r);
        if (r2 > 0.0f) {
          const float invr = 1.0f / r;  // real3 / real multiplies by the reciprocal (utils/vector.cuh:191-193)
          tx += invr * (fmod * rij.x); ty += invr * (fmod * rij.y); tz += invr * (fmod * rij.z);
        }
      } else {
        const float phi = pj.w * table_get1(tabP, r2);
        float ex = 0.f, ey = 0.f, ez = 0.f;
        if (r2 > 0.0f) {
          const float r = sqrtf(r2);
          const float fmod = -pj.w * table_get1(tabF, r);
          const float invr = 1.0f / r;
          ex = invr * (fmod * rij.x); ey = invr * (fmod * rij.y); ez = invr * (fmod * rij.z);
        }
        tx += ex; ty += ey; tz += ez; tw += phi;
      }
     }
    }
  }
  const int ori = groupIndex[id];
  if (MODE == 1) {
    out[ori] += tw;
  } else {
    float4 *o = reinterpret_cast<float4 *>(out) + ori;
    float4 v = *o;
    v.x += tx; v.y += ty; v.z += tz; v.w += (MODE == 0 ? 0.0f : tw);
    *o = v;
  }
}

// (Round 3 measured two restructurings of this walk at 1e6 charges, 1.76 ms per call: eight lanes per particle with a flat candidate
// sequence and compacted hits — the scheme of pse.hip k_pse_near8 — 1.68 ms; one lane per particle testing four candidates per step into
// a per-lane FIFO in LDS with the listed pairs evaluated wave-wide, 1.98 ms.  At this size the walk is bound by its instruction
// throughput, not by latency or by the divergence of the hit branch; neither was kept.)
static Table1 view(const DeviceBuffer &b, int ntable, float rmax) {
  Table1 t;
  t.table = (const float *)b.ptr;
  t.Nm1 = ntable - 1;
  t.rmax = rmax;
  t.interval = (float)(1.0 / (rmax - 0.0f));
  t.dr = (float)(1.0 / (float)t.Nm1);
  return t;
}

static int poisson_make_plans(Poisson *p) {
  if (int e = rocfft_setup_once()) return e;
  const size_t nx = p->cells[0], ny = p->cells[1], nz = p->cells[2], nkx = nx / 2 + 1;
  const size_t lengths[3] = {nx, ny, nz};
  const size_t rstr[3] = {1, (size_t)p->nxpad, (size_t)p->nxpad * ny};
  const size_t cstr[3] = {1, nkx, nkx * ny};
  rocfft_plan_description d = nullptr;
  UH_ROCFFT(rocfft_plan_description_create(&d));
  UH_ROCFFT(rocfft_plan_description_set_data_layout(d, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, nullptr,
                                                     nullptr, 3, rstr, p->planeReal, 3, cstr, p->planeCplx));
  UH_ROCFFT(rocfft_plan_create(&p->fwd, rocfft_placement_inplace, rocfft_transform_type_real_forward, rocfft_precision_single, 3,
                               lengths, 1, d));
  UH_ROCFFT(rocfft_plan_description_destroy(d));
  UH_ROCFFT(rocfft_plan_description_create(&d));
  UH_ROCFFT(rocfft_plan_description_set_data_layout(d, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, nullptr,
                                                     nullptr, 3, cstr, p->planeCplx, 3, rstr, p->planeReal));
  UH_ROCFFT(rocfft_plan_create(&p->inv, rocfft_placement_inplace, rocfft_transform_type_real_inverse, rocfft_precision_single, 3,
                               lengths, 4, d));
  UH_ROCFFT(rocfft_plan_description_destroy(d));
  size_t wf = 0, wi = 0;
  UH_ROCFFT(rocfft_plan_get_work_buffer_size(p->fwd, &wf));
  UH_ROCFFT(rocfft_plan_get_work_buffer_size(p->inv, &wi));
  const size_t w = std::max(wf, wi);
  UH_ROCFFT(rocfft_execution_info_create(&p->info));
  if (w) {
    if (int e = p->work.reserve(w)) return e;
    UH_ROCFFT(rocfft_execution_info_set_work_buffer(p->info, p->work.ptr, w));
  }
  return 0;
}

// counting sort of the packed particles by the tile of their stencil origin
static int poisson_bin(Poisson *p, Poisson::TileSet &ts, int N, hipStream_t st) {
  const int nt = ts.numTiles();
  const TileGeom tg{ts.tile, ts.ntiles, ts.row, ts.plane};
  if (int e = ts.rank.reserve(sizeof(int) * 2 * (size_t)N)) return e;
  if (int e = ts.pos.reserve(sizeof(float4) * (size_t)N)) return e;
  if (int e = ts.idx.reserve(sizeof(int) * (size_t)N)) return e;
  UH_CHECK(hipMemsetAsync(ts.count.ptr, 0, sizeof(int) * (size_t)nt, st));
  hipLaunchKernelGGL(k_poisson_tile_count, dim3((N + 255) / 256), dim3(256), 0, st, (const float4 *)p->packed.ptr, N, p->grid, p->kern,
                     tg, make_fastdiv(ts.tile.x), make_fastdiv(ts.tile.y), make_fastdiv(ts.tile.z), (int *)ts.count.ptr,
                     (int *)ts.rank.ptr);
  hipLaunchKernelGGL(k_poisson_tile_scan, dim3(1), dim3(1024), 0, st, (const int *)ts.count.ptr, (int *)ts.start.ptr, nt);
  hipLaunchKernelGGL(k_poisson_tile_place, dim3((N + 255) / 256), dim3(256), 0, st, (const float4 *)p->packed.ptr,
                     (const int *)p->cl.index.ptr, N, (const int *)ts.start.ptr, (const int *)ts.rank.ptr, (float4 *)ts.pos.ptr,
                     (int *)ts.idx.ptr);
  UH_CHECK(hipGetLastError());
  return 0;
}

// farField (.cu:332-360): any of d_force / d_energy / d_fieldPotential may be null
// (the particles have been sorted and packed by poisson_list)
static int poisson_far(Poisson *p, int N, float *d_force, float *d_energy, float *d_fieldPotential, hipStream_t st) {
  float *gq = (float *)p->gridQ.ptr;
  const FastDiv dsx = make_fastdiv(p->kern.support.x), dsxy = make_fastdiv(p->kern.support.x * p->kern.support.y);
  const int s = p->kern.support.x;
  UH_CHECK(hipMemsetAsync(gq, 0, sizeof(float) * p->planeReal, st));
  const bool tileSpread = p->spreadTiles.on && !p->forceAtomicSpread;
  if (tileSpread) {
    Poisson::TileSet &ts = p->spreadTiles;
    if (int e = poisson_bin(p, ts, N, st)) return e;
    const TileGeom tg{ts.tile, ts.ntiles, ts.row, ts.plane};
    const size_t lds = sizeof(float) * (size_t)ts.waves * ts.plane * (ts.tile.z + s - 1);
    const float4 *tp = (const float4 *)ts.pos.ptr;
    const int *tst = (const int *)ts.start.ptr;
    const dim3 g(ts.numTiles()), b(64 * ts.waves);
#define UH_SPREAD(S_) hipLaunchKernelGGL((k_poisson_spread_tile<S_>), g, b, lds, st, tp, tst, gq, p->grid, p->nxpad, p->kern, tg)
    switch (s) {
      case 7: UH_SPREAD(7); break;
      case 8: UH_SPREAD(8); break;
      case 9: UH_SPREAD(9); break;
      case 10: UH_SPREAD(10); break;
      case 11: UH_SPREAD(11); break;
      case 12: UH_SPREAD(12); break;
      default: UH_SPREAD(0); break;
    }
#undef UH_SPREAD
  } else {
    hipLaunchKernelGGL(k_poisson_spread, dim3((N + 3) / 4), dim3(256), 0, st, (const float4 *)p->packed.ptr, gq, N, p->grid,
                       p->nxpad, p->kern, dsx, dsxy);
  }
  UH_ROCFFT(rocfft_execution_info_set_stream(p->info, (void *)st));
  void *bq[1] = {gq};
  UH_ROCFFT(rocfft_execute(p->fwd, bq, nullptr, p->info));
  const int3 n = p->grid.cellDim;
  const int nkx = n.x / 2 + 1;
  const uint total = (uint)p->planeCplx;
  hipLaunchKernelGGL(k_poisson_convolve, dim3((total + 255) / 256), dim3(256), 0, st, (const float2 *)gq, (float2 *)p->planes.ptr,
                     p->planeCplx, n, real3f{p->L[0], p->L[1], p->L[2]}, p->par.epsilon, make_fastdiv(nkx), make_fastdiv(n.y));
  void *bp[1] = {p->planes.ptr};
  UH_ROCFFT(rocfft_execute(p->inv, bp, nullptr, p->info));
  const uint nreal = (uint)p->planeReal;
  hipLaunchKernelGGL(k_poisson_interleave, dim3((nreal + 255) / 256), dim3(256), 0, st, (const float *)p->planes.ptr, p->planeReal,
                     (float4 *)p->inter.ptr, nreal);
  {
    const int cols = p->kern.support.x * p->kern.support.y, wts = p->kern.support.x + p->kern.support.y + p->kern.support.z;
    const bool colForm = !getenv("UAMMD_POISSON_FLAT_GATHER") && cols <= 128 && wts <= 64 &&
                         (size_t)p->nxpad * p->grid.cellDim.y * p->grid.cellDim.z < ((size_t)1 << 31);
#define UH_PGATHER(K, ...) hipLaunchKernelGGL(K, dim3((N + 3) / 4), dim3(256), 0, st, (const float4 *)p->packed.ptr, (const int *)p->cl.index.ptr, \
                                              (const float4 *)p->inter.ptr, (float4 *)d_force, d_energy, (float4 *)d_fieldPotential, N, p->grid,  \
                                              p->nxpad, p->kern, __VA_ARGS__)
    if (colForm && cols <= 64) UH_PGATHER(k_poisson_gather_col<1>, dsx);
    else if (colForm) UH_PGATHER(k_poisson_gather_col<2>, dsx);
    else UH_PGATHER(k_poisson_gather, dsx, dsxy);
#undef UH_PGATHER
  }
  UH_CHECK(hipGetLastError());
  return 0;
}

// nl->update(box, nearFieldCutOff) + pack (x, y, z, q) in sorted order
// Without splitting there is no near field; the list is still built (cells of half a window) because the spread and the
// gather want the particles in Morton order.
static int poisson_list(Poisson *p, const float *d_pos, const float *d_charge, int N, hipStream_t st) {
  const float rc = p->par.split > 0 ? p->cutoff : 0.5f * (float)p->kern.support.x * p->grid.cellSize.x;
  const float rc3[3] = {rc, rc, rc};
  const int per[3] = {1, 1, 1};
  int cd[3], gper[3];
  float gL[3];
  if (int e = uammd_celllist_create_grid(p->L, per, rc3, cd, gL, gper)) return e;
  if (int e = p->cl.update((const float4 *)d_pos, N, gL, gper, cd, st)) return e;
  if (int e = p->packed.reserve(sizeof(float4) * (size_t)N)) return e;
  hipLaunchKernelGGL(k_poisson_pack, dim3((N + 255) / 256), dim3(256), 0, st, (const float4 *)p->cl.sortPos.ptr,
                     (const int *)p->cl.index.ptr, d_charge, (float4 *)p->packed.ptr, N);
  UH_CHECK(hipGetLastError());
  return 0;
}

template <int MODE> static int poisson_near(Poisson *p, int N, float *d_out, hipStream_t st) {
  const int per[3] = {1, 1, 1};
  const BoxT<float> box = make_box<float>(p->L, per);
  hipLaunchKernelGGL((k_poisson_near<MODE>), dim3((N + 127) / 128), dim3(128), 0, st, (const float4 *)p->packed.ptr,
                     (const int *)p->cl.index.ptr, (const uint *)p->cl.cellStart.ptr, (const int *)p->cl.cellEnd.ptr,
                     p->cl.validCell, N, p->cl.grid, box, view(p->tableField, p->ntable, p->cutoff),
                     view(p->tablePotential, p->ntable, p->cutoff * p->cutoff), d_out);
  UH_CHECK(hipGetLastError());
  return 0;
}

}  // namespace uammd_hip

using namespace uammd_hip;

extern "C" {

int uammd_poisson_create(const uammd_poisson_parameters *par, uammd_poisson **out, uammd_poisson_info *info) {
  if (!par || !out) { set_last_error("uammd_poisson_create: null argument"); return -1; }
  if (!(par->boxSize[0] > 0) || !(par->boxSize[1] > 0) || !(par->boxSize[2] > 0) || !(par->epsilon > 0) || !(par->gw > 0) ||
      !(par->tolerance > 0)) {
    set_last_error("uammd_poisson_create: box, epsilon, gw and tolerance must be positive");
    return -2;
  }
  Poisson *p = new (std::nothrow) Poisson();
  if (!p) { set_last_error("uammd_poisson_create: out of host memory"); return -3; }
  p->par = *par;
  const float gw = par->gw, split = par->split, epsilon = par->epsilon, tolerance = par->tolerance;
  for (int a = 0; a < 3; ++a) p->L[a] = par->boxSize[a];
  // grid (.cu:75-90)
  const double fw = far_width(gw, split);
  double h;
  if (par->upsampling > 0) h = 1.0 / par->upsampling;
  else h = (1.3 - std::min((double)((-log10f(tolerance)) / 10.0), 0.9)) * fw;
  h = std::min(h, p->L[0] / 32.0);
  const float hr = (float)h;
  for (int a = 0; a < 3; ++a) p->cells[a] = next_fft_wise((int)(p->L[a] / hr));
  const int per[3] = {1, 1, 1};
  p->grid = make_grid(make_box<float>(p->L, per), make_int3(p->cells[0], p->cells[1], p->cells[2]));
  const float hx = p->grid.cellSize.x;
  // window (SpectralEwaldPoisson.cuh:65-70, .cu:93-104)
  const float width = (float)fw;
  const float prefactor = (float)cbrt(pow(2 * M_PI * width * width, -1.5));
  const float tau = (float)(-1.0 / (2.0 * width * width));
  const float rmax = (float)sqrt(log(tolerance * sqrt(2 * M_PI * width * width)) / tau);
  int support = std::max(3, (int)(2 * rmax / hx + 0.5));
  if (support > p->cells[0] / 2 - 1) {
    set_last_error("[Poisson] Kernel support (%d) is too large for this configuration (max is %d), try increasing splitting "
                   "parameter or decrasing tolerance", support, p->cells[0] / 2 - 1);
    delete p;
    return -2;
  }
  support = std::min(support, p->cells[0] / 2 - 2);
  if (support > kMaxSupport) {
    set_last_error("uammd_poisson_create: window support %d exceeds the %d nodes per axis one wave evaluates (single precision "
                   "tolerances below ~1e-7 are not meaningful anyway: the reference's support-41 quadrupole test runs in double)", support, kMaxSupport);
    delete p;
    return -2;
  }
  p->kernel.kind = UAMMD_IBM_KERNEL_GAUSSIAN;
  p->kernel.support[0] = p->kernel.support[1] = p->kernel.support[2] = support;
  p->kernel.prefactor = prefactor;
  p->kernel.tau = tau;
  p->kernel.rmax = INFINITY;  // Poisson_ns::Gaussian::phi has no cut
  p->kernel.invh[0] = p->kernel.invh[1] = p->kernel.invh[2] = 0.f;
  p->kern = to_dev(p->kernel);
  // near field cut-off and tables (.cu:105-118, :140-160)
  if (split > 0) {
    long double E = 1;
    long double r = fw;
    while (fabsl(E) > tolerance) {
      r += 0.001l * gw;
      E = greens((float)(r * r), gw, split, epsilon);
    }
    p->cutoff = (float)r;
    if (p->cutoff > p->L[0] / 2.0) {
      set_last_error("[Poisson] Near field cut off is too large, increase splitting parameter.");
      delete p;
      return -2;
    }
    p->ntable = std::max(4096, std::min(1 << 16, (int)(p->cutoff / (gw * tolerance * 1e3))));
    const int Nm1 = p->ntable - 1;
    std::vector<float> tf(p->ntable), tp(p->ntable);
    const float rmaxF = p->cutoff, rmaxP = p->cutoff * p->cutoff;
    for (int i = 0; i <= Nm1; ++i) {  // TabulatedFunction ctor, misc/TabulatedFunction.cuh:103-117
      const double xf = (i / (double)Nm1) * (rmaxF - 0.0f) + 0.0f;
      tf[i] = greens_field((float)xf, gw, split, epsilon);
      const double xp = (i / (double)Nm1) * (rmaxP - 0.0f) + 0.0f;
      tp[i] = greens((float)xp, gw, split, epsilon);
    }
    int e = p->tableField.reserve(sizeof(float) * tf.size());
    if (!e) e = p->tablePotential.reserve(sizeof(float) * tp.size());
    if (e) { delete p; return e; }
    if (hipMemcpy(p->tableField.ptr, tf.data(), sizeof(float) * tf.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p->tablePotential.ptr, tp.data(), sizeof(float) * tp.size(), hipMemcpyHostToDevice) != hipSuccess) {
      set_last_error("uammd_poisson_create: table upload failed");
      delete p;
      return -4;
    }
  }
  p->nxpad = 2 * (p->cells[0] / 2 + 1);
  p->planeReal = (size_t)p->nxpad * p->cells[1] * p->cells[2];
  p->planeCplx = (size_t)(p->cells[0] / 2 + 1) * p->cells[1] * p->cells[2];
  int e = p->gridQ.reserve(sizeof(float) * p->planeReal);
  if (!e) e = p->planes.reserve(sizeof(float) * 4 * p->planeReal);
  if (!e) e = p->inter.reserve(sizeof(float4) * p->planeReal);
  if (!e) e = poisson_make_plans(p);
  // Spread tiles: per axis the divisor of the grid in [4, 12] closest to 8, as many private float copies (waves) as fit 64 KB.
  // (An LDS-tile GATHER of the float4 grid was built too: the halo-extended float4 tile only fits for 4-node-thick tiles,
  // whose 12-27x halo reload made it no faster than the global gather — 1.2-1.9 ms against 1.6 ms — so it was dropped.)
  {
    auto closest = [&](int n, int target) {
      int best = 0;
      for (int d = 4; d <= 12; ++d)
        if (n % d == 0 && (!best || std::abs(d - target) < std::abs(best - target))) best = d;
      return best;
    };
    const int t[3] = {closest(p->cells[0], 8), closest(p->cells[1], 8), closest(p->cells[2], 8)};
    Poisson::TileSet &ts = p->spreadTiles;
    if (t[0] && t[1] && t[2]) {
      // rows padded to an odd stride; planes padded so that row r = jj + s*kk of a pass starts at bank (row*r) mod 32:
      // the 64 lanes of a pass then cover every bank twice, the minimum
      ts.row = (t[0] + support - 1) | 1;
      ts.plane = ts.row * (t[1] + support - 1);
      while ((ts.plane - ts.row * support) % 32 != 0) ++ts.plane;
      const size_t copyBytes = sizeof(float) * (size_t)ts.plane * (t[2] + support - 1);
      ts.waves = (int)std::min<size_t>(4, 65536 / copyBytes);
      ts.on = ts.waves >= 1;
      if (ts.on) {
        ts.tile = make_int3(t[0], t[1], t[2]);
        ts.ntiles = make_int3(p->cells[0] / t[0], p->cells[1] / t[1], p->cells[2] / t[2]);
        const size_t nt = (size_t)ts.numTiles();
        if (!e) e = ts.count.reserve(sizeof(int) * nt);
        if (!e) e = ts.start.reserve(sizeof(int) * (nt + 1));
      }
    }
  }
  if (e) { delete p; return e; }
  if (info) {
    for (int a = 0; a < 3; ++a) info->cells[a] = p->cells[a];
    info->support = support;
    info->nearFieldCutOff = p->cutoff;
    info->nTable = p->ntable;
    info->h = hx;
  }
  *out = reinterpret_cast<uammd_poisson *>(p);
  return 0;
}

int uammd_poisson_set_option(uammd_poisson *h, const char *name, int value) {
  if (!h || !name) { set_last_error("uammd_poisson_set_option: null argument"); return -1; }
  Poisson *p = reinterpret_cast<Poisson *>(h);
  if (std::string(name) == "atomic_spread") { p->forceAtomicSpread = value != 0; return 0; }
  set_last_error("uammd_poisson_set_option: unknown option %s", name);
  return -1;
}

int uammd_poisson_destroy(uammd_poisson *h) {
  delete reinterpret_cast<Poisson *>(h);
  return 0;
}

int uammd_poisson_sum(uammd_poisson *h, const float *d_pos, const float *d_charge, int N, float *d_force, float *d_energy,
                      int nearForce, int nearEnergy, void *stream) {
  if (!h) { set_last_error("uammd_poisson_sum: null argument"); return -1; }
  if (N <= 0) return 0;
  if (!d_pos || !d_charge) { set_last_error("uammd_poisson_sum: null argument"); return -1; }
  if ((nearForce && !d_force) || (nearEnergy && !d_energy)) { set_last_error("uammd_poisson_sum: missing output array"); return -1; }
  if (N <= 0) return 0;
  Poisson *p = reinterpret_cast<Poisson *>(h);
  hipStream_t st = (hipStream_t)stream;
  if (int e = poisson_list(p, d_pos, d_charge, N, st)) return e;
  if (int e = poisson_far(p, N, d_force, d_energy, nullptr, st)) return e;
  if (p->par.split > 0 && (nearForce || nearEnergy)) {
    if (nearForce)
      if (int e = poisson_near<0>(p, N, d_force, st)) return e;
    if (nearEnergy)
      if (int e = poisson_near<1>(p, N, d_energy, st)) return e;
  }
  return 0;
}

int uammd_poisson_field_potential(uammd_poisson *h, const float *d_pos, const float *d_charge, int N, float *d_fieldPotential,
                                  float *d_force, float *d_energy, void *stream) {
  if (!h) { set_last_error("uammd_poisson_field_potential: null argument"); return -1; }
  if (N <= 0) return 0;
  if (!d_pos || !d_charge || !d_fieldPotential) { set_last_error("uammd_poisson_field_potential: null argument"); return -1; }
  if (N <= 0) return 0;
  Poisson *p = reinterpret_cast<Poisson *>(h);
  hipStream_t st = (hipStream_t)stream;
  if (int e = poisson_list(p, d_pos, d_charge, N, st)) return e;
  if (int e = poisson_far(p, N, d_force, d_energy, d_fieldPotential, st)) return e;
  if (p->par.split > 0) {
    if (int e = poisson_near<2>(p, N, d_fieldPotential, st)) return e;
  }
  return 0;
}

}  // extern "C"
