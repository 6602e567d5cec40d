// Staggered-grid pieces shared by BDHI::FIB (fib.hip) and Hydro::ICM (icm.hip): the fluid random numbers, the stochastic
// stress divergence and the (component, node) walk of the 3-point Peskin window on the three face-centred grids.
#pragma once
#include "ibm.hpp"
#include "saru.hpp"

namespace uammd_hip {

// 6 standard normals per cell, slot-major: random[slot * ncells + cell]
static __global__ void __launch_bounds__(256) k_fib_noise(float *__restrict__ random, int ncells, uint seed, uint step) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 3 * ncells) return;  // each thread draws a pair
  Saru rng((uint)i, seed, step);
  const float2 g = rng.gf(0.0f, 1.0f);
  random[2 * (size_t)i] = g.x;
  random[2 * (size_t)i + 1] = g.y;
}

// D~ W of cell (x, y, z): the stochastic stress divergence on the staggered grid (FIB.cu:274-391 = ICM.cu:507-590)
UH_D real3f noise_divergence(const GridT<float> &grid, int x, int y, int z, const float *__restrict__ random) {
  const int3 n = grid.cellDim;
  const int nc = n.x * n.y * n.z;
  const int ic = x + n.x * (y + n.y * z);
  auto idx = [&](int a, int b, int c) { return grid.pbc_x(a) + n.x * (grid.pbc_y(b) + n.y * grid.pbc_z(c)); };
  const float sqrt2 = 1.41421356237310f;
  float dx = 0.f, dy = 0.f, dz = 0.f;
  dx += sqrt2 * grid.invCellSize.x * (random[idx(x + 1, y, z)] - random[ic]);
  dy += sqrt2 * grid.invCellSize.y * (random[idx(x, y + 1, z) + nc] - random[ic + nc]);
  dz += sqrt2 * grid.invCellSize.z * (random[idx(x, y, z + 1) + 2 * nc] - random[ic + 2 * nc]);
  const float wxy = random[ic + 3 * nc], wxz = random[ic + 4 * nc], wyz = random[ic + 5 * nc];
  dx += grid.invCellSize.y * (wxy - random[idx(x, y - 1, z) + 3 * nc]);
  dy += grid.invCellSize.x * (wxy - random[idx(x - 1, y, z) + 3 * nc]);
  dx += grid.invCellSize.z * (wxz - random[idx(x, y, z - 1) + 4 * nc]);
  dz += grid.invCellSize.x * (wxz - random[idx(x - 1, y, z) + 4 * nc]);
  dy += grid.invCellSize.z * (wyz - random[idx(x, y, z - 1) + 5 * nc]);
  dz += grid.invCellSize.y * (wyz - random[idx(x, y - 1, z) + 5 * nc]);
  return real3f{dx, dy, dz};
}

// addRandomAdvection (FIB.cu:274-391): thread per cell, the three planes of g
static __global__ void __launch_bounds__(256) k_fib_random_advection(float *__restrict__ g, size_t plane, int nxpad, GridT<float> grid,
                                                              float noisePrefactor, const float *__restrict__ random) {
  const int ic = blockIdx.x * 256 + threadIdx.x;
  const int3 n = grid.cellDim;
  if (ic >= n.x * n.y * n.z) return;
  const int x = ic % n.x, y = (ic / n.x) % n.y, z = ic / (n.x * n.y);
  const real3f d = noise_divergence(grid, x, y, z, random);
  const size_t node = (size_t)x + (size_t)nxpad * ((size_t)y + (size_t)n.y * (size_t)z);
  g[node] += d.x * noisePrefactor;
  g[plane + node] += d.y * noisePrefactor;
  g[2 * plane + node] += d.z * noisePrefactor;
}

UH_D float peskin3(float invh, float r) { return phi_peskin3(invh, r); }

// the (component, node) pair of lane slot l in [0, 81): staggered cell of the component, node index and window value
struct StagNode { size_t node; float w; };
UH_D StagNode stag_node(const GridT<float> &grid, int nxpad, float invh, real3f pi, int l) {
  const int c = l / 27, i = l - 27 * c;
  real3f ps = pi;  // position seen from the grid of component c: shifted half a cell (:540-548)
  if (c == 0) ps.x = pi.x - 0.5f * grid.cellSize.x;
  if (c == 1) ps.y = pi.y - 0.5f * grid.cellSize.y;
  if (c == 2) ps.z = pi.z - 0.5f * grid.cellSize.z;
  const int3 cell = grid.getCell(ps);
  const int3 cj = make_int3(grid.pbc_x(cell.x + i % 3 - 1), grid.pbc_y(cell.y + (i / 3) % 3 - 1), grid.pbc_z(cell.z + i / 9 - 1));
  const real3f r = grid.distanceToCellCenter(ps, cj);
  StagNode s;
  s.node = (size_t)cj.x + (size_t)nxpad * ((size_t)cj.y + (size_t)grid.cellDim.y * (size_t)cj.z);
  s.w = peskin3(invh, r.x) * peskin3(invh, r.y) * peskin3(invh, r.z);
  return s;
}


// spreadParticleForces (:528-597): one wave per particle, 81 atomics
// (ICM spreads force * dt/rho: `scale`, applied to the force first as the reference does, ICM.cu:95)
static __global__ void __launch_bounds__(256) k_fib_spread(const float4 *__restrict__ pos, const float4 *__restrict__ force,
                                                           float *__restrict__ g, size_t plane, int nxpad, int N, GridT<float> grid,
                                                           float invh, float scale) {
  const int lane = threadIdx.x & 63;
  const int id = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (id >= N) return;
  const float4 p = pos[id], f = force[id];
  for (int l = lane; l < 81; l += 64) {
    const StagNode s = stag_node(grid, nxpad, invh, real3f{p.x, p.y, p.z}, l);
    const int c = l / 27;
    const float fc = (c == 0 ? f.x : (c == 1 ? f.y : f.z)) * scale;
    unsafeAtomicAdd(&g[c * plane + s.node], s.w * fc);
  }
}

// midPointStep (:726-823).  MODE 0 predictor, 1 corrector, 2 euler
template <int MODE>
static __global__ void __launch_bounds__(256) k_fib_midpoint(float4 *__restrict__ pos, float4 *__restrict__ posOld, const float *__restrict__ g,
                                                      size_t plane, int nxpad, int N, GridT<float> grid, float invh, float dt) {
  const int lane = threadIdx.x & 63;
  const int id = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (id >= N) return;
  const float4 p = pos[id];
  const float dV = grid.cellSize.x * grid.cellSize.y * grid.cellSize.z;
  float acc[3] = {0.f, 0.f, 0.f};
  for (int l = lane; l < 81; l += 64) {
    const StagNode s = stag_node(grid, nxpad, invh, real3f{p.x, p.y, p.z}, l);
    const int c = l / 27;
    const float v = s.w * g[c * plane + s.node] * dV;
    acc[0] += c == 0 ? v : 0.0f; acc[1] += c == 1 ? v : 0.0f; acc[2] += c == 2 ? v : 0.0f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    acc[0] += __shfl_xor(acc[0], o, 64); acc[1] += __shfl_xor(acc[1], o, 64); acc[2] += __shfl_xor(acc[2], o, 64);
  }
  if (lane != 0) return;
  if (MODE == 0) {
    posOld[id] = p;
    const float pref = dt * 0.5f;
    pos[id] = make_float4(p.x + pref * acc[0], p.y + pref * acc[1], p.z + pref * acc[2], p.w);
  } else {
    const float4 po = posOld[id];
    pos[id] = make_float4(po.x + dt * acc[0], po.y + dt * acc[1], po.z + dt * acc[2], po.w);
  }
}

// solveStokesFourier on the three complex planes [nz][ny][nkx].  ICM = false: FIB.cu:667-724, eta^-1 (-L)^-1 P, k = 0 zeroed.
// ICM = true: ICM.cu:349-411, (I - dt eta/(2 rho) L)^-1 P with dtOverRho = dt/rho; k = 0 zeroed or only normalised.
template <bool ICM>
static __global__ void __launch_bounds__(256) k_fib_stokes(float2 *__restrict__ g, size_t planeCplx, int3 n, real3f L, float viscosity,
                                                           FastDiv dkx, FastDiv dny, float dtOverRho, bool removeTotalMomentum) {
  const uint id = blockIdx.x * 256 + threadIdx.x;
  const int nkx = n.x / 2 + 1;
  if (id >= (uint)(nkx * n.y * n.z)) return;
  const uint row = dkx.div(id);
  const int cx = (int)(id - row * (uint)nkx);
  const int cz = (int)dny.div(row);
  const int cy = (int)(row - (uint)cz * (uint)n.y);
  float2 v[3] = {g[id], g[planeCplx + id], g[2 * planeCplx + id]};
  if (id == 0) {
    if (!ICM || removeTotalMomentum) {
      v[0] = v[1] = v[2] = make_float2(0.f, 0.f);
    } else {
      const float nc = (float)(n.x * n.y * n.z);
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = make_float2(v[c].x / nc, v[c].y / nc);
    }
  } else {
    const float hx = L.x / (float)n.x, hy = L.y / (float)n.y, hz = L.z / (float)n.z;
    const float px = 2.0f * (float)M_PI / L.x, py = 2.0f * (float)M_PI / L.y, pz = 2.0f * (float)M_PI / L.z;
    float kx = (float)cx * px, ky = (float)cy * py, kz = (float)cz * pz;  // cellToWaveNumber with the (n+1)/2 threshold (:603-617)
    if (cx >= (n.x + 1) / 2) kx -= (float)n.x * px;
    if (cy >= (n.y + 1) / 2) ky -= (float)n.y * py;
    if (cz >= (n.z + 1) / 2) kz -= (float)n.z * pz;
    float sn[3], cs[3];
    sincosf(kx * hx * 0.5f, &sn[0], &cs[0]);
    sincosf(ky * hy * 0.5f, &sn[1], &cs[1]);
    sincosf(kz * hz * 0.5f, &sn[2], &cs[2]);
    const real3f keff{2.0f * (1.0f / hx) * sn[0], 2.0f * (1.0f / hy) * sn[1], 2.0f * (1.0f / hz) * sn[2]};
    float re[3], im[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {  // faces -> centres: phase (cos, -sin)
      re[c] = v[c].x * cs[c] - v[c].y * (-sn[c]);
      im[c] = v[c].y * cs[c] + v[c].x * (-sn[c]);
    }
    const float k2 = dot3(keff, keff);
    float pref;
    if (ICM) {
      const float Lk = -k2;
      pref = 1.0f / (1.0f - dtOverRho * 0.5f * viscosity * Lk);
    } else {
      const float invL = -1.0f / k2;
      pref = -1.0f * invL / viscosity;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { re[c] *= pref; im[c] *= pref; }
    const float invk2 = 1.0f / k2;
    const float kfr = dot3(keff, real3f{re[0], re[1], re[2]}) * invk2, kfi = dot3(keff, real3f{im[0], im[1], im[2]}) * invk2;
    const float ke[3] = {keff.x, keff.y, keff.z};
    const float norm = 1.0f / (float)(n.x * n.y * n.z);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float tr = re[c] - ke[c] * kfr, ti = im[c] - ke[c] * kfi;
      v[c] = make_float2(norm * (tr * cs[c] - ti * sn[c]), norm * (ti * cs[c] + tr * sn[c]));  // centres -> faces, FFT normalisation
    }
  }
  g[id] = v[0];
  g[planeCplx + id] = v[1];
  g[2 * planeCplx + id] = v[2];
}


}  // namespace uammd_hip
