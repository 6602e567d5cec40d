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

// addRandomAdvection (:274-391): thread per cell, the three planes of g
static __global__ void __launch_bounds__(256) k_fib_random_advection(float *__restrict__ g, size_t plane, int nxpad, GridT<float> grid,
                                                              float noisePrefactor, const float *__restrict__ random) {
  const int ic = blockIdx.x * 256 + threadIdx.x;
  const int3 n = grid.cellDim;
  const int nc = n.x * n.y * n.z;
  if (ic >= nc) return;
  const int x = ic % n.x, y = (ic / n.x) % n.y, z = ic / (n.x * n.y);
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
  const size_t node = (size_t)x + (size_t)nxpad * ((size_t)y + (size_t)n.y * (size_t)z);
  g[node] += dx * noisePrefactor;
  g[plane + node] += dy * noisePrefactor;
  g[2 * plane + node] += dz * noisePrefactor;
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


}  // namespace uammd_hip
