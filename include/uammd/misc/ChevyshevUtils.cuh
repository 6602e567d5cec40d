// Chebyshev grids for the IBM template: the include path, names and members of the reference's src/misc/ChevyshevUtils.cuh
// (chebyshev::clencurt :13-31, chebyshev::doublyperiodic::QuadratureWeights :35-72 and ::Grid :75-176), written for this build.
// The doubly periodic solvers that use these grids in the reference are outside this build; the types are here because
// the reference's own unit test of the spreading template, test/misc/ibm/test_ibm.cu, exercises IBM<Kernel, Grid> through them: a grid
// that is not uniform, a per-particle support and quadrature weights that are not the cell volume.
#pragma once
#include "../uammd.h"
#include <memory>
#include <vector>
namespace uammd {
namespace chebyshev {

// Clenshaw–Curtis weight of node i of the n + 1 extrema cos(pi i / n) of T_n on [-1, 1]:
//   w_i = (c_i / n) (1 - sum_{k = 1}^{floor(n / 2)} b_k cos(2 pi k i / n) / (4 k^2 - 1)),  b_k = 1 for 2 k = n and 2 otherwise, c_i = 2 inside;
// the two ends carry 1 / (n^2 - 1) for even n and 1 / n^2 for odd n.
inline UAMMD_HOSTDEV real clencurt(int i, int n) {
  const bool even = (n % 2) == 0;
  if (i == 0 || i == n) return real(1.0) / (even ? (real(n) * n - real(1.0)) : (real(n) * n));
  real s = 1;
  const int whole = even ? n / 2 - 1 : (n - 1) / 2;   // the terms with b_k = 2
  for (int k = 1; k <= whole; ++k) s -= real(2.0) * cos(real(2.0 * M_PI) * k * i / n) / (real(4.0) * k * k - real(1.0));
  if (even) s -= cos(real(M_PI) * i) / (real(n) * n - real(1.0));   // k = n / 2
  return real(2.0) * s / n;
}

namespace doublyperiodic {

// qw(cell, grid) = hx hy (H / 2) w_{cell.z}: a plane's area element times the Clenshaw–Curtis weight of the plane's height.
// The table lives on the device and is owned by the first object; copies (the kernel's argument among them) share it.
struct QuadratureWeights {
  QuadratureWeights(real H, real cellSizex, real cellSizey, int nz) : planeArea(cellSizex * cellSizey) {
    std::vector<real> w(nz + 1, real(0));
    for (int i = 0; i < nz; ++i) w[i] = real(0.5) * H * clencurt(i, nz - 1);
    void *d = nullptr;
    uammd::detail::hipCheck(hipMalloc(&d, w.size() * sizeof(real)), "QuadratureWeights: hipMalloc");
    owner = std::shared_ptr<void>(d, [](void *p) { (void)hipFree(p); });
    uammd::detail::hipCheck(hipMemcpy(d, w.data(), w.size() * sizeof(real), hipMemcpyHostToDevice), "QuadratureWeights: hipMemcpy");
    table = static_cast<const real *>(d);
  }
  template <class Grid> inline __device__ real operator()(int3 cell, const Grid &) const { return planeArea * table[cell.z]; }

private:
  real planeArea;
  const real *table = nullptr;
  std::shared_ptr<void> owner;
};

// Uniform and periodic in x and y; cellDim.z planes at the heights (Lz / 2) cos(pi k / (cellDim.z - 1)), k = 0 at the top, walls in z.
struct Grid {
  int3 gridPos2CellIndex;
  int3 cellDim;
  real2 cellSize;
  real2 invCellSize;
  Box box;

  Grid() : Grid(Box(), make_int3(0, 0, 0)) {}
  Grid(Box box, real3 minCellSize) : Grid(box, make_int3(box.boxSize / minCellSize)) {}
  Grid(Box box, real minCellSize) : Grid(box, make_real3(minCellSize)) {}
  Grid(Box in_box, int3 in_cellDim) : cellDim(in_cellDim), box(in_box) {
    box.setPeriodicity(1, 1, 0);
    cellSize = make_real2(box.boxSize.x / cellDim.x, box.boxSize.y / cellDim.y);
    invCellSize = make_real2(real(1.0) / cellSize.x, real(1.0) / cellSize.y);
    gridPos2CellIndex = make_int3(1, cellDim.x, cellDim.x * cellDim.y);
  }

  // the plane at or above the position (truncation of the Chebyshev angle), the uniform cell in the plane
  template <class VecType> inline UAMMD_HOSTDEV int3 getCell(const VecType &r) const {
    const real3 p = box.apply_pbc(make_real3(r));
    const real angle = acos(real(2.0) * p.z / box.boxSize.z) / real(M_PI);
    int3 c = make_int3(int((p.x + real(0.5) * box.boxSize.x) * invCellSize.x), int((p.y + real(0.5) * box.boxSize.y) * invCellSize.y),
                       int((cellDim.z - 1) * angle));
    if (c.x == cellDim.x) c.x = 0;
    if (c.y == cellDim.y) c.y = 0;
    return c;
  }
  inline UAMMD_HOSTDEV int getCellIndex(const int3 &cell) const {
    return cell.x * gridPos2CellIndex.x + cell.y * gridPos2CellIndex.y + cell.z * gridPos2CellIndex.z;
  }
  // x, y wrap once; a plane outside [0, cellDim.z) does not exist: -1
  template <int coordinate> inline UAMMD_HOSTDEV int pbc_cell_coord(int cell) const {
    if (coordinate == 2) return (cell >= 0 && cell < cellDim.z) ? cell : -1;
    const int n = coordinate == 0 ? cellDim.x : cellDim.y;
    return cell < 0 ? cell + n : (cell >= n ? cell - n : cell);
  }
  inline UAMMD_HOSTDEV int3 pbc_cell(const int3 &cell) const {
    return make_int3(pbc_cell_coord<0>(cell.x), pbc_cell_coord<1>(cell.y), pbc_cell_coord<2>(cell.z));
  }
  inline UAMMD_HOSTDEV int getNumberCells() const { return cellDim.x * cellDim.y * cellDim.z; }
  // (the planes have no thickness of their own: the z extent of a cell is reported as zero, as the reference does)
  inline UAMMD_HOSTDEV real3 getCellSize(int3) const { return make_real3(cellSize.x, cellSize.y, real(0)); }
  inline UAMMD_HOSTDEV real getCellVolume(int3 cell) const {
    const real3 s = getCellSize(cell);
    return s.x * s.y * s.z;
  }
  inline UAMMD_HOSTDEV real cellHeight(int cellz) const { return real(0.5) * box.boxSize.z * cospi(real(cellz) / (cellDim.z - 1)); }
  // a cell's reference point: its lower corner in the plane, its plane's height
  inline UAMMD_HOSTDEV real3 getCellCenter(int3 cell) const {
    return make_real3(cellSize.x * cell.x - real(0.5) * box.boxSize.x, cellSize.y * cell.y - real(0.5) * box.boxSize.y, cellHeight(cell.z));
  }
  inline UAMMD_HOSTDEV real3 distanceToCellCenter(real3 pos, int3 cell) const { return box.apply_pbc(pos - getCellCenter(cell)); }
};

}  // namespace doublyperiodic
}  // namespace chebyshev
}  // namespace uammd
