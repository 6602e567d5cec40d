// misc/ChevyshevUtils.cuh in a plain C++ translation unit (DOUBLE_PRECISION): the Clenshaw–Curtis weights integrate every polynomial of
// degree <= n exactly on the n + 1 Chebyshev extrema (the property chebyshev::clencurt is used for, ChevyshevUtils.cuh:13-31), the
// doubly periodic grid's planes, cells and wrapping (:75-176), and the quadrature functor's table shared between copies (:35-72).
#include "misc/ChevyshevUtils.cuh"
#include <cmath>
#include <cstdio>
using namespace uammd;
static_assert(std::is_same<real, double>::value, "built with -DDOUBLE_PRECISION");

static int bad = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++bad; } } while (0)

int main() {
  for (int n : {1, 2, 7, 8, 15, 16, 33, 127}) {
    double worst = 0;
    for (int degree = 0; degree <= n; ++degree) {
      double sum = 0;
      for (int i = 0; i <= n; ++i) sum += chebyshev::clencurt(i, n) * std::pow(std::cos(M_PI * i / n), degree);
      const double exact = (degree % 2) ? 0.0 : 2.0 / (degree + 1);
      worst = std::max(worst, std::abs(sum - exact));
    }
    std::printf("clencurt n = %3d: worst quadrature error over degrees 0..n %.2e\n", n, worst);
    CHECK(worst < 2e-14 * (n + 1));
    for (int i = 0; i <= n; ++i) CHECK(chebyshev::clencurt(i, n) > 0 && std::abs(chebyshev::clencurt(i, n) - chebyshev::clencurt(n - i, n)) < 1e-15);
  }
  using Grid = chebyshev::doublyperiodic::Grid;
  const int3 n = make_int3(16, 12, 9);
  Grid grid(Box(make_real3(4.0, 3.0, 2.0)), n);
  CHECK(!grid.box.isPeriodicZ() && grid.box.isPeriodicX() && grid.box.isPeriodicY());
  CHECK(grid.getNumberCells() == 16 * 12 * 9 && grid.getCellIndex(make_int3(3, 2, 1)) == 3 + 16 * (2 + 12 * 1));
  CHECK(std::abs(grid.cellHeight(0) - 1.0) < 1e-15 && std::abs(grid.cellHeight(8) + 1.0) < 1e-15 && grid.cellHeight(4) == 0.0);
  for (int k = 0; k + 1 < n.z; ++k) {
    CHECK(grid.cellHeight(k) > grid.cellHeight(k + 1));
    const double between = 0.5 * (grid.cellHeight(k) + grid.cellHeight(k + 1));
    const int3 c = grid.getCell(make_real3(-1.99, 1.49, between));      // between two planes: the upper one's index
    CHECK(c.x == 0 && c.y == 11 && c.z == k);
    const real3 d = grid.distanceToCellCenter(make_real3(-1.99, 1.49, between), c);
    CHECK(std::abs(d.x - 0.01) < 1e-14 && std::abs(d.y - (1.49 - (-1.5 + 11 * 0.25))) < 1e-14 && std::abs(d.z - (between - grid.cellHeight(k))) < 1e-15);
  }
  CHECK(grid.getCell(make_real3(2.0 + 0.1, 0, 0)).x == grid.getCell(make_real3(-2.0 + 0.1, 0, 0)).x);   // periodic in x
  CHECK(grid.pbc_cell(make_int3(-1, 12, -1)).x == 15 && grid.pbc_cell(make_int3(-1, 12, -1)).y == 0 && grid.pbc_cell(make_int3(-1, 12, -1)).z == -1);
  CHECK(grid.pbc_cell_coord<2>(9) == -1 && grid.pbc_cell_coord<2>(8) == 8);
  CHECK(grid.getCellSize(make_int3(0, 0, 0)).z == 0 && grid.getCellVolume(make_int3(0, 0, 0)) == 0);
  CHECK(std::abs(cospi(0.5)) == 0.0 && cospi(1.0) == -1.0 && std::abs(cospi(1.0 / 3) - 0.5) < 2.5e-16 && std::abs(sinpi(1.0 / 6) - 0.5) < 2.5e-16);
  int devices = 0;
  if (hipGetDeviceCount(&devices) != hipSuccess || devices == 0) std::printf("no device here: the quadrature table is not exercised\n");
  else {
    chebyshev::doublyperiodic::QuadratureWeights qw(2.0, 0.25, 0.25, n.z);
    auto copy = qw;                                                        // shares the device table; both go away without a double free
    chebyshev::doublyperiodic::QuadratureWeights again(copy);
    (void)again;
  }
  std::printf(bad ? "chebyshev_grid: FAILED\n" : "chebyshev_grid: ok\n");
  return bad;
}
