// IBM<Kernel> and Grid in library mode (no ParticleData): the shape of the reference's test/misc/ibm programs on the host classes of
// misc/IBM.cuh and utils/Grid.cuh.
//   1. spreading a unit quantity per particle with a Peskin window conserves it: sum_nodes grid * cellVolume == N;
//   2. gathering a constant field returns the constant at every particle (the window is a partition of unity);
//   3. <Jq, v> == <q, Sv>: gather is the adjoint of spread (random field, random particle quantity, real3);
//   4. Grid::getCell / getCellIndex agree with where the spread quantity of a single particle lands.
#include <misc/IBM.cuh>
#include <misc/IBM_kernels.cuh>
#include <utils/Grid.cuh>
#include <cstdio>
#include <random>
#include <vector>
using namespace uammd;

template <class T> static detail::DeviceArray<T> upload(const std::vector<T> &h) {
  detail::DeviceArray<T> d(h.size());
  detail::hipCheck(hipMemcpy(d.d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice), "hipMemcpy");
  return d;
}
template <class T> static std::vector<T> download(const detail::DeviceArray<T> &d) {
  std::vector<T> h(d.size());
  detail::hipCheck(hipMemcpy(h.data(), d.d, sizeof(T) * h.size(), hipMemcpyDeviceToHost), "hipMemcpy");
  return h;
}

int main() {
  const int N = 20000;
  const real3 L = make_real3(32, 24, 40);
  const int3 cells = make_int3(32, 24, 40);
  Grid grid(Box(L), cells);
  if (grid.getNumberCells() != 32 * 24 * 40 || grid.cellSize.x != real(1)) { std::printf("FAIL grid\n"); return 1; }
  std::mt19937 gen(1234);
  std::uniform_real_distribution<real> u(-0.5, 0.5);
  std::vector<real4> pos(N);
  for (auto &p : pos) p = make_real4(make_real3(u(gen), u(gen), u(gen)) * L, 0);
  auto d_pos = upload(pos);
  int fails = 0;
  {  // 1 + 2: Peskin three-point window, scalar quantity
    auto kernel = std::make_shared<IBM_kernels::Peskin::threePoint>(grid.cellSize.x);
    IBM<IBM_kernels::Peskin::threePoint> ibm(kernel, grid);
    std::vector<real> ones(N, real(1));
    auto d_v = upload(ones);
    detail::DeviceArray<real> d_grid((size_t)grid.getNumberCells());
    ibm.spread(d_pos.d, d_v.d, d_grid.d, N);
    double total = 0;
    for (real g : download(d_grid)) total += g;
    total *= grid.getCellVolume();
    std::printf("spread of N unit quantities: sum grid dV = %.4f (N = %d)\n", total, N);
    if (std::fabs(total - N) > 1e-3 * N) { ++fails; std::printf("FAIL conservation\n"); }
    std::vector<real> field((size_t)grid.getNumberCells(), real(2.5));
    auto d_field = upload(field);
    detail::DeviceArray<real> d_out((size_t)N);
    ibm.gather(d_pos.d, d_out.d, d_field.d, N);
    double worst = 0;
    for (real o : download(d_out)) worst = std::max(worst, (double)std::fabs(o - 2.5));
    std::printf("gather of a constant field 2.5: max deviation %.2e\n", worst);
    if (worst > 1e-4) { ++fails; std::printf("FAIL partition of unity\n"); }
  }
  {  // 3: adjointness with a Gaussian window of 7 nodes, real3 quantity
    auto kernel = std::make_shared<IBM_kernels::Gaussian>(real(0.9), 7);
    IBM<IBM_kernels::Gaussian> ibm(kernel, grid);
    std::vector<real3> q(N), field((size_t)grid.getNumberCells());
    for (auto &x : q) x = make_real3(u(gen), u(gen), u(gen));
    for (auto &x : field) x = make_real3(u(gen), u(gen), u(gen));
    auto d_q = upload(q), d_field = upload(field);
    detail::DeviceArray<real3> d_grid(field.size()), d_J((size_t)N);
    ibm.spread(d_pos.d, d_q.d, d_grid.d, N);
    ibm.gather(d_pos.d, d_J.d, d_field.d, N);
    double lhs = 0, rhs = 0;
    auto J = download(d_J);
    auto S = download(d_grid);
    for (int i = 0; i < N; ++i) lhs += dot(J[i], q[i]);
    for (size_t c = 0; c < field.size(); ++c) rhs += dot(field[c], S[c]) * grid.getCellVolume();
    std::printf("adjointness <J v, q> = %.6f, <v, S q> dV = %.6f\n", lhs, rhs);
    if (std::fabs(lhs - rhs) > 2e-4 * std::fabs(lhs) + 1e-3) { ++fails; std::printf("FAIL adjoint\n"); }
  }
  {  // 4: one particle: the heaviest node of its three-point spread is the cell Grid::getCell names
    const real4 one = make_real4(3.3f, -7.8f, 12.2f, 0);
    std::vector<real4> p1(1, one);
    auto d_p1 = upload(p1);
    auto kernel = std::make_shared<IBM_kernels::Peskin::threePoint>(grid.cellSize.x);
    IBM<IBM_kernels::Peskin::threePoint> ibm(kernel, grid);
    std::vector<real> v(1, real(1));
    auto d_v = upload(v);
    detail::DeviceArray<real> d_grid((size_t)grid.getNumberCells());
    ibm.spread(d_p1.d, d_v.d, d_grid.d, 1);
    auto g = download(d_grid);
    size_t best = 0;
    for (size_t c = 0; c < g.size(); ++c) if (g[c] > g[best]) best = c;
    const int3 cell = grid.getCell(one);
    std::printf("particle in cell (%d %d %d) = node %d; heaviest node %zu\n", cell.x, cell.y, cell.z, grid.getCellIndex(cell), best);
    if ((size_t)grid.getCellIndex(cell) != best) { ++fails; std::printf("FAIL getCell\n"); }
    const int3 wrapped = grid.pbc_cell(make_int3(-1, 24, 40));
    if (!(wrapped == make_int3(31, 0, 0))) { ++fails; std::printf("FAIL pbc_cell\n"); }
  }
  std::printf(fails ? "FAILED\n" : "ok\n");
  return fails;
}
