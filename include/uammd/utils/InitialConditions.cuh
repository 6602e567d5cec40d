// utils/InitialConditions.cuh (reference: src/utils/InitialConditions.cuh:17-32) — initLattice(L, N, lattice): N positions on a Bravais
// lattice that fills the box, the input of the reference's benchmark and tutorials (examples/misc/benchmark.cu:21,63,
// examples/basic_concepts/8-, 10-, 11-).
//
// The reference hands the work to its vendored generator (third_party/bravais/bravais.h) and then shifts every coordinate by 0.56.  This
// is an own statement of the same construction — a table of lattice vectors and basis sites per lattice kind, a grid of ncells =
// ceil(N / sites) unit cells shaped after the box's aspect ratio, a per-axis stretch so that the lattice fills the box exactly, nodes
// visited x-slowest / site-fastest, the first N kept — with the generator's arithmetic types (float products and sums, the node counts
// and the box offset in double), so that it returns the same float positions: tests/test_initial_conditions.py compares it bit for bit
// with vectors made by the reference's own header (tests/golden/bravais_lattices.npz).  uammd_amd/initial_conditions.py is its twin.
#ifndef UAMMD_MI355X_UTILS_INITIALCONDITIONS_CUH
#define UAMMD_MI355X_UTILS_INITIALCONDITIONS_CUH

#include "vector.cuh"

#include <cmath>
#include <stdexcept>
#include <vector>

// the lattice kinds, at global scope under the generator's names (user code says `initLattice(L, N, fcc)`)
typedef enum { sc, bcc, fcc, dia, hcp, sq, tri } BRAVAISLAT;

namespace uammd {
namespace initial_conditions_detail {
struct Lattice {
  float e[3][3];      // lattice vectors (rows)
  int nsites;
  float site[8][3];   // basis sites in units of the lattice vectors' cell
  bool planar;        // sq, tri: z = 0, Lz does not enter the volume
  bool wrapX;         // tri, hcp: nodes pushed past + Lx / 2 by the skewed second vector come back in
};
inline Lattice describe(BRAVAISLAT kind) {
  Lattice l{};
  const float h = 0.5f, q = 0.25f, t = 0.75f;
  auto cubic = [&] { l.e[0][0] = l.e[1][1] = l.e[2][2] = 1; };
  auto sites = [&](std::initializer_list<std::initializer_list<float>> s) {
    l.nsites = 0;
    for (const auto &r : s) { int d = 0; for (float v : r) l.site[l.nsites][d++] = v; ++l.nsites; }
  };
  sites({{0, 0, 0}});
  switch (kind) {
    case sc: cubic(); break;
    case bcc: cubic(); sites({{0, 0, 0}, {h, h, h}}); break;
    case fcc: cubic(); sites({{0, 0, 0}, {h, h, 0}, {h, 0, h}, {0, h, h}}); break;
    case dia:  // (the generator's table as it stands: its seventh site is (1/4, 1/4, 3/4) and its eighth (0, 3/4, 3/4))
      cubic(); sites({{0, 0, 0}, {h, h, 0}, {h, 0, h}, {0, h, h}, {q, q, q}, {t, t, q}, {q, q, t}, {0, t, t}}); break;
    case hcp:
      l.e[0][0] = 1; l.e[1][0] = h; l.e[1][1] = (float)(std::sqrt(3) / 2); l.e[2][2] = (float)(2 * std::sqrt(6) / 3);
      sites({{0, 0, 0}, {h, q, (float)(std::sqrt(6.) / 3)}}); l.wrapX = true; break;
    case sq: l.e[0][0] = l.e[1][1] = 1; l.planar = true; break;
    case tri: l.e[0][0] = 1; l.e[1][0] = h; l.e[1][1] = (float)(std::sqrt(3) / 2); l.planar = l.wrapX = true; break;
    default: throw std::invalid_argument("initLattice: unknown lattice");
  }
  return l;
}
}  // namespace initial_conditions_detail

inline std::vector<real4> initLattice(real3 L, uint N, BRAVAISLAT lat) {
  using namespace initial_conditions_detail;
  const Lattice l = describe(lat);
  float box[3] = {(float)L.x, (float)L.y, l.planar ? 1.0f : (float)L.z};   // (the generator works in float whatever `real` is: bravais.h)
  const int ncells = (int)std::ceil((float)N / (1.f * l.nsites));
  const float volume = box[0] * box[1] * box[2];
  const double density = (double)((float)ncells / volume);
  int n[3];
  if (l.planar) {
    n[0] = (int)std::ceil(std::sqrt(density) * box[0]);
    n[1] = (int)std::ceil((float)ncells / (1.0f * n[0]));
    n[2] = 1;
  } else {
    const double c = std::pow(density, 1 / 3.);
    n[0] = (int)std::ceil(c * box[0]);
    n[1] = (int)std::ceil(c * box[1]);
    n[2] = (int)std::ceil((float)ncells / (1.0f * n[0] * n[1]));
  }
  float stretch[3];
  for (int d = 0; d < 3; ++d) stretch[d] = box[d] / (n[d] * l.e[d][d]);
  const float shift = 0.56f;  // InitialConditions.cuh:24
  std::vector<real4> pos(N, make_real4(0));
  uint node = 0;
  for (int i = 0; i < n[0] && node < N; ++i)
    for (int j = 0; j < n[1] && node < N; ++j)
      for (int k = 0; k < n[2] && node < N; ++k)
        for (int s = 0; s < l.nsites && node < N; ++s, ++node) {
          float r[3];
          for (int d = 0; d < 3; ++d) {
            const float inCells = i * l.e[0][d] + j * l.e[1][d] + k * l.e[2][d] + l.site[s][d];
            r[d] = (float)(-box[d] / 2. + stretch[d] * inCells);
          }
          if (l.wrapX && r[0] > box[0] / 2) r[0] -= box[0];
          if (l.planar) r[2] = 0;
          pos[node] = make_real4(r[0] + shift, r[1] + shift, L.z == real(0.0) ? 0.0f : r[2] + shift, real(0.0));
        }
  return pos;
}

}  // namespace uammd
#endif
