"""initLattice (utils/InitialConditions.cuh:17-32) against the reference's own lattice generator (tests/golden/bravais_lattices.npz,
made by tests/golden/make_bravais_golden.py from the reference's third_party/bravais/bravais.h): the Python mirror
(uammd_amd/initial_conditions.py) and the C++ header (include/uammd/utils/InitialConditions.cuh) bit for bit."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G = np.load(os.path.join(ROOT, "tests", "golden", "bravais_lattices.npz"))
NCASES = sum(1 for k in G.files if k.startswith("case") and k.endswith("_pos"))
NBIG = sum(1 for k in G.files if k.startswith("big") and k.endswith("_sha256"))
SHIFT = np.float32(0.56)


def _expected(c):
    pos = G[f"case{c}_pos"].copy()
    L = G[f"case{c}_L"]
    pos[:, :3] += SHIFT                      # initLattice's shift, in float32
    if L[2] == 0:
        pos[:, 2] = 0
    pos[:, 3] = 0
    return str(G[f"case{c}_kind"]), L, pos


@pytest.mark.parametrize("c", range(NCASES))
def test_python_init_lattice_matches_reference_generator(c):
    from uammd_amd.initial_conditions import init_lattice
    kind, L, ref = _expected(c)
    got = init_lattice(L, len(ref), kind)
    assert got.dtype == np.float32 and got.shape == ref.shape
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (kind, np.abs(got - ref).max())


@pytest.mark.parametrize("c", range(NBIG))
def test_python_init_lattice_benchmark_sizes(c):
    """the reference benchmark's input (fcc, 2^20 particles, L = 128: examples/misc/benchmark.cu:21,63) and the C3-sized simple cubic
    lattice, by digest of the generator's output (shift = 0: a float32 subtraction would not invert the + 0.56)"""
    from uammd_amd.initial_conditions import init_lattice
    kind, n, L = str(G[f"big{c}_kind"]), int(G[f"big{c}_n"]), G[f"big{c}_L"]
    got = init_lattice(L, n, kind, shift=0.0)
    assert hashlib.sha256(got.tobytes()).hexdigest() == str(G[f"big{c}_sha256"])


def test_cxx_init_lattice_matches_reference_generator(tmp_path):
    """include/uammd/utils/InitialConditions.cuh compiled by plain g++ -std=c++14 (host-only header)"""
    src = tmp_path / "lat.cpp"
    src.write_text(r'''
#include "utils/InitialConditions.cuh"
#include <cstdio>
#include <cstdlib>
#include <cstring>
using namespace uammd;
int main(int argc, char **argv) {
  const char *names[] = {"sc", "bcc", "fcc", "dia", "hcp", "sq", "tri"};
  const BRAVAISLAT types[] = {sc, bcc, fcc, dia, hcp, sq, tri};
  BRAVAISLAT t = sc;
  for (int i = 0; i < 7; ++i) if (!strcmp(argv[1], names[i])) t = types[i];
  auto pos = initLattice(make_real3(atof(argv[3]), atof(argv[4]), atof(argv[5])), (uint)atoi(argv[2]), t);
  fwrite(pos.data(), sizeof(real4), pos.size(), stdout);
  return 0;
}
''')
    exe = tmp_path / "lat"
    subprocess.run(["g++", "-std=c++14", "-O1", "-I", os.path.join(ROOT, "include", "uammd"), "-I", os.path.join(ROOT, "include"),
                    "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", str(src), "-o", str(exe)], check=True)
    for c in range(NCASES):
        kind, L, ref = _expected(c)
        out = subprocess.run([str(exe), kind, str(len(ref))] + [repr(float(x)) for x in L], check=True, capture_output=True).stdout
        got = np.frombuffer(out, np.float32).reshape(-1, 4)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), kind
