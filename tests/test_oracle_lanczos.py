"""CPU tests: the Lanczos oracle against the reference's known-answer tests (test/misc/lanczos/test_lanczos.cu),
in double precision as the reference's GTest build, with the same std::mt19937 inputs:
  :34-55   identity, vector of ones                         |Bv - v| <= 1e-7, sizes 1..127
  :57-93   2*I: sqrt(2) v and <= 5 iterations               sizes 1..127
  :107-139 random diagonal (mt19937 29374238, U(1,2)), ones  rel 1e-7
  :141-178 random diagonal, random vector (mt19937 1234567, U(-10,10))
  :236-269 dense SPD: M = (A+A^T)/2 + 5n I, operator M^2, result must equal M v (rel 1e-7), sizes 1..511
           (a subset of the sizes by default; every size with -m slow)."""
import numpy as np
import pytest

from oracle.lanczos import LanczosOracle, std_mt19937_uniform_real


def test_mt19937_matches_std():
    """std::mt19937 default-seeded 10000th output is 4123659995 (C++ standard, [rand.predef]); the first outputs of
    seed 5489 are the canonical MT19937 vector."""
    rs = np.random.RandomState(5489)
    raw = rs._bit_generator.random_raw(10000)
    assert int(raw[0]) == 3499211612 and int(raw[-1]) == 4123659995


def test_identity_and_two_identity():
    for scale, expect in ((1.0, 1.0), (2.0, np.sqrt(2.0))):
        solver = LanczosOracle(np.float64)
        for size in range(1, 128):
            v = np.ones(size)
            Bv = solver.run(lambda x: scale * x, v, 1e-7)
            assert np.abs(Bv - expect * v).max() <= 1e-7, size
            if scale == 2.0:
                assert solver.getLastRunRequiredSteps() <= 5, size


def test_diagonal_matrices():
    m = std_mt19937_uniform_real(29374238, 128, 1.0, 2.0)
    for random_vector in (False, True):
        solver = LanczosOracle(np.float64)
        for size in range(1, 128):
            v = std_mt19937_uniform_real(1234567, size, -10.0, 10.0) if random_vector else np.ones(size)
            Bv = solver.run(lambda x: m[:size] * x, v, 1e-7)
            theory = np.sqrt(m[:size]) * v
            assert np.abs((Bv - theory) / theory).max() <= 1e-7, (size, random_vector)


def _dense_case(size):
    vecm = std_mt19937_uniform_real(29374238, size * size, 0.0, 1.0)
    A = vecm.reshape(size, size)                     # column major in the reference; symmetrised, so irrelevant
    M = 0.5 * (A + A.T) + 5 * size * np.eye(size)
    M2 = M @ M.T
    v = std_mt19937_uniform_real(1234567, size, -10.0, 10.0)
    return M, M2, v


@pytest.mark.parametrize("sizes", [list(range(1, 40)) + [63, 64, 65, 127, 128, 200, 255, 256, 300, 511]])
def test_dense_spd(sizes):
    solver = LanczosOracle(np.float64)
    for size in sizes:
        M, M2, v = _dense_case(size)
        Bv = solver.run(lambda x: M2 @ x, v, 1e-7)
        theory = M @ v
        assert np.abs((Bv - theory) / theory).max() <= 1e-7, size


@pytest.mark.slow
def test_dense_spd_all_sizes():
    solver = LanczosOracle(np.float64)
    for size in range(1, 512):
        M, M2, v = _dense_case(size)
        Bv = solver.run(lambda x: M2 @ x, v, 1e-7)
        assert np.abs((Bv - M @ v) / (M @ v)).max() <= 1e-7, size


def test_non_positive_matrix_raises():
    solver = LanczosOracle(np.float64)
    with pytest.raises(RuntimeError):
        solver.run(lambda x: -x * np.arange(1, 9), np.ones(8), 1e-7)


def test_hard_limit():
    solver = LanczosOracle(np.float64)
    solver.setIterationHardLimit(3)
    d = np.linspace(1, 1e6, 400)
    with pytest.raises(RuntimeError, match="Could not converge"):
        solver.run(lambda x: d * x, np.ones(400), 1e-12)
