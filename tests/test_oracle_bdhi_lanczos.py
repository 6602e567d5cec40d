"""CPU: BDHI::Lanczos restatement (oracle_rpy_nbody_mdot + LanczosOracle) against textbook Rotne-Prager-Yamakawa:
isolated-particle mobility 1/(6 pi eta a), the far/overlap pair tensors, symmetry and positive definiteness of the dense
matrix (what makes M^(1/2) exist), and sqrt(M) dW by Lanczos against a dense eigendecomposition.  The reference's own test
for this module (test/BDHI/Lanczos_Cholesky/test.bash) is a long statistical run; its cuRAND noise stream is unpinned."""
import math

import numpy as np


def _dense(o, pos, visc, rh=-1.0, radius=None):
    from oracle.pse import rpy_nbody_mdot
    n = len(pos)
    M = np.zeros((3 * n, 3 * n))
    for c in range(3 * n):
        v = np.zeros((n, 3))
        v[c // 3, c % 3] = 1.0
        M[:, c] = rpy_nbody_mdot(o, pos, v, visc, rh, radius).reshape(-1)
    return M


def test_rpy_known_values(o64):
    visc, a = 1.3, 0.9
    M0 = 1 / (6 * math.pi * visc * a)
    pos = np.zeros((2, 4))
    pos[1, :3] = [3.0, 0, 0]                                   # r > 2a
    M = _dense(o64, pos, visc, rh=a)
    assert abs(M[0, 0] - M0) < 1e-15 and abs(M[4, 4] - M0) < 1e-15
    r = 3.0
    f = M0 * (3 * a / (4 * r)) * (1 + 2 * a * a / (3 * r * r))
    g = M0 * (3 * a / (4 * r)) * (1 - 2 * a * a / (r * r))      # coefficient of rr/r^2
    assert abs(M[0, 3] - (f + g)) < 1e-14 and abs(M[1, 4] - f) < 1e-14 and abs(M[0, 4]) < 1e-16
    pos[1, :3] = [0, 1.2, 0]                                   # overlapping, r < 2a
    M = _dense(o64, pos, visc, rh=a)
    r = 1.2
    f = M0 * (1 - 9 * r / (32 * a))
    g = M0 * (3 * r / (32 * a))
    assert abs(M[0, 3] - f) < 1e-14 and abs(M[1, 4] - (f + g)) < 1e-14


def test_dense_matrix_spd_and_sqrt_by_lanczos(o64):
    from oracle.lanczos import LanczosOracle
    from oracle.pse import rpy_nbody_mdot
    rng = np.random.default_rng(4)
    n, visc = 60, 0.8
    pos = np.zeros((n, 4))
    pos[:, :3] = rng.uniform(-6, 6, (n, 3))
    radius = rng.uniform(0.5, 1.2, n)                          # different sizes: Zuk et al. 2014 formulas
    M = _dense(o64, pos, visc, radius=radius)
    assert np.abs(M - M.T).max() < 1e-14
    lam, P = np.linalg.eigh(M)
    assert lam.min() > 0
    z = rng.normal(0, 1, 3 * n)
    exact = P @ (np.sqrt(lam) * (P.T @ z))
    lz = LanczosOracle(np.float64)
    got = lz.run(lambda v: rpy_nbody_mdot(o64, pos, v.reshape(n, 3), visc, radius=radius).reshape(-1), z, 1e-6)
    assert np.linalg.norm(got - exact) <= 1e-5 * np.linalg.norm(exact)
