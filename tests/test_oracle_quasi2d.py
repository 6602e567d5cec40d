"""Oracle pins for BDHI::True2D / BDHI::Quasi2D from the reference's test/BDHI/quasi2D/quasi2d_test.cu:
  * SelfMobilityQuasi2D (:94-115)   M = 1/(6 pi a) / (1 + 4.41 a/L)   +- 1e-3, cross term < 1e-4
  * SelfMobilityTrue2D  (:117-138)  M = (ln(L/a) - 1.3105329259115095183)/(4 pi)   +- 1e-3
  * ObeysFluctuationDissipation{Quasi2D,True2D} (:140-190)   <dr^2>/(2 T dt) = M within 1 % (fewer samples here: 3 %)
"""
import math

import numpy as np
import pytest

from oracle.quasi2d import BDHI2DOracle


def self_mobility(o, mode, lbox, a, direction, ntest=20, seed=0):
    """computeSelfMobility (:56-92): viscosity 1.12312, dt 0.1, F = 1 on a particle at random positions, T = 0."""
    visc, dt = 1.12312, 0.1
    rng = np.random.default_rng(seed)
    sim = BDHI2DOracle(o, mode, lbox, a, visc, 0.0, dt)
    M = np.zeros(2)
    for _ in range(ntest):
        pos = np.zeros((1, 4), o.real)
        pos[0, :2] = rng.uniform(-lbox / 2, lbox / 2, 2)
        f = np.zeros((1, 2), o.real)
        f[0, direction] = 1.0
        p0 = pos.copy()
        sim.forwardTime(pos, f)
        M += (pos[0, :2] - p0[0, :2]).astype(np.float64)
    return visc * M / (ntest * dt * 1.0)


@pytest.mark.parametrize("mode", ["Quasi2D", "True2D"])
def test_self_mobility(o64, mode):
    a = 1.21312
    for lbox in (32, 96, 224):
        for d in (0, 1):
            M = self_mobility(o64, mode, lbox * a, a, d)
            if mode == "Quasi2D":
                theo = 1.0 / (6 * math.pi * a) * (1 / (1 + 4.41 / lbox))
            else:
                theo = 1.0 / (4 * math.pi) * (math.log(lbox) - 1.3105329259115095183)
            assert abs(M[d] - theo) < 1e-3, (mode, lbox, d, M, theo)
            assert abs(M[1 - d]) < 1e-4


@pytest.mark.slow
@pytest.mark.parametrize("mode", ["Quasi2D", "True2D"])
def test_fluctuation_dissipation(o32, mode):
    """:140-190 with 6000 samples instead of 50000 (statistical error ~1.8 %): sigma/(2 T dt) = self mobility."""
    a, T, dt, navg = 1.21312, 1.012312, 0.9, 6000
    lbox = 128 * a
    sim = BDHI2DOracle(o32, mode, lbox, a, 1.0, T, dt, seed=4321)
    rng = np.random.default_rng(5)
    acc = np.zeros(2)
    for _ in range(navg):
        pos = np.zeros((1, 4), np.float32)
        pos[0, :2] = rng.uniform(-lbox / 2, lbox / 2, 2)
        p0 = pos.astype(np.float64)
        sim.forwardTime(pos)
        r = pos[0, :2].astype(np.float64) - p0[0, :2]
        acc += r * r
    d0 = acc / navg / (2 * T * dt)
    Mtheo = self_mobility(o32, mode, lbox, a, 0)[0]    # computeSelfMobility returns viscosity * M = the mobility at viscosity 1
    assert abs(d0[0] / Mtheo - 1) < 0.06 and abs(d0[1] / Mtheo - 1) < 0.06, (d0, Mtheo)


def test_kernels_and_window(o64):
    """Closed forms: Quasi2D f_k, g_k -> their k a -> 0 limits (1/(4 k^3), 1/(2 k^3)); windows integrate to 1 / carry the drift
    prefactor; support = 2 (int(3 a n / L) + 1) + 1."""
    out = np.zeros(2)
    o64.lib.oracle_q2d_hydro_kernel(1, o64.creal(1e-6), o64.creal(1.0), out.ctypes.data_as(__import__("ctypes").c_void_p))
    k = 1e-3
    assert abs(out[0] * 4 * k ** 3 - 1) < 5e-3 and abs(out[1] * 2 * k ** 3 - 1) < 5e-3
    sim = BDHI2DOracle(o64, "Quasi2D", 64.0, 1.0, 1.0, 0.0, 0.1)
    assert list(sim.cells) == [80, 80, 1] and sim.support == 9
    g = o64.ibm_spread(np.array([[0.3, -0.2, 0.0]]), np.ones((1, 2)), sim.L, [1, 1, 0], sim.cells, sim.kernel)
    assert abs(g[..., 0].sum() * 0.8 * 0.8 - 1.0) < 2e-4      # the 9-node window holds the Gaussian to ~1e-4
    with pytest.raises(RuntimeError):
        BDHI2DOracle(o64, "True2D", 64.0, -1.0, 1.0, 0.0, 0.1)
