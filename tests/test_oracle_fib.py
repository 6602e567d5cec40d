"""Oracle pins for BDHI::FIB.  The reference's tests (test/BDHI/FIB/test.bash) are plotted statistical runs against
FIB::getSelfMobility (FIB.cuh:152-163, "expect +-1 %"); the checks here are their automatable core:
  * selfMobilityCubicBox: a pulled particle moves with getSelfMobility() (staggered grid: +-1 % over positions in a cell)
  * the Stokes operator leaves a divergence-free field with zero mean, and J = dV S^T (adjointness of spread / interpolate)
  * selfDiffusionCubicBox: <dr^2>/(6 dt) = kT M at T > 0 (statistical, slow)
"""
import numpy as np
import pytest

from oracle.fib import FIBOracle
from oracle.oracle import _p


def test_self_mobility_cubic_box(o64):
    rng = np.random.default_rng(0)
    for L, a in ((32.0, 1.0), (64.0, 1.3)):
        fib = FIBOracle(o64, L, 0.0, 1.7, 0.01, hydrodynamicRadius=a)
        assert abs(float(fib.hydrodynamicRadius) / a - 1) < 0.08          # "the most approximate one" the FFT-friendly grid allows
        M0 = fib.getSelfMobility()
        vs = []
        for _ in range(40):
            pos = np.zeros((1, 4))
            pos[0, :3] = rng.uniform(-L / 2, L / 2, 3)
            f = np.zeros((1, 4))
            f[0, 0] = 1.0
            p0 = pos.copy()
            fib.forwardTime(pos, f)
            vs.append((pos[0, :3] - p0[0, :3]) / fib.dt)
        vs = np.array(vs)
        assert abs(vs[:, 0].mean() / M0 - 1) < 0.01, (L, vs[:, 0].mean(), M0)
        assert np.abs(vs[:, 0] / M0 - 1).max() < 0.03
        assert np.abs(vs[:, 1:]).max() < 0.02 * M0


def test_stokes_operator_and_adjointness(o64):
    rng = np.random.default_rng(1)
    L, n = 16.0, 50
    fib = FIBOracle(o64, L, 0.0, 1.0, 0.01, cells=[18, 16, 20])
    pos = np.zeros((n, 4))
    pos[:, :3] = rng.uniform(-L, L, (n, 3))
    force = np.zeros((n, 4))
    force[:, :3] = rng.normal(0, 1, (n, 3))
    v = fib.fluid_velocity(pos, force)                                   # [nz][ny][nx][3], face centred
    h = fib.L / fib.cells
    # the faces carrying v_c(i) and v_c(i-1) bracket the point where the projector's discrete divergence vanishes
    div = ((v[..., 0] - np.roll(v[..., 0], 1, axis=2)) / h[0] + (v[..., 1] - np.roll(v[..., 1], 1, axis=1)) / h[1] +
           (v[..., 2] - np.roll(v[..., 2], 1, axis=0)) / h[2])
    assert np.abs(div).max() <= 1e-12 * np.abs(v).max() / h.min()        # discrete divergence on the staggered grid
    assert np.abs(v.reshape(-1, 3).mean(axis=0)).max() <= 1e-14
    # J v . F = dV * (S F) . v: interpolate a random field with the midpoint kernel (euler mode, dt = 1, posOld = 0)
    field = rng.normal(0, 1, v.shape)
    p = pos.copy()
    old = np.zeros_like(pos)
    o64.lib.oracle_fib_midpoint_step(2, _p(p), _p(old), _p(field), n, _p(fib.L), _p(fib.cells), o64.creal(fib.h), o64.creal(1.0))
    Jv = p[:, :3]
    sf = np.zeros_like(field)
    o64.lib.oracle_fib_spread(_p(pos), _p(force), n, _p(fib.L), _p(fib.cells), o64.creal(fib.h), _p(sf))
    assert abs((Jv * force[:, :3]).sum() - np.prod(h) * (sf * field).sum()) <= 1e-10 * abs((Jv * force[:, :3]).sum())


@pytest.mark.slow
def test_self_diffusion(o32):
    """selfDiffusionCubicBox: free particles at T > 0 diffuse with D = kT M (noise: numpy normals, the reference's are cuRAND's)."""
    rng = np.random.default_rng(2)
    L, T, visc, dt, n, steps = 16.0, 1.3, 1.0, 0.05, 200, 400
    fib = FIBOracle(o32, L, T, visc, dt, hydrodynamicRadius=1.0, noise_fn=lambda nc: rng.normal(0, 1, (6, nc)).astype(np.float32))
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(-L / 2, L / 2, (n, 3))
    acc = 0.0
    for _ in range(steps):
        p0 = pos.astype(np.float64)
        fib.forwardTime(pos)
        acc += ((pos[:, :3].astype(np.float64) - p0[:, :3]) ** 2).sum()
    D = acc / (steps * n) / (6 * dt)
    assert abs(D / (T * fib.getSelfMobility()) - 1) < 0.04, (D, T * fib.getSelfMobility())
