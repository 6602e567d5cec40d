"""Oracle pins for Hydro::ICM.  The reference ships no test for the incompressible ICM, so the restatement is pinned on the
physics its header promises (ICM.cuh:28-31, :164-171):
  * Stokes limit: a steadily pulled particle reaches the velocity F * getSelfMobility() (+-1 %... the header's own margin),
    in a frame where the total fluid momentum is removed every step;
  * a shear wave decays with the Crank-Nicolson rate (1 - a)/(1 + a), a = dt nu k_eff^2 / 2, and stays divergence free;
  * with removeTotalMomentum = false the fluid momentum is conserved by diffusion and advection;
  * at T > 0 the fluid velocity obeys equipartition, <v^2> = kT/(rho dV) * (2/3)(1 - 1/N) per component (projected field).
"""
import math

import numpy as np
import pytest

from oracle.icm import ICMOracle


def test_stokes_limit_mobility(o64):
    L, a, visc, rho, dt = 32.0, 1.0, 5.0, 0.2, 0.05          # momentum diffuses across the box in L^2 rho/eta ~ 40 time units... steady state needs more
    icm = ICMOracle(o64, L, 0.0, visc, rho, dt, hydrodynamicRadius=a)
    pos = np.zeros((1, 4))
    pos[0, :3] = [0.3, -1.1, 2.2]

    def pull(p):
        f = np.zeros((len(p), 4))
        f[:, 0] = 1.0
        return f
    vel = []
    for s in range(4000):
        p0 = pos.copy()
        icm.forwardTime(pos, pull)
        vel.append((pos[0, 0] - p0[0, 0]) / dt)
    v_inf = np.mean(vel[-200:])
    assert abs(v_inf / icm.getSelfMobility() - 1) < 0.02, (v_inf, icm.getSelfMobility())
    assert abs(vel[-1] - vel[-200]) < 1e-2 * v_inf                # converged; what is left is the +-1 % dependence on the position in the cell
    assert np.abs(icm.v.reshape(-1, 3).mean(axis=0)).max() < 1e-14   # removeTotalMomentum


def test_shear_wave_decay_and_incompressibility(o64):
    L, visc, rho, dt, n = 16.0, 0.7, 1.3, 0.1, 16
    x = (np.arange(n) + 0.5) * (L / n)
    v0 = np.zeros((n, n, n, 3))
    v0[..., 1] = 0.01 * np.sin(2 * math.pi * x / L)[None, None, :]      # v_y(x): transverse, divergence free, advection-free
    icm = ICMOracle(o64, L, 0.0, visc, rho, dt, cells=[n, n, n], initial_velocity=v0)
    pos = np.zeros((1, 4))
    amp0 = np.abs(icm.v[..., 1]).max()
    icm.forwardTime(pos)
    keff2 = (2 / (L / n) * math.sin(math.pi / n)) ** 2
    a = dt * visc / rho * keff2 / 2
    assert abs(np.abs(icm.v[..., 1]).max() / amp0 - (1 - a) / (1 + a)) < 1e-9
    h = L / n
    v = icm.v
    div = (v[..., 0] - np.roll(v[..., 0], 1, axis=2) + v[..., 1] - np.roll(v[..., 1], 1, axis=1) + v[..., 2] - np.roll(v[..., 2], 1, axis=0)) / h
    assert np.abs(div).max() < 1e-14


def test_momentum_conservation_without_removal(o64):
    rng = np.random.default_rng(0)
    L, n = 12.0, 12
    v0 = rng.normal(0, 0.05, (n, n, n, 3)) + np.array([0.3, -0.2, 0.1])
    icm = ICMOracle(o64, L, 0.0, 0.5, 1.0, 0.02, cells=[n, n, n], initial_velocity=v0, removeTotalMomentum=False)
    pos = np.zeros((1, 4))
    p0 = icm.v.reshape(-1, 3).mean(axis=0)
    for _ in range(5):
        icm.forwardTime(pos)
    assert np.abs(icm.v.reshape(-1, 3).mean(axis=0) - p0).max() < 1e-13


@pytest.mark.slow
def test_equipartition(o32):
    rng = np.random.default_rng(1)
    L, n, T, visc, rho, dt = 16.0, 16, 1.2, 1.0, 1.0, 0.2
    icm = ICMOracle(o32, L, T, visc, rho, dt, cells=[n, n, n], noise_fn=lambda nc: rng.normal(0, 1, (6, nc)).astype(np.float32))
    pos = np.zeros((1, 4), np.float32)
    acc, cnt = 0.0, 0
    for s in range(300):
        icm.forwardTime(pos)
        if s >= 100:
            acc += float((icm.v.astype(np.float64) ** 2).mean())
            cnt += 1
    dV = (L / n) ** 3
    expect = T / (rho * dV) * (2.0 / 3.0)        # two of three modes per wave vector survive the projection
    assert abs(acc / cnt / expect - 1) < 0.05, (acc / cnt, expect)
