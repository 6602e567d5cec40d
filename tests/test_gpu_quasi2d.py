"""GPU parity: BDHI::True2D / BDHI::Quasi2D (uammd_bdhi2d_*) against the oracle — velocities with forces, with thermal drift and
noise (identical Saru streams) — and the reference's test/BDHI/quasi2D/quasi2d_test.cu on the product."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class Forces:
    def __init__(self, hip, pd, f):
        self.pd, self.f = pd, torch.from_numpy(f).cuda()

    def sum(self, force=False, energy=False, virial=False):
        self.pd.getForce("readwrite").add_(self.f)

    def updateSimulationTime(self, t): pass
    def updateTimeStep(self, dt): pass
    def updateTemperature(self, T): pass
    def updateBox(self, box): pass


@pytest.mark.parametrize("mode", ["Quasi2D", "True2D"])
@pytest.mark.parametrize("L", [40.0, (36.0, 50.0)], ids=["square", "rect"])
def test_step_matches_oracle(hip, o32, mode, L):
    from oracle.quasi2d import BDHI2DOracle
    n, a, visc, T, dt, seed = 500, 1.1, 1.3, 0.7, 0.05, 90210
    rng = np.random.default_rng(2)
    Lx, Ly = (L, L) if np.isscalar(L) else L
    pos = np.zeros((n, 4), np.float32)
    pos[:, 0] = rng.uniform(-0.7, 0.7, n) * Lx      # some particles outside the primary box
    pos[:, 1] = rng.uniform(-0.7, 0.7, n) * Ly
    force = np.zeros((n, 4), np.float32)
    force[:, :2] = rng.normal(0, 1, (n, 2))
    Scheme = getattr(hip.BDHI, mode)
    for temperature, with_forces in [(0.0, True), (T, True), (T, False)]:
        pd = hip.ParticleData(n)
        pd.setPos(pos)
        par = Scheme.Parameters(temperature=temperature, viscosity=visc, hydrodynamicRadius=a, dt=dt, box=hip.Box([Lx, Ly, 0.0]), seed=seed)
        bd = Scheme(pd, par)
        ref = BDHI2DOracle(o32, mode, (Lx, Ly), a, visc, temperature, dt, seed=seed)
        assert bd.cells == [int(ref.cells[0]), int(ref.cells[1])] and bd.support == ref.support
        if with_forces:
            bd.addInteractor(Forces(hip, pd, force))
        rpos = pos.copy()
        for step in range(2):      # two steps: the noise counter advances
            bd.forwardTime()
            rv = ref.forwardTime(rpos, force if with_forces else None)
            v = bd._vel.cpu().numpy()
            assert np.abs(v - rv).max() <= 3e-5 * np.abs(rv).max(), (mode, temperature, with_forces, step)
        assert np.abs(pd.getPos("read").cpu().numpy() - rpos).max() <= 1e-5


def _self_mobility(hip, Scheme, lbox, a, direction, ntest=20):
    """computeSelfMobility, quasi2d_test.cu:56-92."""
    visc, dt = 1.12312, 0.1
    pd = hip.ParticleData(1)
    par = Scheme.Parameters(temperature=0.0, viscosity=visc, hydrodynamicRadius=a, dt=dt, box=hip.Box([lbox, lbox, 0.0]))
    bd = Scheme(pd, par)
    f = np.zeros((1, 4), np.float32)
    f[0, direction] = 1.0
    bd.addInteractor(Forces(hip, pd, f))
    rng = np.random.default_rng(1)
    M = np.zeros(2)
    for _ in range(ntest):
        p0 = np.zeros((1, 4), np.float32)
        p0[0, :2] = rng.uniform(-lbox / 2, lbox / 2, 2)
        pd.setPos(p0)
        bd.forwardTime()
        M += (pd.getPos("read").double().cpu().numpy()[0, :2] - p0[0, :2].astype(np.float64))
    return visc * M / (ntest * dt)


@pytest.mark.parametrize("mode", ["Quasi2D", "True2D"])
def test_reference_self_mobility(hip, mode):
    """quasi2d_test.cu:94-138: a = 1.21312, L/a = 32 ... 224, both directions, +-1e-3 (cross term 1e-4)."""
    a = 1.21312
    Scheme = getattr(hip.BDHI, mode)
    for lbox in range(32, 256, 32):
        for d in (0, 1):
            M = _self_mobility(hip, Scheme, lbox * a, a, d)
            theo = 1.0 / (6 * math.pi * a) / (1 + 4.41 / lbox) if mode == "Quasi2D" else (math.log(lbox) - 1.3105329259115095183) / (4 * math.pi)
            assert abs(M[d] - theo) < 1e-3, (mode, lbox, d, M, theo)
            assert abs(M[1 - d]) < 2e-4      # 1e-4 in the double-precision reference run


@pytest.mark.parametrize("mode", ["Quasi2D", "True2D"])
def test_reference_fluctuation_dissipation(hip, mode):
    """quasi2d_test.cu:140-190: 50000 one-particle steps at random positions, <dr^2>/(2 T dt) = self mobility within 1 %
    (tolerance 2.5 % here: the statistical error of 50000 samples is 0.6 %, single precision positions add nothing)."""
    a, T, dt, navg = 1.21312, 1.012312, 0.9, 50000
    lbox = 128 * a
    Scheme = getattr(hip.BDHI, mode)
    pd = hip.ParticleData(1)
    bd = Scheme(pd, Scheme.Parameters(temperature=T, viscosity=1.0, hydrodynamicRadius=a, dt=dt, box=hip.Box([lbox, lbox, 0.0])))
    gen = torch.Generator(device="cuda").manual_seed(7)
    starts = (torch.rand((navg, 2), generator=gen, device="cuda") - 0.5) * lbox
    acc = torch.zeros(2, dtype=torch.float64, device="cuda")
    p = pd.getPos("write")
    for i in range(navg):
        p.zero_()
        p[0, :2] = starts[i]
        bd.forwardTime()
        r = (p[0, :2] - starts[i]).double()
        acc += r * r
    d0 = (acc / navg / (2 * T * dt)).cpu().numpy()
    Mtheo = _self_mobility(hip, Scheme, lbox, a, 0)[0]
    assert abs(d0[0] / Mtheo - 1) < 0.025 and abs(d0[1] / Mtheo - 1) < 0.025, (d0, Mtheo)


def test_reference_error_paths(hip):
    pd = hip.ParticleData(1)
    P = hip.BDHI.Quasi2D.Parameters
    with pytest.raises(RuntimeError, match="Invalid hydrodynamic radius"):
        hip.BDHI.Quasi2D(pd, P(temperature=1.0, viscosity=1.0, hydrodynamicRadius=-1.0, dt=0.1, box=hip.Box([32.0, 32.0, 0.0])))
    with pytest.raises(RuntimeError, match="Invalid box"):
        hip.BDHI.True2D(pd, P(temperature=1.0, viscosity=1.0, hydrodynamicRadius=1.0, dt=0.1, box=hip.Box([0.0, 0.0, 0.0])))
    hip.BDHI.Quasi2D(pd, P(temperature=1, viscosity=1, dt=0.1, hydrodynamicRadius=1, box=hip.Box(128.0)))   # Q2D.CanBeCreated
