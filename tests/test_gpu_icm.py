"""GPU parity: Hydro::ICM (uammd_icm_*) against the oracle step by step (same initial fluid, forces, fluid noise and RFD
streams), the collocated velocity export, and the physical pins on the product (Stokes-limit mobility, equipartition)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class Spring:
    """Harmonic tether to the origin: a position-dependent force, so that evaluating it at q^{n+1/2} matters."""

    def __init__(self, pd, k):
        self.pd, self.k = pd, k

    def sum(self, force=False, energy=False, virial=False):
        p = self.pd.getPos("read")
        f = self.pd.getForce("readwrite")
        f[:, :3] -= self.k * p[:, :3]

    def updateSimulationTime(self, t): pass
    def updateTimeStep(self, dt): pass
    def updateTemperature(self, T): pass
    def updateBox(self, box): pass


@pytest.mark.parametrize("cells,L", [([16, 16, 16], 16.0), ([18, 12, 20], (17.0, 13.0, 21.0))], ids=["cube", "noncubic"])
@pytest.mark.parametrize("T,drift", [(0.0, False), (0.9, False), (0.9, True)], ids=["T0", "T", "T+drift"])
def test_steps_match_oracle(hip, o32, cells, L, T, drift):
    from oracle.icm import ICMOracle
    n, visc, rho, dt, seed, k = 300, 0.8, 1.3, 0.05, 4711, 0.3
    rng = np.random.default_rng(3)
    Lv = np.broadcast_to(np.asarray(L, np.float32), (3,))
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(-0.6, 0.6, (n, 3)) * Lv
    nx, ny, nz = cells
    v0 = rng.normal(0, 0.05, (nz, ny, nx, 3)).astype(np.float32)
    pd = hip.ParticleData(n)
    pd.setPos(pos)
    par = hip.Hydro.ICM.Parameters(temperature=T, viscosity=visc, density=rho, dt=dt, box=hip.Box(L), cells=cells, sumThermalDrift=drift,
                                   seed=seed)
    icm = hip.Hydro.ICM(pd, par)
    icm.addInteractor(Spring(pd, k))
    icm.setFluidVelocities(torch.from_numpy(v0).cuda())
    ref = ICMOracle(o32, L, T, visc, rho, dt, cells=cells, sumThermalDrift=drift, seed=seed, initial_velocity=v0)
    assert abs(icm.hydrodynamicRadius - float(ref.hydrodynamicRadius)) < 1e-6

    def spring(p):
        f = np.zeros_like(p)
        f[:, :3] = -np.float32(k) * p[:, :3]
        return f
    rpos = pos.copy()
    # the RFD drift is a difference of two window values 1e-4 a apart divided by that distance: single-precision rounding of
    # the windows (6e-8) comes back multiplied by ~1e4, on both sides
    tol = 1e-3 if drift else 3e-5
    for step in range(3):
        noise = rng.normal(0, 1, (6, nx * ny * nz)).astype(np.float32)
        icm.set_noise(torch.from_numpy(noise).cuda() if T > 0 else None)
        icm.forwardTime()
        ref.forwardTime(rpos, spring, noise=noise if T > 0 else None)
        gv = icm.getFluidVelocities(collocated=False).cpu().numpy()
        assert np.abs(gv - ref.v).max() <= tol * np.abs(ref.v).max(), (step,)
        got = pd.getPos("read").cpu().numpy()
        assert np.abs(got - rpos).max() <= tol * np.abs(rpos - pos).max() + 2e-6, (step,)
    col = icm.getFluidVelocities().cpu().numpy()
    assert np.abs(col - ref.getFluidVelocities()).max() <= tol * np.abs(ref.v).max()
    assert float(pd.getForce("read").abs().max()) == 0.0           # the corrector leaves the forces reset


def test_stokes_limit_mobility(hip):
    """ICM.cuh:28-31, :164-168: a steadily pulled particle settles at F * getSelfMobility() (+-2 %)."""
    L, a, visc, rho, dt = 32.0, 1.0, 5.0, 0.2, 0.05
    pd = hip.ParticleData(1)
    p0 = np.zeros((1, 4), np.float32)
    p0[0, :3] = [0.3, -1.1, 2.2]
    pd.setPos(p0)
    icm = hip.Hydro.ICM(pd, hip.Hydro.ICM.Parameters(temperature=0.0, viscosity=visc, density=rho, dt=dt, box=hip.Box(L), hydrodynamicRadius=a))

    class Pull:
        def sum(self, force=False, energy=False, virial=False):
            pd.getForce("readwrite")[0, 0] += 1.0
        def updateSimulationTime(self, t): pass
        def updateTimeStep(self, dt): pass
        def updateTemperature(self, T): pass
        def updateBox(self, box): pass
    icm.addInteractor(Pull())
    for _ in range(3800):
        icm.forwardTime()
    x0 = float(pd.getPos("read")[0, 0])
    for _ in range(200):
        icm.forwardTime()
    v = (float(pd.getPos("read")[0, 0]) - x0) / (200 * dt)
    assert abs(v / icm.getSelfMobility() - 1) < 0.02, (v, icm.getSelfMobility())
    assert float(icm.getFluidVelocities(collocated=False).mean(dim=(0, 1, 2)).abs().max()) < 1e-6     # removeTotalMomentum
    with pytest.raises(RuntimeError, match="fluid density"):
        hip.Hydro.ICM(pd, hip.Hydro.ICM.Parameters(viscosity=1.0, dt=dt, box=hip.Box(L), hydrodynamicRadius=a))
    with pytest.raises(RuntimeError, match="not both"):
        hip.Hydro.ICM(pd, hip.Hydro.ICM.Parameters(viscosity=1.0, density=1.0, dt=dt, box=hip.Box(L), hydrodynamicRadius=a, cells=[32, 32, 32]))


def test_equipartition_with_builtin_noise(hip):
    """<v^2> per component = (2/3) kT/(rho dV) for the projected fluctuating field, built-in Saru noise and thermal start."""
    L, n, T, visc, rho, dt = 32.0, 32, 1.2, 1.0, 1.0, 0.2
    pd = hip.ParticleData(1)
    icm = hip.Hydro.ICM(pd, hip.Hydro.ICM.Parameters(temperature=T, viscosity=visc, density=rho, dt=dt, box=hip.Box(L), cells=[n, n, n], seed=5))
    acc, cnt = 0.0, 0
    for s in range(300):
        icm.forwardTime()
        if s >= 100 and s % 5 == 0:
            acc += float((icm.getFluidVelocities(collocated=False).double() ** 2).mean())
            cnt += 1
    expect = T / (rho * (L / n) ** 3) * (2.0 / 3.0)
    assert abs(acc / cnt / expect - 1) < 0.03, (acc / cnt, expect)
