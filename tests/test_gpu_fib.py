"""GPU parity: BDHI::FIB (uammd_fib_*) against the oracle step by step (same forces, same fluid noise), the self mobility the
reference's test.bash plots against FIB::getSelfMobility, and self diffusion with the built-in Saru noise."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class Forces:
    def __init__(self, pd, f):
        self.pd, self.f = pd, torch.from_numpy(f).cuda()

    def sum(self, force=False, energy=False, virial=False):
        self.pd.getForce("readwrite").add_(self.f)

    def updateSimulationTime(self, t): pass
    def updateTimeStep(self, dt): pass
    def updateTemperature(self, T): pass
    def updateBox(self, box): pass


@pytest.mark.parametrize("cells,L", [([32, 32, 32], 32.0), ([18, 16, 20], (17.0, 16.0, 21.0)), ([15, 9, 21], 14.0)],
                         ids=["cube", "noncubic", "odd"])
def test_steps_match_oracle(hip, o32, cells, L):
    from oracle.fib import FIBOracle
    n, visc, dt, T = 400, 1.4, 0.02, 0.8
    rng = np.random.default_rng(3)
    Lv = np.broadcast_to(np.asarray(L, np.float32), (3,))
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(-0.7, 0.7, (n, 3)) * Lv          # some particles outside the primary box
    pos[:, 3] = rng.integers(0, 3, n)
    force = np.zeros((n, 4), np.float32)
    force[:, :3] = rng.normal(0, 1, (n, 3))
    ncells = int(np.prod(cells))
    for temperature in (0.0, T):
        pd = hip.ParticleData(n)
        pd.setPos(pos)
        fib = hip.BDHI.FIB(pd, hip.BDHI.FIB.Parameters(temperature=temperature, viscosity=visc, dt=dt, box=hip.Box(L), cells=cells))
        fib.addInteractor(Forces(pd, force))
        ref = FIBOracle(o32, L, temperature, visc, dt, cells=cells)
        assert fib.cells == [int(c) for c in ref.cells] and abs(fib.hydrodynamicRadius - float(ref.hydrodynamicRadius)) < 1e-6
        rpos = pos.copy()
        for step in range(3):
            noise = rng.normal(0, 1, (6, ncells)).astype(np.float32)
            dn = torch.from_numpy(noise).cuda()
            fib.set_noise(dn)
            fib.forwardTime()
            ref.forwardTime(rpos, force, noise=noise)
            got = pd.getPos("read").cpu().numpy()
            disp = np.abs(rpos[:, :3] - pos[:, :3]).max()
            assert np.abs(got - rpos).max() <= 2e-5 * disp + 2e-6, (temperature, step)
            assert np.array_equal(got[:, 3], pos[:, 3])              # the type column rides along


def test_self_mobility(hip):
    """selfMobilityCubicBox of test/BDHI/FIB/test.bash: velocity of a pulled particle vs FIB::getSelfMobility(), +-1 %."""
    L, a, visc, dt = 64.0, 1.0, 1.0, 0.01
    pd = hip.ParticleData(1)
    fib = hip.BDHI.FIB(pd, hip.BDHI.FIB.Parameters(temperature=0.0, viscosity=visc, dt=dt, box=hip.Box(L), hydrodynamicRadius=a))
    f = np.zeros((1, 4), np.float32)
    f[0, 0] = 1.0
    fib.addInteractor(Forces(pd, f))
    rng = np.random.default_rng(0)
    vs = []
    for _ in range(64):
        p0 = np.zeros((1, 4), np.float32)
        p0[0, :3] = rng.uniform(-L / 2, L / 2, 3)
        pd.setPos(p0)
        fib.forwardTime()
        vs.append((pd.getPos("read").double().cpu().numpy()[0, :3] - p0[0, :3].astype(np.float64)) / dt)
    vs = np.array(vs)
    M0 = fib.getSelfMobility()
    assert abs(vs[:, 0].mean() / M0 - 1) < 0.01 and np.abs(vs[:, 0] / M0 - 1).max() < 0.03, (vs[:, 0].mean(), M0)
    assert np.abs(vs[:, 1:]).max() < 0.02 * M0
    with pytest.raises(RuntimeError, match="not both"):
        hip.BDHI.FIB(pd, hip.BDHI.FIB.Parameters(viscosity=1.0, dt=dt, box=hip.Box(L), hydrodynamicRadius=a, cells=[32, 32, 32]))
    with pytest.raises(RuntimeError, match="either the hydrodynamic radius or the number of cells"):
        hip.BDHI.FIB(pd, hip.BDHI.FIB.Parameters(viscosity=1.0, dt=dt, box=hip.Box(L)))


def test_self_diffusion_with_builtin_noise(hip):
    """selfDiffusionCubicBox: free particles at T > 0, <dr^2>/(6 dt) = kT M (Saru noise of the library), 3 %."""
    L, T, visc, dt, n, steps = 32.0, 1.3, 1.0, 0.05, 4000, 300
    rng = np.random.default_rng(2)
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(-L / 2, L / 2, (n, 3))
    pd = hip.ParticleData(n)
    pd.setPos(pos)
    fib = hip.BDHI.FIB(pd, hip.BDHI.FIB.Parameters(temperature=T, viscosity=visc, dt=dt, box=hip.Box(L), hydrodynamicRadius=1.0, seed=99))
    acc = torch.zeros((), dtype=torch.float64, device="cuda")
    for _ in range(steps):
        p0 = pd.getPos("read")[:, :3].double().clone()
        fib.forwardTime()
        acc += ((pd.getPos("read")[:, :3].double() - p0) ** 2).sum()
    D = float(acc) / (steps * n) / (6 * dt)
    assert abs(D / (T * fib.getSelfMobility()) - 1) < 0.03, (D, T * fib.getSelfMobility())
