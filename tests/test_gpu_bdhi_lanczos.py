"""GPU parity: BDHI::Lanczos (uammd_rpy_nbody_mdot, uammd_rpy_lanczos_bdw) vs the oracle.  The all-pairs product keeps the
reference's j order and the oracle's FMA placement: expected bit-identical, asserted <= 1e-6 of max|Mv|.  The Lanczos result
stops at a relative change <= tolerance: two implementations agree to a few times that."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sizes", ["equal", "different"])
def test_rpy_mdot_and_bdw(hip, o32, sizes):
    from oracle.lanczos import LanczosOracle
    from oracle.pse import rpy_nbody_mdot
    rng = np.random.default_rng(6)
    n, visc, tol = 1500, 1.1, 1e-3
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(-12, 12, (n, 3))
    pos[1, :3] = pos[0, :3]                                   # coincident pair: r = 0 branch between different particles
    radius = rng.uniform(0.4, 1.1, n).astype(np.float32) if sizes == "different" else None
    rh = -1.0 if sizes == "different" else 0.8
    pd = hip.ParticleData(n, seed=5)
    pd.setPos(pos)
    if radius is not None:
        pd.getRadius("write").copy_(torch.from_numpy(radius).cuda())
    par = hip.BDHI.Lanczos.Parameters(temperature=1.0, viscosity=visc, hydrodynamicRadius=rh, tolerance=tol, dt=0.01)
    lz = hip.BDHI.Lanczos(pd, par)
    f4 = np.zeros((n, 4), np.float32)
    f4[:, :3] = rng.normal(0, 1, (n, 3))
    pd.getForce("write").copy_(torch.from_numpy(f4).cuda())
    MF = torch.full((n, 3), 3.0, dtype=torch.float32, device="cuda")     # set() overwrites
    lz.computeMF(MF)
    expect = rpy_nbody_mdot(o32, pos, f4, visc, rh, radius)
    got = MF.cpu().numpy()
    assert np.abs(got - expect).max() <= 1e-6 * np.abs(expect).max()
    print("differing words:", np.count_nonzero(got.view(np.uint32) != expect.view(np.uint32)), "of", got.size)
    noise = rng.normal(0, 1, (n, 3)).astype(np.float32)
    BdW = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    lz.computeBdW(BdW, noise=torch.from_numpy(noise).cuda())
    ref = LanczosOracle(np.float32).run(lambda v: rpy_nbody_mdot(o32, pos, v.reshape(n, 3), visc, rh, radius).reshape(-1),
                                        noise.reshape(-1), tol).reshape(n, 3)
    assert np.linalg.norm(BdW.cpu().numpy() - ref) <= 5 * tol * np.linalg.norm(ref)
    assert 1 <= lz.lastIterations <= 60
    if sizes == "equal":
        assert abs(lz.getSelfMobility() - 1 / (6 * math.pi * visc * rh)) < 1e-12


def test_euler_maruyama_lanczos_free_diffusion(hip):
    """BDHI::EulerMaruyama<BDHI::Lanczos> on far-apart particles: <dx^2> = 2 T M0 dt per component (statistical, 3 %)."""
    n, visc, a, T, dt = 4096, 1.0, 1.0, 0.7, 0.02
    rng = np.random.default_rng(1)
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = (np.stack(np.meshgrid(*[np.arange(16)] * 3, indexing="ij"), -1).reshape(-1, 3) * 400.0).astype(np.float32)
    pd = hip.ParticleData(n, seed=8)
    pd.setPos(pos)
    par = hip.BDHI.Lanczos.Parameters(temperature=T, viscosity=visc, hydrodynamicRadius=a, tolerance=1e-3, dt=dt)
    integ = hip.BDHI.EulerMaruyama(pd, par, Method=hip.BDHI.Lanczos)
    integ.forwardTime()
    d = pd.getPos().cpu().numpy()[:, :3] - pos[:, :3]
    m0 = 1 / (6 * math.pi * visc * a)
    # hydrodynamic coupling at r = 400 a is 3a/(4r) ~ 2e-3: negligible against the 3 % statistical bar
    assert abs((d ** 2).mean() / (2 * T * m0 * dt) - 1) < 0.03
