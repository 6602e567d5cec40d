"""GPU parity: BDHI::Cholesky (uammd_bdhi_cholesky_*: HIP matrix fill, rocBLAS symv / trmv, rocSOLVER potrf) vs the oracle,
the reference's Lanczos_Cholesky cross-mobility test in single precision, and BDHI::EulerMaruyama<Cholesky>."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _system(n, seed, two_sizes=True):
    rng = np.random.default_rng(seed)
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(-2, 2, (n, 3))
    radius = np.full(n, 0.38173, np.float32)
    if two_sizes:
        radius[0] = 1.89538
    return pos, radius


def test_mf_and_bdw_match_oracle(hip, o32):
    from oracle.pse import CholeskyOracle
    n, visc = 300, 1.2131
    pos, radius = _system(n, 3)
    rng = np.random.default_rng(4)
    force = np.zeros((n, 4), np.float32)
    force[:, :3] = rng.normal(0, 1, (n, 3))
    noise = rng.normal(0, 1, 3 * n).astype(np.float32)
    pd = hip.ParticleData(n)
    pd.setPos(pos)
    pd.getRadius("write").copy_(torch.from_numpy(radius))
    pd.getForce("write").copy_(torch.from_numpy(force))
    par = hip.BDHI.Cholesky.Parameters(temperature=1.0, viscosity=visc, hydrodynamicRadius=-1.0, dt=0.01)
    ch = hip.BDHI.Cholesky(pd, par, noise_fn=lambda: torch.from_numpy(noise).cuda())
    ref = CholeskyOracle(o32, -1.0, visc)
    MF = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    BdW = torch.zeros((n + 1, 3), dtype=torch.float32, device="cuda")
    ch.setup_step()
    ch.computeMF(MF)
    rMF = ref.computeMF(pos, force, radius)
    assert np.abs(MF.cpu().numpy() - rMF).max() <= 1e-5 * np.abs(rMF).max()
    ch.computeBdW(BdW)
    rB = ref.computeBdW(pos, noise, radius)
    assert np.abs(BdW[:n].cpu().numpy() - rB).max() <= 2e-4 * np.abs(rB).max()     # single-precision potrf of a 900 x 900 matrix
    # computeBdW consumed M: computeMF rebuilds it (BDHI_Cholesky.cu:200-209) and gives the same product
    MF2 = torch.zeros_like(MF)
    ch.computeMF(MF2)
    assert torch.equal(MF, MF2)
    # hydrodynamicRadius > 0 overrides the particle radii (:102-104)
    ch1 = hip.BDHI.Cholesky(pd, hip.BDHI.Cholesky.Parameters(viscosity=visc, hydrodynamicRadius=0.5))
    ch1.computeMF(MF2)
    r1 = CholeskyOracle(o32, 0.5, visc).computeMF(pos, force, radius)
    assert np.abs(MF2.cpu().numpy() - r1).max() <= 1e-5 * np.abs(r1).max()
    assert abs(ch1.getSelfMobility() - 1 / (6 * np.pi * visc * 0.5)) < 1e-12 and ch.getSelfMobility() == -1.0


def test_reference_cross_mobility(hip):
    """test/BDHI/Lanczos_Cholesky/test.bash in single precision, for BOTH modes: pull particle 0 (radius a2) among particles of
    radius a1 with BDHI::EulerMaruyama at T = 0; f(r), g(r) from one step's displacements vs RPY_differentSizes."""
    from tests.test_oracle_bdhi_cholesky import rpy_theory
    n, visc, dt = 2000, 1.2131, 10.0
    pos, radius = _system(n, 7)

    class Pull(hip.Interactor):
        def __init__(self, pd):
            self.pd = pd

        def sum(self, force=False, energy=False, virial=False):
            self.pd.getForce("readwrite")[0, 0] += 1.0

    for Method in (hip.BDHI.Cholesky, hip.BDHI.Lanczos):
        pd = hip.ParticleData(n)
        pd.setPos(pos)
        pd.getRadius("write").copy_(torch.from_numpy(radius))
        par = Method.Parameters(temperature=0.0, viscosity=visc, hydrodynamicRadius=-1.0, dt=dt, tolerance=1e-8)
        integ = hip.BDHI.EulerMaruyama(pd, par, Method=Method)
        integ.addInteractor(Pull(pd))
        integ.forwardTime()
        v = (pd.getPos("read").double().cpu().numpy()[:, :3] - pos[:, :3].astype(np.float64)) / dt
        wf = wg = 0.0
        for i in range(1, n):
            rij = (pos[i, :3] - pos[0, :3]).astype(np.float64)
            if abs(rij[0] * rij[1]) < 0.05:
                continue
            r = np.linalg.norm(rij)
            f_th, g_th = rpy_theory(r, radius[0], radius[i], visc)
            g = v[i, 1] / (rij[0] * rij[1])
            f = v[i, 0] - g * rij[0] * rij[0]
            wf = max(wf, abs((f - f_th) / f_th))
            # g changes sign near r^2 = a1^2 + a2^2, so its error is measured against the size of the tensor, f/r^2
            wg = max(wg, abs(g - g_th) * r * r / f_th)
        # 1e-7 in the double-precision reference run; positions of magnitude 2 advanced by ~0.1 in float32 leave ~1e-5
        assert wf < 5e-4 and wg < 5e-3, (Method.__name__, wf, wg)


def test_factor_reproduces_mobility(hip, o32):
    """B B^T = M: the factor applied to the unit vectors (small N)."""
    from oracle.pse import rpy_dense
    n = 24
    pos, radius = _system(n, 11, two_sizes=False)
    pd = hip.ParticleData(n)
    pd.setPos(pos)
    par = hip.BDHI.Cholesky.Parameters(temperature=1.0, viscosity=1.0, hydrodynamicRadius=0.4, dt=0.01)
    cols = []
    BdW = torch.zeros((n + 1, 3), dtype=torch.float32, device="cuda")
    for k in range(3 * n):
        e = torch.zeros(3 * n, dtype=torch.float32, device="cuda")
        e[k] = 1.0
        ch = hip.BDHI.Cholesky(pd, par, noise_fn=lambda e=e: e)
        ch.computeBdW(BdW)
        cols.append(BdW[:n].cpu().numpy().reshape(-1).copy())
    B = np.stack(cols, axis=1).astype(np.float64)
    M = rpy_dense(o32, pos, None, 0.4, 1.0).astype(np.float64)
    assert np.abs(B @ B.T - M).max() <= 2e-5 * np.abs(M).max()
