"""GPU parity: the triply periodic Poisson Interactor (uammd_poisson_*) against the oracle, stage by stage and end to end,
plus the reference's SingleSimulationTest in single precision."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _system(n, L, seed):
    rng = np.random.default_rng(seed)
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(-0.5, 0.5, (n, 3)) * L
    q = rng.normal(0, 1, n).astype(np.float32)
    q -= q.mean()
    return pos, q


def _make(hip, pos, q, L, eps, gw, tol, split):
    pd = hip.ParticleData(len(pos))
    pd.setPos(pos)
    pd.getCharge("write").copy_(torch.from_numpy(q))
    par = hip.Poisson.Parameters(box=hip.Box(L), epsilon=eps, gw=gw, tolerance=tol, split=split)
    return pd, hip.Poisson(pd, par)


@pytest.mark.parametrize("n,L,gw,tol,split", [(2000, 32.0, 0.5, 1e-4, 0.8), (3000, (30.0, 36.0, 42.0), 0.4, 1e-5, 1.0),
                                              (500, 20.0, 0.6, 1e-3, -1.0)], ids=["cube", "noncubic", "nosplit"])
def test_sum_and_field_match_oracle(hip, o32, n, L, gw, tol, split):
    from oracle.poisson import PoissonOracle
    eps = 1.7
    Lmax = L if np.isscalar(L) else max(L)
    pos, q = _system(n, np.asarray(L, np.float32), 5)
    pd, poisson = _make(hip, pos, q, L, eps, gw, tol, split)
    ref = PoissonOracle(o32, L, eps, gw, tol, split)
    # the construction heuristics are host code on both sides: identical
    assert poisson.cells == [int(c) for c in ref.cells] and poisson.support == ref.support
    if split > 0:
        assert poisson.nearFieldCutOff == float(ref.nearFieldCutOff) and poisson.nTable == ref.ntable
    rf, re = np.zeros((n, 4), np.float32), np.zeros(n, np.float32)
    ref.sum(pos, q, rf, re, force=True, energy_flag=True)
    pd.getForce("write").zero_()
    pd.getEnergy("write").zero_()
    poisson.sum(force=True, energy=True)
    torch.cuda.synchronize()
    f, e = pd.getForce("read").cpu().numpy(), pd.getEnergy("read").cpu().numpy()
    assert np.abs(f - rf).max() <= 2e-5 * np.abs(rf).max()
    assert np.abs(e - re).max() <= 2e-5 * np.abs(re).max()
    assert np.all(f[:, 3] == 0)
    # the global-atomic spread gives the same forces as the tile-owned one
    poisson.set_option("atomic_spread", 1)
    pd.getForce("write").zero_()
    pd.getEnergy("write").zero_()
    poisson.sum(force=True, energy=True)
    fa = pd.getForce("read").cpu().numpy()
    assert np.abs(fa - f).max() <= 1e-5 * np.abs(f).max()
    poisson.set_option("atomic_spread", 0)
    # force only: the far field still adds the energy (reference behaviour), the near field does not
    rf2, re2 = np.zeros((n, 4), np.float32), np.zeros(n, np.float32)
    ref.sum(pos, q, rf2, re2, force=True, energy_flag=False)
    pd.getForce("write").zero_()
    pd.getEnergy("write").zero_()
    poisson.sum(force=True)
    e2 = pd.getEnergy("read").cpu().numpy()
    assert np.abs(e2 - re2).max() <= 2e-5 * np.abs(re2).max() + 1e-7
    if split > 0:
        assert np.abs(e2 - e).max() > 1e-4 * np.abs(e).max()
    # field and potential
    fp = poisson.computeFieldPotentialAtParticles().cpu().numpy()
    rfp = ref.computeFieldPotentialAtParticles(pos, q)
    assert np.abs(fp - rfp).max() <= 2e-5 * np.abs(rfp).max()
    assert np.abs(f[:, :3] - q[:, None] * fp[:, :3]).max() <= 1e-4 * np.abs(f).max()


def test_reference_single_simulation(hip):
    """test/Potentials/Poisson/TriplyPeriodic/test_poisson.cu:189-222 in single precision: three charges, L = 100, r = 2,
    tolerance 1e-7, gw = 1e-3, split 0.2: force and field on the first charge within 1e-3 of the free-space field."""
    L, r, tol, gw, split = 100.0, 2.0, 1e-7, 0.001, 0.2
    th = -math.exp(-r * r / (4.0 * gw * gw)) / (4 * math.pi * math.sqrt(math.pi) * gw * r) - math.erf(r / (2.0 * gw)) / (4 * math.pi * r * r)
    ori = np.random.default_rng(0).uniform(-0.5, 0.5, 3) * L
    pos = np.zeros((3, 4), np.float32)
    pos[:, :3] = np.array([[-r * 0.5, 0, 0], [r * 0.5, 0, 0], [r * 0.5, 0, 0]]) + ori
    q = np.array([1.0, -0.5, -0.5], np.float32)
    pd, poisson = _make(hip, pos, q, L, 1.0, gw, tol, split)
    pd.getForce("write").zero_()
    poisson.sum(force=True)
    f = pd.getForce("read").cpu().numpy()[0]
    fp = poisson.computeFieldPotentialAtParticles().cpu().numpy()[0]
    for v in (f, fp):
        assert abs(v[1]) < 1e-6 and abs(v[2]) < 1e-6 and v[0] > 0     # 1e-10 in the double-precision reference test
        assert abs(1.0 - abs(v[0] / th)) < 1e-3


def test_reference_error_paths(hip):
    with pytest.raises(ValueError, match="Kernel support .* is too large"):       # .cu:95-102
        hip.Poisson(hip.ParticleData(4), hip.Poisson.Parameters(box=hip.Box(8.0), epsilon=1.0, gw=0.5, tolerance=1e-6, split=0.2))
    with pytest.raises(ValueError, match="Near field cut off is too large"):    # .cu:111-116
        hip.Poisson(hip.ParticleData(4), hip.Poisson.Parameters(box=hip.Box(12.0), epsilon=1e-12, gw=0.3, tolerance=1e-3, split=1.0))
    with pytest.raises(RuntimeError, match="not implemented"):
        pd, p = _make(hip, *_system(10, 16.0, 1), 16.0, 1.0, 0.5, 1e-3, 1.0)
        p.sum(force=True, virial=True)


def test_large_system_properties(hip):
    """Size-independent properties at 2e5 charges: total force vanishes (Newton's third law through both the grid and
    the pair pass), sum_i q_i phi_i equals the summed energies, and two splittings agree."""
    n, L = 200000, 64.0
    pos, q = _system(n, L, 9)
    res = []
    for split in (0.8, 1.1):
        pd, poisson = _make(hip, pos, q, L, 1.0, 0.5, 1e-4, split)
        pd.getForce("write").zero_()
        pd.getEnergy("write").zero_()
        poisson.sum(force=True, energy=True)
        f = pd.getForce("read").double().cpu().numpy()
        e = pd.getEnergy("read").double().cpu().numpy()
        fp = poisson.computeFieldPotentialAtParticles().double().cpu().numpy()
        res.append((f, e))
        assert np.abs(f[:, :3].sum(axis=0)).max() <= 1e-4 * np.abs(f[:, :3]).sum(axis=0).max()
        assert abs((q * fp[:, 3]).sum() - e.sum()) <= 1e-5 * np.abs(e).sum()
    (f0, e0), (f1, e1) = res
    assert np.abs(f0 - f1).max() <= 2e-2 * np.abs(f0).max()   # tolerance 1e-4 -> h = 0.9 sigma: aliasing ~2e-3 per split
    assert abs(e0.sum() - e1.sum()) <= 2e-2 * abs(e0.sum())
