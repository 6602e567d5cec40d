"""GPU: ParticleGroup — index iterators through the C ABI (d_globalIndex of the traversal, d_index of the
integrators) against the oracle on the same subset, and membership that survives ParticleData::sortParticles."""
import numpy as np
import pytest
import torch

from util import lattice_positions

pytestmark = pytest.mark.gpu


def test_group_pairforces_and_integrator(hip, o32):
    n, L, rc = 6000, 20.0, 2.5
    pos = lattice_positions(n, L, seed=3, jitter=0.12, ntypes=2)          # pos.w = type 0 / 1
    pd = hip.ParticleData(n, seed=4)
    pd.setPos(pos)
    box = hip.Box(L)
    pot = hip.Potential.LJ()
    pot.setPotParameters(0, 0, pot.InputPairParameters(rc, 1.0, 1.0, False))
    pgA = hip.ParticleGroup(pd, hip.ParticleGroup.Type(1), "type1")
    members = np.nonzero(pos[:, 3] == 1)[0]
    assert pgA.getNumberParticles() == len(members)
    assert np.array_equal(pgA.getIndexIterator().cpu().numpy(), members)
    # forces among the members only, written at the members' indices
    pf = hip.PairForces(pgA, box, pot, algo=9)      # EXACT: the reference's summation order (bitwise against the oracle)
    pd.getForce("write").zero_()
    pf.sum(force=True)
    got = pd.getForce().cpu().numpy()
    pd.getForce("write").zero_()
    hip.PairForces(pgA, box, pot).sum(force=True)   # AUTO: the tile kernel, rounding-level differences
    auto = pd.getForce().cpu().numpy()
    assert np.abs(auto - got).max() <= 1e-5 * np.abs(got).max() and np.all(auto[np.setdiff1d(np.arange(n), np.nonzero(pos[:, 3] == 1)[0])] == 0)
    sub = np.ascontiguousarray(pos[members])
    cd, oL, oper = o32.celllist_create_grid(L, 1, rc)
    cl = o32.celllist_build(sub, oL, oper, cd)
    ref, _, _ = o32.lj_transverse_celllist(cl, L, 1, pot.table, 1, len(sub))
    assert np.array_equal(got[members].view(np.uint32), ref.view(np.uint32))
    others = np.setdiff1d(np.arange(n), members)
    assert np.all(got[others] == 0)
    # sortParticles moves everybody; the group follows its ids
    ids_before = pd.id.cpu().numpy()[pgA.getIndexIterator().cpu().numpy()]
    pd.sortParticles()
    idx = pgA.getIndexIterator().cpu().numpy()
    assert np.array_equal(np.sort(pd.id.cpu().numpy()[idx]), np.sort(ids_before))
    assert np.all(pd.getPos().cpu().numpy()[idx, 3] == 1)
    # an integrator on the group moves the members only
    before = pd.getPos().cpu().numpy().copy()
    par = hip.VerletNVT.GronbechJensen.Parameters(temperature=1.0, dt=0.002, friction=1.0)
    integ = hip.VerletNVT.GronbechJensen(pgA, par)
    integ.addInteractor(hip.PairForces(pgA, box, pot))
    for _ in range(3):
        integ.forwardTime()
    after = pd.getPos().cpu().numpy()
    moved = np.abs(after[:, :3] - before[:, :3]).max(axis=1) > 0
    assert moved[idx].all() and not moved[np.setdiff1d(np.arange(n), idx)].any()
    assert np.isfinite(after).all()


def test_group_selectors(hip):
    pd = hip.ParticleData(100, seed=1)
    pd.setPos(np.zeros((100, 4), np.float32))
    assert hip.ParticleGroup(pd).getIndexIterator() is None and hip.ParticleGroup(pd).getNumberParticles() == 100
    g = hip.ParticleGroup(pd, hip.ParticleGroup.IDRange(10, 19))
    assert g.getIndexIterator().cpu().tolist() == list(range(10, 20))
    g = hip.ParticleGroup(pd, [5, 3, 99])
    assert g.getIndexIterator().cpu().tolist() == [3, 5, 99]
    assert g.getPropertyIterator(pd.id).cpu().tolist() == [3, 5, 99]
