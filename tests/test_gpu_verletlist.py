"""GPU parity, SURVEY row a15: the Verlet list (uammd_verletlist_* + uammd_lj_transverse_verletlist) vs the oracle
restatement (oracle/src/lj.c K7/K8 + oracle/verletlist.py host logic).

Bars: the list itself (numberNeighbours, every entry neighbourList[k*N+i], capacity, sorted order) is integer work ->
EXACT.  Forces keep the reference's summation order (list order) and the oracle's FMA placement -> expected
bit-identical; asserted <= 1e-5 max|F| and the differing-word count is asserted 0.  The rebuild schedule
(which step rebuilds) is a function of float comparisons on bit-identical trajectories -> EXACT.
"""
import math

import numpy as np
import pytest
import torch

from util import lattice_positions

pytestmark = pytest.mark.gpu


def _pot(hip, rc, ntypes=1):
    pot = hip.Potential.LJ()
    for ti in range(ntypes):
        for tj in range(ti, ntypes):
            pot.setPotParameters(ti, tj, pot.InputPairParameters(rc * (1.0 - 0.04 * ti), 1.0 + 0.03 * tj, 1.0 + 0.1 * ti + 0.2 * tj,
                                                                 (ti + tj) % 2 == 1))
    return pot


def _compare_lists(h, ref):
    n = len(ref.numberNeighbours)
    assert h["stride"] == n
    assert h["maxNeighboursPerParticle"] == ref.maxNeighboursPerParticle
    assert np.array_equal(h["index"], ref.groupIndex)
    assert np.array_equal(h["numberNeighbours"], ref.numberNeighbours)
    m = ref.maxNeighboursPerParticle
    a = h["neighbourList"].reshape(m + 1, n)
    b = ref.neighbourList.reshape(m + 1, n)
    k = np.arange(m + 1)[:, None]
    valid = k < ref.numberNeighbours[None, :]
    assert np.array_equal(a[valid], b[valid])
    assert np.array_equal(h["sortPos"], ref.sortPos)


@pytest.mark.parametrize("n,L,rc,periodic,ntypes", [
    (6400, 20.0, 2.5, (1, 1, 1), 1),                 # liquid density: capacity must grow 32 -> 64 -> 96
    (3000, (14.0, 17.0, 21.0), 2.5, (1, 1, 1), 3),   # non cubic, 3 types
    (2000, (16.0, 16.0, 9.0), 2.5, (1, 1, 0), 1),    # non periodic z, collapsed z dimension (9/2.7 = 3 cells -> 1)
    (500, 30.0, 1.2, (1, 1, 1), 1),                  # dilute: stays at capacity 32
])
def test_list_and_forces_match_oracle(hip, o32, n, L, rc, periodic, ntypes):
    from oracle.verletlist import VerletListOracle
    pos = lattice_positions(n, L, seed=77, jitter=0.15, ntypes=ntypes)
    Lb = np.broadcast_to(np.asarray(L, np.float32), (3,))
    for k in range(3):
        if not periodic[k]:
            pos[:, k] = np.clip(pos[:, k], -Lb[k] / 2 + 1e-3, Lb[k] / 2 - 1e-3)
    box = hip.Box(L, periodic)
    pot = _pot(hip, rc, ntypes)
    rcp = float(pot.getCutOff())
    ref = VerletListOracle(o32)
    ref.update(pos, L, [int(p) for p in periodic], rcp)
    vl = hip.VerletList()
    d_pos = torch.from_numpy(pos).cuda()
    vl.update(box, rcp, d_pos)
    h = vl.to_host()
    _compare_lists(h, ref)
    if n == 6400:
        assert ref.maxNeighboursPerParticle == 96
    if n == 500:
        assert ref.maxNeighboursPerParticle == 32
    # the particle is its own neighbour, the reference keeps it
    own = (h["neighbourList"].reshape(-1, n)[:, :] == np.arange(n)[None, :]) & \
          (np.arange(h["maxNeighboursPerParticle"] + 1)[:, None] < h["numberNeighbours"][None, :])
    assert np.all(own.sum(axis=0) == 1)
    for fev in [(True, False, False), (True, True, True)]:
        f = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
        e = torch.zeros(n, dtype=torch.float32, device="cuda") if fev[1] else None
        v = torch.zeros(n, dtype=torch.float32, device="cuda") if fev[2] else None
        vl.transverse_lj(pot.device_table(), pot.ntypes, box, f, e, v)
        torch.cuda.synchronize()
        rf, re, rv = ref.lj_forces(L, [int(p) for p in periodic], pot.table, pot.ntypes, want_energy=fev[1], want_virial=fev[2])
        fmax = np.abs(rf).max()
        assert np.abs(f.cpu().numpy() - rf).max() <= 1e-5 * fmax
        assert np.count_nonzero(f.cpu().numpy().view(np.uint32) != rf.view(np.uint32)) == 0
        if fev[1]:
            assert np.count_nonzero(e.cpu().numpy().view(np.uint32) != re.view(np.uint32)) == 0
            assert np.count_nonzero(v.cpu().numpy().view(np.uint32) != rv.view(np.uint32)) == 0


def test_forces_equal_celllist_forces_up_to_rounding(hip):
    """Same pairs as the CellList path, different summation order -> 1e-5 of max|F|."""
    n, L, rc = 8000, 21.5443469, 2.5
    pos = lattice_positions(n, L, seed=5, jitter=0.12)
    box, pot = hip.Box(L), _pot(hip, rc)
    d_pos = torch.from_numpy(pos).cuda()
    vl, cl = hip.VerletList(), hip.CellList()
    vl.update(box, rc, d_pos)
    cl.update(box, rc, d_pos)
    f1 = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    f2 = torch.zeros_like(f1)
    vl.transverse_lj(pot.device_table(), 1, box, f1)
    cl.transverse_lj(pot.device_table(), 1, box, f2)
    assert (f1 - f2).abs().max().item() <= 1e-5 * f2.abs().max().item()


def test_rebuild_schedule_and_trajectory_match_oracle(hip, o32):
    """VerletNVT::GronbechJensen at T = 0 (deterministic) with PairForces<LJ, VerletList>: the oracle loop with the restated
    host logic must rebuild at the same steps, and positions stay bit-identical."""
    from oracle.verletlist import VerletListOracle
    n, L, rc, dt, nsteps = 4096, 17.3, 2.5, 0.005, 40
    pos = lattice_positions(n, L, seed=21, jitter=0.05)
    rng = np.random.default_rng(9)
    vel = rng.normal(0, 1.2, (n, 3)).astype(np.float32)
    # reference loop
    ref = VerletListOracle(o32)
    par = o32.lj_params(rc, 1.0, 1.0)
    p, v = pos.copy(), vel.copy()

    def ref_forces():
        ref.handlePosWriteRequested()              # the integrator asked for pos with write access
        ref.update(p, L, 1, rc)
        return ref.lj_forces(L, 1, par, 1)[0]
    f = ref_forces()
    ref_rebuild_steps = []
    for s in range(1, nsteps + 1):
        o32.verletnvt_gj(1, p, v, f, dt, 1.0, 0.0, s, 1234)
        before = ref.rebuilds
        f = ref_forces()
        if ref.rebuilds != before:
            ref_rebuild_steps.append(s)
        o32.verletnvt_gj(2, p, v, f, dt, 1.0, 0.0, s, 1234)
    # product
    pd = hip.ParticleData(n, seed=1)
    pd.setPos(pos)
    pd.getVel("write").copy_(torch.from_numpy(vel).cuda())
    box, pot = hip.Box(L), _pot(hip, rc)
    par_i = hip.VerletNVT.GronbechJensen.Parameters(temperature=0.0, dt=dt, friction=1.0, initVelocities=False)
    integ = hip.VerletNVT.GronbechJensen(pd, par_i)
    integ.seed = 1234
    vl = hip.VerletList(pd)
    integ.addInteractor(hip.PairForces(pd, box, pot, nl=vl))
    steps = []
    for s in range(1, nsteps + 1):
        before = vl.rebuilds
        integ.forwardTime()
        if vl.rebuilds != before and s > 1:
            steps.append(s)
        elif s == 1 and vl.rebuilds > 1:
            steps.append(1)
    torch.cuda.synchronize()
    assert len(ref_rebuild_steps) >= 2, "the configuration must exercise the drift rebuild"
    assert steps == ref_rebuild_steps
    assert np.array_equal(pd.getPos().cpu().numpy().view(np.uint32), p.view(np.uint32))
    assert vl.getNumberOfStepsSinceLastUpdate() == ref.getNumberOfStepsSinceLastUpdate()


def test_wrapper_logic(hip):
    """VerletList::needsRebuild (VerletList.cuh:184-200) and VerletListBase::needsRebuild (:158-175)."""
    n, L, rc = 1000, 12.0, 2.5
    pos = torch.from_numpy(lattice_positions(n, L, seed=2, jitter=0.1)).cuda()
    box = hip.Box(L)
    vl = hip.VerletList()
    vl.update(box, rc, pos)
    assert vl.rebuilds == 1 and vl.getNumberOfStepsSinceLastUpdate() == 0
    vl.update(box, rc, pos)                      # nothing changed, no pos write: the wrapper does not even reach the base
    assert vl.rebuilds == 1 and vl.getNumberOfStepsSinceLastUpdate() == 0
    vl._handle_pos_write()
    vl.update(box, rc, pos)                      # positions "written" but nobody moved: drift check says no rebuild
    assert vl.rebuilds == 1 and vl.getNumberOfStepsSinceLastUpdate() == 1
    moved = pos.clone()
    moved[17, 0] += 0.04 * rc + 1e-3             # one particle past the threshold (1.08 rc - rc)/2
    vl._handle_pos_write()
    vl.update(box, rc, moved)
    assert vl.rebuilds == 2 and vl.getNumberOfStepsSinceLastUpdate() == 0
    vl.update(box, 2.0, moved)                   # new cut-off
    assert vl.rebuilds == 3
    vl.update(hip.Box(L + 1.0), 2.0, moved)      # new box
    assert vl.rebuilds == 4
    vl.setCutOffMultiplier(1.2)
    vl.update(hip.Box(L + 1.0), 2.0, moved)
    assert vl.rebuilds == 5
    vl._handle_reorder()
    vl.update(hip.Box(L + 1.0), 2.0, moved)
    assert vl.rebuilds == 6
    vl.setCutOffMultiplier(1.0)                  # threshold 0 <= 1e-6: every update rebuilds (VerletListBase.cuh:181-183)
    vl.update(hip.Box(L + 1.0), 2.0, moved)
    vl._handle_pos_write()
    vl.update(hip.Box(L + 1.0), 2.0, moved)
    assert vl.rebuilds == 8
    with pytest.raises(RuntimeError):
        vl.update(box, [2.5, 2.5, 3.0], pos)
