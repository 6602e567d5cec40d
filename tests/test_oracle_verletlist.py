"""CPU: the Verlet-list restatement (oracle/src/lj.c K7/K8, oracle/verletlist.py) against brute force.  The reference has
no test of its own for the list ("cell-list indexing and traversal have no reference test"), so the pin is
an O(N^2) enumeration of the same predicate plus the documented quirks (self in the list, `<=`, capacity 32 + 32k)."""
import numpy as np
import pytest

from util import lattice_positions


def _brute(pos, L, periodic, rc2):
    d = pos[None, :, :3].astype(np.float32) - pos[:, None, :3].astype(np.float32)
    for k in range(3):
        if periodic[k]:
            d[..., k] += np.floor(d[..., k] * np.float32(-1.0 / L[k]) + np.float32(0.5)) * np.float32(L[k])
    r2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    return r2 <= np.float32(rc2)


@pytest.mark.parametrize("L,periodic,n", [((12.0, 12.0, 12.0), (1, 1, 1), 1300), ((11.0, 14.0, 9.0), (1, 0, 1), 900)])
def test_list_is_the_set_of_pairs_within_the_verlet_radius(o32, L, periodic, n):
    from oracle.verletlist import VerletListOracle
    pos = lattice_positions(n, L, seed=4, jitter=0.2)
    for k in range(3):
        if not periodic[k]:
            pos[:, k] = np.clip(pos[:, k], -L[k] / 2 + 1e-3, L[k] / 2 - 1e-3)
    v = VerletListOracle(o32)
    rc = 2.5
    v.update(pos, L, periodic, rc)
    rcut = np.float32(np.float32(rc) * np.float32(1.08))
    sp = v.cl["sortPos"]
    near = _brute(sp, L, periodic, rcut * rcut)
    m = v.maxNeighboursPerParticle
    nl = v.neighbourList.reshape(m + 1, n)
    # fma-free brute force vs the oracle's dot = fma(z,z,fma(y,y,x*x)): pairs within 1 ulp of the radius may differ
    mism = 0
    for i in range(n):
        mine = set(nl[:v.numberNeighbours[i], i].tolist())
        want = set(np.nonzero(near[i])[0].tolist())
        assert i in mine                                   # the particle itself (r2 = 0 <= cutOff2)
        mism += len(mine ^ want)
        assert len(mine) == v.numberNeighbours[i]          # no duplicates
    assert mism <= 2
    assert v.numberNeighbours.max() < m and m % 32 == 0 and m >= 32
    assert v.numberNeighbours.max() >= m - 32 or m == 32   # grew only as far as needed


def test_capacity_growth_and_flag(o32):
    pos = lattice_positions(1000, 10.0, seed=1, jitter=0.1)       # density 1.0: ~83 neighbours within 2.7
    cd, gL, gp = o32.celllist_create_grid(10.0, 1, 2.7)
    cl = o32.celllist_build(pos, gL, gp, cd)
    flag, nl, nn = o32.verletlist_fill(cl, 10.0, 1, np.float32(2.7 * 2.7), 32, 1000)
    assert flag == 32                                             # atomicMax(nneigh) with nneigh == max at the abort
    flag, nl, nn = o32.verletlist_fill(cl, 10.0, 1, np.float32(2.7 * 2.7), 128, 1000)
    assert flag == 0 and 60 < nn.mean() < 110


def test_drift_check(o32):
    L, rc = 10.0, 2.5
    pos = lattice_positions(500, L, seed=3, jitter=0.1)
    cur = pos.copy()
    thr = np.float32((np.float32(1.08) * np.float32(rc) - np.float32(rc)) / 2.0)
    assert o32.verletlist_check_drift(cur, pos, thr, L, 1) == 0
    cur[3, 0] += thr * np.float32(1.01)
    cur[9, 2] -= np.float32(L)                                    # a whole box length is no drift (apply_pbc)
    cur[11, 1] += thr * np.float32(0.5)
    assert o32.verletlist_check_drift(cur, pos, thr, L, 1) == 1
    assert o32.verletlist_check_drift(cur, pos, thr, L, 0) == 2   # non periodic: the image shift counts


def test_host_logic_schedule(o32):
    from oracle.verletlist import VerletListOracle
    L, rc = 12.0, 2.5
    pos = lattice_positions(800, L, seed=2, jitter=0.1)
    v = VerletListOracle(o32)
    v.update(pos, L, 1, rc)
    assert v.rebuilds == 1 and v.getNumberOfStepsSinceLastUpdate() == 0
    v.update(pos, L, 1, rc)
    assert v.rebuilds == 1 and v.getNumberOfStepsSinceLastUpdate() == 0     # wrapper: nothing changed
    v.handlePosWriteRequested()
    v.update(pos, L, 1, rc)
    assert v.rebuilds == 1 and v.getNumberOfStepsSinceLastUpdate() == 1     # base: no drift
    p2 = pos.copy()
    p2[5, 1] += 0.11
    v.handlePosWriteRequested()
    v.update(p2, L, 1, rc)
    assert v.rebuilds == 2 and v.getNumberOfStepsSinceLastUpdate() == 0
    f_list = v.lj_forces(L, 1, o32.lj_params(rc, 1.0, 1.0), 1)[0]
    cd, gL, gp = o32.celllist_create_grid(L, 1, rc)
    cl = o32.celllist_build(p2, gL, gp, cd)
    f_cell = o32.lj_transverse_celllist(cl, L, 1, o32.lj_params(rc, 1.0, 1.0), 1, len(p2))[0]
    assert np.abs(f_list - f_cell).max() <= 1e-5 * np.abs(f_cell).max()
