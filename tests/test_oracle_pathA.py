"""CPU tests (no GPU): the oracle's path-A restatement against the known answers the reference ships and against
independent brute-force computations.  Reference tests for this path: the sorter test
(test/utils/ParticleSorter.cu:23-46) and the LJ contact force (examples/uammd_as_a_library/wrapper.py:22-32); the
reference has NO cell-list / traversal unit test, so those are pinned by brute-force cross-checks."""
import numpy as np
import pytest

from util import canon_cell_tables, lattice_positions


def test_sorter_reference_test(o32):
    """ParticleSorter.cu:23-46: 163840 reversed keys -> ascending; max_hash = n."""
    n = 163840
    keys = np.arange(n, dtype=np.uint32)[::-1].copy()
    k, v = o32.stable_sort_pairs(keys, np.arange(n), o32.sort_end_bit(n))
    assert np.array_equal(k, np.arange(n)) and np.array_equal(keys[v], np.arange(n))


def test_sort_is_stable_and_respects_end_bit(o32):
    rng = np.random.default_rng(0)
    keys = rng.integers(0, 64, 5000).astype(np.uint32) | np.uint32(0xFF000000)
    k, v = o32.stable_sort_pairs(keys, np.arange(5000), 6)
    order = np.argsort(keys & 63, kind="stable")
    assert np.array_equal(v, order) and np.array_equal(k, keys[order])
    assert o32.sort_end_bit(0) == 0 and o32.sort_end_bit(1) == 1 and o32.sort_end_bit(0xFFFFFFFF) == 32


def test_morton_hash(o32):
    def ref(x, y, z):
        h = 0
        for b in range(10):
            h |= ((x >> b) & 1) << (3 * b) | ((y >> b) & 1) << (3 * b + 1) | ((z >> b) & 1) << (3 * b + 2)
        return h
    for c in [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (5, 9, 3), (1023, 1023, 1023), (42, 42, 0), (1024 + 3, 2, 1)]:
        assert o32.morton_hash(*c) == ref(c[0] & 1023, c[1] & 1023, c[2] & 1023)


@pytest.mark.parametrize("L,rc,expect", [(50.0, 2.5, [20, 20, 20]), (107.7217345, 2.5, [43, 43, 43]),
                                          ((30.0, 30.0, 9.0), 2.5, [12, 12, 1]), (7.0, 2.5, [1, 1, 1]),
                                          ((10.0, 10.1, 12.49), 2.5, [4, 4, 4])])
def test_create_update_grid(o32, L, rc, expect):
    """CellList::createUpdateGrid: C truncation in float; <= 3 cells collapse to 1 (CellList.cuh:116-124).  L=50 -> 20^3,
    L=107.7217345 -> 43^3 as SURVEY §8 states."""
    cd, _, _ = o32.celllist_create_grid(L, 1, rc)
    assert list(cd) == expect


def test_infinite_box(o32):
    cd, Lo, per = o32.celllist_create_grid([np.finfo(np.float32).max, 20.0, 20.0], 1, 2.5)
    # numeric_limits<real>::max() is not isinf(): the dimension stays periodic (CellList.cuh:104-115, Box.cuh:25-31)
    assert list(cd) == [64, 8, 8] and Lo[0] == np.float32(160.0) and list(per) == [1, 1, 1]
    cd, Lo, per = o32.celllist_create_grid([np.inf, 20.0, 20.0], 1, 2.5)
    assert list(cd) == [64, 8, 8] and Lo[0] == np.float32(160.0) and list(per) == [0, 1, 1]


@pytest.mark.parametrize("L,periodic", [(16.0, (1, 1, 1)), ((30.0, 26.0, 40.0), (1, 1, 1)), ((30.0, 30.0, 9.0), (1, 1, 1)),
                                        ((40.0, 40.0, 40.0), (1, 0, 1))])
def test_celllist_tables_against_bruteforce(o32, L, periodic):
    n, rc = 5000, 2.5
    pos = lattice_positions(n, L, seed=3, jitter=0.3)
    L3 = np.broadcast_to(np.asarray(L, np.float32), (3,))
    for k in range(3):
        if not periodic[k]:
            pos[:, k] = np.clip(pos[:, k], -L3[k] / 2 + 0.01, L3[k] / 2 - 0.01)
    pos[:5, 0] += L3[0]
    cd, oL, oper = o32.celllist_create_grid(L, periodic, rc)
    cl = o32.celllist_build(pos, oL, oper, cd)
    assert cl["error"] == 0
    assert sorted(cl["index"].tolist()) == list(range(n))                       # a permutation
    assert np.array_equal(cl["sortPos"], pos[cl["index"]])
    assert np.all(np.diff(cl["hash"].astype(np.int64)) >= 0)                    # sorted by Morton key
    same = cl["hash"][1:] == cl["hash"][:-1]
    assert np.all(np.diff(cl["index"])[same] > 0)                               # stable inside a key
    ic = o32.cell_of(cl["sortPos"], oL, oper, cd)
    start, end = canon_cell_tables(cl)
    for c in range(int(np.prod(cd))):
        m = np.nonzero(ic == c)[0]
        if len(m) == 0:
            assert start[c] == -1
        else:
            assert start[c] == m[0] and end[c] == m[-1] + 1 and np.all(np.diff(m) == 1)
    # the cell of a position: independent float64 computation agrees except within an ulp of a cell face
    p64 = pos[cl["index"]].astype(np.float64)[:, :3]
    Ld = oL.astype(np.float64)
    folded = p64 - np.floor(p64 / Ld + 0.5) * Ld * np.array(oper)
    c64 = np.floor((folded + Ld / 2) / (Ld / cd)).astype(np.int64) % np.maximum(cd, 1)
    lin = c64[:, 0] + cd[0] * (c64[:, 1] + cd[1] * c64[:, 2])
    assert (lin != ic).mean() < 1e-3


def test_valid_cell_epoch(o32):
    st = np.array([-1, -1], np.int64)
    assert o32.next_valid_cell(1000, st) == (1000, True)
    assert o32.next_valid_cell(1000, st) == (2000, False)
    assert o32.next_valid_cell(1000, st) == (3000, False)
    assert o32.next_valid_cell(999, st) == (999, True)          # N changed -> reset
    st = np.array([-1, -1], np.int64)
    n = 2 ** 30
    assert o32.next_valid_cell(n, st) == (n, True)
    assert o32.next_valid_cell(n, st) == (2 * n, False)
    assert o32.next_valid_cell(n, st)[1] is True               # (counter+2)*N would overflow uint -> reset


def test_lj_contact_force(o32):
    """wrapper.py:22-32: two particles at r = sigma: force -/+ 24 eps/sigma; half pair energy 0; virial F.r12."""
    par = o32.lj_params(2.5, 1.0, 1.0)
    p2 = np.array([[0, 0, 0, 0], [1, 0, 0, 0]], np.float32)
    f, e, v = o32.lj_transverse_nbody(p2, 0, 0, par, 1, True, True, True)
    assert np.array_equal(f[:, 0], [-24.0, 24.0]) and np.all(f[:, 1:] == 0)
    assert np.all(e == 0) and np.array_equal(v, [-24.0, -24.0])
    # the energy is HALF the pair energy (Potential.cuh:64): at r = 2^(1/6) sigma, E_pair = -eps
    p2[1, 0] = 2 ** (1 / 6)
    _, e, _ = o32.lj_transverse_nbody(p2, 0, 0, par, 1, False, True, False)
    assert np.allclose(e, -0.5, atol=1e-6)
    # shift: energy(rc) = 0
    par_s = o32.lj_params(2.5, 1.0, 1.0, True)
    p2[1, 0] = np.float32(2.5) - np.float32(1e-6)
    _, e, _ = o32.lj_transverse_nbody(p2, 0, 0, par_s, 1, False, True, False)
    assert np.all(np.abs(e) < 1e-6)


@pytest.mark.parametrize("L", [16.0, (22.0, 18.0, 26.0), (30.0, 30.0, 9.0)])
def test_lj_celllist_equals_all_pairs(o32, L):
    """Cell-list traversal vs all pairs (float, different summation order) and vs the float64 yardstick."""
    rc = 2.5
    n = int(0.8 * np.prod(np.broadcast_to(L, (3,))))
    pos = lattice_positions(n, L, seed=5, jitter=0.12)
    par = o32.lj_params(rc, 1.0, 1.0)
    cd, oL, oper = o32.celllist_create_grid(L, 1, rc)
    cl = o32.celllist_build(pos, oL, oper, cd)
    f, e, v = o32.lj_transverse_celllist(cl, L, 1, par, 1, n, True, True, True)
    fn, en, vn = o32.lj_transverse_nbody(pos, L, 1, par, 1, True, True, True)
    fd = o32.lj_nbody_f64(pos, L, 1, rc, 1.0, 1.0)
    scale = np.abs(fd).max()
    assert np.abs(f[:, :3] - fn[:, :3]).max() <= 1e-5 * scale   # 1e-5 of max|F|
    assert np.abs(f[:, :3] - fd).max() <= 1e-5 * scale
    assert np.abs(e - en).max() <= 1e-5 * np.abs(en).max() and np.abs(v - vn).max() <= 1e-5 * np.abs(vn).max()
    # Newton's third law over the whole system
    assert np.abs(fd.sum(0)).max() <= 1e-9 * scale * n


def test_lj_multitype_table(o32):
    n, L, rc = 1500, 12.5, 2.5
    pos = lattice_positions(n, L, seed=9, jitter=0.1, ntypes=3)
    tbl = np.zeros((9, 4), np.float32)
    for ti in range(3):
        for tj in range(ti, 3):
            p = o32.lj_params(rc * (1 - 0.05 * ti), 1 + 0.03 * tj, 1 + 0.1 * ti + 0.2 * tj, (ti + tj) % 2 == 1)
            tbl[ti + 3 * tj] = p
            tbl[tj + 3 * ti] = p
    cd, oL, oper = o32.celllist_create_grid(L, 1, rc)
    cl = o32.celllist_build(pos, oL, oper, cd)
    f, _, _ = o32.lj_transverse_celllist(cl, L, 1, tbl, 3, n)
    fn, _, _ = o32.lj_transverse_nbody(pos, L, 1, tbl, 3)
    assert np.abs(f - fn).max() <= 1e-5 * np.abs(fn).max()
    # symmetric table -> momentum conserved up to float rounding of O(sqrt(n)) terms
    assert np.abs(f[:, :3].astype(np.float64).sum(0)).max() <= 1e-6 * np.abs(fn).max() * np.sqrt(n)
