"""GPU parity: cell-list construction through the C ABI vs the oracle — bit exact.

Contract: hash[], index[] (sort order), sortPos[], cellStart-VALID_CELL, cellEnd are
identical to the stable-radix-sort reference for both build paths of the library (counting sort
and rocPRIM radix sort).
"""
import numpy as np
import pytest
import torch

from util import canon_cell_tables, lattice_positions

pytestmark = pytest.mark.gpu

CASES = [
    # N, L, rc, periodic, note
    (4096, 16.0, 2.5, (1, 1, 1), "c2small golden config"),
    (20000, (30.0, 26.0, 40.0), 2.5, (1, 1, 1), "non cubic"),
    (3000, (30.0, 30.0, 9.0), 2.5, (1, 1, 1), "z collapses to one cell"),
    (5000, (40.0, 40.0, 40.0), 2.5, (1, 0, 1), "non periodic y"),
    (1000, (7.0, 7.0, 7.0), 2.5, (1, 1, 1), "all dims collapse: single cell"),
    (100000, 50.0, 2.5, (1, 1, 1), "C2"),
]


def _build_both(hip, o32, pos, L, rc, periodic, force_radix):
    box = hip.Box(L, periodic)
    cd, ubox = hip.CellList.create_update_grid(box, rc)
    ocd, oL, oper = o32.celllist_create_grid(box.boxSize, [int(p) for p in box.periodic], rc)
    assert list(ocd) == list(cd)
    assert np.array_equal(oL, ubox.boxSize) and [bool(p) for p in oper] == list(ubox.periodic)
    ref = o32.celllist_build(pos, oL, oper, ocd)
    cl = hip.CellList()
    cl.set_option("force_radix", int(force_radix))
    d_pos = torch.from_numpy(pos).cuda()
    cl.update_grid(d_pos, ubox, cd)
    got = cl.to_host()
    return ref, got


@pytest.mark.parametrize("force_radix", [0, 1])
@pytest.mark.parametrize("case", CASES, ids=[c[4] for c in CASES])
def test_celllist_bit_exact(hip, o32, case, force_radix):
    n, L, rc, periodic, _ = case
    Lmin = np.min(np.broadcast_to(L, (3,)))
    pos = lattice_positions(n, L, seed=1234, jitter=0.3)
    # a few particles outside the primary box in periodic directions and exactly on the faces
    pos[:7, 0] += np.float32(np.broadcast_to(L, (3,))[0])
    pos[7, 0] = -np.float32(np.broadcast_to(L, (3,))[0]) / 2
    pos[8, 2] = np.float32(np.broadcast_to(L, (3,))[2]) / 2 if periodic[2] else pos[8, 2]
    L3 = np.broadcast_to(np.asarray(L, dtype=np.float32), (3,))
    for k in range(3):
        if not periodic[k]:  # a non periodic box has no image to fold an outside particle into
            pos[:, k] = np.clip(pos[:, k], -L3[k] / 2 + 0.01, L3[k] / 2 - 0.01)
    ref, got = _build_both(hip, o32, pos, L, rc, periodic, force_radix)
    assert got["validCell"] == ref["validCell"] == n
    assert np.array_equal(got["hash"], ref["hash"])
    assert np.array_equal(got["index"], ref["index"])
    assert np.array_equal(got["sortPos"].view(np.uint32), ref["sortPos"].view(np.uint32))
    gs, ge = canon_cell_tables(got)
    rs, re = canon_cell_tables(ref)
    assert np.array_equal(gs, rs) and np.array_equal(ge, re)
    assert Lmin > 0


def test_epoch_and_rebuild(hip, o32):
    """VALID_CELL follows CellListBase::updateCurrentValidCell (N, 2N, 3N, ... reset on resize) and a
    rebuild after moving particles equals a fresh oracle build."""
    n, L, rc = 4096, 16.0, 2.5
    pos = lattice_positions(n, L, seed=7, jitter=0.2)
    box = hip.Box(L)
    cd, ubox = hip.CellList.create_update_grid(box, rc)
    cl = hip.CellList()
    state = np.array([-1, -1], np.int64)
    rng = np.random.default_rng(3)
    for it in range(4):
        v, _ = o32.next_valid_cell(n, state)
        d_pos = torch.from_numpy(pos).cuda()
        cl.update_grid(d_pos, ubox, cd)
        got = cl.to_host()
        assert got["validCell"] == v == n * (it + 1)
        ref = o32.celllist_build(pos, ubox.boxSize, [int(p) for p in ubox.periodic], cd, valid_cell=v)
        assert np.array_equal(got["index"], ref["index"])
        gs, ge = canon_cell_tables(got)
        rs, re = canon_cell_tables(ref)
        assert np.array_equal(gs, rs) and np.array_equal(ge, re)
        pos[:, :3] += rng.normal(0, 0.5, (n, 3)).astype(np.float32)
    # changing N resets the epoch
    pos2 = pos[:1000].copy()
    cl.update_grid(torch.from_numpy(pos2).cuda(), ubox, cd)
    assert cl.to_host()["validCell"] == 1000


def test_sorter_reference_test(hip):
    """test/utils/ParticleSorter.cu:23-46: 163840 reversed keys come back ascending."""
    import ctypes as C
    from uammd_amd._lib import check, load
    lib = load()
    n = 163840
    keys = torch.arange(n - 1, -1, -1, dtype=torch.int32, device="cuda")
    vals = torch.arange(n, dtype=torch.int32, device="cuda")
    end_bit = int(n).bit_length()
    check(lib.uammd_sort_pairs(C.c_void_p(keys.data_ptr()), C.c_void_p(vals.data_ptr()), n, end_bit, None))
    torch.cuda.synchronize()
    assert torch.equal(keys.cpu(), torch.arange(n, dtype=torch.int32))
    assert torch.equal(vals.cpu(), torch.arange(n - 1, -1, -1, dtype=torch.int32))


def test_sort_pairs_stable_and_end_bit(hip, o32):
    """Stability and the end_bit contract: only bits [0,end_bit) take part."""
    import ctypes as C
    from uammd_amd._lib import check, load
    lib = load()
    rng = np.random.default_rng(5)
    for n, end_bit in [(1, 8), (37, 3), (5000, 6), (200000, 11), (200000, 32)]:
        keys = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
        if end_bit < 32:
            keys &= np.uint32((1 << end_bit) - 1) | np.uint32(0xF0000000)  # junk above end_bit must be ignored
        vals = np.arange(n, dtype=np.int32)
        rk, rv = o32.stable_sort_pairs(keys, vals, end_bit)
        dk = torch.from_numpy(keys.view(np.int32)).cuda()
        dv = torch.from_numpy(vals).cuda()
        check(lib.uammd_sort_pairs(C.c_void_p(dk.data_ptr()), C.c_void_p(dv.data_ptr()), n, end_bit, None))
        torch.cuda.synchronize()
        assert np.array_equal(dv.cpu().numpy(), rv)
        assert np.array_equal(dk.cpu().numpy().view(np.uint32), rk)


@pytest.mark.parametrize("n,L,note", [(300000, 250.0, "2^21 keys: 256 scan workgroups of 8192 keys"),
                                       (40000, (20.0, 20.0, 640.0), "256 cells along z only: a 2^24-key envelope falls back to the radix build"),
                                       (70000, (45.0, 40.0, 160.0), "2^18 keys, most of them not cells")])
def test_counting_build_key_scan_sizes(hip, n, L, note):
    """k_key_scan (the counting build's one-launch scan of the key counters) over key spaces of other sizes than the bench's: the
    tables equal the radix build's, build after build on the same handle (the per-workgroup totals are tagged with a generation, never
    reset)."""
    pos = lattice_positions(n, L, seed=21, jitter=0.4)
    box = hip.Box(L)
    cd, ubox = hip.CellList.create_update_grid(box, 2.5)
    cl, ref_cl = hip.CellList(), hip.CellList()
    ref_cl.set_option("force_radix", 1)
    rng = np.random.default_rng(9)
    for it in range(3):
        d_pos = torch.from_numpy(pos).cuda()
        cl.update_grid(d_pos, ubox, cd)
        ref_cl.update_grid(d_pos, ubox, cd)
        a, b = cl.to_host(), ref_cl.to_host()
        for k in ("hash", "index"):
            assert np.array_equal(a[k], b[k]), (it, k)
        assert np.array_equal(a["sortPos"].view(np.uint32), b["sortPos"].view(np.uint32))
        sa, ea = canon_cell_tables(a)
        sb, eb = canon_cell_tables(b)
        assert np.array_equal(sa, sb) and np.array_equal(ea, eb)
        pos = (pos + rng.normal(0, 0.3, pos.shape).astype(np.float32)).astype(np.float32)
