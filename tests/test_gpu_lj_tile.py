"""GPU parity of the tile kernel (uammd_amd/csrc/lj_tile.hip: cell-pair tiles, distance test on the matrix pipe) vs the oracle.

Same pairs and the reference's per-pair arithmetic, another summation order: forces, energies and virials to the stated
tolerance of SURVEY 8d (|dF| <= 1e-5 max|F| per particle) instead of bits.  Grids: cubic, odd cell counts (a last unpaired
x cell), non cubic, minimal (4 x 3 x 3), collapsed z (2D-like), non periodic, unwrapped coordinates, several types, a dense
system (several 32-owner sub-tiles and more than one 512-candidate chunk), energy + virial, groups, ghosts.
"""
import numpy as np
import pytest
import torch

from test_gpu_lj import _check_force, _oracle, _run, _setup

pytestmark = pytest.mark.gpu
TILE = 8      # four waves per 2 x 2 x 2 brick of cells (AUTO)
TILE1 = 10    # one wave per pair of cells
SHAPES = pytest.mark.parametrize("TILE", [8, 10], ids=["brick4", "solo"])


@SHAPES
@pytest.mark.parametrize("L", [16.0, 27.7, (33.0, 22.0, 45.5), (10.5, 10.2, 11.0), (12.7, 12.7, 12.7)],
                         ids=["L16-6cells", "L27.7-11cells-odd", "noncubic", "minimal-4x4x4", "5cells"])
@pytest.mark.parametrize("outside", [False, True])
def test_tile_force_parity(hip, o32, L, outside, TILE):
    rc = 2.5
    vol = float(np.prod(np.broadcast_to(L, (3,))))
    n = int(0.8 * vol)
    pos, box, pot = _setup(hip, o32, n, L, rc, outside=outside)
    (ref, _, _), cd = _oracle(o32, pos, box, pot, rc)
    got, _, _ = _run(hip, pos, box, pot, rc, TILE)
    _check_force(got, ref, f"tile cellDim={list(cd)} outside={outside}", reordered=True)
    if not outside:
        fd = o32.lj_nbody_f64(pos, box.boxSize, [1, 1, 1], rc, 1.0, 1.0)
        err64 = np.abs(got[:, :3] - fd).max() / np.abs(fd).max()
        assert err64 <= 3e-5   # float cancellation of the f32 paths themselves (the exact kernels: 1e-5)


@SHAPES
def test_tile_energy_virial_multitype(hip, o32, TILE):
    n, L, rc = 12000, 25.0, 2.5
    pos, box, pot = _setup(hip, o32, n, L, rc, ntypes=3, seed=99)
    (rf, re, rv), cd = _oracle(o32, pos, box, pot, rc, (True, True, True))
    gf, ge, gv = _run(hip, pos, box, pot, rc, TILE, (True, True, True))
    _check_force(gf, rf, "tile multitype", reordered=True)
    assert np.abs(ge - re).max() <= 1e-5 * np.abs(re).max()
    assert np.abs(gv - rv).max() <= 1e-5 * np.abs(rv).max()


@SHAPES
@pytest.mark.parametrize("periodic", [(1, 1, 0), (0, 1, 1), (0, 0, 0)], ids=["z-open", "x-open", "open"])
def test_tile_non_periodic(hip, o32, periodic, TILE):
    n, L, rc = 9000, (24.0, 21.0, 26.0), 2.5
    pos, box, pot = _setup(hip, o32, n, L, rc, periodic=periodic, jitter=0.1)
    (ref, _, _), cd = _oracle(o32, pos, box, pot, rc)
    got, _, _ = _run(hip, pos, box, pot, rc, TILE)
    _check_force(got, ref, f"tile periodic={periodic} cellDim={list(cd)}", reordered=True)


@SHAPES
def test_tile_collapsed_periodic_dimension_falls_back(hip, o32, TILE):
    """z has fewer than 4 cells -> one cell (CellList.cuh:100-126).  With one PERIODIC cell the nearest image of a pair is not a
    function of the two centred coordinates the matrix prefilter multiplies: TILE refuses, AUTO takes the exact walk."""
    from uammd_amd._lib import UammdHipError
    n, L, rc = 3000, (30.0, 30.0, 6.0), 2.5
    pos, box, pot = _setup(hip, o32, n, L, rc)
    (ref, _, _), cd = _oracle(o32, pos, box, pot, rc)
    assert cd[2] == 1
    with pytest.raises(UammdHipError, match="tile kernel"):
        _run(hip, pos, box, pot, rc, TILE)
    got, _, _ = _run(hip, pos, box, pot, rc, 0)
    assert _check_force(got, ref, f"auto on collapsed z cellDim={list(cd)}") == 0


@SHAPES
def test_tile_open_slab_one_cell_thick(hip, o32, TILE):
    """One NON-periodic cell along z (a 2D-like open slab): the 9-cell neighbourhood, no images."""
    n, L, rc = 3000, (30.0, 30.0, 6.0), 2.5
    pos, box, pot = _setup(hip, o32, n, L, rc, periodic=(1, 1, 0))
    (ref, _, _), cd = _oracle(o32, pos, box, pot, rc)
    assert cd[2] == 1
    got, _, _ = _run(hip, pos, box, pot, rc, TILE)
    _check_force(got, ref, f"tile open slab cellDim={list(cd)}", reordered=True)


@SHAPES
def test_tile_dense_system_subtiles_and_chunks(hip, o32, TILE):
    """rho = 3.75: ~59 particles per cell -> ~118 owners per tile (4 sub-tiles of 32) and ~2100 candidates (5 chunks of 512)."""
    n, L, rc = 30000, 20.0, 2.5
    pos, box, pot = _setup(hip, o32, n, L, rc, jitter=0.02, seed=11)
    pot = hip.Potential.LJ()
    pot.setPotParameters(0, 0, pot.InputPairParameters(rc, 0.5, 1.0, False))
    (ref, _, _), cd = _oracle(o32, pos, box, pot, rc)
    got, _, _ = _run(hip, pos, box, pot, rc, TILE)
    # a jittered lattice: the ~500 pair forces of a particle cancel to a small total, so the bar is the SYSTEM's largest force
    # (the summation-order difference is ~1e-7 of the sum of the magnitudes)
    err = np.abs(got[:, :3] - ref[:, :3]).max() / np.abs(ref[:, :3]).max()
    print(f"[tile dense] max |dF| / max|F| = {err:.2e}")
    assert err <= 1e-5 and np.all(got[:, 3] == 0)


@SHAPES
def test_tile_sparse_and_empty_cells(hip, o32, TILE):
    n, L, rc = 300, 30.0, 2.5
    rng = np.random.default_rng(3)
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(-L / 2, L / 2, (n, 3))
    pos[:40, :3] = rng.normal(0, 0.7, (40, 3)) + np.array([L / 2 - 0.3, 0.0, -L / 2 + 0.2])   # a cluster on a box edge
    box = hip.Box(L)
    pot = hip.Potential.LJ()
    pot.setPotParameters(0, 0, pot.InputPairParameters(rc, 1.0, 1.0, True))
    (ref, re, _), cd = _oracle(o32, pos, box, pot, rc, (True, True, False))
    gf, ge, _ = _run(hip, pos, box, pot, rc, TILE, (True, True, False))
    fmax = np.abs(ref[:, :3]).max()
    assert np.abs(gf[:, :3] - ref[:, :3]).max() <= 1e-5 * fmax
    assert np.abs(ge - re).max() <= 1e-5 * np.abs(re).max()


def test_tile_accumulates_into_existing_forces_and_auto_selects_it(hip, o32):
    n, L, rc = 20000, 30.0, 2.5
    pos, box, pot = _setup(hip, o32, n, L, rc)
    (ref, _, _), cd = _oracle(o32, pos, box, pot, rc)
    d_pos = torch.from_numpy(pos).cuda()
    cl = hip.CellList()
    cdd, ubox = hip.CellList.create_update_grid(box, rc)
    cl.update_grid(d_pos, ubox, cdd)
    f = torch.ones((n, 4), dtype=torch.float32, device="cuda")
    cl.transverse_lj(pot.device_table(), 1, box, f, None, None, None, 8)
    f2 = torch.ones((n, 4), dtype=torch.float32, device="cuda")
    cl.transverse_lj(pot.device_table(), 1, box, f2, None, None, None, 0)            # AUTO
    torch.cuda.synchronize()
    got = f.cpu().numpy()
    assert np.abs(got[:, :3] - 1.0 - ref[:, :3]).max() <= 1e-5 * np.abs(ref[:, :3]).max() and np.all(got[:, 3] == 1.0)
    assert np.array_equal(f2.cpu().numpy(), got), "AUTO must resolve to the tile kernel on this grid"
    f3 = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    cl.transverse_lj(pot.device_table(), 1, box, f3, None, None, None, 9)           # EXACT: the reference's summation order
    torch.cuda.synchronize()
    assert np.array_equal(f3.cpu().numpy()[:, :3].view(np.uint32), ref[:, :3].view(np.uint32))


def test_cutoff_longer_than_a_cell_edge_takes_the_exact_walk(hip, o32):
    """uammd_celllist_update accepts any cellDim.  With a cell edge SHORTER than the cut-off the 27 cells no longer hold every pair
    inside rc, the reference visits exactly those 27 cells all the same — and the tile kernel's 4-cell x halo would add pairs the
    reference never sees while its prefilter margin (derived for rc <= edge) could drop others.  TILE must refuse, AUTO must give
    the reference's (27-cell) answer bit for bit through the exact walk."""
    from uammd_amd._lib import UammdHipError
    n, L, rc = 6400, 20.0, 2.5
    pos, box, pot = _setup(hip, o32, n, L, rc)
    cd = [10, 10, 10]                                   # edge 2.0 < rc
    ref_cl = o32.celllist_build(pos, [L] * 3, [1, 1, 1], cd)
    ref, _, _ = o32.lj_transverse_celllist(ref_cl, box.boxSize, [1, 1, 1], pot.table, 1, n, True, False, False)
    d_pos = torch.from_numpy(pos).cuda()
    cl = hip.CellList()
    cl.update_grid(d_pos, box, cd)
    f = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    cl.transverse_lj(pot.device_table(), 1, box, f, None, None, None, 0)               # AUTO
    torch.cuda.synchronize()
    assert np.array_equal(f.cpu().numpy()[:, :3].view(np.uint32), ref[:, :3].view(np.uint32))
    for algo in (TILE, TILE1):
        with pytest.raises(UammdHipError, match="cut-off"):
            cl.transverse_lj(pot.device_table(), 1, box, f, None, None, None, algo)
    # the same list with a table whose cut-off fits the edge: the tile kernel again
    pot2 = hip.Potential.LJ()
    pot2.setPotParameters(0, 0, pot2.InputPairParameters(1.9, 1.0, 1.0, False))
    ref2, _, _ = o32.lj_transverse_celllist(ref_cl, box.boxSize, [1, 1, 1], pot2.table, 1, n, True, False, False)
    f2 = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    cl.transverse_lj(pot2.device_table(), 1, box, f2, None, None, None, TILE)
    torch.cuda.synchronize()
    _check_force(f2.cpu().numpy(), ref2, "tile, rc 1.9 on 2.0 cells", reordered=True)


def test_stale_parameter_table_is_reported_not_computed(hip, o32):
    """The host remembers a device table's largest cut-off by pointer.  A caller that REWRITES the table in place with a longer
    cut-off gets an error from the list (the kernel raises the host-mapped flag and computes nothing), never a wrong force."""
    from uammd_amd._lib import UammdHipError
    n, L = 6400, 20.0
    pos, box, pot = _setup(hip, o32, n, L, 1.9)
    pot.setPotParameters(0, 0, pot.InputPairParameters(1.9, 1.0, 1.0, False))
    cd = [10, 10, 10]
    d_pos = torch.from_numpy(pos).cuda()
    cl = hip.CellList()
    cl.update_grid(d_pos, box, cd)
    tbl = pot.device_table()
    f = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    cl.transverse_lj(tbl, 1, box, f, None, None, None, 0)
    torch.cuda.synchronize()
    before = f.clone()
    big = hip.Potential.LJ()
    big.setPotParameters(0, 0, big.InputPairParameters(2.5, 1.0, 1.0, False))
    tbl.copy_(torch.from_numpy(big.table.copy()).cuda())   # same pointer, longer cut-off, and nobody told the library (uammd_lj_table_changed)
    cl.transverse_lj(tbl, 1, box, f, None, None, None, 0)
    torch.cuda.synchronize()
    assert torch.equal(f, before)                      # nothing was accumulated
    with pytest.raises(UammdHipError, match="rewritten in place"):
        cl.update_grid(d_pos, box, cd)
    cl.update_grid(d_pos, box, cd)                     # the flag is consumed; the next traversal re-reads the table -> exact walk
    ref_cl = o32.celllist_build(pos, [L] * 3, [1, 1, 1], cd)
    ref, _, _ = o32.lj_transverse_celllist(ref_cl, box.boxSize, [1, 1, 1], big.table, 1, n, True, False, False)
    f.zero_()
    cl.transverse_lj(tbl, 1, box, f, None, None, None, 0)
    torch.cuda.synchronize()
    assert np.array_equal(f.cpu().numpy()[:, :3].view(np.uint32), ref[:, :3].view(np.uint32))


@SHAPES
@pytest.mark.parametrize("ratio", [4.0, 4.13, 4.37, 4.5, 4.77, 4.999])
def test_prefilter_margin_adversarial_pairs_at_the_cutoff(hip, o32, ratio, TILE):
    """The f16 matrix prefilter must be a SUPERSET of r < rc.  Its margin (lj_tile.hip tile_margin) is a hand-derived bound for
    r <= rc <= cell edge; this sweeps L / rc over [4, 5): 4 cells per side, cell edge from 1.0 rc up to 1.25 rc (where the relative
    margin is thinnest), and plants pairs at r = rc (1 - 1e-6) — inside the cut-off by less than any half-precision rounding —
    in random directions, at random places of the cells including faces, edges, corners and across the periodic boundary.  Every
    such pair carries a force of 0.0389 that a dropped hit would remove: compared against the oracle's cell-list walk."""
    rc = 2.5
    L = ratio * rc
    rng = np.random.default_rng(int(ratio * 1000) + TILE)
    npairs = 250
    edge = L / 4
    pts = np.zeros((0, 3))
    while len(pts) < 2 * npairs:
        a = rng.uniform(-L / 2, L / 2, 3)
        # a third of the first partners sit on cell faces / edges / corners (multiples of the cell edge), within one float ulp
        for k in range(int(rng.integers(0, 4))):
            a[k % 3] = np.round(a[k % 3] / edge) * edge + rng.choice([-1e-6, 0.0, 1e-6])
        d = rng.normal(size=3)
        b = a + d / np.linalg.norm(d) * (rc * (1 - 1e-6))
        both = np.stack([a, b])
        both -= np.round(both / L) * L                           # fold (f64), stored as f32 below
        if len(pts):                                             # no overlaps: every other pair force stays O(1)
            dd = both[:, None, :] - pts[None, :, :]
            dd -= np.round(dd / L) * L
            if (np.linalg.norm(dd, axis=2) < 0.97).any():
                continue
        pts = np.concatenate([pts, both])
    pos = np.zeros((2 * npairs, 4), np.float32)
    pos[:, :3] = pts
    box = hip.Box(L)
    pot = hip.Potential.LJ()
    pot.setPotParameters(0, 0, pot.InputPairParameters(rc, 1.0, 1.0, False))
    (ref, _, _), cd = _oracle(o32, pos, box, pot, rc)
    assert list(cd) == [4, 4, 4]
    got, _, _ = _run(hip, pos, box, pot, rc, TILE)
    # which planted pairs are inside the cut-off IN FLOAT (the reference's own test: r2 >= rc2 -> 0, Potential.cuh:37-50)?
    p32 = pos[:, :3]
    r = p32[1::2] - p32[0::2]
    r = r - np.floor(r / np.float32(L) + np.float32(0.5)) * np.float32(L)
    inside = (r.astype(np.float32) ** 2).sum(axis=1) < np.float32(rc * rc)
    assert inside.sum() > npairs // 3                           # the construction does put pairs on both sides of the float test
    fmax = np.abs(ref[:, :3]).max()
    err = np.abs(got[:, :3] - ref[:, :3]).max()
    print(f"[margin L/rc={ratio}] {int(inside.sum())} planted pairs inside rc; max |dF| = {err:.3e} (pair force at rc: 3.9e-2, max|F| {fmax:.3e})")
    # a dropped pair at r ~ rc changes a force by |F(rc)| = 24 (2/rc^13 - 1/rc^7) ~ 0.039: two orders above this bar (no two
    # particles are closer than 0.97, so every force is O(10)).
    assert fmax < 100.0 and err <= 2e-4
    _check_force(got, ref, f"tile margin sweep L/rc={ratio}", reordered=True)


@pytest.mark.parametrize("sigma,eps", [(1.0, 1.0), (0.9, 1.3), (1.0, 0.7)], ids=["reduced-units", "sigma0.9-eps1.3", "eps0.7"])
def test_tile_unit_and_general_parameter_instantiations(hip, o32, sigma, eps):
    """AUTO launches the brick kernel in two single-type instantiations: the general one (products by sigma^2 and epsilon / sigma^2 per
    pair) and, for sigma = epsilon = 1, one without them (the same bits: x * 1.0f == x).  Liquid density on the kernel's main path
    (no dense-brick fallback: checked with the stats hook), forces alone (the fused step's launch) and with energy + virial."""
    n, L, rc = 30000, 33.5, 2.5
    pos, box, _ = _setup(hip, o32, n, L, rc, seed=5)
    pot = hip.Potential.LJ()
    pot.setPotParameters(0, 0, pot.InputPairParameters(rc, sigma, eps, False))
    (rf, re, rv), cd = _oracle(o32, pos, box, pot, rc, (True, True, True))
    d_pos = torch.from_numpy(pos).cuda()
    cl = hip.CellList()
    cdd, ubox = hip.CellList.create_update_grid(box, rc)
    cl.update_grid(d_pos, ubox, cdd)
    f = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    cl.tile_stats(True)
    cl.transverse_lj(pot.device_table(), 1, box, f, None, None, None, 0)
    torch.cuda.synchronize()
    st = cl.tile_stats(False)
    assert st["bricks"] > 0 and st["fallback_bricks"] == 0
    _check_force(f.cpu().numpy(), rf, f"tile forces only, sigma {sigma} eps {eps}", reordered=True)
    gf, ge, gv = _run(hip, pos, box, pot, rc, 0, (True, True, True))
    _check_force(gf, rf, f"tile F + E + V, sigma {sigma} eps {eps}", reordered=True)
    assert np.abs(ge - re).max() <= 1e-5 * np.abs(re).max() and np.abs(gv - rv).max() <= 1e-5 * np.abs(rv).max()


def test_tile_parameter_table_rewritten_in_place_is_caught(hip, o32):
    """The host picks the reduced-units instantiation from its cached copy of the device table (read back once per pointer).  A program
    that rewrites the SAME device buffer with other units must not get forces in the wrong units: the kernel checks the table it is
    given, raises the list's error flag instead of computing, and the next call reads the table again."""
    from uammd_amd._lib import UammdHipError
    n, L, rc = 20000, 30.0, 2.5
    pos, box, pot = _setup(hip, o32, n, L, rc)          # sigma = epsilon = 1
    d_pos = torch.from_numpy(pos).cuda()
    cl = hip.CellList()
    cdd, ubox = hip.CellList.create_update_grid(box, rc)
    cl.update_grid(d_pos, ubox, cdd)
    tbl = pot.device_table()
    f = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    cl.transverse_lj(tbl, 1, box, f, None, None, None, 0)        # caches: unit table at this pointer
    torch.cuda.synchronize()
    other = hip.Potential.LJ()
    other.setPotParameters(0, 0, other.InputPairParameters(rc, 0.8, 2.0, False))
    tbl.copy_(torch.from_numpy(other.table.copy()).cuda())        # same pointer, other units, unannounced (no uammd_lj_table_changed)
    f.zero_()
    cl.transverse_lj(tbl, 1, box, f, None, None, None, 0)        # launched as "unit": the kernel refuses
    torch.cuda.synchronize()
    assert float(f.abs().max()) == 0.0
    with pytest.raises(UammdHipError, match="rewritten in place"):
        cl.check_errors()                                         # (the next update / get of the list reports it as well)
    (ref, _, _), _ = _oracle(o32, pos, box, other, rc)
    f.zero_()
    cl.transverse_lj(tbl, 1, box, f, None, None, None, 0)        # table read again: the general instantiation, right forces
    torch.cuda.synchronize()
    _check_force(f.cpu().numpy(), ref, "after the table was re-read", reordered=True)


def test_parameter_table_rewritten_in_place_and_announced_is_followed(hip, o32):
    """Round 5: Potential::LJ::setPotParameters between steps re-uploads into the SAME device buffer (legal in the reference,
    Potential.cuh:60-82).  Both host layers announce it (uammd_lj_table_changed): the list then reads its cached view of the table again
    and the traversal computes with the new parameters at once — other units (the reduced-units instantiation is dropped) and a longer
    cut-off — with no error raised."""
    from uammd_amd._lib import check, load
    n, L, rc = 20000, 30.0, 2.5
    pos, box, pot = _setup(hip, o32, n, L, rc)          # sigma = epsilon = 1
    d_pos = torch.from_numpy(pos).cuda()
    cl = hip.CellList()
    cdd, ubox = hip.CellList.create_update_grid(box, rc)
    cl.update_grid(d_pos, ubox, cdd)
    tbl = pot.device_table()
    f = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    cl.transverse_lj(tbl, 1, box, f, None, None, None, 0)        # caches: unit table at this pointer
    other = hip.Potential.LJ()
    other.setPotParameters(0, 0, other.InputPairParameters(rc, 0.8, 2.0, False))
    tbl.copy_(torch.from_numpy(other.table.copy()).cuda())        # same pointer, other units ...
    check(load().uammd_lj_table_changed())                        # ... announced
    f.zero_()
    cl.transverse_lj(tbl, 1, box, f, None, None, None, 0)
    torch.cuda.synchronize()
    cl.check_errors()
    (ref, _, _), _ = _oracle(o32, pos, box, other, rc)
    _check_force(f.cpu().numpy(), ref, "after an announced rewrite", reordered=True)
    # the host mirror's own path: the same Potential object changed between two traversals
    pot.setPotParameters(0, 0, pot.InputPairParameters(rc, 1.1, 0.5, False))
    f.zero_()
    cl.transverse_lj(pot.device_table(), 1, box, f, None, None, None, 0)
    torch.cuda.synchronize()
    cl.check_errors()
    (ref, _, _), _ = _oracle(o32, pos, box, pot, rc)
    _check_force(f.cpu().numpy(), ref, "after setPotParameters between traversals", reordered=True)
