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
