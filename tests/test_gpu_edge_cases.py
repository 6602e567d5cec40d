"""Edge cases through the C ABI: empty and single-particle inputs, particles on the box faces and far outside the box, everything
in one cell, non-finite positions, and a 4.2e6-particle build checked through size-independent properties (the list is a
permutation, hashes are sorted and stable, every cell range is consistent)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lj(hip, rc=2.5):
    pot = hip.Potential.LJ()
    pot.setPotParameters(0, 0, pot.InputPairParameters(rc, 1.0, 1.0, False))
    return pot


def _forces(hip, pos, L, rc=2.5, nl=None):
    n = len(pos)
    pd = hip.ParticleData(n)
    if n:
        pd.setPos(pos)
    box = hip.Box(L)
    pf = hip.PairForces(pd, box, _lj(hip, rc), nl=nl(pd) if nl else None)
    pd.getForce("write").zero_()
    pf.sum(force=True)
    torch.cuda.synchronize()
    return pd.getForce("read").cpu().numpy()


@pytest.mark.parametrize("nl", [None, "verlet"])
def test_empty_and_single_particle(hip, nl):
    mk = (lambda pd: hip.VerletList(pd)) if nl else None
    assert _forces(hip, np.zeros((0, 4), np.float32), 20.0, nl=mk).shape == (0, 4)
    f = _forces(hip, np.array([[1.0, -2.0, 3.0, 0.0]], np.float32), 20.0, nl=mk)
    assert np.all(f == 0)
    # two particles through the periodic image: r = sigma across the face -> -/+ 24 eps / sigma
    f = _forces(hip, np.array([[9.5, 0, 0, 0], [-9.5, 0, 0, 0]], np.float32), 20.0, nl=mk)
    assert abs(f[0, 0] + 24.0) < 1e-4 and abs(f[1, 0] - 24.0) < 1e-4 and np.abs(f[:, 1:]).max() == 0


def test_faces_far_images_and_one_crowded_cell(hip, o32):
    L, rc = 25.0, 2.5
    rng = np.random.default_rng(5)
    pos = np.zeros((3000, 4), np.float32)
    pos[:1000, :3] = rng.uniform(-L / 2, L / 2, (1000, 3))
    pos[1000:1200, :3] = np.float32(L / 2) * rng.choice([-1.0, 1.0], (200, 3))          # exactly on the corners / faces
    pos[1200:1400, :3] = rng.uniform(-L / 2, L / 2, (200, 3)) + np.float32(L) * rng.integers(-1000, 1000, (200, 3))   # far images
    pos[1400:, :3] = rng.uniform(0, 2.4, (1600, 3)) + np.float32(3.0)                      # 1600 particles in one cell
    cd, oL, oper = o32.celllist_create_grid([L] * 3, [1, 1, 1], rc)
    ref, _, _ = o32.lj_transverse_celllist(o32.celllist_build(pos, oL, oper, cd), [L] * 3, [1, 1, 1], o32.lj_params(rc, 1.0, 1.0), 1,
                                           len(pos))
    got = _forces(hip, pos, L, rc)
    fin = np.isfinite(ref).all(axis=1)
    assert np.array_equal(np.isfinite(got).all(axis=1), fin)
    scale = np.abs(ref[fin]).max()
    assert np.abs(got[fin] - ref[fin]).max() <= 1e-5 * scale


def test_non_finite_position_is_contained_and_reported(hip):
    """A NaN position poisons the pairs it takes part in (as in the reference, whose cut-off test keeps NaN pairs) and nothing else
    crashes: far-away particles keep finite forces.  The reference raises a device flag for such a particle and throws in
    UAMMD_DEBUG builds (CellListBase.cuh:82-85,258-264); here the flag lives in host-mapped memory: the next call on the list
    fails (lazy, no per-step sync), at once with strict_errors, never with report_errors = 0 (the reference's release build)."""
    from uammd_amd._lib import UammdHipError
    L = 30.0
    rng = np.random.default_rng(6)
    pos = np.zeros((2000, 4), np.float32)
    pos[:, :3] = rng.uniform(-L / 2, L / 2, (2000, 3))
    pos[7, 0] = np.nan
    box = hip.Box(L)
    pot = hip.Potential.LJ()
    pot.setPotParameters(0, 0, pot.InputPairParameters(2.5, 1.0, 1.0, False))
    cd, ubox = hip.CellList.create_update_grid(box, 2.5)
    dp = torch.from_numpy(pos).cuda()
    cl = hip.CellList()
    cl.set_option("report_errors", 0)
    cl.update_grid(dp, ubox, cd)
    f = torch.zeros((2000, 4), dtype=torch.float32, device="cuda")
    cl.transverse_lj(pot.device_table(), 1, box, f)
    torch.cuda.synchronize()
    cl.update_grid(dp, ubox, cd)                       # silent, like a release build of the reference
    assert np.isfinite(f.cpu().numpy()).all(axis=1).sum() >= 1900
    cl = hip.CellList()
    cl.update_grid(dp, ubox, cd)                       # lazy: this call returns before the flag can be seen
    torch.cuda.synchronize()
    with pytest.raises(UammdHipError, match="NaN positions"):
        cl.update_grid(dp, ubox, cd)
    good = dp.clone()
    good[7, 0] = 0.0
    cl.update_grid(good, ubox, cd)                     # the flag was consumed; a clean build passes
    torch.cuda.synchronize()
    cl.update_grid(good, ubox, cd)
    cl = hip.CellList()
    cl.set_option("strict_errors", 1)
    with pytest.raises(UammdHipError, match="NaN positions"):
        cl.update_grid(dp, ubox, cd)
    # a particle outside a NON-periodic box has no cell either
    pbox = hip.Box(L, (True, True, False))
    cd2, ubox2 = hip.CellList.create_update_grid(pbox, 2.5)
    out = good.clone()
    out[3, 2] = L
    cl = hip.CellList()
    cl.set_option("strict_errors", 1)
    with pytest.raises(UammdHipError, match="non-periodic"):
        cl.update_grid(out, ubox2, cd2)


def test_fcm_and_ibm_with_no_particles(hip):
    k, a = hip.Kernels.Gaussian(1.0, 1e-3)
    fcm = hip.BDHI.FCM_impl(hip.Box(32.0), [32, 32, 32], k, 1.0, 12, a)
    empty = torch.zeros((0, 4), dtype=torch.float32, device="cuda")
    v = fcm.computeHydrodynamicDisplacements(empty, empty, 0, 1.0, 1.0)
    assert v.shape == (0, 3)
    one = torch.tensor([[0.2, 0.4, -0.3, 0.0]], dtype=torch.float32, device="cuda")
    f1 = torch.tensor([[1.0, 0.0, 0.0, 0.0]], dtype=torch.float32, device="cuda")
    v = fcm.computeHydrodynamicDisplacements(one, f1, 1, 0.0, 0.0).cpu().numpy()
    assert abs(v[0, 0] / fcm.getSelfMobility() - 1) < 2e-3
    ibm = hip.IBM(k, hip.Box(32.0), [32, 32, 32])
    g = torch.zeros((32, 32, 32, 3), dtype=torch.float32, device="cuda")
    ibm.spread(empty, torch.zeros((0, 3), dtype=torch.float32, device="cuda"), g)
    assert float(g.abs().max()) == 0.0


def test_four_million_particles_properties(hip):
    """2^22 particles at rho* = 0.8: the build's outputs satisfy the invariants the parity tests check on small inputs."""
    n = 1 << 22
    L = float((n / 0.8) ** (1 / 3))
    gen = torch.Generator(device="cuda").manual_seed(3)
    pos = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    pos[:, :3] = (torch.rand((n, 3), generator=gen, device="cuda") - 0.5) * L
    cl = hip.CellList()
    cd, ubox = hip.CellList.create_update_grid(hip.Box(L), 2.5)
    cl.update_grid(pos, ubox, cd)
    d = cl.getCellList()
    idx = cl._wrap(d.d_groupIndex, (n,), torch.int32).long()
    assert int(torch.bincount(idx, minlength=n).max()) == 1 and int(idx.min()) == 0 and int(idx.max()) == n - 1   # a permutation
    assert torch.equal(cl._wrap(d.d_sortPos, (n, 4), torch.float32), pos[idx])
    h = cl._wrap(d.d_sortHash, (n,), torch.int32).long()
    assert bool((h[1:] >= h[:-1]).all())
    same = h[1:] == h[:-1]
    assert bool((idx[1:][same] > idx[:-1][same]).all())      # stable: equal keys keep the input order
    nc = int(d.cellDim[0] * d.cellDim[1] * d.cellDim[2])
    cs = cl._wrap(d.d_cellStart, (nc,), torch.int32).long()
    ce = cl._wrap(d.d_cellEnd, (nc,), torch.int32).long()
    occ = cs >= int(d.VALID_CELL)
    assert int((ce[occ] - (cs[occ] - int(d.VALID_CELL))).sum()) == n     # the cell ranges tile the sorted array
    # forces: Newton's third law over the whole system
    pd = hip.ParticleData(n)
    pd.getPos("write").copy_(pos)
    pf = hip.PairForces(pd, hip.Box(L), _lj(hip))
    pd.getForce("write").zero_()
    pf.sum(force=True)
    f = pd.getForce("read")[:, :3].double()
    ok = torch.isfinite(f).all(dim=1) & (f.abs().max(dim=1).values < 1e6)     # random overlaps give huge but paired forces
    assert float(f[ok].sum(dim=0).abs().max()) <= 1e-3 * float(f[ok].abs().sum(dim=0).max()) + 1e3
