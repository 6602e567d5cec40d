"""The fused MD step (uammd_verletnvt_gj_lj_step: GronbechJensen's first half step inside the cell list's hash kernel, the second inside
the tile traversal's store) against the plain sequence uammd_verletnvt_gj(1) -> CellList::update -> PairForces::sum -> uammd_verletnvt_gj(2)
it replaces (GronbechJensen.cu:88-115): positions, velocities and forces must be the SAME BITS after every step, with noise, on grids that
take the fused kernels and on grids that fall back (radix build, exact traversal, all-pairs box)."""
import numpy as np
import pytest
import torch

from util import lattice_positions

pytestmark = pytest.mark.gpu


def _sim(hip, n, L, fuse, T=1.0, algo=0, ntypes=1, mass=None, is2D=False, periodic=(1, 1, 1), force_radix=False):
    pd = hip.ParticleData(n, seed=77)
    pos = lattice_positions(n, L, seed=5, jitter=0.1, ntypes=ntypes)
    if is2D:   # a square lattice in the plane z = 0
        m = int(round(n ** 0.5))
        g = np.stack(np.meshgrid(np.arange(m), np.arange(m), indexing="ij"), -1).reshape(-1, 2)
        rng = np.random.default_rng(5)
        pos[:, :2] = (g + 0.5) / m * np.asarray(L[:2]) - np.asarray(L[:2]) / 2 + rng.uniform(-0.1, 0.1, (n, 2))
        pos[:, 2] = 0.0
    pd.setPos(pos)
    if mass is not None:
        pd.getMass("write").copy_(torch.from_numpy(mass).cuda())
    box = hip.Box(L, periodic)
    pot = hip.Potential.LJ()
    for a in range(ntypes):
        for b in range(a, ntypes):
            pot.setPotParameters(a, b, pot.InputPairParameters(2.5 - 0.2 * a, 1.0 + 0.05 * b, 1.0 + 0.1 * (a + b), (a + b) % 2 == 1))
    par = hip.VerletNVT.GronbechJensen.Parameters(temperature=T, dt=0.004, friction=1.0, initVelocities=True, is2D=is2D)
    integ = hip.VerletNVT.GronbechJensen(pd, par)
    integ.fuse = fuse
    pf = hip.PairForces(pd, box, pot, algo=algo)
    if force_radix:
        pf.nl = hip.CellList(pd)
        pf.nl.set_option("force_radix", 1)
    integ.addInteractor(pf)
    return pd, integ


def _same(a, b):
    return np.array_equal(a.cpu().numpy().view(np.uint32), b.cpu().numpy().view(np.uint32))


@pytest.mark.parametrize("case", ["cubic", "noncubic-multitype", "mass-array", "exact-algo", "radix-build", "small-box-nbody", "2D"])
def test_fused_step_is_bit_identical(hip, case):
    kw, n, L = {}, 20000, 29.3
    if case == "noncubic-multitype":
        kw, n, L = dict(ntypes=3), 24000, (33.0, 27.5, 35.2)
    elif case == "mass-array":
        kw = dict(mass=np.random.default_rng(1).uniform(0.5, 2.0, n).astype(np.float32))
    elif case == "exact-algo":
        kw = dict(algo=9)
    elif case == "radix-build":
        kw = dict(force_radix=True)
    elif case == "small-box-nbody":
        n, L = 300, 7.2
    elif case == "2D":
        n, L, kw = 2500, (56.0, 56.0, 10.0), dict(is2D=True, periodic=(1, 1, 0))
    pa, ia = _sim(hip, n, L, True, **kw)
    pb, ib = _sim(hip, n, L, False, **kw)
    for step in range(25):
        ia.forwardTime()
        ib.forwardTime()
        if step in (0, 1, 9, 24):
            assert _same(pa.getPos("read"), pb.getPos("read")), (case, step)
            assert _same(pa.getVel("read"), pb.getVel("read")), (case, step)
            fa, fb = pa.getForce("read").cpu().numpy(), pb.getForce("read").cpu().numpy()
            assert np.array_equal(fa, fb), (case, step)          # (== : +0 and -0 compare equal)
    assert np.isfinite(pa.getPos("read").cpu().numpy()).all()
    # a later plain call on the fused simulation's list sees a valid, current list
    pf = ia.interactors[0]
    pa.getForce("write").zero_()
    pf.sum(force=True)
    assert np.allclose(pa.getForce("read").cpu().numpy(), fa, rtol=0, atol=0) or case == "small-box-nbody"


def test_fused_step_sort_and_restart(hip):
    """sortParticles between fused steps (the reorder signal) and switching the fusion off and on mid-run keep the two runs identical."""
    n, L = 20000, 29.3
    pa, ia = _sim(hip, n, L, True)
    pb, ib = _sim(hip, n, L, False)
    for step in range(30):
        if step == 10:
            pa.sortParticles()
            pb.sortParticles()
        if step == 20:
            ia.fuse = False
        if step == 25:
            ia.fuse = True
        ia.forwardTime()
        ib.forwardTime()
    ida, idb = pa.id.cpu().numpy(), pb.id.cpu().numpy()
    assert np.array_equal(ida, idb)
    assert _same(pa.getPos("read"), pb.getPos("read")) and _same(pa.getVel("read"), pb.getVel("read"))


@pytest.mark.parametrize("case", ["tile", "tile-mass-array", "exact-fallback", "multitype", "no-ghosts"])
def test_traversal_with_second_half_step_on_a_list_with_ghosts(hip, case):
    """uammd_lj_transverse_celllist_gj2 (the slab drivers' step: the list holds ghosts, the traversal's store applies GronbechJensen's second
    half step, GronbechJensen.cu:28-62, to the owned rows) against uammd_lj_transverse_celllist -> uammd_verletnvt_gj(2) on the owned rows:
    forces and velocities the same bits, ghost rows of the force array and every row beyond the owned velocities untouched."""
    import ctypes as C
    from uammd_amd._lib import check, load
    lib = load()
    n, L, ntypes = 30000, (31.0, 31.0, 36.0), 1
    if case == "multitype":
        ntypes = 2
    pos = torch.from_numpy(lattice_positions(n, L, seed=11, jitter=0.1, ntypes=ntypes)).cuda()
    # a slab: non-periodic z, the rows with |z| beyond 14 are ghosts and sit at the END of the array (as the halo tail does)
    ghost = pos[:, 2].abs() > 14.0
    order = torch.argsort(ghost.to(torch.int8), stable=True)
    pos = pos[order].contiguous()
    n_owned = n if case == "no-ghosts" else int((~ghost).sum())
    assert case == "no-ghosts" or 0 < n_owned < n
    box = hip.Box(L, (1, 1, 0))
    pot = hip.Potential.LJ()
    for a in range(ntypes):
        for b in range(a, ntypes):
            pot.setPotParameters(a, b, pot.InputPairParameters(2.5 - 0.2 * a, 1.0 + 0.05 * b, 1.0 + 0.1 * (a + b), False))
    cd, ubox = hip.CellList.create_update_grid(box, 2.5)
    algo = 9 if case == "exact-fallback" else 0
    mass = torch.from_numpy(np.random.default_rng(3).uniform(0.5, 2.0, n).astype(np.float32)).cuda() if case == "tile-mass-array" else None
    dt = 0.004
    vel0 = torch.randn((n_owned + 7, 3), dtype=torch.float32, device="cuda")   # 7 spare rows that nobody may touch
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())

    def run(fused):
        cl = hip.CellList()
        cl.update_grid(pos, ubox, cd)
        cl.set_option("num_owned", -1 if case == "no-ghosts" else n_owned)
        f = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
        f[n_owned:] = 123.0                       # ghost rows: must come back as they went in
        v = vel0.clone()
        if fused:
            cl.transverse_lj_gj2(pot.device_table(), ntypes, box, f, v, dt, mass, 1.0, False, algo)
        else:
            cl.transverse_lj(pot.device_table(), ntypes, box, f, None, None, None, algo)
            check(lib.uammd_verletnvt_gj(2, None, p(v), p(f), p(mass), 0.0 if mass is not None else 1.0, None, n_owned, dt, 1.0, 0, 0.0, 0, 0, st))
        torch.cuda.synchronize()
        return f, v

    f0, v0 = run(False)
    f1, v1 = run(True)
    assert float(f0[:n_owned, :3].abs().max()) > 1.0 and not _same(v0[:n_owned], vel0[:n_owned])
    assert _same(f0, f1) and _same(v0, v1)
    assert _same(v1[n_owned:], vel0[n_owned:]) and bool((f1[n_owned:] == 123.0).all())


@pytest.mark.parametrize("wait", [False, True], ids=["queued", "waited"])
@pytest.mark.parametrize("seed", [1, 2])
def test_md_random_call_sequences(hip, seed, wait):
    """The MD step keeps state from call to call as well (a cell list that rebuilds lazily on the position-write and reorder signals, the
    fused step's own build standing in for update(), the sorter's hint).  A random sequence of 50 calls on VerletNVT::GronbechJensen with
    PairForces<LJ, CellList> — steps, the whole system shifted by up to three sigma, particles jiggled one by one, sortParticles, plain
    force evaluations — with the tile kernel (AUTO) against the same sequence with the exact kernels: the same trajectory to rounding
    (the thermostat's streams are keyed by array row on both, and both permute their arrays alike); queued and with a wait after every
    call."""
    from util import lattice_positions
    n = 20000
    L = (n / 0.8) ** (1 / 3)
    rng = np.random.default_rng(300 + seed)
    pos0 = lattice_positions(n, L, seed=5, jitter=0.1)
    ops = []
    for _ in range(50):
        u = rng.uniform()
        if u < 0.62: ops.append(("step",))
        elif u < 0.72: ops.append(("shift", float(rng.uniform(0.05, 3.0))))
        elif u < 0.82: ops.append(("sort",))
        elif u < 0.92: ops.append(("jiggle", int(rng.integers(1 << 30))))
        else: ops.append(("forces",))

    def run(algo, wait_):
        pd = hip.ParticleData(n, seed=1234)
        pd.setPos(pos0.copy())
        box = hip.Box(L)
        pot = hip.Potential.LJ()
        pot.setPotParameters(0, 0, pot.InputPairParameters(2.5, 1.0, 1.0, False))
        integ = hip.VerletNVT.GronbechJensen(pd, hip.VerletNVT.GronbechJensen.Parameters(temperature=1.0, dt=0.002, friction=1.0))
        pf = hip.PairForces(pd, box, pot)
        pf.algo = algo
        integ.addInteractor(pf)
        pd.hintSortByHash(box, [2.5] * 3)
        for op in ops:
            if op[0] == "step":
                integ.forwardTime()
            elif op[0] == "shift":
                pd.getPos("readwrite")[:, :3] += op[1]
            elif op[0] == "sort":
                pd.sortParticles()
            elif op[0] == "jiggle":
                g = torch.Generator(device="cuda").manual_seed(op[1])
                pd.getPos("readwrite")[:, :3] += 0.02 * torch.randn((n, 3), generator=g, device="cuda")
            else:
                pd.getForce("write").zero_()
                pf.sum(force=True)
            if wait_:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        out = np.zeros((n, 3))
        out[pd.id.cpu().numpy().astype(np.int64)] = pd.getPos("read").cpu().numpy()[:, :3]
        return out, integ.fused_steps

    ref, _ = run(9, True)
    got, fused = run(0, wait)
    assert fused == sum(1 for op in ops if op[0] == "step") and np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 5e-5
