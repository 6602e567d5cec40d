"""GPU parity, SURVEY rows a28/a29: BDHI::PSE (uammd_pse_near_*, uammd_pse_far_*) vs the oracle, plus the reference's
own known-answer tests (test/BDHI/PSE/pse_test.cu) run on the product.

Tolerances: near-field products keep the reference's pair order and the oracle's FMA placement; tables come from the
same double closed form -> expected bit-identical, asserted <= 1e-6 of max|Mv|.  Far field: 1e-5 relative L2 (FFT
factorisation, spread order).  Lanczos results: the iteration is stopped at a relative change <= tolerance, so two
implementations agree to a few times that tolerance.  Saru Gaussians use device log/sin/cos: 1e-6.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RH, VISC = 1.012312, 1.12321


def _pair(hip, o32, L, tol, psi, n, seed=1, shear=0.0):
    from oracle.pse import PSEOracle
    pd = hip.ParticleData(n, seed=seed)
    rng = np.random.default_rng(7)
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(-L / 2, L / 2, (n, 3))
    pd.setPos(pos)
    par = hip.BDHI.PSE.Parameters(psi=psi, shearStrain=shear, temperature=0.0, viscosity=VISC, hydrodynamicRadius=RH,
                                  tolerance=tol, dt=1.0, box=hip.Box(L))
    pse = hip.BDHI.PSE(pd, par)
    # the same System::rng() stream gives the oracle its seeds
    r = hip.md.Xorshift128plus() if hasattr(hip, "md") else None
    from uammd_amd.md import Xorshift128plus
    r = Xorshift128plus()
    r.set_seed(seed)
    ref = PSEOracle(o32, [L] * 3, RH, VISC, tol, psi, shearStrain=shear, seed_near=r.next32(), seed_far=r.next32())
    return pd, pse, ref, pos, r


@pytest.mark.parametrize("psi,tol,shear", [(0.6, 1e-3, 0.0), (1.0, 1e-4, 0.0), (0.8, 1e-3, 0.15)])
def test_setup_and_near_dot(hip, o32, psi, tol, shear):
    L, n = 20.0, 3000
    pd, pse, ref, pos, _ = _pair(hip, o32, L, tol, psi, n, shear=shear)
    assert pse.nPointsTable == ref.nPointsTable and abs(pse.rcut - float(ref.rcut)) == 0.0
    assert list(pse.cells) == list(ref.cells) and pse.support == ref.support
    assert abs(pse.eta - float(ref.eta)) <= 1e-6 * float(ref.eta)
    rng = np.random.default_rng(2)
    v = rng.normal(0, 1, (n, 3)).astype(np.float32)
    d_v = torch.from_numpy(v).cuda()
    out = torch.full((n, 3), 7.0, dtype=torch.float32, device="cuda")      # the Dotctor zeroes Mv first
    from uammd_amd._lib import check
    from uammd_amd.md import _ptr, current_stream
    check(pse.lib.uammd_pse_near_dot(pse.near, _ptr(pd.getPos()), _ptr(d_v), n, _ptr(out), current_stream()))
    expect = np.zeros((n, 3), np.float32)
    cl = ref._near_list(pos)
    ref._near_dot(cl, v, 3, expect)
    got = out.cpu().numpy()
    # default product: eight lanes per particle (same pairs, same per-pair arithmetic, another summation order)
    assert np.abs(got - expect).max() <= 1e-6 * np.abs(expect).max()
    print("8-lane product, differing words:", np.count_nonzero(got.view(np.uint32) != expect.view(np.uint32)), "of", got.size)
    # "exact_order": the reference's summation order, bit for bit
    check(pse.lib.uammd_pse_near_set_option(pse.near, b"exact_order", 1))
    out.fill_(7.0)
    check(pse.lib.uammd_pse_near_dot(pse.near, _ptr(pd.getPos()), _ptr(d_v), n, _ptr(out), current_stream()))
    got = out.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), expect.view(np.uint32))
    check(pse.lib.uammd_pse_near_set_option(pse.near, b"exact_order", 0))
    # Mdot with real4 forces ACCUMULATES into MF
    f4 = np.zeros((n, 4), np.float32)
    f4[:, :3] = v
    MF = torch.ones((n, 3), dtype=torch.float32, device="cuda")
    check(pse.lib.uammd_pse_near_mdot(pse.near, _ptr(pd.getPos()), _ptr(torch.from_numpy(f4).cuda()), n, _ptr(MF), current_stream()))
    assert np.abs(MF.cpu().numpy() - (1.0 + expect)).max() <= 2e-6 * max(1.0, np.abs(expect).max())


@pytest.mark.parametrize("psi,tol,shear", [(0.6, 1e-3, 0.0), (1.0, 1e-4, 0.0), (0.8, 1e-3, 0.15)])
def test_far_field_deterministic_and_noise(hip, o32, psi, tol, shear):
    L, n = 20.0, 2000
    pd, pse, ref, pos, _ = _pair(hip, o32, L, tol, psi, n, shear=shear)
    rng = np.random.default_rng(4)
    f4 = np.zeros((n, 4), np.float32)
    f4[:, :3] = rng.normal(0, 1, (n, 3))
    d_f = torch.from_numpy(f4).cuda()
    from uammd_amd._lib import check
    from uammd_amd.md import _ptr, current_stream
    for T, pref, force, seed2 in [(0.0, 0.0, d_f, 0), (0.9, 1.7, d_f, 4242), (0.9, 1.7, None, 77)]:
        MF = torch.full((n, 3), 0.5, dtype=torch.float32, device="cuda")       # the gather ADDS
        check(pse.lib.uammd_pse_far_displacements(pse.far, _ptr(pd.getPos()), _ptr(force), n, T, pref, seed2, _ptr(MF),
                                                  current_stream()))
        expect = np.full((n, 3), 0.5, np.float32)
        ref.far(pos, None if force is None else f4, expect, T, pref, seed2)
        got = MF.cpu().numpy()
        err = np.linalg.norm(got - expect) / np.linalg.norm(expect - 0.5)
        assert err <= 1e-5, (T, force is None, err)


def test_near_noise_and_lanczos(hip, o32):
    L, n, tol, psi = 16.0, 1500, 1e-3, 0.7
    pd, pse, ref, pos, _ = _pair(hip, o32, L, tol, psi, n)
    from uammd_amd._lib import check
    from uammd_amd.md import _ptr, current_stream
    nz = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    check(pse.lib.uammd_pse_near_noise(pse.near, n, 1.3, 999, _ptr(nz), current_stream()))
    exp_noise = ref.near_noise(n, 1.3, 999)
    assert np.abs(nz.cpu().numpy() - exp_noise).max() <= 1e-6 * np.abs(exp_noise).max()
    import ctypes as C
    BdW = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    it = C.c_int(0)
    check(pse.lib.uammd_pse_near_stochastic(pse.near, _ptr(pd.getPos()), n, 0.8, 1.1, 555, _ptr(BdW), current_stream(), C.byref(it)))
    exp = ref.near_stochastic(pos, 0.8, 1.1, 555)
    got = BdW.cpu().numpy()
    assert 1 <= it.value <= 30
    assert np.linalg.norm(got - exp) <= 5 * tol * np.linalg.norm(exp)
    # the default path iterates on vectors in cell order; "exact_order" keeps the caller's order: same Krylov space, same iteration
    # count, results equal to rounding
    check(pse.lib.uammd_pse_near_set_option(pse.near, b"exact_order", 1))
    B2 = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    it2 = C.c_int(0)
    check(pse.lib.uammd_pse_near_stochastic(pse.near, _ptr(pd.getPos()), n, 0.8, 1.1, 555, _ptr(B2), current_stream(), C.byref(it2)))
    check(pse.lib.uammd_pse_near_set_option(pse.near, b"exact_order", 0))
    assert it2.value == it.value
    assert np.linalg.norm(B2.cpu().numpy() - got) <= 1e-5 * np.linalg.norm(got)
    # M^(1/2) property: |B dW|^2 / |dW|^2 is a Rayleigh quotient of M_near -> between its extreme eigenvalues (all > 0)
    assert 0 < np.linalg.norm(got) < np.linalg.norm(ref.near_noise(n, 1.1 * math.sqrt(2 * 0.8), 555))


def test_self_mobility_reference_test(hip):
    """pse_test.cu:64-118 on the product: pulling one particle gives the Hasimoto mobility within the tolerance (float:
    tolerance 1e-4 and a floor of 2e-6 for single precision), 6 random positions x 3 directions."""
    tol, L = 1e-4, 32 * RH
    pd = hip.ParticleData(1, seed=3)
    par = hip.BDHI.PSE.Parameters(psi=1.0, temperature=0.0, viscosity=VISC, hydrodynamicRadius=RH, tolerance=tol, dt=1.0,
                                  box=hip.Box(L))
    pse = hip.BDHI.PSE(pd, par)
    m0 = pse.getSelfMobility()
    rng = np.random.default_rng(1234)
    MF = torch.zeros((1, 3), dtype=torch.float32, device="cuda")
    for j in range(6):
        p = np.zeros((1, 4), np.float32)
        p[0, :3] = rng.uniform(-0.5, 0.5, 3) * L
        pd.setPos(p)
        for d in range(3):
            f = torch.zeros((1, 4), dtype=torch.float32, device="cuda")
            f[0, d] = 1.0
            pse.computeHydrodynamicDisplacements(f, MF, 0.0, 0.0)
            expect = np.zeros(3)
            expect[d] = m0
            assert np.abs(MF.cpu().numpy()[0] - expect).max() <= tol


def test_self_diffusion_reference_tests(hip):
    """pse_test.cu:121-205: <dx^2> = 2 T M0 within 1e-2, through computeHydrodynamicDisplacements (T = 1, prefactor 1) and
    through computeMF + computeBdW (dt = 1)."""
    L = 32 * RH
    temperature = 1.12312
    pd = hip.ParticleData(1, seed=99)
    par = hip.BDHI.PSE.Parameters(psi=1.0, temperature=temperature, viscosity=VISC, hydrodynamicRadius=RH, tolerance=1e-4,
                                  dt=1.0, box=hip.Box(L))
    pse = hip.BDHI.PSE(pd, par)
    m0 = pse.getSelfMobility()
    rng = np.random.default_rng(5)
    ntest = 1000
    acc1 = torch.zeros(3, dtype=torch.float64, device="cuda")
    acc2 = torch.zeros(3, dtype=torch.float64, device="cuda")
    MF = torch.zeros((1, 3), dtype=torch.float32, device="cuda")
    BdW = torch.zeros((1, 3), dtype=torch.float32, device="cuda")
    pd.getForce("write").zero_()
    for j in range(ntest):
        p = np.zeros((1, 4), np.float32)
        p[0, :3] = rng.uniform(-0.5, 0.5, 3) * L
        pd.setPos(p)
        pse.computeHydrodynamicDisplacements(None, MF, 1.0, 1.0)
        acc1 += MF[0].double() ** 2
        pse.computeMF(MF)
        pse.computeBdW(BdW)
        acc2 += (MF[0] + BdW[0]).double() ** 2
    d1 = (acc1 / ntest).cpu().numpy()
    d2 = (acc2 / ntest).cpu().numpy()
    assert np.abs(d1 - 2.0 * m0).max() <= 1e-2
    # computeMF carries the far noise with prefactor 1/sqrt(dt); computeBdW the near noise with prefactor 1: with dt = 1 and the
    # integrator's sqrt(2 T dt) on BdW only, the reference test sums them as is (pse_test.cu:189-203) and compares with 2 T M0
    assert np.abs(d2 - 2.0 * temperature * m0).max() <= 1.5e-2


def test_euler_maruyama_pse_step_matches_oracle(hip, o32):
    """BDHI::EulerMaruyama<PSE>::forwardTime at T = 0 with a constant-force interactor and a shear matrix K."""
    L, n, tol, psi, dt = 18.0, 800, 1e-3, 0.8, 0.05
    pd, pse, ref, pos, _ = _pair(hip, o32, L, tol, psi, n)
    rng = np.random.default_rng(8)
    f4 = np.zeros((n, 4), np.float32)
    f4[:, :3] = rng.normal(0, 1, (n, 3))

    class Const:
        def __init__(self, pd, f):
            self.pd, self.f = pd, f

        def sum(self, force=True, energy=False, virial=False):
            self.pd.getForce("readwrite").add_(self.f)

        def updateSimulationTime(self, t): pass
        def updateTimeStep(self, dt): pass
        def updateTemperature(self, T): pass
        def updateBox(self, b): pass
    K = [[0.0, 0.3, 0.0], [0.0, 0.0, 0.0], [0.1, 0.0, -0.2]]
    par = hip.BDHI.PSE.Parameters(psi=psi, temperature=0.0, viscosity=VISC, hydrodynamicRadius=RH, tolerance=tol, dt=dt,
                                  box=hip.Box(L))
    par.K = K
    integ = hip.BDHI.EulerMaruyama(pd, par, method=pse)
    integ.addInteractor(Const(pd, torch.from_numpy(f4).cuda()))
    integ.forwardTime()
    got = pd.getPos().cpu().numpy()
    MF = np.zeros((n, 3), np.float32)
    ref.far(pos, f4, MF, 0.0, 0.0, 0)
    ref.near_mdot(pos, f4, MF)
    import ctypes as C
    from oracle.oracle import _p
    p = pos.copy()
    Kf = np.asarray(K, np.float32).reshape(-1)
    o32.lib.oracle_bdhi_euler_maruyama(_p(p), None, _p(MF), None, _p(Kf), n, C.c_float(0.0), C.c_float(dt), 0)
    assert np.abs(got - p).max() <= 1e-5 * np.abs(p - pos).max() + 1e-6


def test_lazy_list_follows_position_writes(hip, o32):
    """CellList::update rebuilds only after a position write (CellList.cuh:94-98,134-136): the PSE class wires ParticleData's write
    signal to uammd_pse_near_positions_changed.  Moving the particles through the ParticleData interface must be seen by the next
    product; a second product without a write reuses the list and gives the same bits."""
    L, n, tol, psi = 20.0, 3000, 1e-3, 0.6
    pd, pse, ref, pos, _ = _pair(hip, o32, L, tol, psi, n)
    from uammd_amd._lib import check
    from uammd_amd.md import _ptr, current_stream
    rng = np.random.default_rng(12)
    f4 = np.zeros((n, 4), np.float32)
    f4[:, :3] = rng.normal(0, 1, (n, 3))
    d_f = torch.from_numpy(f4).cuda()

    def product():
        MF = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
        check(pse.lib.uammd_pse_near_mdot(pse.near, _ptr(pd.getPos()), _ptr(d_f), n, _ptr(MF), current_stream()))
        return MF.cpu().numpy()
    a = product()
    assert np.array_equal(product(), a)
    pos2 = pos.copy()
    pos2[:, :3] = np.random.default_rng(99).uniform(-L / 2, L / 2, (n, 3))
    pd.getPos("write").copy_(torch.from_numpy(pos2).cuda())              # same array, new contents: the signal invalidates the list
    b = product()
    expect = np.zeros((n, 3), np.float32)
    ref.near_mdot(pos2, f4, expect)
    assert np.abs(b - expect).max() <= 1e-6 * np.abs(expect).max() and not np.allclose(a, b)


def test_pair_records_match_the_scan_product(hip, o32):
    """The pair records (k_pse_pairs_build / k_pse_near_pairs: the default once the list is kept between products) against the product that
    scans the cells every time (k_pse_near8, option pair_list = 0): the same pairs, (G - F) / r^2 rounded once more per pair; caller-order
    and accumulate forms; a sheared box; after a position write the records follow the new list; and a cluster with more neighbours than a
    hit list holds sends the handle back to the scanning product (same results)."""
    from uammd_amd._lib import check
    from uammd_amd.md import _ptr, current_stream
    for shear, cluster in ((0.0, False), (0.2, False), (0.0, True)):
        L, n, tol, psi = 40.0, 4000, 1e-3, 0.6   # ~22 neighbours per particle (at L = 20 every particle has more than a hit list holds)
        pd, pse, ref, pos, _ = _pair(hip, o32, L, tol, psi, n, shear=shear)
        if cluster:
            pos = pos.copy()
            pos[:300, :3] = np.random.default_rng(5).normal(0, 0.3, (300, 3)).astype(np.float32)   # 300 particles within one cut-off
            pd.setPos(pos)
        rng = np.random.default_rng(3)
        f4 = np.zeros((n, 4), np.float32)
        f4[:, :3] = rng.normal(0, 1, (n, 3))
        d_f = torch.from_numpy(f4).cuda()
        d_v = torch.from_numpy(np.ascontiguousarray(f4[:, :3])).cuda()

        def products():
            MF = torch.ones((n, 3), dtype=torch.float32, device="cuda")
            check(pse.lib.uammd_pse_near_mdot(pse.near, _ptr(pd.getPos()), _ptr(d_f), n, _ptr(MF), current_stream()))
            out = torch.full((n, 3), 7.0, dtype=torch.float32, device="cuda")
            check(pse.lib.uammd_pse_near_dot(pse.near, _ptr(pd.getPos()), _ptr(d_v), n, _ptr(out), current_stream()))
            return MF.cpu().numpy(), out.cpu().numpy()
        a = products()
        import ctypes as C
        used = C.c_longlong(0)
        check(pse.lib.uammd_pse_near_pair_records(pse.near, C.byref(used), None))
        assert (used.value == 0) if cluster else (15 * n < used.value < 40 * n)   # the records are what ran (or, for the cluster, are not)
        a2 = products()                                    # the records are reused: the same bits
        assert np.array_equal(a[0], a2[0]) and np.array_equal(a[1], a2[1])
        check(pse.lib.uammd_pse_near_set_option(pse.near, b"pair_list", 0))
        b = products()
        check(pse.lib.uammd_pse_near_set_option(pse.near, b"pair_list", 1))
        expect = np.zeros((n, 3), np.float32)
        ref.near_mdot(pos, f4, expect)
        scale = np.abs(expect).max()
        for x, y in zip(a, b):
            # (the accumulate form adds to ones: an ulp of 1 on top)
            assert np.abs(x - y).max() <= 1e-6 * scale + 2.5e-7, (shear, cluster, float(np.abs(x - y).max()), float(scale))
        assert np.abs(a[1] - expect).max() <= 1e-6 * scale and np.abs(a[0] - 1.0 - expect).max() <= 2e-6 * max(1.0, scale)
        if cluster:
            assert np.array_equal(a[1], b[1])              # more than a hit list holds: both runs took the scanning product
        # a position write: new list, new records
        pos2 = pos.copy()
        pos2[:, :3] = np.random.default_rng(99).uniform(-L / 2, L / 2, (n, 3))
        pd.getPos("write").copy_(torch.from_numpy(pos2).cuda())
        c = products()
        expect2 = np.zeros((n, 3), np.float32)
        ref.near_mdot(pos2, f4, expect2)
        assert np.abs(c[1] - expect2).max() <= 1e-6 * np.abs(expect2).max()


@pytest.mark.gpu
def test_near_prepare_changes_no_result(hip, o32):
    """uammd_pse_near_prepare queues the list and the pair records' build ahead of the products (the records' host read then overlaps whatever
    the caller queues in between): the products give the bits they give without it; a prepare for positions that are then written again is
    superseded by the next call's list; capacity growth still works when the first launch came from prepare."""
    from uammd_amd._lib import check
    from uammd_amd.md import _ptr, current_stream
    import ctypes as C
    L, n, tol, psi = 40.0, 4000, 1e-3, 0.6       # ~22 neighbours per particle: the records are in use
    pd, pse, ref, pos, _ = _pair(hip, o32, L, tol, psi, n)

    def records():
        used, cap = C.c_longlong(0), C.c_longlong(0)
        check(pse.lib.uammd_pse_near_pair_records(pse.near, C.byref(used), C.byref(cap)))
        return used.value, cap.value
    rng = np.random.default_rng(3)
    f4 = np.zeros((n, 4), np.float32)
    f4[:, :3] = rng.normal(0, 1, (n, 3))
    d_f = torch.from_numpy(f4).cuda()

    def mdot(prepare, between=None):
        check(pse.lib.uammd_pse_near_positions_changed(pse.near))
        if prepare:
            check(pse.lib.uammd_pse_near_prepare(pse.near, _ptr(pd.getPos()), n, current_stream()))
        if between is not None:
            between()
        MF = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
        check(pse.lib.uammd_pse_near_mdot(pse.near, _ptr(pd.getPos()), _ptr(d_f), n, _ptr(MF), current_stream()))
        return MF.cpu().numpy()
    plain = mdot(False)
    assert 15 * n < records()[0] < 30 * n
    scratch = torch.zeros(1 << 20, device="cuda")
    assert np.array_equal(mdot(True), plain)
    assert np.array_equal(mdot(True, lambda: scratch.add_(1.0)), plain)      # other work between the two halves of the build
    expect = np.zeros((n, 3), np.float32)
    ref.near_mdot(pos, f4, expect)
    assert np.abs(plain - expect).max() <= 1e-6 * np.abs(expect).max()
    # prepared, then written: the products follow the positions they are called with
    check(pse.lib.uammd_pse_near_prepare(pse.near, _ptr(pd.getPos()), n, current_stream()))
    pos2 = pos.copy()
    pos2[:, :3] = np.random.default_rng(99).uniform(-L / 2, L / 2, (n, 3))
    pd.getPos("write").copy_(torch.from_numpy(pos2).cuda())
    MF = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    check(pse.lib.uammd_pse_near_mdot(pse.near, _ptr(pd.getPos()), _ptr(d_f), n, _ptr(MF), current_stream()))
    expect2 = np.zeros((n, 3), np.float32)
    ref.near_mdot(pos2, f4, expect2)
    assert np.abs(MF.cpu().numpy() - expect2).max() <= 1e-6 * np.abs(expect2).max()
    # a denser configuration (~54 neighbours, at most 91) than the first allocation holds (48 records per particle): the build started by prepare is
    # repeated with more room
    assert records()[1] == 48 * n
    pos3 = pos.copy()
    pos3[:, :3] = np.random.default_rng(7).uniform(-14.0, 14.0, (n, 3))
    pd.getPos("write").copy_(torch.from_numpy(pos3).cuda())
    check(pse.lib.uammd_pse_near_prepare(pse.near, _ptr(pd.getPos()), n, current_stream()))
    MF.zero_()
    check(pse.lib.uammd_pse_near_mdot(pse.near, _ptr(pd.getPos()), _ptr(d_f), n, _ptr(MF), current_stream()))
    expect3 = np.zeros((n, 3), np.float32)
    ref.near_mdot(pos3, f4, expect3)
    assert np.abs(MF.cpu().numpy() - expect3).max() <= 2e-6 * np.abs(expect3).max()
    assert records()[0] > 48 * n and records()[1] >= records()[0]


@pytest.mark.gpu
def test_records_read_after_the_solve(hip, o32):
    """"optimistic_records": the near-field noise streams the records of a build whose counters it reads only after the Lanczos run.  Same
    bits and iteration counts as reading them first — also when the build did not fit (first allocation forced small: the solve is repeated
    on the larger build from the same noise, the solver's adaptive schedule put back), over a run of calls whose schedule adapts."""
    from uammd_amd._lib import check
    from uammd_amd.md import _ptr, current_stream
    import ctypes as C
    L, n, tol, psi = 40.0, 4000, 1e-3, 0.6

    def run(optimistic, capacity):
        pd, pse, ref, pos, _ = _pair(hip, o32, L, tol, psi, n)
        check(pse.lib.uammd_pse_near_set_option(pse.near, b"optimistic_records", optimistic))
        if capacity:
            check(pse.lib.uammd_pse_near_set_option(pse.near, b"pair_capacity", capacity))
        outs, its = [], []
        for call in range(6):
            if call == 3:      # new positions: new list, new records
                pos2 = pos.copy()
                pos2[:, :3] = np.random.default_rng(99).uniform(-L / 2, L / 2, (n, 3))
                pd.getPos("write").copy_(torch.from_numpy(pos2).cuda())
            BdW = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
            it = C.c_int(0)
            check(pse.lib.uammd_pse_near_stochastic(pse.near, _ptr(pd.getPos()), n, 1.0, 1.0, 1000 + call, _ptr(BdW), current_stream(),
                                                    C.byref(it)))
            outs.append(BdW.cpu().numpy())
            its.append(it.value)
        used, cap = C.c_longlong(0), C.c_longlong(0)
        check(pse.lib.uammd_pse_near_pair_records(pse.near, C.byref(used), C.byref(cap)))
        return outs, its, used.value, cap.value
    base = run(0, 0)
    assert base[2] > 15 * n and all(np.isfinite(o).all() and np.abs(o).max() > 0 for o in base[0])
    for optimistic, capacity in ((1, 0), (1, 8 * n), (0, 8 * n)):
        got = run(optimistic, capacity)
        assert got[1] == base[1], (optimistic, capacity, got[1], base[1])
        assert all(np.array_equal(a, b) for a, b in zip(got[0], base[0])), (optimistic, capacity)
        assert got[2] == base[2] and (capacity == 0 or got[3] >= got[2] > capacity)


@pytest.mark.gpu
def test_far_field_interleaved_with_the_solve(hip, o32):
    """PSE.computeMFandBdW (what EulerMaruyama runs when T > 0: the far field queued from inside the Lanczos solve, behind its convergence
    check) against computeMF followed by computeBdW on a twin with the same seeds: the same noise bit for bit, the same M F.  The callback runs exactly once per solve,
    and a product of the near-field handle from inside it is refused."""
    from uammd_amd._lib import check, INTERLEAVE_FN
    from uammd_amd.md import _ptr, current_stream
    import ctypes as C
    L, n = 40.0, 4000

    def twin():
        pd = hip.ParticleData(n, seed=77)
        rng = np.random.default_rng(7)
        pos = np.zeros((n, 4), np.float32)
        pos[:, :3] = rng.uniform(-L / 2, L / 2, (n, 3))
        pd.setPos(pos)
        f4 = np.zeros((n, 4), np.float32)
        f4[:, :3] = np.random.default_rng(3).normal(0, 1, (n, 3))
        pd.getForce("write").copy_(torch.from_numpy(f4).cuda())
        par = hip.BDHI.PSE.Parameters(psi=0.6, temperature=1.3, viscosity=VISC, hydrodynamicRadius=RH, tolerance=1e-3, dt=0.01,
                                      box=hip.Box(L))
        return pd, hip.BDHI.PSE(pd, par)
    pd_a, a = twin()
    pd_b, b = twin()
    for step in range(3):
        MFa, Ba = torch.zeros((n, 3), device="cuda"), torch.zeros((n + 1, 3), device="cuda")
        MFb, Bb = torch.zeros((n, 3), device="cuda"), torch.zeros((n + 1, 3), device="cuda")
        a.computeMF(MFa)
        a.computeBdW(Ba)
        b.computeMFandBdW(MFb, Bb)
        assert a.lastLanczosIterations == b.lastLanczosIterations, (step, a.lastLanczosIterations, b.lastLanczosIterations)
        assert torch.equal(Ba, Bb), (step, float((Ba - Bb).abs().max()), float(Ba.abs().max()))
        # (the far field of this small box spreads with f32 atomics — no tile edge fits its grid — whose order varies from run to run)
        assert float((MFa - MFb).abs().max()) <= 2e-6 * float(MFa.abs().max()), (step, float((MFa - MFb).abs().max()))
        assert float(MFa.abs().max()) > 0 and float(Ba.abs().max()) > 0
        for pd, pse in ((pd_a, a), (pd_b, b)):     # move on: new list, new records
            pd.getPos("write")[:, :3] += 0.05 * torch.randn((n, 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(step))
    # the callback: once per solve; the handle refuses its own products from inside it
    calls, refused = [], []

    def inside(_ctx, _stream):
        calls.append(1)
        MF = torch.zeros((n, 3), device="cuda")
        refused.append(b.lib.uammd_pse_near_mdot(b.near, _ptr(pd_b.getPos()), _ptr(pd_b.getForce()), n, _ptr(MF), current_stream()))
        return 0
    cb = INTERLEAVE_FN(inside)
    check(b.lib.uammd_pse_near_positions_changed(b.near))
    check(b.lib.uammd_pse_near_set_interleave(b.near, C.cast(cb, C.c_void_p), None))
    out = torch.zeros((n, 3), device="cuda")
    for _ in range(2):
        check(b.lib.uammd_pse_near_stochastic(b.near, _ptr(pd_b.getPos()), n, 1.0, 1.0, 5, _ptr(out), current_stream(), None))
    assert len(calls) == 1 and refused[0] != 0


@pytest.mark.gpu
def test_interleaved_work_is_queued_once_on_every_path(hip, o32):
    """uammd_pse_near_set_interleave: the callback runs exactly once per uammd_pse_near_stochastic whatever the call does — a solve, nothing to
    solve (T = 0), the exact-order solve, a refused call — and no registration is left for the call after."""
    from uammd_amd._lib import INTERLEAVE_FN
    from uammd_amd.md import _ptr, current_stream
    import ctypes as C
    L, n = 40.0, 2000
    pd, pse, ref, pos, _ = _pair(hip, o32, L, 1e-3, 0.6, n)
    calls, order = [], []
    cb = INTERLEAVE_FN(lambda _c, _s: (calls.append(1), order.append("late")) and 0)
    cb_early = INTERLEAVE_FN(lambda _c, _s: order.append("early") or 0)
    out = torch.zeros((n, 3), device="cuda")

    def stochastic(T, register=True, out_ptr=None):
        if register:
            assert pse.lib.uammd_pse_near_set_interleave(pse.near, C.cast(cb, C.c_void_p), None) == 0
            assert pse.lib.uammd_pse_near_set_interleave_early(pse.near, C.cast(cb_early, C.c_void_p), None) == 0
        return pse.lib.uammd_pse_near_stochastic(pse.near, _ptr(pd.getPos()), n, T, 1.0, 7, _ptr(out) if out_ptr is None else out_ptr,
                                                 current_stream(), None)
    assert stochastic(1.0) == 0 and len(calls) == 1
    assert stochastic(0.0) == 0 and len(calls) == 2              # nothing to solve: queued on the way out
    assert stochastic(1.0, register=False) == 0 and len(calls) == 2   # one-shot
    assert stochastic(1.0, out_ptr=C.c_void_p(0)) != 0 and len(calls) == 3   # a refused call still hands the stream over
    assert pse.lib.uammd_pse_near_set_option(pse.near, b"exact_order", 1) == 0
    assert stochastic(1.0) == 0 and len(calls) == 4
    assert stochastic(1.0, register=False) == 0 and len(calls) == 4
    torch.cuda.synchronize()
    assert order == ["early", "late"] * 4          # the early callback before the late one on every path, each once


@pytest.mark.gpu
def test_kept_candidate_lists_follow_the_particles(hip, o32):
    """Option "list_skin_percent": the particle order and the candidates within rc + skin of a list build are kept while the particles
    move, a step refreshes the sorted positions and makes its pair records from the candidates.  Against the ORACLE at every step of a walk
    of small displacements (deterministic product and the Lanczos noise's iteration count against a twin that builds its list per step),
    with the counters showing which path ran; then a jump larger than skin / 2: the bound is measured broken on the device, the build and
    the solve that streamed its records are repeated from a fresh list — the result is the fresh handle's, bit for bit."""
    from uammd_amd._lib import check
    from uammd_amd.md import _ptr, current_stream
    import ctypes as C
    L, n, tol, psi = 40.0, 4000, 1e-3, 0.6

    def stats(pse):
        s = (C.c_longlong * 4)()
        check(pse.lib.uammd_pse_near_list_stats(pse.near, s))
        return [int(x) for x in s]
    pd, pse, ref, pos, _ = _pair(hip, o32, L, tol, psi, n)
    pd2, twin, _, _, _ = _pair(hip, o32, L, tol, psi, n)
    check(twin.lib.uammd_pse_near_set_option(twin.near, b"list_skin_percent", 0))
    assert stats(pse)[3] == 1 and stats(twin)[3] == 0
    rng = np.random.default_rng(3)
    f4 = np.zeros((n, 4), np.float32)
    f4[:, :3] = rng.normal(0, 1, (n, 3))
    d_f = torch.from_numpy(f4).cuda()
    skin = 0.4 * pse.rcut
    walk = np.random.default_rng(11)

    def step(p, handle, positions, seed2):
        p.getPos("write").copy_(torch.from_numpy(positions).cuda())
        BdW = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
        it = C.c_int(0)
        check(handle.lib.uammd_pse_near_stochastic(handle.near, _ptr(p.getPos()), n, 1.0, 1.0, seed2, _ptr(BdW), current_stream(), C.byref(it)))
        MF = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
        check(handle.lib.uammd_pse_near_mdot(handle.near, _ptr(p.getPos()), _ptr(d_f), n, _ptr(MF), current_stream()))
        return MF.cpu().numpy(), BdW.cpu().numpy(), it.value
    cur = pos.copy()
    for k in range(8):
        if k:   # every particle moves by at most 0.02 skin per axis: the bound holds for all seven steps
            cur = cur.copy()
            cur[:, :3] += walk.uniform(-0.02 * skin, 0.02 * skin, (n, 3)).astype(np.float32)
        a = step(pd, pse, cur, 500 + k)
        b = step(pd2, twin, cur, 500 + k)
        expect = np.zeros((n, 3), np.float32)
        ref.near_mdot(cur, f4, expect)
        scale = np.abs(expect).max()
        assert np.abs(a[0] - expect).max() <= 1e-6 * scale, (k, float(np.abs(a[0] - expect).max() / scale))
        assert a[2] == b[2] and np.abs(a[1] - b[1]).max() <= 2e-6 * np.abs(b[1]).max(), (k, a[2], b[2])
    s = stats(pse)
    assert s[0] == 1 and s[1] == 7 and s[2] == 0, s          # one list from scratch, seven record builds from its candidates, no repeat
    assert stats(twin)[0] == 8 and stats(twin)[1] == 0
    # positions in another periodic image are the same particles: the displacement is measured with the minimum image
    cur = cur.copy()
    cur[:100, 1] += L
    a = step(pd, pse, cur, 998)
    s = stats(pse)
    assert s[0] == 1 and s[1] == 8 and s[2] == 0, s
    expect = np.zeros((n, 3), np.float32)
    ref.near_mdot(cur, f4, expect)
    assert np.abs(a[0] - expect).max() <= 1e-6 * np.abs(expect).max()
    # a jump: half the particles move by 0.6 skin along x
    cur = cur.copy()
    cur[::2, 0] += 0.6 * skin
    a = step(pd, pse, cur, 999)
    s = stats(pse)
    assert s[2] == 1 and s[0] == 2, s                          # measured broken, repeated from a fresh list
    pd3, fresh, _, _, _ = _pair(hip, o32, L, tol, psi, n)
    c = step(pd3, fresh, cur, 999)
    assert a[2] == c[2] and np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])
    expect = np.zeros((n, 3), np.float32)
    ref.near_mdot(cur, f4, expect)
    assert np.abs(a[0] - expect).max() <= 1e-6 * np.abs(expect).max()
    # (the jump is remembered as the largest one-step increase: the next steps build from scratch instead of risking a repeat)
    cur = cur.copy()
    cur[:, :3] += walk.uniform(-0.02 * skin, 0.02 * skin, (n, 3)).astype(np.float32)
    step(pd, pse, cur, 1000)
    assert stats(pse)[2] == 1 and stats(pse)[0] == 3


@pytest.mark.gpu
def test_mdot_rides_on_the_solve(hip, o32):
    """uammd_pse_near_set_mdot_rider: the next uammd_pse_near_stochastic also adds M_near F to MF — inside the solve's first product where
    that streams pair records, by the plain product otherwise (T = 0; the scanning products).  Against the two separate calls on a twin:
    the same noise bit for bit, M_near F against the oracle; one-shot (a second solve without re-arming leaves MF alone)."""
    from uammd_amd._lib import check
    from uammd_amd.md import _ptr, current_stream
    import ctypes as C
    L, n, tol, psi = 40.0, 4000, 1e-3, 0.6
    rng = np.random.default_rng(3)
    f4 = np.zeros((n, 4), np.float32)
    f4[:, :3] = rng.normal(0, 1, (n, 3))
    d_f = torch.from_numpy(f4).cuda()
    for records, T in ((1, 1.0), (0, 1.0), (1, 0.0)):
        pd, pse, ref, pos, _ = _pair(hip, o32, L, tol, psi, n)
        pd2, twin, _, _, _ = _pair(hip, o32, L, tol, psi, n)
        for h in (pse, twin):
            check(h.lib.uammd_pse_near_set_option(h.near, b"pair_list", records))
        expect = np.zeros((n, 3), np.float32)
        ref.near_mdot(pos, f4, expect)
        scale = np.abs(expect).max()
        MF = torch.full((n, 3), 0.5, dtype=torch.float32, device="cuda")
        BdW = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
        it = C.c_int(0)
        check(pse.lib.uammd_pse_near_set_mdot_rider(pse.near, _ptr(d_f), _ptr(MF)))
        check(pse.lib.uammd_pse_near_stochastic(pse.near, _ptr(pd.getPos()), n, T, 1.0, 77, _ptr(BdW), current_stream(), C.byref(it)))
        MF2 = torch.full((n, 3), 0.5, dtype=torch.float32, device="cuda")
        BdW2 = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
        it2 = C.c_int(0)
        check(twin.lib.uammd_pse_near_stochastic(twin.near, _ptr(pd2.getPos()), n, T, 1.0, 77, _ptr(BdW2), current_stream(), C.byref(it2)))
        check(twin.lib.uammd_pse_near_mdot(twin.near, _ptr(pd2.getPos()), _ptr(d_f), n, _ptr(MF2), current_stream()))
        assert it.value == it2.value and np.array_equal(BdW.cpu().numpy(), BdW2.cpu().numpy()), (records, T)
        got = MF.cpu().numpy() - 0.5
        assert np.abs(got - expect).max() <= 1e-6 * scale + 1.5e-7, (records, T, float(np.abs(got - expect).max() / scale))
        assert np.abs(MF.cpu().numpy() - MF2.cpu().numpy()).max() <= 1e-6 * scale + 1.5e-7
        # one-shot
        before = MF.clone()
        check(pse.lib.uammd_pse_near_stochastic(pse.near, _ptr(pd.getPos()), n, T, 1.0, 78, _ptr(BdW), current_stream(), C.byref(it)))
        assert torch.equal(MF, before)


@pytest.mark.gpu
def test_kept_lists_step_aside_in_a_box_of_three_cells(hip, o32):
    """A box that holds three cells of the cut-off per direction but not three of cut-off + skin: the 27-cell scan on the wider grid would
    meet a cell twice.  The handle keeps the reference's grid and builds its list per step (the mechanism reports itself off); products
    against the oracle before and after a move."""
    from uammd_amd._lib import check
    from uammd_amd.md import _ptr, current_stream
    import ctypes as C
    L, n, tol, psi = 16.0, 2000, 1e-3, 0.6        # rc = 4.38: 3 cells of 5.33; rc + 40 % = 6.13: 2 cells
    pd, pse, ref, pos, _ = _pair(hip, o32, L, tol, psi, n)
    assert int(L / pse.rcut) == 3 and int(L / (1.4 * pse.rcut)) == 2
    rng = np.random.default_rng(3)
    f4 = np.zeros((n, 4), np.float32)
    f4[:, :3] = rng.normal(0, 1, (n, 3))
    d_f = torch.from_numpy(f4).cuda()
    cur = pos.copy()
    for k in range(3):
        if k:
            cur = cur.copy()
            cur[:, :3] += rng.uniform(-0.01, 0.01, (n, 3)).astype(np.float32)
            pd.getPos("write").copy_(torch.from_numpy(cur).cuda())
        MF = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
        check(pse.lib.uammd_pse_near_mdot(pse.near, _ptr(pd.getPos()), _ptr(d_f), n, _ptr(MF), current_stream()))
        expect = np.zeros((n, 3), np.float32)
        ref.near_mdot(cur, f4, expect)
        assert np.abs(MF.cpu().numpy() - expect).max() <= 1e-6 * np.abs(expect).max(), k
    s = (C.c_longlong * 4)()
    check(pse.lib.uammd_pse_near_list_stats(pse.near, s))
    assert s[3] == 0 and s[1] == 0 and s[0] == 3


@pytest.mark.parametrize("wait", [False, True], ids=["queued", "waited"])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_pse_random_call_sequences(hip, seed, wait):
    """The PSE near field keeps state from call to call (a kept candidate list with its displacement bound, pair records built ahead and
    verified after the solve, the Lanczos check schedule, the far field queued from inside the solve) and reads the device's reports
    without waiting.  A random sequence of 40 calls on EulerMaruyama<PSE> at T > 0 — steps, particles moved together or jiggled one by
    one behind the lists' back, sortParticles, plain computeMF calls — against the same sequence on a handle with the pair records, the
    optimistic builds, the deferred checks and the fused recurrence switched off: the same trajectory to rounding (the draws are keyed by
    particle and by the System generator's stream, which both handles consume alike).  Queued without a wait and with a wait after every
    call: the two take different paths through the handle's state (tests/test_gpu_ibm_fcm.py::test_fcm_step_random_call_sequences found
    two bugs of the FCM handle that way)."""
    import math
    from uammd_amd._lib import check
    n, L = 20000, 64.0
    rng = np.random.default_rng(200 + seed)
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(-0.5, 0.5, (n, 3)) * L
    force = torch.from_numpy(rng.normal(0, 1, (n, 4)).astype(np.float32)).cuda()
    force[:, 3] = 0
    ops = []
    for _ in range(40):
        u = rng.uniform()
        if u < 0.6: ops.append(("step",))
        elif u < 0.72: ops.append(("move", float(rng.uniform(0.05, 2.0))))
        elif u < 0.82: ops.append(("sort",))
        elif u < 0.92: ops.append(("mf",))
        else: ops.append(("jiggle", int(rng.integers(1 << 30))))

    class Forces:
        def __init__(self, pd): self.pd = pd
        def sum(self, force_=False, energy=False, virial=False, **kw): self.pd.getForce("readwrite").add_(force)
        def updateSimulationTime(self, t): pass
        def updateTimeStep(self, dt): pass
        def updateTemperature(self, T): pass
        def updateBox(self, box): pass

    def run(tuned, wait_):
        pd = hip.ParticleData(n, seed=77)
        pd.setPos(pos.copy())
        par = hip.BDHI.PSE.Parameters(temperature=0.5, viscosity=1.0, hydrodynamicRadius=1.0, dt=0.01, box=hip.Box(L), tolerance=1e-3, psi=0.5)
        em = hip.BDHI.EulerMaruyama(pd, par, Method=hip.BDHI.PSE)
        pse = em.bdhi
        if not tuned:
            for name in ("pair_list", "optimistic_records", "defer_checks", "fuse_recurrence"):
                check(pse.lib.uammd_pse_near_set_option(pse.near, name.encode(), 0))
        em.addInteractor(Forces(pd))
        MF = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
        for op in ops:
            if op[0] == "step":
                em.forwardTime()
            elif op[0] == "move":
                pd.getPos("readwrite")[:, :3] += op[1]
            elif op[0] == "sort":
                pd.hintSortByHash(hip.Box(L), [5.3] * 3)
                pd.sortParticles()
            elif op[0] == "mf":
                pd.getForce("write").zero_()
                em.interactors[0].sum(True)
                pse.computeMF(MF)
            else:
                g = torch.Generator(device="cuda").manual_seed(op[1])
                pd.getPos("readwrite")[:, :3] += 0.3 * torch.randn((n, 3), generator=g, device="cuda")
            if wait_:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return pd.getPos("read").cpu().numpy()   # (the same calls permute both handles' arrays alike)

    ref = run(False, True)
    out = run(True, wait)
    d = np.abs(out[:, :3] - ref[:, :3])
    d = np.minimum(d, L - d).max(axis=1)
    nsteps = sum(1 for op in ops if op[0] == "step")
    assert nsteps > 10 and np.isfinite(out).all()
    # a step's noise displacement is ~ sqrt(2 T M0 dt) = 0.02: a lost or stale list shows at that size; rounding stays below 5e-5
    assert d.max() <= 5e-5, (float(d.max()), int((d > 5e-5).sum()))
