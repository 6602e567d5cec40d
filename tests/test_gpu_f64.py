"""The reference's double-precision known answers ON THE GPU, at the reference's own tolerances, through the DOUBLE_PRECISION build of the
library (the `_f64` entry points of include/uammd_hip.h; global/defines.h:9-11 makes `real` a build switch and test/CMakeLists.txt:9 /
test/BDHI/FCM/Makefile:9 compile the reference's tests with it):

  test/misc/ibm/test_ibm_regular.cu:16-64     constant window: 27 / 27 / 8 nodes                                  exact
  test/misc/ibm/test_ibm_regular.cu:113-136   Peskin spreading against the brute-force weights                    1e-10
  test/misc/ibm/test_ibm_regular.cu:240-274   interpolation of a random field, 128 particles, 32^3                1e-10
  test/BDHI/FCM/fcm_test.cu:85-144            Hasimoto self mobility, tolerance 1e-8, L = 96 h ceil(a/h) = 288^3   1e-8
  test/misc/lanczos/test_lanczos.cu:34-93,236-269   identity, 2 I in <= 5 steps, dense SPD operator M^2 -> M v     1e-7
  test/BDHI/PSE/pse_test.cu:64-117            PSE self mobility = Hasimoto, tolerance 1e-8, L = 128 a, psi = 1      1e-8

plus the same kernels against the double-precision oracle on random inputs (1e-12: same arithmetic, another summation order)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def f64(hip):
    from uammd_amd import f64 as m
    return m


def _peskin3(h):
    def phi(rr):
        r = abs(rr) / h
        if r < 0.5:
            return (1 + math.sqrt(1 - 3 * r * r)) / (3 * h)
        if r < 1.5:
            return (5 - 3 * r - math.sqrt(1 - 3 * (1 - r) ** 2)) / (6 * h)
        return 0.0
    return phi


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def test_constant_kernel_counts(f64):
    """test_ibm_regular.cu:16-64."""
    k = f64.Kernels.Constant(3)
    for L, pos, per, expect in [(1.0, (0, 0, 0), 1, 27), (3.0, (-1, -1, -1), 1, 27), (3.0, (-1, -1, -1), 0, 8)]:
        ibm = f64.IBM(k, L, per, [3, 3, 3])
        g = torch.zeros((3, 3, 3), dtype=torch.float64, device="cuda")
        ibm.spread(_dev([pos]), _dev([1.0]), g)
        assert float(g.sum()) == expect


def test_peskin_spread_bruteforce(f64):
    """test_ibm_regular.cu:113-136, tolerance 1e-10."""
    n, L = 8, 16.0
    h = L / n
    ibm = f64.IBM(f64.Kernels.Peskin3pt(h), L, 1, [n] * 3)
    g = torch.zeros((n, n, n), dtype=torch.float64, device="cuda")
    ibm.spread(_dev(np.zeros((1, 3))), _dev([1.0]), g)
    phi = _peskin3(h)
    w = np.array([phi(-L / 2 + (i + 0.5) * h) for i in range(n)])
    assert np.abs(g.cpu().numpy() - w[:, None, None] * w[None, :, None] * w[None, None, :]).max() <= 1e-10


def test_interpolation_random_field(f64):
    """test_ibm_regular.cu:240-274: 128 particles kept 2 h from the faces, 32^3, random field, against brute force, 1e-10."""
    n, L, N = 32, 16.0, 128
    h = L / n
    rng = np.random.default_rng(123)
    pos = rng.uniform(-L / 2 + 2 * h, L / 2 - 2 * h, (N, 3))
    field = rng.uniform(-L / 2 + 2 * h, L / 2 - 2 * h, (n, n, n))
    ibm = f64.IBM(f64.Kernels.Peskin3pt(h), L, 1, [n] * 3)
    out = torch.zeros(N, dtype=torch.float64, device="cuda")
    ibm.gather(_dev(pos), out, _dev(field))
    out = out.cpu().numpy()
    phi = _peskin3(h)
    c = -L / 2 + (np.arange(n) + 0.5) * h
    for i in range(N):
        wx = np.array([phi(x - pos[i, 0]) for x in c]); wy = np.array([phi(y - pos[i, 1]) for y in c])
        wz = np.array([phi(z - pos[i, 2]) for z in c])
        exp = np.einsum("k,j,i,kji->", wz, wy, wx, field) * h ** 3
        assert abs(out[i] - exp) <= 1e-10


@pytest.mark.parametrize("kind", ["gaussian-tol1e-8", "peskin4", "gaussian-3comp-nonperiodic"])
def test_ibm_against_the_double_oracle(f64, o64, kind):
    rng = np.random.default_rng(5)
    cells, L, per = [24, 20, 28], np.array([12.0, 10.0, 14.0]), 1
    N, ncomp = 300, 1
    if kind == "gaussian-tol1e-8":
        kg, _ = f64.Kernels.Gaussian(0.5, 1e-8)
        ko = o64.fcm_gaussian(0.5, 1e-8)["kernel"]
    elif kind == "peskin4":
        kg = f64.Kernels.Peskin4pt(L / np.asarray(cells))
        ko = o64.ibm_kernel("peskin4", 4, invh=list(np.asarray(cells) / L))
    else:
        kg, _ = f64.Kernels.Gaussian(0.5, 1e-4)
        ko = o64.fcm_gaussian(0.5, 1e-4)["kernel"]
        per, ncomp = [1, 0, 1], 3
    pos = rng.uniform(-0.5, 0.5, (N, 3)) * L * 1.2        # some outside the primary box
    if kind == "gaussian-3comp-nonperiodic":
        pos[:, 1] = rng.uniform(-0.5, 0.5, N) * L[1] * 0.99
    q = rng.normal(0, 1, (N, ncomp))
    shape = (cells[2], cells[1], cells[0], ncomp)
    ibm = f64.IBM(kg, L, per, cells)
    g = torch.zeros(shape, dtype=torch.float64, device="cuda")
    ibm.spread(_dev(pos), _dev(q if ncomp > 1 else q[:, 0]), g)
    ref = o64.ibm_spread(pos, q, L, per, cells, ko)
    assert np.abs(g.cpu().numpy() - ref).max() <= 1e-12 * np.abs(ref).max()
    field = rng.normal(0, 1, shape)
    out = torch.zeros((N, ncomp) if ncomp > 1 else (N,), dtype=torch.float64, device="cuda")
    ibm.gather(_dev(pos), out, _dev(field))
    refg = o64.ibm_gather(pos, field, L, per, cells, ko)
    assert np.abs(out.cpu().numpy().reshape(N, ncomp) - refg).max() <= 1e-12 * np.abs(refg).max()


def _mobility_error(f64, o64, mult, positions):
    """fcm_test.cu:85-144: a = 1.012312, eta = 1.12321, tolerance 1e-8, h from adviseGridSize, L = mult h ceil(a / h), positions from
    Saru(1234).f(-0.5, 0.5) L, unit force along each axis, the pulled particle must move with the Hasimoto mobility."""
    a, eta, tol = 1.012312, 1.12321, 1e-8
    h = f64.Kernels.adviseGridSize(a, tol)
    L = mult * h * math.ceil(a / h)
    cells = int(L / h)
    k, a_eff = f64.Kernels.Gaussian(L / cells, tol)
    assert abs(a_eff - a) < 1e-12 and k.support[0] == 17
    fcm = f64.FCM_impl(L, [cells] * 3, k, eta, a_eff)
    m0 = fcm.getSelfMobility()
    u = o64.saru_f_range(1234, -0.5, 0.5, 3 * positions).reshape(positions, 3)       # the reference's own random positions
    worst = 0.0
    for j in range(positions):
        for d in range(3):
            f = np.zeros((1, 4)); f[0, d] = 1.0
            v = fcm.computeHydrodynamicDisplacements(_dev([[u[j, 0] * L, u[j, 1] * L, u[j, 2] * L, 0.0]]), _dev(f)).cpu().numpy()[0]
            exp = np.zeros(3); exp[d] = m0
            worst = max(worst, np.abs(v - exp).max())
    return worst, cells


def test_fcm_hasimoto_self_mobility_reference_configuration(f64, o64):
    """EXACTLY fcm_test.cu:85-144: 288^3 grid (3 x 288^3 doubles = 573 MB), support 17, error <= 1e-8, the reference's ten positions."""
    worst, cells = _mobility_error(f64, o64, 96, 10)
    print(f"[f64 FCM Hasimoto, {cells}^3] max |M - M0| = {worst:.2e}")
    assert cells == 288 and worst <= 1e-8


def test_fcm_against_the_double_oracle(f64, o64):
    from oracle.fcm import FCMOracle
    cells, L, n, tol, eta = [40, 36, 48], np.array([20.0, 18.0, 24.0]), 500, 1e-6, 0.9
    rng = np.random.default_rng(9)
    pos = np.zeros((n, 4)); pos[:, :3] = rng.uniform(-0.7, 0.7, (n, 3)) * L
    force = np.zeros((n, 4)); force[:, :3] = rng.normal(0, 1, (n, 3))
    h = float(min(L / np.asarray(cells)))
    k, a_eff = f64.Kernels.Gaussian(h, tol)
    fcm = f64.FCM_impl(L, cells, k, eta, a_eff)
    v = fcm.computeHydrodynamicDisplacements(_dev(pos), _dev(force)).cpu().numpy()
    ref = FCMOracle(o64, L, cells, tolerance=tol, viscosity=eta).displacements(pos, force)
    assert np.linalg.norm(v - ref) <= 1e-11 * np.linalg.norm(ref)


def _dense_case(size):
    from oracle.lanczos import std_mt19937_uniform_real
    A = std_mt19937_uniform_real(29374238, size * size, 0.0, 1.0).reshape(size, size)
    M = 0.5 * (A + A.T) + 5 * size * np.eye(size)
    return M, M @ M.T, std_mt19937_uniform_real(1234567, size, -10.0, 10.0)


def test_lanczos_reference_tests(f64):
    """test_lanczos.cu: identity and 2 I for every size up to 128 (2 I in <= 5 steps), random diagonal matrices, and the dense SPD
    operator M M^T whose square root applied to v is M v — all to the reference's 1e-7, with the reference's mt19937 inputs."""
    from oracle.lanczos import std_mt19937_uniform_real
    solver = f64.LanczosSolver()
    for scale, expect in ((1.0, 1.0), (2.0, math.sqrt(2.0))):
        for size in list(range(1, 40)) + [64, 127, 128]:
            v, Bv = _dev(np.ones(size)), torch.zeros(size, dtype=torch.float64, device="cuda")
            it = solver.run(lambda x, y: y.copy_(scale * x), Bv, v, 1e-7)
            assert np.abs(Bv.cpu().numpy() - expect).max() <= 1e-7, (scale, size)
            if scale == 2.0:
                assert it <= 5 and solver.getLastRunRequiredSteps() <= 5          # test_lanczos.cu:78-93
    m = std_mt19937_uniform_real(29374238, 128, 1.0, 2.0)
    for size in (1, 2, 3, 17, 64, 100, 127):
        d, v = _dev(m[:size]), std_mt19937_uniform_real(1234567, size, -10.0, 10.0)
        Bv = torch.zeros(size, dtype=torch.float64, device="cuda")
        solver.run(lambda x, y: torch.mul(d, x, out=y), Bv, _dev(v), 1e-7)
        theory = np.sqrt(m[:size]) * v
        assert np.abs((Bv.cpu().numpy() - theory) / theory).max() <= 1e-7, size
    for size in (1, 2, 5, 31, 64, 65, 128, 200, 255, 256, 300, 511):                     # test_lanczos.cu:236-269
        M, M2, v = _dense_case(size)
        dM2, Bv = _dev(M2), torch.zeros(size, dtype=torch.float64, device="cuda")
        solver.run(lambda x, y: torch.mv(dM2, x, out=y), Bv, _dev(v), 1e-7)
        theory = M @ v
        assert np.abs((Bv.cpu().numpy() - theory) / theory).max() <= 1e-7, size


def test_lanczos_golden_vector(f64):
    """tests/golden/lanczos_dense.npz: Bv and the iteration count of the double-precision solve."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lanczos_dense.npz"))
    dM2, Bv = _dev(g["M2"]), torch.zeros(int(g["size"]), dtype=torch.float64, device="cuda")
    it = f64.LanczosSolver().run(lambda x, y: torch.mv(dM2, x, out=y), Bv, _dev(g["v"]), float(g["tolerance_f64"]))
    assert it == int(g["iterations_f64"])
    assert np.abs(Bv.cpu().numpy() - g["Bv_f64"]).max() <= 1e-11 * np.abs(g["Bv_f64"]).max()


RH, VISC = 1.012312, 1.12321


def test_pse_self_mobility_reference_configuration(f64, o64):
    """pse_test.cu:64-117: tolerance 1e-8, L = 128 a, psi = 1 (360^3 far-field grid, support 13: 1.1 GB of doubles), one particle at
    the reference's Saru(1234) position pulled along x, y and z: M F = Hasimoto's mobility within the tolerance."""
    L = 128 * RH
    p = f64.PSE(L, VISC, RH, 1e-8, 1.0)
    m0 = p.getSelfMobility()
    u = o64.saru_f_range(1234, -0.5, 0.5, 3)
    pos = np.zeros((1, 4)); pos[0, :3] = u * L
    worst = 0.0
    for d in range(3):
        f = np.zeros((1, 4)); f[0, d] = 1.0
        MF = p.computeMF(_dev(pos), _dev(f)).cpu().numpy()[0]
        exp = np.zeros(3); exp[d] = m0
        worst = max(worst, np.abs(MF - exp).max())
    print(f"[f64 PSE self mobility, grid {p.cells}, support {p.support}] max |M - M0| = {worst:.2e}")
    assert list(p.cells) == [360, 360, 360] and worst <= 1e-8


def test_pse_against_the_double_oracle(f64, o64):
    from oracle.pse import PSEOracle
    L, n, tol, psi = 24.0, 60, 1e-6, 0.8
    rng = np.random.default_rng(3)
    pos = np.zeros((n, 4)); pos[:, :3] = rng.uniform(-L / 2, L / 2, (n, 3))
    f = np.zeros((n, 4)); f[:, :3] = rng.normal(0, 1, (n, 3))
    p = f64.PSE(L, VISC, RH, tol, psi)
    ref = PSEOracle(o64, [L] * 3, RH, VISC, tol, psi)
    assert p.nPointsTable == ref.nPointsTable and abs(p.rcut - float(ref.rcut)) <= 1e-15 and list(p.cells) == list(ref.cells)
    assert p.support == ref.support and abs(p.eta - float(ref.eta)) <= 1e-14 * float(ref.eta)
    MF = p.computeMF(_dev(pos), _dev(f)).cpu().numpy()
    exp = ref.computeHydrodynamicDisplacements(pos, f, 0.0, 0.0)
    assert np.abs(MF - exp).max() <= 1e-11 * np.abs(exp).max()


@pytest.mark.parametrize("mode", ["Quasi2D", "True2D"])
def test_bdhi2d_step_vs_oracle(f64, o64, mode):
    """uammd_bdhi2d_*_f64 against the double-precision oracle (Integrator/Hydro/BDHI_quasi2D.cu:179-541): forces only, forces + thermal
    drift + noise (the same Saru streams), noise only; two steps each (the noise counter advances).  Same arithmetic, another summation
    order in the spread: 1e-11 of the largest velocity at T = 0; with noise 2e-6 — the draws are Saru's single-precision Gaussians in both
    (promoted to double, as the reference's DOUBLE_PRECISION build does), and Box-Muller's logf / sincosf differ between libm and the
    device by an ulp of a float."""
    from oracle.quasi2d import BDHI2DOracle
    n, a, visc, T, dt, seed, L = 300, 1.1, 1.3, 0.7, 0.05, 90210, (36.0, 50.0)
    rng = np.random.default_rng(2)
    pos = np.zeros((n, 4))
    pos[:, 0] = rng.uniform(-0.7, 0.7, n) * L[0]      # some particles outside the primary box
    pos[:, 1] = rng.uniform(-0.7, 0.7, n) * L[1]
    force = np.zeros((n, 4))
    force[:, :2] = rng.normal(0, 1, (n, 2))
    for temperature, with_forces in [(0.0, True), (T, True), (T, False)]:
        bd = f64.BDHI2D(mode, L, a, visc, temperature, dt, seed)
        ref = BDHI2DOracle(o64, mode, L, a, visc, temperature, dt, seed=seed)
        assert bd.cells == [int(ref.cells[0]), int(ref.cells[1])] and bd.support == ref.support
        dpos, rpos = _dev(pos), pos.copy()
        for step in range(2):
            v = bd.forwardTime(dpos, _dev(force) if with_forces else None).cpu().numpy()
            rv = ref.forwardTime(rpos, force if with_forces else None)
            assert np.abs(v - rv).max() <= (2e-6 if temperature > 0 else 1e-11) * np.abs(rv).max(), (mode, temperature, with_forces, step)
        assert np.abs(dpos.cpu().numpy() - rpos).max() <= (1e-6 if temperature > 0 else 1e-12)


def test_poisson_vs_oracle(f64, o64):
    """uammd_poisson_*_f64 against the double-precision oracle (Interactor/SpectralEwaldPoisson.cu): forces, energies and the field /
    potential at the particles of a neutral random set; grid, window support, near cut-off and table size equal."""
    from oracle.poisson import PoissonOracle
    n, L, eps, gw, tol, split = 400, 24.0, 1.7, 0.35, 1e-6, 0.9
    rng = np.random.default_rng(5)
    pos = np.zeros((n, 4))
    pos[:, :3] = rng.uniform(-0.6, 0.6, (n, 3)) * L
    q = rng.normal(0, 1, n)
    q -= q.mean()
    ps = f64.Poisson(L, eps, gw, tol, split)
    ref = PoissonOracle(o64, L, eps, gw, tol, split)
    assert ps.cells == [int(c) for c in ref.cells] and ps.support == ref.support and ps.ntable == ref.ntable
    assert abs(ps.nearFieldCutOff - float(ref.nearFieldCutOff)) <= 1e-12 * ps.nearFieldCutOff
    f, e = torch.zeros((n, 4), dtype=torch.float64, device="cuda"), torch.zeros(n, dtype=torch.float64, device="cuda")
    ps.sum(_dev(pos), _dev(q), f, e, True, True)
    rf, re = np.zeros((n, 4)), np.zeros(n)
    ref.sum(pos, q, rf, re, True, True)
    assert np.abs(f.cpu().numpy() - rf).max() <= 1e-10 * np.abs(rf).max()
    assert np.abs(e.cpu().numpy() - re).max() <= 1e-10 * np.abs(re).max()
    fp = ps.computeFieldPotentialAtParticles(_dev(pos), _dev(q)).cpu().numpy()
    rfp = ref.computeFieldPotentialAtParticles(pos, q)
    assert np.abs(fp - rfp).max() <= 1e-10 * np.abs(rfp).max()


@pytest.mark.parametrize("sizes", ["equal", "different"])
def test_rpy_open_boundary_f64_vs_oracle(hip, o64, sizes):
    """uammd_rpy_nbody_mdot_f64 (BDHI::Lanczos::computeMF with real = double, BDHI_Lanczos.cu:120-160) against the double-precision oracle:
    the same j order and FMA placement, 1e-13 of max|Mv|; uammd_bdhi_cholesky_*_f64 (BDHI_Cholesky.cu: dense matrix, dsymv) gives the same
    product to the rounding of another summation order; its factor (dpotrf, applied by dtrmv to unit vectors on a set of 24 spheres)
    reproduces the oracle's dense matrix: B B^T = M to 1e-12."""
    import ctypes as C
    from oracle.pse import rpy_nbody_mdot
    from uammd_amd import _lib
    check, lib = _lib.check, _lib.load()
    rng = np.random.default_rng(6)
    n, visc = 700, 1.1
    pos = np.zeros((n, 4))
    pos[:, :3] = rng.uniform(-9, 9, (n, 3))
    pos[1, :3] = pos[0, :3]                                   # coincident pair: the r = 0 branch between different particles
    radius = rng.uniform(0.4, 1.1, n) if sizes == "different" else None
    rh = -1.0 if sizes == "different" else 0.8
    f4 = np.zeros((n, 4))
    f4[:, :3] = rng.normal(0, 1, (n, 3))
    dpos, df, drad = _dev(pos), _dev(f4), (_dev(radius) if radius is not None else None)
    ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    MF = torch.full((n, 3), 3.0, dtype=torch.float64, device="cuda")
    check(lib.uammd_rpy_nbody_mdot_f64(ptr(dpos), ptr(df), 4, ptr(drad), rh, visc, n, ptr(MF), None))
    torch.cuda.synchronize()
    expect = rpy_nbody_mdot(o64, pos, f4, visc, rh, radius)
    got = MF.cpu().numpy()
    assert np.abs(got - expect).max() <= 1e-13 * np.abs(expect).max()
    h = C.c_void_p()
    check(lib.uammd_bdhi_cholesky_create_f64(n, visc, rh, C.byref(h)))
    try:
        check(lib.uammd_bdhi_cholesky_setup_step_f64(h, ptr(dpos), None, ptr(drad), None))
        MF2 = torch.zeros((n, 3), dtype=torch.float64, device="cuda")
        check(lib.uammd_bdhi_cholesky_mf_f64(h, ptr(dpos), ptr(df), None, ptr(drad), ptr(MF2), None))
        torch.cuda.synchronize()
        assert np.abs(MF2.cpu().numpy() - expect).max() <= 1e-12 * np.abs(expect).max()
    finally:
        lib.uammd_bdhi_cholesky_destroy_f64(h)
    # the factor: B = U^T column by column (computeBdW maps the draws w to B w, BDHI_Cholesky.cu:235-262)
    from oracle.pse import rpy_dense
    m = 24
    p24 = np.zeros((m, 4))
    p24[:, :3] = rng.uniform(-3, 3, (m, 3))
    r24 = rng.uniform(0.4, 1.1, m) if sizes == "different" else None
    M = rpy_dense(o64, p24, r24, rh, visc).astype(np.float64)
    M = np.triu(M) + np.triu(M, 1).T          # (the reference fills the upper triangle)
    h = C.c_void_p()
    check(lib.uammd_bdhi_cholesky_create_f64(m, visc, rh, C.byref(h)))
    try:
        d24, dr24 = _dev(p24), (_dev(r24) if r24 is not None else None)
        B = np.zeros((3 * m, 3 * m))
        for k in range(3 * m):
            e = np.zeros(3 * m)
            e[k] = 1.0
            de = _dev(e)
            check(lib.uammd_bdhi_cholesky_setup_step_f64(h, ptr(d24), None, ptr(dr24), None))
            check(lib.uammd_bdhi_cholesky_bdw_f64(h, ptr(d24), None, ptr(dr24), ptr(de), None))
            torch.cuda.synchronize()
            B[:, k] = de.cpu().numpy()
        assert np.abs(B @ B.T - M).max() <= 1e-12 * np.abs(M).max()
    finally:
        lib.uammd_bdhi_cholesky_destroy_f64(h)


@pytest.mark.parametrize("T", [0.0, 1.0])
@pytest.mark.parametrize("scheme", ["EulerMaruyama", "MidPoint", "AdamsBashforth", "Leimkuhler"])
def test_bd_schemes_f64_vs_oracle(hip, o64, scheme, T):
    """The four BD schemes with real = double (Integrator/BrownianDynamics.cu:119-144, :178-214, :262-289, :313-345 as test/BD/Makefile:2
    builds them) through uammd_bd_scheme_step_f64 against the double build of the oracle's restatement, three steps as forwardTime strings
    the calls: a proper subgroup, shear, per-particle radii, forces that change between the calls.  T = 0: the same bits.  T > 0: the draws
    are FLOAT Gaussians in both (Saru::gf) and differ by libm vs device logf / sinf — 4e-5 of the noise amplitude, as in single precision."""
    import ctypes as C
    from uammd_amd._lib import check, load
    lib = load()
    n, dt, M, seed = 6000, 0.1, 1.0 / (6 * math.pi), 24680
    rng = np.random.default_rng(3)
    pos = np.zeros((n, 4), np.float64)
    pos[:, :3] = rng.uniform(-0.5, 0.5, (n, 3))
    pos[:, 3] = rng.integers(0, 3, n)
    radius = rng.uniform(0.5, 1.5, n)
    index = np.sort(rng.choice(n, 4000, replace=False)).astype(np.int32)
    original = rng.permutation(n).astype(np.int32)
    K = np.array([[0, 0.1, 0], [0, 0, 0], [0.2, 0, 0]], np.float64)
    Kc = (C.c_double * 9)(*[float(x) for x in K.reshape(9)])
    code = {"EulerMaruyama": 0, "MidPoint": 1, "AdamsBashforth": 2, "Leimkuhler": 3}[scheme]
    rp, dp = pos.copy(), torch.from_numpy(pos.copy()).cuda()
    raux, daux = np.zeros((len(index), 4), np.float64), torch.zeros((len(index), 4), dtype=torch.float64, device="cuda")
    dindex, dorig, dradius = torch.from_numpy(index).cuda(), torch.from_numpy(original).cuda(), torch.from_numpy(radius).cuda()
    P = lambda t: C.c_void_p(t.data_ptr())

    def forces(k):
        f = np.zeros((n, 4), np.float64)
        f[:, :3] = np.random.default_rng(100 + k).normal(0, 1, (n, 3))
        return f

    def gpu(sub, f, step):
        check(lib.uammd_bd_scheme_step_f64(code, sub, P(dp), P(daux), P(dindex), P(dorig), P(f), Kc, M, P(dradius), dt, 0, T, len(index), step, seed, None))

    B = math.sqrt(2 * T * M * dt / 0.5)   # the largest noise amplitude among the members (radius >= 0.5)
    for step in range(1, 4):
        f0, f1 = forces(2 * step), forces(2 * step + 1)
        if scheme == "EulerMaruyama":
            o64.bd_euler_maruyama(rp, f0, M, dt, T, step, seed, K=K, radius=radius, index=index)
            gpu(0, torch.from_numpy(f0).cuda(), step)
        elif scheme == "MidPoint":
            o64.bd_midpoint(0, rp, raux, f0, M, dt, T, step, seed, K=K, radius=radius, index=index)
            gpu(0, torch.from_numpy(f0).cuda(), step)
            o64.bd_midpoint(1, rp, raux, f1, M, dt, T, step, seed, K=K, radius=radius, index=index)
            gpu(1, torch.from_numpy(f1).cuda(), step)
        elif scheme == "AdamsBashforth":
            raux[:] = f0[index]
            daux.copy_(torch.from_numpy(f0[index]))
            o64.bd_adams_bashforth(rp, raux, f1, M, dt, T, step, seed, K=K, radius=radius, index=index)
            gpu(0, torch.from_numpy(f1).cuda(), step)
        else:
            o64.bd_leimkuhler(rp, f0, M, dt, T, step, seed, K=K, radius=radius, index=index, original_index=original)
            gpu(0, torch.from_numpy(f0).cuda(), step)
        torch.cuda.synchronize()
        got = dp.cpu().numpy()
        if T == 0:
            assert np.array_equal(got, rp), np.abs(got - rp).max()
        else:
            assert np.abs(got - rp).max() <= 4e-5 * B * step
            assert np.array_equal(got[:, 3], rp[:, 3])
        others = np.setdiff1d(np.arange(n), index)
        assert np.array_equal(got[others], pos[others])      # nobody outside the group moved


@pytest.mark.parametrize("scheme", ["EulerMaruyama", "MidPoint", "AdamsBashforth", "Leimkuhler"])
def test_bd_python_mirror_f64(f64, scheme):
    """uammd_amd.f64.BD (the Python mirror of the double-precision BD classes): a constant force at T = 0 moves every particle by
    n dt M F to rounding, whatever the scheme; at T > 0 free particles spread with <x^2> = 2 T M dt n per axis (Leimkuhler: n - 1/2)."""
    n, dt, steps = 4096, 0.125, 8
    rng = np.random.default_rng(5)
    start = np.zeros((n, 4)); start[:, :3] = rng.uniform(-3, 3, (n, 3))
    pull = torch.tensor([0.25, -0.5, 1.0, 0.0], dtype=torch.float64, device="cuda")

    def forces(pos, force):
        force += pull

    bd = f64.BD(scheme, 0.0, 1.0 / (6 * math.pi), 1.0, dt, 1234, forces=forces)
    pos = _dev(start)
    for _ in range(steps):
        bd.forwardTime(pos)
    torch.cuda.synchronize()
    expect = start[:, :3] + steps * dt * np.array([0.25, -0.5, 1.0])
    assert np.abs(pos.cpu().numpy()[:, :3] - expect).max() < 1e-13
    T = 0.7
    bd = f64.BD(scheme, T, 1.0 / (6 * math.pi), 1.0, dt, 4321)
    pos = torch.zeros((16384, 4), dtype=torch.float64, device="cuda")
    for _ in range(steps):
        bd.forwardTime(pos)
    msd = float((pos[:, :3] ** 2).mean())
    expected = 2 * T * dt * (steps - 0.5 if scheme == "Leimkuhler" else steps)
    assert abs(msd / expected - 1) < 0.03, (msd, expected)      # 49152 samples of a sum of 8 steps: the estimate's own deviation is ~0.7 %
