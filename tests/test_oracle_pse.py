"""CPU: the PSE restatement (oracle/src/pse.c + oracle/pse.py) pinned against the reference's own known answers
(test/BDHI/PSE/pse_test.cu): the pulled-particle self mobility equals the Hasimoto-corrected value within `tolerance`
(:64-118, the reference asserts DoubleNear(m0, tolerance)), and the self-diffusion <dx^2> = 2 T M0 within 1e-2
(:121-158).  Plus properties the method must have: independence of the Ewald splitting psi, symmetry of M_near, the
closed-form F, G against the plain RPY tensor."""
import math

import numpy as np
import pytest

RH, VISC = 1.012312, 1.12321      # pse_test.cu:66-67


def _pse(o, L, tol, psi, **kw):
    from oracle.pse import PSEOracle
    return PSEOracle(o, [L] * 3, RH, VISC, tol, psi, **kw)


@pytest.mark.parametrize("tol,Lmult", [(1e-4, 32), (1e-6, 32)])
def test_self_mobility_reference_test(o64, tol, Lmult):
    """pse_test.cu:64-118 at a CPU-sized box: positions from Saru(1234).f(-0.5, 0.5), the three directions."""
    L = Lmult * RH
    p = _pse(o64, L, tol, 1.0)
    u = o64.saru_f_range(1234, -0.5, 0.5, 6).reshape(2, 3)
    m0 = p.getSelfMobility()
    for j in range(2):
        pos = np.zeros((1, 4))
        pos[0, :3] = u[j] * L
        for d in range(3):
            f = np.zeros((1, 4))
            f[0, d] = 1.0
            MF = p.computeHydrodynamicDisplacements(pos, f, 0.0, 0.0)
            expect = np.zeros(3)
            expect[d] = m0
            assert np.abs(MF[0] - expect).max() <= tol, (j, d, MF[0] - expect)


@pytest.mark.slow
def test_self_mobility_reference_configuration(o64):
    """The reference's exact configuration: tolerance 1e-8, L = 128 a, psi = 1 (360^3 grid, support 13)."""
    L = 128 * RH
    p = _pse(o64, L, 1e-8, 1.0)
    u = o64.saru_f_range(1234, -0.5, 0.5, 3)
    pos = np.zeros((1, 4))
    pos[0, :3] = u * L
    f = np.zeros((1, 4))
    f[0, 0] = 1.0
    MF = p.computeHydrodynamicDisplacements(pos, f, 0.0, 0.0)
    assert abs(MF[0, 0] - p.getSelfMobility()) <= 1e-8 and np.abs(MF[0, 1:]).max() <= 1e-8


def test_result_does_not_depend_on_the_splitting(o64):
    """M = M_near(psi) + M_far(psi) for any psi.  The reference's cut-off heuristic rcut = sqrt(-ln tol)/psi drops a
    near-field tail of relative size ~ tol^(1/2..1) (exp(-psi^2 r^2) times a polynomial), so two splittings agree to a few
    1e-4 at tol = 1e-5 and the difference shrinks with the tolerance; the closed forms themselves are checked to 1e-7
    against direct Ewald sums in test_splitting_identity_direct_sums."""
    L, n = 24.0, 40
    rng = np.random.default_rng(3)
    pos = np.zeros((n, 4))
    pos[:, :3] = rng.uniform(-L / 2, L / 2, (n, 3))
    f = np.zeros((n, 4))
    f[:, :3] = rng.normal(0, 1, (n, 3))
    err = []
    for tol in (1e-5, 1e-7):
        a = _pse(o64, L, tol, 0.6).computeHydrodynamicDisplacements(pos, f, 0.0, 0.0)
        b = _pse(o64, L, tol, 1.1).computeHydrodynamicDisplacements(pos, f, 0.0, 0.0)
        err.append(np.abs(a - b).max() / np.abs(a).max())
    assert err[0] <= 2e-3 and err[1] <= 2e-4 and err[1] < err[0] / 4, err


def test_splitting_identity_direct_sums(o64):
    """Closed-form near field (summed over images) + the far-field greens function summed directly over k (no grid):
    independent of psi to 1e-7 -> F, G (RPY_PSE.cuh) and B(k) (FarField.cuh:85-119) are consistent with each other."""
    import ctypes as C
    from oracle.oracle import _p
    a, eta_v, L = 1.0, 1.0, 12.0

    def near_sum(r, psi, nimg=3):
        M = np.zeros((3, 3))
        fg = np.zeros(2)
        for i in range(-nimg, nimg + 1):
            for j in range(-nimg, nimg + 1):
                for k in range(-nimg, nimg + 1):
                    d = np.array(r) + L * np.array([i, j, k])
                    rr = np.linalg.norm(d)
                    o64.lib.oracle_pse_rpy_near_fandg(C.c_double(rr), C.c_double(a), C.c_double(psi), C.c_double(1e9), _p(fg))
                    M += (fg[0] * np.eye(3) + (fg[1] - fg[0]) * np.outer(d, d) / rr ** 2) / (6 * math.pi * a * eta_v)
        return M

    def far_sum(r, psi, nk=24):
        rng = np.arange(-nk, nk + 1)
        I, J, K = np.meshgrid(rng, rng, rng, indexing="ij")
        kk = np.stack([I.ravel(), J.ravel(), K.ravel()], 1).astype(float) * 2 * math.pi / L
        k2 = (kk ** 2).sum(1)
        kk, k2 = kk[k2 > 0], k2[k2 > 0]
        km = np.sqrt(k2)
        B = (np.sin(km * a) / (km * a)) ** 2 * (1 + k2 / (4 * psi ** 2)) * np.exp(-k2 / (4 * psi ** 2)) / (eta_v * L ** 3 * k2)
        w = B * np.cos(kk @ np.array(r))
        return np.eye(3) * w.sum() - np.einsum("n,ni,nj->ij", w / k2, kk, kk)
    for r in ([2.5, 0.7, -1.1], [1.2, 0.3, 0.4]):          # r > 2a and the overlapping branch r < 2a
        tot = [near_sum(r, psi) + far_sum(r, psi) for psi in (0.6, 0.9, 1.1)]
        assert np.abs(tot[0] - tot[1]).max() <= 1e-7 * np.abs(tot[0]).max()
        assert np.abs(tot[0] - tot[2]).max() <= 1e-7 * np.abs(tot[0]).max()


def test_near_matrix_is_symmetric_and_matches_closed_form(o64):
    L, tol, psi, n = 20.0, 1e-4, 0.7, 60
    p = _pse(o64, L, tol, psi)
    rng = np.random.default_rng(5)
    pos = np.zeros((n, 4))
    pos[:, :3] = rng.uniform(-L / 2, L / 2, (n, 3))
    pos[1, :3] = pos[0, :3] + [1.3 * RH, 0.2, -0.1]                  # an overlapping pair (r < 2a branch)
    M = np.zeros((3 * n, 3 * n))
    for c in range(3 * n):
        v = np.zeros((n, 4))
        v[c // 3, c % 3] = 1.0
        out = np.zeros((n, 3))
        p.near_mdot(pos, v, out)
        M[:, c] = out.reshape(-1)
    assert np.abs(M - M.T).max() <= 1e-12
    # one block against the closed form + table-free evaluation
    d = pos[1, :3] - pos[0, :3]
    r = np.linalg.norm(d)
    fg = np.zeros(2)
    import ctypes as C
    from oracle.oracle import _p
    o64.lib.oracle_pse_rpy_near_fandg(C.c_double(r), C.c_double(RH), C.c_double(psi), C.c_double(float(p.rcut)), _p(fg))
    norm = 6 * math.pi * RH * VISC
    block = (fg[0] * np.eye(3) + (fg[1] - fg[0]) * np.outer(d, d) / r ** 2) / norm
    assert np.abs(M[0:3, 3:6] - block).max() <= 1e-6 * np.abs(block).max()     # table interpolation error only
    assert np.all(np.linalg.eigvalsh(M) > 0)                                    # positively split: M_near is SPD


def test_closed_form_limits(o64):
    """psi -> 0 turns the near field into the plain RPY tensor: F = 1 - 9r/32a ... (r < 2a), 3a/4r (1 + 2a^2/3r^2) (r > 2a)."""
    import ctypes as C
    from oracle.oracle import _p
    a, psi = 1.0, 1e-3
    fg = np.zeros(2)
    for r in (0.5, 1.5, 2.5, 6.0):
        o64.lib.oracle_pse_rpy_near_fandg(C.c_double(r), C.c_double(a), C.c_double(psi), C.c_double(1e9), _p(fg))
        if r > 2 * a:
            F = 3 * a / (4 * r) * (1 + 2 * a * a / (3 * r * r))
            G = 3 * a / (4 * r) * (2 - 4 * a * a / (3 * r * r)) / 1.0 * 0.5 * 2 / 2 * 2 / 2     # = 3a/(2r) (1 - 2a^2/(3r^2)) * (1/1)
            G = 3 * a / (2 * r) * (1 - 2 * a * a / (3 * r * r))
        else:
            F = 1 - 9 * r / (32 * a)
            G = 1 - 3 * r / (16 * a) * 1.0 * 1.0 - 0.0
            G = 1 - 3 * r / (16 * a)
        # the far field carries an O(psi) self term, so compare up to that
        assert abs(fg[0] - F) < 5e-3 and abs(fg[1] - G) < 5e-3, (r, fg, F, G)


def test_table_lookup_and_setup(o32):
    from oracle.pse import PSEOracle
    p = PSEOracle(o32, [32.0] * 3, 1.0, 1.0, 1e-3, 0.5)
    assert p.nPointsTable == 1 << 14                                   # rcut/(a tol) = 5257 -> clamped to 2^14
    assert abs(float(p.rcut) - math.sqrt(-math.log(1e-3)) / 0.5) < 1e-6
    assert list(p.cells) == [28, 28, 28] and p.support == 7           # kcut = 2.63 -> h = 2.39 -> 2L/h + 1 = 27 -> 28 = 4*7
    p2 = PSEOracle(o32, [64.0] * 3, 1.0, 1.0, 1e-5, 1.0)
    assert p2.nPointsTable == int(float(p2.rcut) / np.float32(1e-5) + 0.5)
    with pytest.raises(RuntimeError):
        PSEOracle(o32, [8.0] * 3, 1.0, 1.0, 1e-3, 0.5)                  # rcut 5.26 > L/2


def test_self_diffusion_reference_test(o64):
    """pse_test.cu:121-158 (SelfDiffusionIsCorrectUpToToleranceHydroDisp): <dx^2> = 2 T M0 within 1e-2, 1000 samples."""
    L = 32 * RH
    p = _pse(o64, L, 1e-4, 1.0, seed_near=11, seed_far=12)
    u = o64.saru_f_range(1234, -0.5, 0.5, 3 * 1000).reshape(1000, 3)
    m0 = p.getSelfMobility()
    dx2 = np.zeros(3)
    for j in range(1000):
        pos = np.zeros((1, 4))
        pos[0, :3] = u[j] * L
        dx = p.computeHydrodynamicDisplacements(pos, None, 1.0, 1.0, seed2_near=1000 + j, seed2_far=5000 + j)
        dx2 += dx[0] ** 2
    assert np.abs(dx2 / 1000 - 2.0 * m0).max() <= 1e-2
