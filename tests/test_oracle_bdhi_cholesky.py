"""Oracle pin for the open-boundary dense RPY mobility (BDHI::Cholesky, and the same tensor BDHI::Lanczos applies matrix
free): the reference's test/BDHI/Lanczos_Cholesky (test.bash + process.cpp) — N particles of radius a1 and one of radius
a2, zero temperature, particle 0 pulled with F = (1, 0, 0); f(r) and g(r) recovered from every other particle's velocity
must equal RPY_differentSizes within 1e-7 (double precision)."""
import numpy as np

from oracle.pse import CholeskyOracle, rpy_dense


def rpy_theory(r, ai, aj, viscosity):   # process.cpp:79-103 (long double there)
    M0 = 1.0 / (6 * np.pi * viscosity)
    asum, asub = ai + aj, abs(ai - aj)
    if r > asum:
        pref = M0 * 3.0 * 0.25 / r
        denom = (ai * ai + aj * aj) / (3.0 * r * r)
        return pref * (1.0 + denom), pref * (1.0 - 3.0 * denom) / (r * r)
    if r > asub:
        pref = M0 / (ai * aj * 32.0 * r ** 3)
        num = asub * asub + 3.0 * r * r
        c1 = pref * (16.0 * r ** 3 * asum - num * num)
        num = asub * asub - r * r
        return c1, pref * (3.0 * num * num) / (r * r)
    return M0 / max(ai, aj), 0.0


def test_cross_mobility_of_two_sizes(o64):
    """test.bash:8-21: boxSize 4, radius_min 0.38173, radius_max 1.89538, viscosity 1.2131 (N reduced from 5000)."""
    rng = np.random.default_rng(0)
    n, visc, a1, a2 = 400, 1.2131, 0.38173, 1.89538
    pos = np.zeros((n, 4))
    pos[:, :3] = rng.uniform(-2, 2, (n, 3))
    radius = np.full(n, a1)
    radius[0] = a2
    force = np.zeros((n, 4))
    force[0, 0] = 1.0
    ch = CholeskyOracle(o64, -1.0, visc)
    v = ch.computeMF(pos, force, radius)
    worst_f = worst_g = 0.0
    for i in range(1, n):
        rij = pos[i, :3] - pos[0, :3]
        r = np.linalg.norm(rij)
        f_th, g_th = rpy_theory(r, a2, a1, visc)
        # v_i = f e_x + g (rij . e_x) rij: the y component isolates g, then x gives f  (process.cpp:150-176)
        if abs(rij[0] * rij[1]) < 1e-3:
            continue
        g = v[i, 1] / (rij[0] * rij[1])
        f = v[i, 0] - g * rij[0] * rij[0]
        worst_f = max(worst_f, abs((f - f_th) / f_th))
        if g_th != 0:
            worst_g = max(worst_g, abs((g - g_th) / g_th))
    assert worst_f < 1e-7 and worst_g < 1e-7, (worst_f, worst_g)
    # the pulled particle moves with its own Stokes mobility
    assert abs(v[0, 0] * 6 * np.pi * visc * a2 - 1) < 1e-12 and abs(v[0, 1]) < 1e-15


def test_matrix_is_spd_and_factor_reproduces_it(o64, o32):
    rng = np.random.default_rng(1)
    n = 60
    pos = np.zeros((n, 4))
    pos[:, :3] = rng.uniform(-4, 4, (n, 3))
    for o, tol in ((o64, 1e-12), (o32, 2e-5)):
        M = rpy_dense(o, pos, None, 1.0, 1.0)
        assert np.array_equal(M, M.T) and np.linalg.eigvalsh(M.astype(np.float64)).min() > 0
        ch = CholeskyOracle(o, 1.0, 1.0)
        B = np.stack([ch.computeBdW(pos, e).reshape(-1) for e in np.eye(3 * n)], axis=1)   # columns B e_k
        assert np.abs(B @ B.T - M).max() <= tol * np.abs(M).max()
        # same tensor as the matrix-free product of BDHI::Lanczos
        from oracle.pse import rpy_nbody_mdot
        v = rng.normal(0, 1, (n, 3)).astype(o.real)
        assert np.abs(rpy_nbody_mdot(o, pos, v, 1.0, radius=np.full(n, 1.0)).reshape(-1) - M @ v.reshape(-1)).max() <= 50 * tol
