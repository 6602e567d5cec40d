"""GPU parity: lanczos::Solver through the C ABI vs the reference's known answers (test/misc/lanczos/test_lanczos.cu)
and vs the oracle.  The library computes in float (UAMMD's default real); the reference tests are compiled in double
with tolerance 1e-7, so here the requested tolerance is 1e-6 and results are held to 1e-5 relative."""
import numpy as np
import pytest
import torch

from oracle.lanczos import LanczosOracle, std_mt19937_uniform_real

pytestmark = pytest.mark.gpu


def _run(hip, solver, matvec, v, tol):
    dv = torch.from_numpy(np.asarray(v, np.float32)).cuda()
    Bv = torch.zeros_like(dv)
    it = solver.run(matvec, Bv, dv, tol)
    torch.cuda.synchronize()
    return Bv.cpu().numpy(), it


def test_identity_and_two_identity(hip):
    for scale, expect in ((1.0, 1.0), (2.0, np.sqrt(2.0))):
        solver = hip.BDHI.LanczosSolver()
        for size in list(range(1, 70)) + [127]:
            Bv, it = _run(hip, solver, lambda x, y: y.copy_(scale * x), np.ones(size), 1e-6)
            assert np.abs(Bv - expect).max() <= 1e-5, size
            if scale == 2.0:
                assert solver.getLastRunRequiredSteps() <= 5 and it <= 5, size   # test_lanczos.cu:78-93


def test_diagonal_random(hip):
    m = std_mt19937_uniform_real(29374238, 128, 1.0, 2.0)
    solver = hip.BDHI.LanczosSolver()
    for size in (1, 2, 3, 17, 64, 100, 127):
        d = torch.from_numpy(m[:size].astype(np.float32)).cuda()
        v = std_mt19937_uniform_real(1234567, size, -10.0, 10.0)
        Bv, _ = _run(hip, solver, lambda x, y: torch.mul(d, x, out=y), v, 1e-6)
        theory = np.sqrt(m[:size]) * v
        assert np.abs((Bv - theory) / theory).max() <= 1e-5, size


@pytest.mark.parametrize("size", [1, 5, 64, 200, 511])
def test_dense_spd_reference_test(hip, size):
    """test_lanczos.cu:236-269: operator M^2, answer M v."""
    vecm = std_mt19937_uniform_real(29374238, size * size, 0.0, 1.0)
    A = vecm.reshape(size, size)
    M = 0.5 * (A + A.T) + 5 * size * np.eye(size)
    M2 = torch.from_numpy((M @ M.T).astype(np.float32)).cuda()
    v = std_mt19937_uniform_real(1234567, size, -10.0, 10.0)
    solver = hip.BDHI.LanczosSolver()
    Bv, it = _run(hip, solver, lambda x, y: torch.mv(M2, x, out=y), v, 1e-6)
    theory = M @ v
    assert np.abs((Bv - theory) / theory).max() <= 2e-5
    # same iteration count and result as the float oracle
    o = LanczosOracle(np.float32)
    M2h = (M @ M.T).astype(np.float32)
    Bo, ito = o.run(lambda x: M2h @ x, v.astype(np.float32), 1e-6, return_all=True)
    assert abs(it - ito) <= 1 and np.abs(Bv - Bo).max() <= 2e-5 * np.abs(Bo).max()


def test_errors(hip):
    solver = hip.BDHI.LanczosSolver()
    d = torch.arange(1, 9, dtype=torch.float32, device="cuda")
    with pytest.raises(hip.UammdHipError, match="NaN"):
        _run(hip, solver, lambda x, y: torch.mul(-d, x, out=y), np.ones(8), 1e-6)
    solver = hip.BDHI.LanczosSolver()
    solver.setIterationHardLimit(3)
    big = torch.linspace(1, 1e6, 400, device="cuda")
    with pytest.raises(hip.UammdHipError, match="Could not converge"):
        _run(hip, solver, lambda x, y: torch.mul(big, x, out=y), np.ones(400), 1e-12)


def test_large_vector(hip):
    """3N = 3e5 (the size of the PSE near-field problem at C4): sqrt of a diagonal SPD operator."""
    n = 300000
    rng = np.random.default_rng(3)
    dm = rng.uniform(0.5, 2.0, n).astype(np.float32)
    d = torch.from_numpy(dm).cuda()
    v = rng.normal(0, 1, n).astype(np.float32)
    solver = hip.BDHI.LanczosSolver()
    Bv, it = _run(hip, solver, lambda x, y: torch.mul(d, x, out=y), v, 1e-5)
    assert np.abs(Bv - np.sqrt(dm) * v).max() <= 1e-4 * np.abs(v).max() and it < 30


@pytest.mark.parametrize("size,spread", [(300, 50.0), (2000, 400.0), (64, 3.0)])
def test_deferred_checks_stop_where_the_reference_does(hip, size, spread):
    """Deferred convergence checks (lanczos_run: the checks before the iteration the previous run stopped at are evaluated together, one
    host round trip and one pass over the Krylov basis) against a check after every iteration: a sequence of runs on matrices whose
    required iteration count moves up and down — the stopping iteration, hence lastRunRequiredSteps and the adaptive first-check step,
    and the result must be the same run by run (the same estimates from the same H and V: bit for bit)."""
    rng = np.random.default_rng(size)
    out = {}
    for defer in (1, 0):
        solver = hip.BDHI.LanczosSolver()
        solver.setOption("defer_checks", defer)
        res = []
        r2 = np.random.default_rng(size + 1)
        for run in range(12):
            # condition number varies from run to run: easy (few iterations), hard (many), easy again
            cond = [2.0, spread, spread, 5.0, 5.0, spread, 1.5, 1.5, spread, spread, 3.0, spread][run]
            d = torch.from_numpy(np.linspace(1.0, cond, size).astype(np.float32)).cuda()
            v = r2.normal(0, 1, size)
            Bv, it = _run(hip, solver, lambda x, y: torch.mul(d, x, out=y), v, 1e-5)
            res.append((Bv, it, solver.getLastRunRequiredSteps()))
            theory = np.sqrt(np.linspace(1.0, cond, size)) * v
            assert np.linalg.norm(Bv - theory) <= 2e-4 * np.linalg.norm(theory)
        out[defer] = res
    for a, b in zip(out[1], out[0]):
        assert a[1] == b[1] and a[2] == b[2]
        assert np.array_equal(a[0], b[0])
