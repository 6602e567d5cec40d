"""ORACLE — TEST INFRASTRUCTURE ONLY.

lanczos::Solver restated with numpy (misc/LanczosAlgorithm/LanczosAlgorithm.cu):
  KrylovSubspace::setFirstBasisVector   :102-112
  KrylovSubspace::nextIteration         :114-157   (incl. the breakdown guard hsup < 1e-3*hdiag/||z|| -> 0, w = e1)
  computeSquareRoot / diagonalizeSubSpace :43-82   (LAPACKE_steqr + cblas_gemv: third-party, absent from
                                                    /root/reference; a symmetric tridiagonal eigenproblem has one
                                                    H^(1/2) e1, taken here from scipy.linalg.eigh_tridiagonal)
  computeCurrentResultEstimation        :162-172   y = ||z|| V_m H^(1/2) e1
  Solver::run                           :202-228   check schedule + hard limit 200 + "Could not converge"
  Solver::computeError                  :230-251   ||Bz_i - Bz_(i-1)|| / ||Bz_(i-1)||
  Solver::registerRequiredStepsForConverge :253-262 (adaptive check_convergence_steps, starts at 3)
Vectors are numpy arrays of the oracle precision; the matrix is a Python callable v -> M v.
"""
import numpy as np

try:
    from scipy.linalg import eigh_tridiagonal
except Exception:  # pragma: no cover
    eigh_tridiagonal = None


class LanczosOracle:
    def __init__(self, real=np.float64):
        self.real = real
        self.check_convergence_steps = 3
        self.iterationHardLimit = 200
        self.lastRunRequiredSteps = 0

    def setIterationHardLimit(self, n):
        self.iterationHardLimit = n

    def getLastRunRequiredSteps(self):
        return self.lastRunRequiredSteps

    # -- KrylovSubspace -------------------------------------------------------------------------------
    def _sqrt_e1(self, hdiag, hsup, m):
        d = np.asarray(hdiag[:m], dtype=np.float64)
        e = np.asarray(hsup[:m - 1], dtype=np.float64)
        if m == 1:
            lam, P = d.copy(), np.ones((1, 1))
        elif eigh_tridiagonal is not None:
            lam, P = eigh_tridiagonal(d, e)
        else:
            T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
            lam, P = np.linalg.eigh(T)
        # htemp[j] = sqrt(lambda_j) * P[0][j];  H^(1/2) e1 = P * htemp   (NaN for a negative eigenvalue, as the reference)
        with np.errstate(invalid="ignore"):
            t = np.sqrt(lam) * P[0, :]
        return (P @ t).astype(self.real)

    def run(self, dot, z, tolerance, return_all=False):
        real = self.real
        z = np.asarray(z, dtype=real)
        N = len(z)
        oldBz = np.zeros(N, real)
        V = [None]
        hdiag, hsup = [], []
        normz = real(np.linalg.norm(z))
        V[0] = (z * real(1.0 / normz)).astype(real)
        checkConvergenceSteps = min(self.check_convergence_steps, self.iterationHardLimit - 2)
        Bz = np.zeros(N, real)
        for i in range(self.iterationHardLimit):
            # nextIteration
            w = np.asarray(dot(V[i]), dtype=real).copy()
            if i > 0:
                w = (w - hsup[i - 1] * V[i - 1]).astype(real)
            hdiag.append(real(np.dot(w, V[i])))
            w = (w - hdiag[i] * V[i]).astype(real)
            hs = real(np.linalg.norm(w))
            tol = real(1e-3) * hdiag[i] / normz
            if hs < tol:
                hs = real(0.0)
            hsup.append(hs)
            if hs > 0:
                w = (w * real(1.0 / hs)).astype(real)
            else:
                w = np.zeros(N, real)
                w[0] = 1
            V.append(w)
            if i >= checkConvergenceSteps:
                m = i + 1
                y = self._sqrt_e1(hdiag, hsup, m)
                Bz = (normz * (np.stack(V[:m], axis=1) @ y)).astype(real)
                if i > 0:
                    with np.errstate(divide="ignore", invalid="ignore"):
                        err = abs(real(np.linalg.norm(Bz - oldBz)) / real(np.linalg.norm(oldBz)))
                    if np.isnan(err):
                        raise RuntimeError(f"[Lanczos] Unknown error (found NaN in result guess) at iteration {i}")
                    if err <= tolerance:
                        self._register(i)
                        return (Bz, i) if return_all else Bz
                oldBz = Bz.copy()
        raise RuntimeError("[Lanczos] Could not converge")

    def _register(self, steps_needed):
        self.lastRunRequiredSteps = steps_needed
        if steps_needed - 2 > self.check_convergence_steps:
            self.check_convergence_steps += 1
        else:
            self.check_convergence_steps = max(1, self.check_convergence_steps - 2)


def std_mt19937_uniform_real(seed, n, a, b, dtype=np.float64):
    """std::mt19937{seed} + std::uniform_real_distribution<real>{a,b} as libstdc++ implements them:
    generate_canonical<double,53> = (r0 + r1*2^32) / 2^64 (two draws), <float,24> = r0 / 2^32 (one draw;
    a result that rounds to 1 is replaced by nextafter(1,0))."""
    rs = np.random.RandomState(seed)  # init_genrand(seed): the same state as std::mt19937(seed)
    if dtype == np.float64:
        raw = rs._bit_generator.random_raw(2 * n).astype(np.float64)
        c = (raw[0::2] + raw[1::2] * 4294967296.0) / 18446744073709551616.0
        c = np.where(c >= 1.0, np.nextafter(1.0, 0.0), c)
        return a + (b - a) * c
    raw = rs._bit_generator.random_raw(n).astype(np.float32)
    c = (raw / np.float32(4294967296.0)).astype(np.float32)
    c = np.where(c >= 1.0, np.nextafter(np.float32(1.0), np.float32(0.0)), c).astype(np.float32)
    return (np.float32(a) + np.float32(b - a) * c).astype(np.float32)
