"""ORACLE — TEST INFRASTRUCTURE ONLY.

BDHI::PSE restated on the CPU (Integrator/BDHI/BDHI_PSE.cuh:79-176, PSE/NearField.cuh, PSE/FarField.cuh) on top of
oracle/src/pse.c, the cell list / IBM oracles and numpy/scipy FFTs:
  NearField::Mdot                          NearField.cuh:239-250   (cell list at rcut * safety factor, RPYNearTransverser)
  NearField::computeStochasticDisplacements NearField.cuh:252-285   (Saru noise -> lanczos::Solver with the near matvec)
  FarField::computeHydrodynamicDisplacements FarField.cuh:569-589   (spread -> FFT -> greens function -> noise -> FFT -> gather)
  PSE::{computeMF, computeBdW, computeHydrodynamicDisplacements}   BDHI_PSE.cuh:92-155
  computeSelfMobility                      PSE/initialization.cu:31-47
Seeds: the reference draws `seed` (near, then far) from System::rng() at construction and a fresh `seed2` from the same
generator at every stochastic call; here they are explicit arguments.
"""
import ctypes as C
import math

import numpy as np

from .fcm import _fft, _kw
from .lanczos import LanczosOracle
from .oracle import _p


def next_fft_wise_size_3d(size):
    """utils/Grid.cuh:142-213 (same restatement as uammd_amd.bdhi.nextFFTWiseSize3D, kept separate: oracle is standalone)."""
    lim = {2: 64, 3: 64, 5: 5, 7: 4, 11: 3}
    out = []
    for s in size:
        c = max(int(s), 1)
        while True:
            m, ok = c, c % 2 == 0
            if ok:
                for p, mx in lim.items():
                    e = 0
                    while m % p == 0:
                        m //= p
                        e += 1
                    ok = ok and e <= mx
                ok = ok and m == 1
            if ok:
                break
            c += 1
        out.append(c)
    return out


def cut_off_sheared_safety_factor(g):      # NearField.cuh:24-27
    return 1 + 0.5 * g * g + 0.5 * math.sqrt(g * g * (g * g + 4.0))


class PSEOracle:
    def __init__(self, oracle, L, hydrodynamicRadius, viscosity, tolerance, psi, shearStrain=0.0, seed_near=1, seed_far=2):
        o = self.o = oracle
        self.real, cr = o.real, o.creal
        self.L = np.broadcast_to(np.asarray(L, dtype=self.real), (3,)).copy()
        self.rh, self.viscosity, self.tolerance, self.psi = hydrodynamicRadius, viscosity, tolerance, psi
        self.shear = shearStrain
        self.seed_near, self.seed_far = seed_near, seed_far
        lib = o.lib
        # ---- near field
        rc = cr(0)
        lib.oracle_pse_near_setup.restype = C.c_int
        self.nPointsTable = lib.oracle_pse_near_setup(cr(hydrodynamicRadius), cr(psi), cr(tolerance), C.byref(rc))
        self.rcut = self.real(rc.value)
        if 0.5 * self.L[0] < self.rcut:
            raise RuntimeError("[BDHI::PSE] Cut off is too large, try increasing psi")      # NearField.cuh:69-78
        self.table = np.zeros((self.nPointsTable, 2), self.real)
        lib.oracle_pse_near_table(cr(hydrodynamicRadius), cr(psi), cr(viscosity), cr(self.rcut), self.nPointsTable, _p(self.table))
        self.lanczos = LanczosOracle(self.real)
        # ---- far field
        raw = np.zeros(3, np.int32)
        lib.oracle_pse_far_raw_cells(_p(self.L), cr(psi), cr(tolerance), _p(raw))
        self.cells = np.asarray(next_fft_wise_size_3d(raw), np.int32)
        sup, eta, pref, tau = C.c_int(0), cr(0), cr(0), cr(0)
        lib.oracle_pse_far_kernel(_p(self.L), _p(self.cells), cr(psi), cr(tolerance), C.byref(sup), C.byref(eta), C.byref(pref),
                                  C.byref(tau))
        self.support, self.eta = int(sup.value), self.real(eta.value)
        self.kernel = o.ibm_kernel("gaussian", self.support, pref.value, tau.value, np.inf)
        self.kernel_prefactor, self.kernel_tau = pref.value, tau.value

    def getSelfMobility(self):
        return self.o.fcm_self_mobility(self.rh, self.viscosity, float(self.L[0]))

    # ---- near ----------------------------------------------------------------------------------------------------
    def _near_list(self, pos):
        o = self.o
        rc = self.real(self.rcut * self.real(cut_off_sheared_safety_factor(self.shear)))
        cd, gL, gper = o.celllist_create_grid(self.L, 1, rc)
        cl = o.celllist_build(o.r(pos), gL, gper, cd)
        assert cl["error"] == 0
        return cl

    def _near_dot(self, cl, v, vstride, out):
        cr = self.o.creal
        self.o.lib.oracle_pse_near_mdot(_p(cl["sortPos"]), _p(cl["index"]), len(cl["index"]), _p(cl["cellStart"]), _p(cl["cellEnd"]),
                                        C.c_uint(cl["validCell"]), _p(cl["L"]), _p(cl["periodic"]), _p(cl["cellDim"]), _p(self.L),
                                        cr(self.shear), cr(self.rcut), _p(self.table), self.nPointsTable, _p(v), int(vstride), _p(out))

    def near_mdot(self, pos, force4, MF):
        """NearField::Mdot: MF += M_near F (force real4[N])."""
        self._near_dot(self._near_list(pos), self.o.r(force4), 4, MF)

    def near_noise(self, n, variance, seed2):
        out = np.zeros((n, 3), self.real)
        self.o.lib.oracle_pse_near_noise(n, self.o.creal(variance), C.c_uint(self.seed_near), C.c_uint(seed2), _p(out))
        return out

    def near_stochastic(self, pos, temperature, prefactor, seed2):
        """NearField::computeStochasticDisplacements -> BdW real3[N] (overwrites, as the Lanczos result does)."""
        n = len(pos)
        if temperature == 0:
            return None
        cl = self._near_list(pos)
        noise = self.near_noise(n, self.real(prefactor * math.sqrt(2 * temperature)), seed2)

        def dot(v):
            out = np.zeros(3 * n, self.real)
            self._near_dot(cl, np.ascontiguousarray(v, dtype=self.real), 3, out)
            return out
        return self.lanczos.run(dot, noise.reshape(-1), self.tolerance).reshape(n, 3)

    # ---- far -----------------------------------------------------------------------------------------------------
    def far(self, pos, force4, MF, temperature, prefactor, seed2, grids=None):
        """FarField::computeHydrodynamicDisplacements: MF += M_far F + noise (IBM gather adds)."""
        o, cr = self.o, self.o.creal
        nx, ny, nz = (int(c) for c in self.cells)
        pos = o.r(pos)
        if force4 is not None:
            gr = o.ibm_spread(pos, o.r(np.asarray(force4)[:, :3]), self.L, 1, self.cells, self.kernel)     # [nz][ny][nx][3]
            gk = np.ascontiguousarray(_fft.rfftn(gr, axes=(0, 1, 2), **_kw).astype(np.complex64 if self.real == np.float32
                                                                                   else np.complex128))
            o.lib.oracle_pse_force_fourier_to_vel(_p(gk), cr(self.shear), cr(self.rh), cr(self.viscosity), cr(self.psi),
                                                  cr(self.eta), _p(self.L), _p(self.cells))
        else:
            gk = np.zeros((nz, ny, nx // 2 + 1, 3), np.complex64 if self.real == np.float32 else np.complex128)
        if temperature > 0:
            dV = self.real(np.prod((self.L / self.cells.astype(self.real)).astype(self.real)))
            npf = self.real(prefactor * math.sqrt(2 * temperature / float(dV)))
            o.lib.oracle_pse_fourier_brownian_noise(_p(gk), _p(self.L), _p(self.cells), cr(npf), cr(self.shear), cr(self.rh),
                                                    cr(self.viscosity), cr(self.psi), cr(self.eta), C.c_uint(self.seed_far),
                                                    C.c_uint(seed2))
        gv = np.ascontiguousarray((_fft.irfftn(gk, s=(nz, ny, nx), axes=(0, 1, 2), **_kw) * (nx * ny * nz)).astype(self.real))
        o.ibm_gather(pos, gv, self.L, 1, self.cells, self.kernel, out=MF)
        if grids is not None:
            grids.update(fourier=gk, velocity=gv)
        return MF

    # ---- BDHI::PSE ----------------------------------------------------------------------------------------------------
    def computeHydrodynamicDisplacements(self, pos, force4, temperature, prefactor, seed2_near=0, seed2_far=0):
        """BDHI_PSE.cuh:135-155, statement for statement (with forces AND T > 0 the Lanczos result overwrites the near
        deterministic part: reproduced, not fixed)."""
        n = len(pos)
        MF = np.zeros((n, 3), self.real)
        if force4 is not None:
            self.near_mdot(pos, force4, MF)
        b = self.near_stochastic(pos, temperature, prefactor, seed2_near)
        if b is not None:
            MF[:] = b
        self.far(pos, force4, MF, temperature, prefactor, seed2_far)
        return MF


def rpy_nbody_mdot(oracle, pos4, v, viscosity, rh=-1.0, radius=None):
    """Lanczos_ns::NbodyMatrixFreeMobilityDot through NBody::transverse (BDHI_Lanczos.cu:56-118): Mv real3[N]."""
    pos4 = oracle.r(pos4)
    v = oracle.r(v)
    n = len(pos4)
    out = np.zeros((n, 3), oracle.real)
    rad = None if radius is None else oracle.r(radius)
    oracle.lib.oracle_rpy_nbody_mdot(_p(pos4), _p(v), int(v.shape[1]), _p(rad), oracle.creal(rh), oracle.creal(viscosity), n, _p(out))
    return out


def rpy_dense(oracle, pos, radius, hydrodynamicRadius, viscosity):
    """BDHI::Cholesky's mobility matrix (BDHI_Cholesky.cu:34-80) as a full symmetric [3N, 3N] array."""
    o = oracle
    pos = o.r(pos)
    n = len(pos)
    M = np.zeros((3 * n, 3 * n), o.real)
    rad = None if radius is None else o.r(radius)
    o.lib.oracle_rpy_dense(_p(pos), _p(rad) if rad is not None else None, o.creal(hydrodynamicRadius), o.creal(viscosity), n, _p(M))
    return M   # symmetric: row/column major agree


class CholeskyOracle:
    """BDHI::Cholesky (BDHI_Cholesky.cu:158-262): MF = M F, BdW = U^T dW with M = U^T U (potrf upper, trmv transposed)."""

    def __init__(self, oracle, hydrodynamicRadius, viscosity):
        self.o, self.rh, self.viscosity = oracle, hydrodynamicRadius, viscosity

    def matrix(self, pos, radius=None):
        return rpy_dense(self.o, pos, None if self.rh > 0 else radius, self.rh, self.viscosity)

    def computeMF(self, pos, force, radius=None):
        f3 = self.o.r(np.asarray(force)[:, :3]).reshape(-1)
        return (self.matrix(pos, radius) @ f3).reshape(-1, 3)

    def computeBdW(self, pos, noise, radius=None):
        L = np.linalg.cholesky(self.matrix(pos, radius).astype(np.float64))   # M = L L^T, L = U^T
        return (L @ np.asarray(noise, np.float64).reshape(-1)).reshape(-1, 3)
