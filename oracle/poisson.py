"""ORACLE — TEST INFRASTRUCTURE ONLY.

Poisson (Interactor/SpectralEwaldPoisson.cu(h)) restated on the CPU on top of oracle/src/poisson.c, the cell list / IBM
oracles and numpy/scipy FFTs:
  Poisson::Poisson                       SpectralEwaldPoisson.cu:71-160   (grid, window, near cut-off, tables)
  Poisson::farField                      :332-360   (spread q -> R2C -> convolve -> 4 x C2R -> gather real4 -> force += qE, energy += q phi)
  Poisson::nearField{Force,Energy,FieldPotential}   :362-408   (CellList at the near cut-off, tabulated Green's functions)
  Poisson::sum / computeFieldPotentialAtParticles   SpectralEwaldPoisson.cuh:110-136
Pinned by the reference's own tests (tests/test_oracle_poisson.py): test/Potentials/Poisson/TriplyPeriodic/test_poisson.cu
(SingleSimulationTest, a thinned InfiniteBoxSizeTest) and test_tp_quadrupole.cu, both in double precision.
"""
import ctypes as C

import numpy as np

from .fcm import _fft, _kw
from .oracle import _p
from .pse import next_fft_wise_size_3d


class PoissonOracle:
    def __init__(self, oracle, L, epsilon, gw, tolerance=1e-5, split=-1.0, upsampling=-1.0):
        o = self.o = oracle
        self.real, cr = o.real, o.creal
        self.cplx = np.complex64 if self.real == np.float32 else np.complex128
        self.L = np.broadcast_to(np.asarray(L, dtype=self.real), (3,)).copy()
        self.epsilon, self.gw, self.tolerance, self.split = epsilon, gw, tolerance, split
        lib = o.lib
        raw = np.zeros(3, np.int32)
        lib.oracle_poisson_raw_cells(_p(self.L), cr(gw), cr(split), cr(tolerance), cr(upsampling), _p(raw))
        self.cells = np.asarray(next_fft_wise_size_3d(raw), np.int32)
        pref, tau, sup, rc, nt = cr(0), cr(0), C.c_int(0), cr(0), C.c_int(0)
        lib.oracle_poisson_setup.restype = C.c_int
        e = lib.oracle_poisson_setup(_p(self.L), _p(self.cells), cr(gw), cr(split), cr(epsilon), cr(tolerance), C.byref(pref),
                                     C.byref(tau), C.byref(sup), C.byref(rc), C.byref(nt))
        if e == -1:
            raise ValueError("[Poisson] Kernel support is too large")          # :95-102
        if e == -2:
            raise ValueError("[Poisson] Near field cut off is too large")      # :111-116
        self.support, self.nearFieldCutOff, self.ntable = int(sup.value), self.real(rc.value), int(nt.value)
        self.kernel = o.ibm_kernel("gaussian", self.support, pref.value, tau.value, np.inf)
        self.kernel_prefactor, self.kernel_tau = pref.value, tau.value
        self.nxpad = 2 * (int(self.cells[0]) // 2 + 1)
        if split > 0:
            self.tableField = np.zeros(self.ntable, self.real)
            self.tablePotential = np.zeros(self.ntable, self.real)
            lib.oracle_poisson_tables(cr(gw), cr(split), cr(epsilon), cr(self.nearFieldCutOff), self.ntable, _p(self.tableField),
                                      _p(self.tablePotential))

    # ---- far field -----------------------------------------------------------------------------------------------
    def far_grid(self, pos, charge, grids=None):
        """The real4 grid (Ex, Ey, Ez, phi) [nz][ny][2(nx/2+1)][4] of :332-349."""
        o = self.o
        nx, ny, nz = (int(c) for c in self.cells)
        gq = o.ibm_spread(o.r(pos), o.r(charge), self.L, 1, self.cells, self.kernel, nx_stride=self.nxpad)[..., 0]
        gk = np.ascontiguousarray(_fft.rfftn(gq[:, :, :nx], axes=(0, 1, 2), **_kw).astype(self.cplx))
        out = np.zeros(gk.shape + (4,), self.cplx)
        o.lib.oracle_poisson_convolve(_p(gk), _p(out), _p(self.L), _p(self.cells), o.creal(self.epsilon))
        g4 = np.zeros((nz, ny, self.nxpad, 4), self.real)
        for c in range(4):  # cuFFT C2R is unnormalised
            g4[:, :, :nx, c] = (_fft.irfftn(out[..., c], s=(nz, ny, nx), axes=(0, 1, 2), **_kw) * (nx * ny * nz)).astype(self.real)
        if grids is not None:
            grids.update(charges=gq, fourier=gk, convolved=out, fieldPotential=g4)
        return g4

    def far(self, pos, charge, force4=None, energy=None, fieldPotential=None):
        o = self.o
        g4 = self.far_grid(pos, charge)
        fp = o.ibm_gather(o.r(pos), g4, self.L, 1, self.cells, self.kernel, nx_stride=self.nxpad)
        q = o.r(charge)
        o.lib.oracle_poisson_apply_charges(_p(fp), _p(q), len(q), _p(force4) if force4 is not None else None,
                                           _p(energy) if energy is not None else None)
        if fieldPotential is not None:   # ibm.gather(pos, fieldPotentialAtParticles, ...) adds
            fieldPotential += fp

    # ---- near field ----------------------------------------------------------------------------------------------
    def _list(self, pos):
        o = self.o
        cd, gL, gper = o.celllist_create_grid(self.L, 1, self.nearFieldCutOff)
        cl = o.celllist_build(o.r(pos), gL, gper, cd)
        assert cl["error"] == 0
        return cl

    def near(self, pos, charge, mode, out):
        if not self.split > 0:
            return
        o, cr = self.o, self.o.creal
        cl = self._list(pos)
        q = o.r(charge)
        o.lib.oracle_poisson_near(_p(cl["sortPos"]), _p(cl["index"]), len(q), _p(cl["cellStart"]), _p(cl["cellEnd"]),
                                  C.c_uint(cl["validCell"]), _p(cl["L"]), _p(cl["periodic"]), _p(cl["cellDim"]), _p(self.L), _p(q),
                                  _p(self.tableField), _p(self.tablePotential), self.ntable, cr(self.nearFieldCutOff), int(mode),
                                  _p(out))

    # ---- Interactor interface -----------------------------------------------------------------------------------
    def sum(self, pos, charge, force4, energy, force=True, energy_flag=False):
        """Poisson::sum: the far field ALWAYS adds to both force and energy (interpolateFields has no flags)."""
        pos4 = np.zeros((len(pos), 4), self.real)
        pos4[:, :3] = np.asarray(pos)[:, :3]
        self.far(pos4, charge, force4=force4, energy=energy)
        if force:
            self.near(pos4, charge, 0, force4)
        if energy_flag:
            self.near(pos4, charge, 1, energy)

    def computeFieldPotentialAtParticles(self, pos, charge):
        """-> real4[N] (Ex, Ey, Ez, phi).  As in the reference, the far-field call also needs force/energy arrays."""
        n = len(pos)
        pos4 = np.zeros((n, 4), self.real)
        pos4[:, :3] = np.asarray(pos)[:, :3]
        fp = np.zeros((n, 4), self.real)
        self.far(pos4, charge, fieldPotential=fp)
        self.near(pos4, charge, 2, fp)
        return fp
