"""ORACLE — TEST INFRASTRUCTURE ONLY.

BDHI::FIB (Integrator/BDHI/FIB/FIB.cu(h)) restated on the CPU on top of oracle/src/fib.c and numpy/scipy FFTs, as the reference
actually runs it: both Scheme values take forwardMidpoint (FIB.cu:1072-1079) and the RFD thermal drift kernel is disabled
(:400).  The fluid noise (cuRAND in the reference, unpinned) is an explicit input: `noise_fn(ncells) -> real[6, ncells]`.
"""
import math

import numpy as np

from .fcm import _fft, _kw
from .oracle import _p
from .pse import next_fft_wise_size_3d


class FIBOracle:
    def __init__(self, oracle, L, temperature, viscosity, dt, hydrodynamicRadius=-1.0, cells=None, noise_fn=None):
        o = self.o = oracle
        self.real, cr = o.real, o.creal
        self.cplx = np.complex64 if self.real == np.float32 else np.complex128
        self.L = np.broadcast_to(np.asarray(L, dtype=self.real), (3,)).copy()
        self.temperature, self.viscosity, self.dt = temperature, viscosity, dt
        if hydrodynamicRadius > 0 and cells is not None:
            raise RuntimeError("[BDHI::FIB] Please provide hydrodynamic radius OR cell dimensions, not both.")   # :95-97
        if cells is None:
            if hydrodynamicRadius < 0:
                raise RuntimeError("[BHDI::FIB] I need either the hydrodynamic radius or the number of cells!")    # :101-103
            raw = np.zeros(3, np.int32)
            o.lib.oracle_fib_raw_cells(_p(self.L), cr(hydrodynamicRadius), _p(raw))
            cells = next_fft_wise_size_3d(raw)
        cells = [int(c) for c in cells]
        cells[0], cells[1] = max(cells[0], 3), max(cells[1], 3)                 # :113-118
        if cells[2] == 2:
            cells[2] = 3
        self.cells = np.asarray(cells, np.int32)
        cs = self.L / self.cells.astype(self.real)
        self.h = self.real(min(cs))                                             # kernel h (:119-120)
        self.hydrodynamicRadius = self.real(cs[0] * self.real(0.91))            # fixHydrodynamicRadius(h, cellSize.x) (:121)
        self.noise_fn = noise_fn
        self.step = 0

    def getSelfMobility(self):   # FIB.cuh:152-163
        rh, L = float(self.hydrodynamicRadius), float(self.L[0])
        a = rh / L
        c, b = 2.83729747948061947666591710460773907, 0.19457
        a6pref = 16.0 * math.pi ** 2 / 45.0 + 630.0 * b * b
        return 1.0 / (6.0 * math.pi * self.viscosity * rh) * (1.0 - c * a + (4.0 / 3.0) * math.pi * a ** 3 - a6pref * a ** 6)

    def fluid_velocity(self, pos4, force4, noise=None):
        """gridVels real3[nz][ny][nx] after applyStokesSolutionOperator (g = noise + S F)."""
        o, cr = self.o, self.o.creal
        nx, ny, nz = (int(c) for c in self.cells)
        g = np.zeros((nz, ny, nx, 3), self.real)
        if self.temperature != 0:
            dV = float(np.prod(self.L / self.cells))
            pref = self.real(math.sqrt(2 * self.viscosity * self.temperature / (self.dt * dV)))     # :973-975
            noise = np.ascontiguousarray(noise, dtype=self.real)
            o.lib.oracle_fib_random_advection(_p(g), _p(self.L), _p(self.cells), cr(pref), _p(noise))
        if force4 is not None:
            o.lib.oracle_fib_spread(_p(o.r(pos4)), _p(o.r(force4)), len(pos4), _p(self.L), _p(self.cells), cr(self.h), _p(g))
        gk = np.ascontiguousarray(_fft.rfftn(g, axes=(0, 1, 2), **_kw).astype(self.cplx))
        o.lib.oracle_fib_solve_stokes(_p(gk), cr(self.viscosity), _p(self.L), _p(self.cells))
        return np.ascontiguousarray((_fft.irfftn(gk, s=(nz, ny, nx), axes=(0, 1, 2), **_kw) * (nx * ny * nz)).astype(self.real))

    def forwardTime(self, pos4, force4=None, noise=None):
        """forwardMidpoint (:965-1000) in place on pos4 (real4[N]); forces are evaluated once, at q^n."""
        o, cr = self.o, self.o.creal
        self.step += 1
        if self.temperature != 0 and noise is None:
            noise = self.noise_fn(int(np.prod(self.cells)))
        v = self.fluid_velocity(pos4, force4, noise)
        old = np.zeros_like(pos4)
        for mode in (0, 1):
            o.lib.oracle_fib_midpoint_step(mode, _p(pos4), _p(old), _p(v), len(pos4), _p(self.L), _p(self.cells), cr(self.h), cr(self.dt))
        return v
