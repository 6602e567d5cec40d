"""ORACLE — TEST INFRASTRUCTURE ONLY.

Hydro::ICM (Integrator/Hydro/ICM.cu(h)), zero excess mass, restated on the CPU on top of oracle/src/icm.c, oracle/src/fib.c (the
staggered spreading / interpolation are FIB's) and numpy/scipy FFTs.  forwardTime (ICM.cu:1191-1224):
  predictor q^{n+1/2} = q^n + dt/2 J(q^n) v^n -> unperturbed fluid update (diffusion, noise, Adams-Bashforth advection) ->
  + (dt/rho) S(q^{n+1/2}) F(q^{n+1/2}) [+ RFD thermal drift] -> (I - dt eta/(2 rho) L)^-1 P in Fourier space ->
  corrector q^{n+1} = q^n + dt J(q^{n+1/2}) v^{n+1}.
The reference has no test for this module (test/Hydro only covers ICM_Compressible): the pins in tests/test_oracle_icm.py
are physical (Stokes-limit mobility = getSelfMobility(), momentum conservation, viscous decay rate, equipartition).
"""
import math

import numpy as np

from .fcm import _fft, _kw
from .oracle import _p
from .pse import next_fft_wise_size_3d


class ICMOracle:
    def __init__(self, oracle, L, temperature, viscosity, density, dt, hydrodynamicRadius=-1.0, cells=None, sumThermalDrift=False,
                 removeTotalMomentum=True, seed=1234, noise_fn=None, initial_velocity=None):
        o = self.o = oracle
        self.real, cr = o.real, o.creal
        self.cplx = np.complex64 if self.real == np.float32 else np.complex128
        self.L = np.broadcast_to(np.asarray(L, dtype=self.real), (3,)).copy()
        self.temperature, self.viscosity, self.density, self.dt = temperature, viscosity, density, dt
        self.sumThermalDrift, self.removeTotalMomentum, self.seed, self.noise_fn = sumThermalDrift, removeTotalMomentum, seed, noise_fn
        if density < 0 or viscosity < 0:
            raise RuntimeError("[Hydro::ICM] Please provide fluid density / viscosity")                      # :833-836
        if hydrodynamicRadius > 0 and cells is not None:
            raise RuntimeError("[Hydro::ICM] Please provide hydrodynamic radius OR cell dimensions, not both.")  # :837-839
        if cells is None:
            if hydrodynamicRadius < 0:
                raise RuntimeError("[BHDI::ICM] I need either the hydrodynamic radius or the number of cells!")   # :869-872
            raw = np.zeros(3, np.int32)
            o.lib.oracle_fib_raw_cells(_p(self.L), cr(hydrodynamicRadius), _p(raw))     # hgrid = rh / 0.91, as FIB
            cells = next_fft_wise_size_3d(raw)
        cells = [int(c) for c in cells]
        cells[0], cells[1] = max(cells[0], 3), max(cells[1], 3)
        if cells[2] == 2:
            cells[2] = 3
        self.cells = np.asarray(cells, np.int32)
        self.h = self.L / self.cells.astype(self.real)
        self.hydrodynamicRadius = self.real(0.91) * self.L[0] / self.real(self.cells[0])     # ICM.cuh:169-171
        self.deltaRFD = self.real((1e-4 if self.real == np.float32 else 1e-6) * float(self.hydrodynamicRadius))
        nx, ny, nz = cells
        self.v = np.zeros((nz, ny, nx, 3), self.real) if initial_velocity is None else np.ascontiguousarray(initial_velocity, self.real)
        self.advOld = np.zeros_like(self.v)
        self.step = 0

    def getSelfMobility(self):   # ICM.cuh:164-168
        rh = float(self.hydrodynamicRadius)
        return 1.0 / (6 * math.pi * self.viscosity * rh) * (1 - 2.837297 * rh / float(self.L[0]))

    def _midpoint(self, mode, pos4, old4):
        o, cr = self.o, self.o.creal
        o.lib.oracle_fib_midpoint_step(mode, _p(pos4), _p(old4), _p(self.v), len(pos4), _p(self.L), _p(self.cells), cr(self.h[0]), cr(self.dt))

    def forwardTime(self, pos4, forces=None, noise=None):
        """pos4 real4[N] advanced in place; `forces(pos4) -> real4[N]` is called at q^{n+1/2} (None: no interactors)."""
        o, cr = self.o, self.o.creal
        nx, ny, nz = (int(c) for c in self.cells)
        self.step += 1
        old = np.zeros_like(pos4)
        self._midpoint(0, pos4, old)
        dV = float(np.prod(self.h))
        amp = 0.0
        if self.temperature != 0:
            if noise is None:
                noise = self.noise_fn(nx * ny * nz)
            noise = np.ascontiguousarray(noise, self.real)
            amp = math.sqrt(2 * self.temperature * self.viscosity * self.dt / dV) / self.density      # :1107
        vnew = np.zeros_like(self.v)
        o.lib.oracle_icm_update_unperturbed(_p(self.v), _p(vnew), _p(self.advOld), _p(self.L), _p(self.cells), cr(self.density),
                                            cr(self.viscosity), cr(amp), cr(self.dt), _p(noise) if noise is not None else None)
        self.v = vnew
        if forces is not None:
            f = np.array(forces(pos4), dtype=self.real)
            f[:, :3] *= self.real(self.dt / self.density)
            o.lib.oracle_fib_spread(_p(pos4), _p(f), len(pos4), _p(self.L), _p(self.cells), cr(self.h[0]), _p(self.v))
        if self.sumThermalDrift and self.temperature > 0:
            o.lib.oracle_icm_thermal_drift(_p(pos4), len(pos4), _p(self.v), _p(self.L), _p(self.cells),
                                           cr((self.dt / self.density) * self.temperature / float(self.deltaRFD)), cr(self.deltaRFD),
                                           self.seed & 0xFFFFFFFF, self.step)
        gk = np.ascontiguousarray(_fft.rfftn(self.v, axes=(0, 1, 2), **_kw).astype(self.cplx))
        o.lib.oracle_icm_solve_stokes(_p(gk), cr(self.viscosity), cr(self.density), cr(self.dt), _p(self.L), _p(self.cells),
                                      int(self.removeTotalMomentum))
        self.v = np.ascontiguousarray((_fft.irfftn(gk, s=(nz, ny, nx), axes=(0, 1, 2), **_kw) * (nx * ny * nz)).astype(self.real))
        self._midpoint(1, pos4, old)

    def getFluidVelocities(self):
        out = np.zeros_like(self.v)
        self.o.lib.oracle_icm_collocate(_p(self.v), _p(out), _p(self.L), _p(self.cells))
        return out
