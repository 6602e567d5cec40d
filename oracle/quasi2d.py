"""ORACLE — TEST INFRASTRUCTURE ONLY.

BDHI::True2D / BDHI::Quasi2D (Integrator/Hydro/BDHI_quasi2D.cu(h)) restated on the CPU on top of oracle/src/quasi2d.c, the IBM
oracle (2D grid, real2 quantities) and numpy/scipy FFTs:
  BDHI2D::BDHI2D            grid h = 0.8 a, window support and Gaussian variance of the hydrodynamic kernel    .cu:18-88
  forwardTime               reset -> forces -> spread (thermal drift + forces) -> FFT -> G_k -> noise -> FFT -> gather -> Euler  .cu:179-205
Pinned by the reference's own tests (tests/test_oracle_quasi2d.py): test/BDHI/quasi2D/quasi2d_test.cu self mobilities of
both kernels and the fluctuation-dissipation checks.
"""
import ctypes as C

import numpy as np

from .fcm import _fft, _kw
from .oracle import _p
from .pse import next_fft_wise_size_3d

MODES = {"True2D": 0, "Quasi2D": 1}


class BDHI2DOracle:
    def __init__(self, oracle, mode, L, hydrodynamicRadius, viscosity, temperature, dt, cells=None, seed=1234):
        o = self.o = oracle
        self.real, cr = o.real, o.creal
        self.cplx = np.complex64 if self.real == np.float32 else np.complex128
        self.mode = MODES[mode]
        Lx, Ly = (L, L) if np.isscalar(L) else (L[0], L[1])
        self.L = np.array([Lx, Ly, 0], self.real)                       # box = Box(Lx, Ly, 0), .cu:30
        self.a, self.viscosity, self.temperature, self.dt, self.seed = hydrodynamicRadius, viscosity, temperature, dt, seed
        if self.L[0] == 0 and self.L[1] == 0:
            raise RuntimeError("Invalid box")                               # .cu:46-52
        if hydrodynamicRadius <= 0:
            raise RuntimeError("Invalid hydrodynamic radius")               # .cu:53-57
        lib = o.lib
        if cells is None:
            raw = np.zeros(2, np.int32)
            lib.oracle_q2d_raw_cells(_p(self.L), cr(self.a), _p(raw))
            c = next_fft_wise_size_3d([raw[0], raw[1], 0])
            cells = [c[0], c[1]]
        self.cells = np.array([cells[0], cells[1], 1], np.int32)
        sup, pref, prefd, tau = C.c_int(0), cr(0), cr(0), cr(0)
        lib.oracle_q2d_window(self.mode, _p(self.L), _p(self.cells), cr(self.a), C.byref(sup), C.byref(pref), C.byref(prefd), C.byref(tau))
        self.support = int(sup.value)
        s3 = [self.support, self.support, 1]
        self.kernel = o.ibm_kernel("gauss2d", s3, pref.value, tau.value, np.inf)
        self.kernelDrift = [o.ibm_kernel("gauss2d_drift_x", s3, prefd.value, tau.value, np.inf),
                            o.ibm_kernel("gauss2d_drift_y", s3, prefd.value, tau.value, np.inf)]
        self.nxpad = 2 * (int(self.cells[0]) // 2 + 1)
        self.hasThermalDrift = self.mode == 1
        self.counter = 0                                                    # `static ullint counter`, .cu:455-456

    def velocities(self, pos, force2=None, grids=None):
        """particleVels real2[N] of one forwardTime (force2 = None: no interactors)."""
        o, cr = self.o, self.o.creal
        nx, ny = int(self.cells[0]), int(self.cells[1])
        pos = o.r(pos)
        n = len(pos)
        per = [1, 1, 0]
        grid = np.zeros((1, ny, self.nxpad, 2), self.real)
        nonzero = False
        if self.hasThermalDrift and self.temperature > 0:                   # spreadThermalDrift, .cu:234-257
            vx = np.zeros((n, 2), self.real)
            vx[:, 0] = -self.temperature
            vy = np.zeros((n, 2), self.real)
            vy[:, 1] = -self.temperature
            o.ibm_spread(pos, vx, self.L, per, self.cells, self.kernelDrift[0], grid=grid, nx_stride=self.nxpad)
            o.ibm_spread(pos, vy, self.L, per, self.cells, self.kernelDrift[1], grid=grid, nx_stride=self.nxpad)
            nonzero = True
        if force2 is not None:                                              # spreadParticleForces, .cu:268-283
            o.ibm_spread(pos, o.r(np.asarray(force2)[:, :2]), self.L, per, self.cells, self.kernel, grid=grid, nx_stride=self.nxpad)
            nonzero = True
        gk = np.zeros((ny, nx // 2 + 1, 2), self.cplx)
        if nonzero:
            gk = np.ascontiguousarray(_fft.rfftn(grid[0, :, :nx, :], axes=(0, 1), **_kw).astype(self.cplx))
            o.lib.oracle_q2d_force_fourier_to_vel(_p(gk), self.mode, cr(self.viscosity), _p(self.L), _p(self.cells), cr(self.a))
        if self.temperature > 0:                                            # addStochastichTermFourier, .cu:450-469
            self.counter += 1
            pref = np.sqrt(2.0 * self.temperature / (self.viscosity * self.dt * float(self.L[0]) * float(self.L[1])))
            o.lib.oracle_q2d_fourier_brownian_noise(_p(gk), self.mode, _p(self.L), _p(self.cells), cr(pref), cr(self.a),
                                                    C.c_uint(self.seed), C.c_uint(self.counter))
        gv = np.zeros((1, ny, self.nxpad, 2), self.real)
        if force2 is not None or self.temperature > 0:
            gv[0, :, :nx, :] = (_fft.irfftn(gk, s=(ny, nx), axes=(0, 1), **_kw) * (nx * ny)).astype(self.real)
        if grids is not None:
            grids.update(spread=grid, fourier=gk, velocity=gv)
        return o.ibm_gather(pos, gv, self.L, per, self.cells, self.kernel, nx_stride=self.nxpad)

    def forwardTime(self, pos, force2=None):
        """pos real4[N] advanced in place: pos += make_real4(vel * dt) (.cu:521-541)."""
        v = self.velocities(pos, force2)
        # one rounding, as the fused multiply-add the GPU issues (the product of two floats is exact in double)
        dt = np.float64(self.real(self.dt))
        pos[:, 0] = (v[:, 0].astype(np.float64) * dt + pos[:, 0].astype(np.float64)).astype(pos.dtype)
        pos[:, 1] = (v[:, 1].astype(np.float64) * dt + pos[:, 1].astype(np.float64)).astype(pos.dtype)
        return v
