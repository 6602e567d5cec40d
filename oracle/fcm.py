"""ORACLE — TEST INFRASTRUCTURE ONLY.

FCM_impl::computeHydrodynamicDisplacements restated on the CPU (Integrator/BDHI/FCM/FCM_impl.cuh:652-693):
spread (oracle/src/ibm.c) -> 3 x 3-D R2C FFT -> forceFourier2Vel -> fourierBrownianNoise -> 3 x C2R FFT ->
gather.  The FFT is scipy/numpy pocketfft (the reference calls cuFFT, a third-party library that is not in
/root/reference; a DFT has one answer up to rounding).  Data layouts follow the reference exactly: real grid
real3[nz][ny][2(nx/2+1)] (padded, FCM_impl.cuh:253), Fourier grid complex3[nz][ny][nx/2+1].
"""
import os

import numpy as np

try:
    import scipy.fft as _fft
    # as many FFT workers as OpenMP threads (bench.py sets OMP_NUM_THREADS to the CPUs the container grants)
    _kw = dict(workers=int(os.environ.get("OMP_NUM_THREADS", "-1")))
except Exception:  # pragma: no cover
    import numpy.fft as _fft
    _kw = {}


class FCMOracle:
    """BDHI::FCM (BDHI_FCM.cuh:84-147) + FCM_impl, deterministic and stochastic parts, no torques."""

    def __init__(self, oracle, L, cells, tolerance=1e-3, viscosity=1.0, seed=1234, kernel=None):
        self.o = oracle
        self.real = oracle.real
        self.cplx = np.complex64 if self.real == np.float32 else np.complex128
        self.L = np.broadcast_to(np.asarray(L, dtype=self.real), (3,)).copy()
        self.cells = np.asarray(cells, dtype=np.int32)
        self.viscosity = viscosity
        self.seed = seed
        self.seed2 = 0  # `static uint seed2` of addBrownianNoise (FCM_impl.cuh:517), incremented per call with T > 0
        h = (self.L / self.cells.astype(self.real)).astype(self.real)
        self.h = float(min(h))           # BDHI_FCM.cuh:53: h = min(cellSize)
        self.kinfo = kernel or oracle.fcm_gaussian(self.h, tolerance)
        self.kernel = self.kinfo["kernel"]
        self.hydrodynamicRadius = self.kinfo["a_eff"]  # fixHydrodynamicRadius returns a (FCM_kernels.cuh:52)
        self.nxpad = 2 * (int(self.cells[0]) // 2 + 1)

    def getSelfMobility(self):
        return self.o.fcm_self_mobility(self.hydrodynamicRadius, self.viscosity, float(self.L[0]))

    # -- pieces ---------------------------------------------------------------------------------------
    def spread(self, pos, f3):
        return self.o.ibm_spread(pos, f3, self.L, 1, self.cells, self.kernel, nx_stride=self.nxpad)

    def forward(self, grid_real):
        nx = int(self.cells[0])
        g = grid_real[:, :, :nx, :]
        out = _fft.rfftn(g, axes=(0, 1, 2), **_kw)  # [nz, ny, nx/2+1, 3] — wait: rfftn halves the LAST listed axis
        return np.ascontiguousarray(out.astype(self.cplx))

    def inverse(self, grid_k):
        nx, ny, nz = (int(c) for c in self.cells)
        out = _fft.irfftn(grid_k, s=(nz, ny, nx), axes=(0, 1, 2), **_kw) * (nx * ny * nz)  # cuFFT C2R is unnormalised
        pad = np.zeros((nz, ny, self.nxpad, 3), self.real)
        pad[:, :, :nx, :] = out.astype(self.real)
        return pad

    def displacements(self, pos, force, temperature=0.0, prefactor=0.0, grids=None):
        """Returns linear velocities real3[N]; `grids` (dict) receives the intermediate grids for the tests."""
        pos = self.o.r(pos)
        f3 = self.o.r(np.asarray(force)[:, :3])
        gr = self.spread(pos, f3)
        gk = self.forward(gr)
        self.o.fcm_force_fourier_to_vel(gk, self.viscosity, self.L, self.cells)
        if temperature > 0:
            self.seed2 += 1
            npf = self.o.fcm_noise_prefactor(prefactor, temperature, self.L, self.cells)
            self.o.fcm_fourier_brownian_noise(gk, self.L, self.cells, npf, self.viscosity, self.seed, self.seed2)
        gv = self.inverse(gk)
        v = self.o.ibm_gather(pos, gv, self.L, 1, self.cells, self.kernel, nx_stride=self.nxpad)
        if grids is not None:
            grids.update(spread=gr, fourier=gk, velocity=gv)
        return v


    # -- torques (FCM_impl.cuh:652-693 with torque != nullptr) ----------------------------------------------------
    def torque_kernel(self, hydrodynamicRadius=None, tolerance=1e-3):
        """detail::initializeKernelTorque (BDHI_FCM.cuh:69-80) -> oracle IBM kernel."""
        import ctypes as C
        from .oracle import _p
        a = self.hydrodynamicRadius if hydrodynamicRadius is None else hydrodynamicRadius
        out = np.zeros(3, self.real)
        self.o.lib.oracle_fcm_torque_gaussian_init.restype = C.c_int
        sup = self.o.lib.oracle_fcm_torque_gaussian_init(self.o.creal(a), self.o.creal(self.h), self.o.creal(tolerance), _p(out))
        return self.o.ibm_kernel("gaussian", int(sup), out[0], out[1], out[2]), int(sup)

    def _half_curl(self, gin, gout, accumulate):
        from .oracle import _p
        self.o.lib.oracle_fcm_half_curl_fourier(_p(gin), _p(gout), _p(self.L), _p(self.cells), int(accumulate))

    def displacements_torque(self, pos, force, torque, kernel_torque, temperature=0.0, prefactor=0.0):
        """Returns (linear real3[N], angular real3[N])."""
        o = self.o
        pos = o.r(pos)
        nx, ny, nz = (int(c) for c in self.cells)
        if force is not None:
            gk = self.forward(self.spread(pos, o.r(np.asarray(force)[:, :3])))
        else:
            gk = np.zeros((nz, ny, nx // 2 + 1, 3), self.cplx)
        gt = o.ibm_spread(pos, o.r(np.asarray(torque)[:, :3]), self.L, 1, self.cells, kernel_torque, nx_stride=self.nxpad)
        gtk = self.forward(gt)
        self._half_curl(gtk, gk, True)
        o.fcm_force_fourier_to_vel(gk, self.viscosity, self.L, self.cells)
        if temperature > 0:
            self.seed2 += 1
            npf = o.fcm_noise_prefactor(prefactor, temperature, self.L, self.cells)
            o.fcm_fourier_brownian_noise(gk, self.L, self.cells, npf, self.viscosity, self.seed, self.seed2)
        gak = np.zeros_like(gk)
        self._half_curl(gk, gak, False)
        ga = self.inverse(gak)
        w = o.ibm_gather(pos, ga, self.L, 1, self.cells, kernel_torque, nx_stride=self.nxpad)
        gv = self.inverse(gk)
        v = o.ibm_gather(pos, gv, self.L, 1, self.cells, self.kernel, nx_stride=self.nxpad)
        return v, w
