"""DOUBLE_PRECISION build of path B (the `_f64` entry points of include/uammd_hip.h): the reference makes `real` a build switch
(global/defines.h:9-11) and compiles every accuracy assertion it ships in double.  Thin mirrors of the same classes, arrays are
torch.float64 on the GPU: positions / forces real4 = double[N, 4], velocities real3 = double[N, 3]."""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from ._lib import FCMParameters64, IBMKernel64, MATVEC64, check


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _d3(v):
    v = np.broadcast_to(np.asarray(v, dtype=np.float64), (3,))
    return (C.c_double * 3)(*[float(x) for x in v])


def _i3(v):
    v = np.broadcast_to(np.asarray(v), (3,))
    return (C.c_int * 3)(*[int(x) for x in v])


class Kernels:
    @staticmethod
    def Gaussian(h, tolerance):
        """FCM_ns::Kernels::Gaussian(h, tolerance) in double -> (kernel, a_eff)  (FCM_kernels.cuh:22-58)."""
        k, a = IBMKernel64(), C.c_double(0)
        check(_lib.load().uammd_fcm_gaussian_kernel_f64(float(h), float(tolerance), C.byref(k), C.byref(a)))
        return k, float(a.value)

    @staticmethod
    def adviseGridSize(hydrodynamicRadius, tolerance):
        return float(_lib.load().uammd_fcm_advise_grid_size_f64(float(hydrodynamicRadius), float(tolerance)))

    @staticmethod
    def Peskin3pt(h):
        h = np.broadcast_to(np.asarray(h, dtype=np.float64), (3,))
        return IBMKernel64(1, (C.c_int * 3)(3, 3, 3), 0.0, 0.0, float("inf"), (C.c_double * 3)(*[1.0 / x if x > 0 else 0.0 for x in h]))

    @staticmethod
    def Peskin4pt(h):
        h = np.broadcast_to(np.asarray(h, dtype=np.float64), (3,))
        return IBMKernel64(2, (C.c_int * 3)(4, 4, 4), 0.0, 0.0, float("inf"), (C.c_double * 3)(*[1.0 / x if x > 0 else 0.0 for x in h]))

    @staticmethod
    def Constant(support):
        s = np.broadcast_to(np.asarray(support), (3,))
        return IBMKernel64(3, (C.c_int * 3)(int(s[0]), int(s[1]), int(s[2])), 0.0, 0.0, float("inf"), (C.c_double * 3)(0, 0, 0))


class IBM:
    """IBM<Kernel, Grid, LinearIndex3D>::spread / gather (misc/IBM.cuh:99-203) on interleaved user grids, in double."""

    def __init__(self, kernel, L, periodic, cellDim, nxStride=None):
        self.lib = _lib.load()
        self.kernel, self.L, self.periodic, self.cellDim = kernel, L, periodic, [int(c) for c in cellDim]
        self.nxStride = int(nxStride) if nxStride is not None else self.cellDim[0]

    def _args(self):
        return _d3(self.L), _i3([int(p) for p in np.broadcast_to(self.periodic, (3,))]), _i3(self.cellDim), self.nxStride, C.byref(self.kernel)

    def spread(self, pos, v, gridData):
        ncomp = 1 if v.dim() == 1 else v.shape[1]
        assert pos.dtype == v.dtype == gridData.dtype == torch.float64
        check(self.lib.uammd_ibm_spread_f64(_ptr(pos), pos.shape[1], _ptr(v), ncomp, pos.shape[0], *self._args(), _ptr(gridData), _stream()))

    def gather(self, pos, Jq, gridData):
        ncomp = 1 if Jq.dim() == 1 else Jq.shape[1]
        assert pos.dtype == Jq.dtype == gridData.dtype == torch.float64
        check(self.lib.uammd_ibm_gather_f64(_ptr(pos), pos.shape[1], _ptr(Jq), ncomp, pos.shape[0], *self._args(), _ptr(gridData), _stream()))


class FCM_impl:
    """FCM_impl<Gaussian> (Integrator/BDHI/FCM/FCM_impl.cuh), deterministic part, in double."""

    def __init__(self, L, cells, kernel, viscosity, hydrodynamicRadius):
        self.lib = _lib.load()
        self.L = np.broadcast_to(np.asarray(L, dtype=np.float64), (3,)).copy()
        self.cells = [int(c) for c in cells]
        self.viscosity, self.hydrodynamicRadius = float(viscosity), float(hydrodynamicRadius)
        p = FCMParameters64()
        for k in range(3):
            p.boxSize[k], p.cells[k] = float(self.L[k]), self.cells[k]
        p.viscosity, p.kernel = self.viscosity, kernel
        h = C.c_void_p()
        check(self.lib.uammd_fcm_create_f64(C.byref(p), C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.uammd_fcm_destroy_f64(self.h)
                self.h = None
        except Exception:
            pass

    def getSelfMobility(self):
        return float(self.lib.uammd_fcm_self_mobility(self.hydrodynamicRadius, self.viscosity, float(self.L[0])))

    def computeHydrodynamicDisplacements(self, pos, force, out=None):
        n = pos.shape[0]
        v = out if out is not None else torch.zeros((n, 3), dtype=torch.float64, device=pos.device)
        check(self.lib.uammd_fcm_displacements_f64(self.h, _ptr(pos), _ptr(force), n, _ptr(v), _stream()))
        return v


class PSE:
    """BDHI::PSE deterministic mobility (BDHI_PSE.cuh:92-120 with T = 0): M F = M_far F + M_near F, in double."""

    def __init__(self, L, viscosity, hydrodynamicRadius, tolerance, psi, shearStrain=0.0):
        from .bdhi import nextFFTWiseSize3D
        self.lib = _lib.load()
        L3 = np.broadcast_to(np.asarray(L, dtype=np.float64), (3,)).copy()
        self.L, self.viscosity, self.rh = L3, float(viscosity), float(hydrodynamicRadius)
        h, rc, npts = C.c_void_p(), C.c_double(0), C.c_int(0)
        check(self.lib.uammd_pse_near_create_f64(_d3(L3), self.viscosity, self.rh, float(tolerance), float(psi), C.byref(h), C.byref(rc), C.byref(npts)))
        self.near, self.rcut, self.nPointsTable = h, float(rc.value), int(npts.value)
        raw = _i3(0)
        check(self.lib.uammd_pse_far_raw_cells_f64(_d3(L3), float(psi), float(tolerance), raw))
        self.cells = nextFFTWiseSize3D(list(raw))
        hf, sup, eta = C.c_void_p(), C.c_int(0), C.c_double(0)
        check(self.lib.uammd_pse_far_create_f64(_d3(L3), _i3(self.cells), self.viscosity, self.rh, float(tolerance), float(psi), float(shearStrain),
                                                C.byref(hf), C.byref(sup), C.byref(eta)))
        self.far, self.support, self.eta = hf, int(sup.value), float(eta.value)

    def __del__(self):
        try:
            if getattr(self, "near", None):
                self.lib.uammd_pse_near_destroy_f64(self.near)
                self.near = None
            if getattr(self, "far", None):
                self.lib.uammd_fcm_destroy_f64(self.far)
                self.far = None
        except Exception:
            pass

    def getSelfMobility(self):
        return float(self.lib.uammd_fcm_self_mobility(self.rh, self.viscosity, float(self.L[0])))

    def computeMF(self, pos, force):
        n = pos.shape[0]
        MF = torch.zeros((n, 3), dtype=torch.float64, device=pos.device)
        check(self.lib.uammd_fcm_displacements_f64(self.far, _ptr(pos), _ptr(force), n, _ptr(MF), _stream()))      # the far field ADDS
        check(self.lib.uammd_pse_near_mdot_f64(self.near, _ptr(pos), _ptr(force), 4, n, _ptr(MF), _stream()))
        return MF


class LanczosSolver:
    """lanczos::Solver with real = double (misc/LanczosAlgorithm.cuh:32-83).  dot(v, Mv) writes Mv = M v on float64 GPU tensors."""

    def __init__(self):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.uammd_lanczos_create_f64(C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.uammd_lanczos_destroy_f64(self.h)
                self.h = None
        except Exception:
            pass

    def setIterationHardLimit(self, n):
        check(self.lib.uammd_lanczos_set_iteration_hard_limit_f64(self.h, int(n)))

    def getLastRunRequiredSteps(self):
        s = C.c_int(0)
        check(self.lib.uammd_lanczos_get_last_run_required_steps_f64(self.h, C.byref(s)))
        return int(s.value)

    def run(self, dot, Bv, v, tolerance):
        n = v.numel()
        err = []

        def _wrap(ptr, count):
            class _Raw:
                pass
            r = _Raw()
            r.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}
            return torch.as_tensor(r, device="cuda")

        def cb(ctx, d_v, d_Mv, nn, stream):
            try:
                dot(_wrap(d_v, nn), _wrap(d_Mv, nn))
                return 0
            except Exception as e:   # noqa: BLE001 - reported through the C status
                err.append(e)
                return -1
        it = C.c_int(0)
        fn = MATVEC64(cb)
        rc = self.lib.uammd_lanczos_run_f64(self.h, fn, None, _ptr(Bv), _ptr(v), float(tolerance), n, _stream(), C.byref(it))
        if err:
            raise err[0]
        check(rc)
        return int(it.value)


class BDHI2D:
    """BDHI::True2D / BDHI::Quasi2D with real = double (Integrator/Hydro/BDHI_quasi2D.cuh:155-257): one step's particle velocities
    (real2[N]) and the position update.  mode: "True2D" | "Quasi2D"."""

    def __init__(self, mode, L, hydrodynamicRadius, viscosity, temperature, dt, seed, cells=(-1, -1)):
        self.lib = _lib.load()
        Lx, Ly = (L, L) if np.isscalar(L) else (L[0], L[1])
        p = _lib.BDHI2DParametersF64()
        p.boxSize[0], p.boxSize[1] = float(Lx), float(Ly)
        p.hydrodynamicRadius, p.viscosity, p.temperature, p.dt = float(hydrodynamicRadius), float(viscosity), float(temperature), float(dt)
        p.cells[0], p.cells[1] = int(cells[0]), int(cells[1])
        p.seed, p.kernel = int(seed), {"True2D": 0, "Quasi2D": 1}[mode]
        h, cd, sup = C.c_void_p(), (C.c_int * 2)(0, 0), C.c_int(0)
        check(self.lib.uammd_bdhi2d_create_f64(C.byref(p), C.byref(h), C.byref(cd), C.byref(sup)))
        self.h, self.cells, self.support, self.dt = h, [int(cd[0]), int(cd[1])], int(sup.value), float(dt)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.uammd_bdhi2d_destroy_f64(self.h)
                self.h = None
        except Exception:
            pass

    def velocities(self, pos, force=None):
        n = pos.shape[0]
        vel = torch.empty((n, 2), dtype=torch.float64, device=pos.device)
        check(self.lib.uammd_bdhi2d_velocities_f64(self.h, _ptr(pos), _ptr(force), n, _ptr(vel), _stream()))
        return vel

    def forwardTime(self, pos, force=None):
        vel = self.velocities(pos, force)
        check(self.lib.uammd_bdhi2d_update_positions_f64(_ptr(pos), _ptr(vel), pos.shape[0], self.dt, _stream()))
        return vel


class Poisson:
    """Poisson with real = double (Interactor/SpectralEwaldPoisson.cuh:83-136): sum(force, energy) and the field / potential at the
    particles.  Near field over all pairs."""

    def __init__(self, L, epsilon, gw, tolerance=1e-5, split=-1.0, upsampling=-1.0):
        self.lib = _lib.load()
        p = _lib.PoissonParametersF64()
        L3 = np.broadcast_to(np.asarray(L, dtype=np.float64), (3,))
        for a in range(3):
            p.boxSize[a] = float(L3[a])
        p.epsilon, p.tolerance, p.gw, p.split, p.upsampling = float(epsilon), float(tolerance), float(gw), float(split), float(upsampling)
        h, info = C.c_void_p(), _lib.PoissonInfoF64()
        if self.lib.uammd_poisson_create_f64(C.byref(p), C.byref(h), C.byref(info)) != 0:
            msg = self.lib.uammd_hip_last_error().decode()
            raise ValueError(msg) if "[Poisson]" in msg else RuntimeError(msg)
        self.h = h
        self.cells, self.support, self.nearFieldCutOff, self.ntable = [int(c) for c in info.cells], int(info.support), float(info.nearFieldCutOff), int(info.nTable)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.uammd_poisson_destroy_f64(self.h)
                self.h = None
        except Exception:
            pass

    def sum(self, pos, charge, force4, energy, force=True, energy_flag=False):
        check(self.lib.uammd_poisson_sum_f64(self.h, _ptr(pos), _ptr(charge), pos.shape[0], _ptr(force4), _ptr(energy), int(force), int(energy_flag), _stream()))

    def computeFieldPotentialAtParticles(self, pos, charge):
        n = pos.shape[0]
        fp = torch.zeros((n, 4), dtype=torch.float64, device=pos.device)
        check(self.lib.uammd_poisson_field_potential_f64(self.h, _ptr(pos), _ptr(charge), n, _ptr(fp), None, None, _stream()))
        return fp


class BD:
    """BD::EulerMaruyama / MidPoint / AdamsBashforth / Leimkuhler with real = double (Integrator/BrownianDynamics.cuh:57-183): positions and
    forces double[N, 4]; `forces(pos)` fills the force array before each position update, as the scheme's interactors would (None: free
    particles).  seed: the reference takes the System generator's third draw."""
    SCHEMES = {"EulerMaruyama": 0, "MidPoint": 1, "AdamsBashforth": 2, "Leimkuhler": 3}

    def __init__(self, scheme, temperature, viscosity, hydrodynamicRadius, dt, seed, K=None, is2D=False, forces=None):
        self.lib = _lib.load()
        self.scheme = self.SCHEMES[scheme]
        self.temperature, self.dt, self.is2D, self.seed = float(temperature), float(dt), bool(is2D), int(seed) & 0xFFFFFFFF
        self.selfMobility = 1.0 / (6.0 * math.pi * float(viscosity) * float(hydrodynamicRadius))
        self.K = None if K is None else (C.c_double * 9)(*[float(x) for x in np.asarray(K, dtype=np.float64).reshape(9)])
        self.forces, self.steps, self.aux, self.force = forces, 0, None, None

    def _eval(self, pos):
        if self.force is None or self.force.shape != pos.shape:
            self.force = torch.zeros_like(pos)
        self.force.zero_()
        if self.forces is not None:
            self.forces(pos, self.force)

    def _advance(self, pos, substep, original_index=None):
        check(self.lib.uammd_bd_scheme_step_f64(self.scheme, substep, _ptr(pos), _ptr(self.aux), None, _ptr(original_index), _ptr(self.force), self.K,
                                                self.selfMobility, None, self.dt, int(self.is2D), self.temperature, pos.shape[0], self.steps, self.seed,
                                                _stream()))

    def forwardTime(self, pos, original_index=None):
        self.steps += 1
        if self.scheme == 1:       # MidPoint: two force evaluations (BrownianDynamics.cu:216-232)
            if self.aux is None:
                self.aux = torch.empty_like(pos)
            self._eval(pos)
            self._advance(pos, 0)
            self._eval(pos)
            self._advance(pos, 1)
        elif self.scheme == 2:     # AdamsBashforth: the previous step's forces are kept (:291-308)
            if self.steps == 1:
                self._eval(pos)
            self.aux = self.force.clone()
            self._eval(pos)
            self._advance(pos, 0)
        else:
            self._eval(pos)
            self._advance(pos, 0, original_index if self.scheme == 3 else None)
