"""Host-side mirror of the UAMMD interfaces on path B (IBM + BDHI::FCM).

Citations (relative to /root/reference/src):
  IBM<Kernel>::spread/gather             misc/IBM.cuh:99-203
  IBM_kernels / FCM_ns::Kernels          misc/IBM_kernels.cuh, Integrator/BDHI/FCM/FCM_kernels.cuh:22-58
  BDHI::Parameters                       Integrator/BDHI/BDHI.cuh:13-24
  BDHI::FCM (Method concept)             Integrator/BDHI/BDHI_FCM.cuh:84-147
  BDHI::FCMIntegrator                    Integrator/BDHI/BDHI_FCM.cuh:149-198, BDHI_FCM.cu:95-119
  BDHI::EulerMaruyama<Method>            Integrator/BDHI/BDHI_EulerMaruyama.cu:82-166
  BDHI::PSE (Method)                     Integrator/BDHI/BDHI_PSE.cuh:79-176, PSE/NearField.cuh, PSE/FarField.cuh
  BDHI::Lanczos (Method)                 Integrator/BDHI/BDHI_Lanczos.cuh:20-67, BDHI_Lanczos.cu
  FCM_impl                               Integrator/BDHI/FCM/FCM_impl.cuh:56-129, :652-693
All compute happens in libuammd_hip.so through the C ABI.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from ._lib import FCMParameters, IBMKernel, check, f3, i3
from .md import Box, Integrator, _ptr, current_stream


def nextFFTWiseSize3D(size):
    """utils/Grid.cuh:142-213: next size of the form 2^a 3^b 5^c 7^d 11^e (a >= 1) per dimension; the `continue`
    in the forbidden-size loop of the reference only continues that inner loop, so forbidden sizes are NOT skipped
    (reproduced)."""
    out = []
    for s in size:
        n = max(int(s), 1)
        c = n
        while True:
            m = c
            if m % 2 == 0:
                for p in (2, 3, 5, 7, 11):
                    while m % p == 0:
                        m //= p
                if m == 1 and _exponents_ok(c):
                    break
            c += 1
        out.append(c)
    return out


def _exponents_ok(c):
    lim = {2: 64, 3: 64, 5: 5, 7: 4, 11: 3}  # loop bounds n5=6, n7=5, n11=4 of Grid.cuh:146-150
    for p, mx in lim.items():
        e = 0
        while c % p == 0:
            c //= p
            e += 1
        if e > mx:
            return False
    return True


class Kernels:
    """Spreading windows that can cross the C ABI (include/uammd_hip.h uammd_ibm_kernel)."""

    @staticmethod
    def Gaussian(h, tolerance):
        """FCM_ns::Kernels::Gaussian(h, tolerance) -> (kernel, a_eff)."""
        lib = _lib.load()
        k = IBMKernel()
        a = C.c_float(0)
        check(lib.uammd_fcm_gaussian_kernel(float(h), float(tolerance), C.byref(k), C.byref(a)))
        return k, float(a.value)

    @staticmethod
    def adviseGridSize(hydrodynamicRadius, tolerance):
        return float(_lib.load().uammd_fcm_advise_grid_size(float(hydrodynamicRadius), float(tolerance)))

    @staticmethod
    def Peskin3pt(h):
        h = np.broadcast_to(np.asarray(h, dtype=np.float32), (3,))
        inv = [float(np.float32(1.0) / x) if x > 0 else 0.0 for x in h]
        return IBMKernel(1, (C.c_int * 3)(3, 3, 3), 0.0, 0.0, float("inf"), (C.c_float * 3)(*inv))

    @staticmethod
    def Peskin4pt(h):
        h = np.broadcast_to(np.asarray(h, dtype=np.float32), (3,))
        inv = [float(np.float32(1.0) / x) if x > 0 else 0.0 for x in h]
        return IBMKernel(2, (C.c_int * 3)(4, 4, 4), 0.0, 0.0, float("inf"), (C.c_float * 3)(*inv))

    @staticmethod
    def BarnettMagland(alpha, beta, support, lengthUnit=1.0):
        """IBM_kernels::BarnettMagland(alpha, beta): phi(r) = exp(beta(sqrt(1-(r/alpha)^2)-1))/norm, 0 beyond alpha
        (r in units of lengthUnit, as FCM_ns::Kernels::BarnettMagland::phi does with the cell size)."""
        k = IBMKernel()
        check(_lib.load().uammd_ibm_barnett_magland_kernel(float(alpha), float(beta), int(support), float(lengthUnit),
                                                           C.byref(k)))
        return k

    @staticmethod
    def GaussianFlexibleSixPoint(h):
        """IBM_kernels::GaussianFlexible::sixPoint(h), support 6."""
        h = np.broadcast_to(np.asarray(h, dtype=np.float32), (3,))
        inv = [float(np.float32(1.0) / x) if x > 0 else 0.0 for x in h]
        return IBMKernel(5, (C.c_int * 3)(6, 6, 6), 0.0, 0.0, float("inf"), (C.c_float * 3)(*inv))

    @staticmethod
    def Constant(support):
        s = np.broadcast_to(np.asarray(support), (3,))
        return IBMKernel(3, (C.c_int * 3)(int(s[0]), int(s[1]), int(s[2])), 0.0, 0.0, float("inf"), (C.c_float * 3)(0, 0, 0))


class FCMKernels:
    """FCM_ns::Kernels::* (BDHI/FCM/FCM_kernels.cuh): each entry builds the window for a cell size h and a tolerance
    and carries the two static helpers FCM needs: adviseGridSize(a, tol) and fixHydrodynamicRadius(a, h).
    Use: kernel, a_eff = FCMKernels.make(name, h, tol); FCM_impl(box, cells, kernel, eta, seed, a_eff)."""

    @staticmethod
    def _bm_support(tolerance):  # BarnettMagland::computeSupport, FCM_kernels.cuh:91-95
        i_w = max(1.5, int(-math.log10(tolerance) + 2) / 2.0)
        i_w = min(float(np.float32(9.0)), i_w)
        return int(math.ceil(np.float32(i_w)))

    @staticmethod
    def _bm_upsampling(w):       # FCM_kernels.cuh:97-101
        return 1.36409985665115 * math.pow(float(np.float32(w)), -0.53028415751646)

    @staticmethod
    def adviseGridSize(name, hydrodynamicRadius, tolerance):
        a = float(hydrodynamicRadius)
        if name == "Gaussian":
            return Kernels.adviseGridSize(a, tolerance)
        if name == "BarnettMagland":   # :140-144
            return float(np.float32(a * FCMKernels._bm_upsampling(FCMKernels._bm_support(tolerance))))
        if name == "Peskin3pt":        # :167-169
            return a
        if name == "Peskin4pt":        # :185-189
            return float(np.float32(a) / np.float32(1.31))
        if name == "GaussianFlexible6pt":  # :207-211
            return float(np.float32(a) / np.float32(1.5195))
        raise ValueError(name)

    @staticmethod
    def make(name, h, tolerance):
        """-> (kernel, fixHydrodynamicRadius(.., h))."""
        h = float(np.float32(h))
        if name == "Gaussian":
            return Kernels.Gaussian(h, tolerance)
        if name == "BarnettMagland":   # :85-89, :130-133, :146-149
            w = FCMKernels._bm_support(tolerance)
            alpha = float(np.float32(w * 0.5))
            k = Kernels.BarnettMagland(alpha, float(np.float32(1.8 * w * 2)), int(math.ceil(2 * alpha)), lengthUnit=h)
            return k, float(np.float32(h / FCMKernels._bm_upsampling(k.support[0])))
        if name == "Peskin3pt":
            return Kernels.Peskin3pt(h), h
        if name == "Peskin4pt":
            return Kernels.Peskin4pt(h), float(np.float32(h) * np.float32(1.31))
        if name == "GaussianFlexible6pt":
            return Kernels.GaussianFlexibleSixPoint(h), float(np.float32(h) * np.float32(1.5195))
        raise ValueError(name)


class IBM:
    """IBM<Kernel, Grid, LinearIndex3D>(kernel, grid[, cell2index]) — misc/IBM.cuh:99-203."""

    def __init__(self, kernel, box, cellDim, nxStride=None):
        self.lib = _lib.load()
        self.kernel, self.box = kernel, box
        self.cellDim = [int(c) for c in cellDim]
        self.nxStride = int(nxStride) if nxStride is not None else self.cellDim[0]

    def _args(self):
        return f3(self.box.boxSize), i3([int(p) for p in self.box.periodic]), i3(self.cellDim), self.nxStride, C.byref(self.kernel)

    def spread(self, pos, v, gridData):
        """gridData += S v.  pos: float[N,3|4]; v: float[N] or float[N,3]; gridData float[nz,ny,nxStride(,3)]."""
        ncomp = 1 if v.dim() == 1 else v.shape[1]
        L, per, cd, nxs, k = self._args()
        check(self.lib.uammd_ibm_spread(_ptr(pos), pos.shape[1], _ptr(v), ncomp, pos.shape[0], L, per, cd, nxs, k,
                                        _ptr(gridData), current_stream()))

    def gather(self, pos, Jq, gridData):
        """Jq += J q."""
        ncomp = 1 if Jq.dim() == 1 else Jq.shape[1]
        L, per, cd, nxs, k = self._args()
        check(self.lib.uammd_ibm_gather(_ptr(pos), pos.shape[1], _ptr(Jq), ncomp, pos.shape[0], L, per, cd, nxs, k,
                                        _ptr(gridData), current_stream()))


class _Parameters:
    """BDHI::Parameters + FCM_impl::Parameters (BDHI.cuh:13-24, FCM_impl.cuh:47-54)."""

    def __init__(self, temperature=0.0, viscosity=1.0, hydrodynamicRadius=-1.0, tolerance=1e-3, dt=0.0, box=None,
                 cells=(-1, -1, -1), seed=0, adaptBoxSize=False):
        self.temperature, self.viscosity, self.hydrodynamicRadius = temperature, viscosity, hydrodynamicRadius
        self.tolerance, self.dt, self.box, self.cells, self.seed = tolerance, dt, box, list(cells), seed
        self.adaptBoxSize = adaptBoxSize


class FCM_impl:
    """FCM_impl<Gaussian, GaussianTorque> without torques: owns the solver handle."""

    def __init__(self, box, cells, kernel, viscosity, seed, hydrodynamicRadius):
        self.lib = _lib.load()
        p = FCMParameters()
        for k in range(3):
            p.boxSize[k] = float(box.boxSize[k])
            p.cells[k] = int(cells[k])
        p.viscosity, p.seed, p.kernel, p.hydrodynamicRadius = float(viscosity), int(seed) & 0xFFFFFFFF, kernel, float(hydrodynamicRadius)
        if box.boxSize[0] <= 0:
            raise RuntimeError("Invalid arguments")  # FCM_impl.cuh:74-77
        if any(int(kernel.support[k]) >= int(cells[k]) for k in range(3)):   # BDHI_FCM.cuh:58-64: the reference says so and goes on
            import sys
            print("[ERROR] [BDHI::FCM] Kernel support is too big, try lowering the tolerance or increasing the box size!.", file=sys.stderr)
        h = C.c_void_p()
        check(self.lib.uammd_fcm_create(C.byref(p), C.byref(h)))
        self.h, self.box, self.cells = h, box, [int(c) for c in cells]
        self.viscosity, self.hydrodynamicRadius = viscosity, hydrodynamicRadius

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.uammd_fcm_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def getHydrodynamicRadius(self):
        return self.hydrodynamicRadius

    def getSelfMobility(self):
        return float(self.lib.uammd_fcm_self_mobility(self.hydrodynamicRadius, self.viscosity, float(self.box.boxSize[0])))

    def getBox(self):
        return self.box

    def computeHydrodynamicDisplacements(self, pos, force, numberParticles, temperature, prefactor, out=None):
        """Returns linear velocities real3[N] (torques are not on this round's path)."""
        if out is None:
            out = torch.empty((numberParticles, 3), dtype=torch.float32, device=pos.device)
        check(self.lib.uammd_fcm_displacements(self.h, _ptr(pos), _ptr(force), int(numberParticles), float(temperature),
                                               float(prefactor), _ptr(out), current_stream()))
        return out

    def stepEulerMaruyama(self, pos, force, numberParticles, temperature, prefactor, dt, out=None, positions_kept=False):
        """computeHydrodynamicDisplacements followed by integrateEulerMaruyamaD (BDHI_FCM.cu:67-92) in one library call
        (uammd_fcm_step_euler_maruyama): pos += v dt in place, inside the interpolation kernel where the solver's gather takes it.
        `out` (real3[N], optional) receives the velocities.  positions_kept: `pos` is exactly what the previous call left (nobody
        asked for write access since): the binning that call did on the way is used (UAMMD_FCM_STEP_POSITIONS_KEPT)."""
        check(self.lib.uammd_fcm_step_euler_maruyama(self.h, _ptr(pos), _ptr(force), int(numberParticles), float(temperature),
                                                     float(prefactor), float(dt), _ptr(out) if out is not None else None,
                                                     1 if positions_kept else 0, current_stream()))
        return out

    def setTorqueKernel(self, kernelTorque=None, tolerance=1e-3):
        """FCM_impl::Parameters::kernelTorque; default = detail::initializeKernelTorque (BDHI_FCM.cuh:69-80)."""
        if kernelTorque is None:
            h = min(float(np.float32(l) / np.float32(c)) for l, c in zip(self.box.boxSize, self.cells))
            kernelTorque = IBMKernel()
            check(self.lib.uammd_fcm_torque_gaussian_kernel(float(self.hydrodynamicRadius), h, float(tolerance), C.byref(kernelTorque)))
        check(self.lib.uammd_fcm_set_torque_kernel(self.h, C.byref(kernelTorque)))
        self.kernelTorque = kernelTorque
        return kernelTorque

    def computeHydrodynamicDisplacementsTorque(self, pos, force, torque, numberParticles, temperature, prefactor):
        """FCM_impl::computeHydrodynamicDisplacements(pos, force, torque, ...) -> (linear real3[N], angular real3[N])."""
        v = torch.empty((numberParticles, 3), dtype=torch.float32, device=pos.device)
        w = torch.empty((numberParticles, 3), dtype=torch.float32, device=pos.device)
        check(self.lib.uammd_fcm_displacements_torque(self.h, _ptr(pos), _ptr(force), _ptr(torque), int(numberParticles),
                                                      float(temperature), float(prefactor), _ptr(v), _ptr(w), current_stream()))
        return v, w

    def fourier_grid(self, pos, force, numberParticles, temperature, prefactor):
        """Test hook: the Fourier grid after the k-space kernel as complex3[nz,ny,nx/2+1,3]."""
        nkx = self.cells[0] // 2 + 1
        out = torch.empty((self.cells[2], self.cells[1], nkx, 6), dtype=torch.float32, device=pos.device)
        check(self.lib.uammd_fcm_displacements_staged(self.h, _ptr(pos), _ptr(force), int(numberParticles),
                                                      float(temperature), float(prefactor), None, 1, current_stream()))
        check(self.lib.uammd_fcm_export_fourier(self.h, _ptr(out), current_stream()))
        return out

    def set_option(self, name, value):
        check(self.lib.uammd_fcm_set_option(self.h, name.encode(), int(value)))

    def seed2(self, value=None):
        if value is None:
            v = C.c_uint(0)
            check(self.lib.uammd_fcm_get_seed2(self.h, C.byref(v)))
            return int(v.value)
        check(self.lib.uammd_fcm_set_seed2(self.h, int(value)))


def _initialize(par, rng):
    """detail::initializeGrid / initializeKernel (BDHI_FCM.cuh:29-66) + the ctor body of BDHI::FCM (:98-110)."""
    if par.seed == 0:
        par.seed = rng.next32()
    box = par.box
    if par.cells[0] <= 0:
        if par.hydrodynamicRadius <= 0:
            raise RuntimeError("[BDHI::FCM] I need an hydrodynamic radius if cell dimensions are not provided!")
        h = np.float32(Kernels.adviseGridSize(par.hydrodynamicRadius, par.tolerance))
        cd = [int(np.float32(l) / h) for l in box.boxSize]
        cd = nextFFTWiseSize3D(cd)
        if par.adaptBoxSize:
            box = Box([np.float32(c) * h for c in cd])
    else:
        cd = [int(c) for c in par.cells]
    cs = [np.float32(l) / np.float32(c) for l, c in zip(box.boxSize, cd)]
    hmin = float(min(cs))
    kernel, a_eff = Kernels.Gaussian(hmin, par.tolerance)
    return box, cd, kernel, a_eff


class FCM:
    """BDHI::FCM — the Method concept used by BDHI::EulerMaruyama (BDHI_FCM.cuh:84-147)."""
    Parameters = _Parameters

    def __init__(self, pd, par):
        self.pd = pd
        self.temperature, self.dt = par.temperature, par.dt
        box, cd, kernel, a_eff = _initialize(par, pd.rng)
        self.fcm = FCM_impl(box, cd, kernel, par.viscosity, par.seed, a_eff)  # fixHydrodynamicRadius returns a_eff

    def setup_step(self):
        pass

    def computeMF(self, MF):
        pd = self.pd
        self.fcm.computeHydrodynamicDisplacements(pd.getPos("read"), pd.getForce("read"), pd.N, self.temperature,
                                                  1.0 / math.sqrt(self.dt), out=MF)

    def computeBdW(self, BdW):
        pass  # included in Fourier space when computing MF (BDHI_FCM.cuh:144-146)

    def finish_step(self):
        pass

    def getHydrodynamicRadius(self):
        return self.fcm.getHydrodynamicRadius()

    def getSelfMobility(self):
        return self.fcm.getSelfMobility()


class FCMIntegrator(Integrator):
    """BDHI::FCMIntegrator (BDHI_FCM.cuh:155-199, BDHI_FCM.cu:7-119): forces (and torques when the particles carry
    orientations) -> linear/angular velocities -> pos += v dt, dir = rotVec2Quaternion(w dt) * dir."""
    Parameters = _Parameters

    def __init__(self, pd, par):
        super().__init__(pd)
        self.temperature, self.dt = par.temperature, par.dt
        box, cd, kernel, a_eff = _initialize(par, pd.rng)
        self.fcm = FCM_impl(box, cd, kernel, par.viscosity, par.seed, a_eff)
        self.fcm.setTorqueKernel(getattr(par, "kernelTorque", None), par.tolerance)  # detail::initializeKernelTorque
        self._v = torch.empty((pd.N, 3), dtype=torch.float32, device=pd.device)
        self._pos_touched = True
        pd.connectPosWrite(self._on_pos_write)
        pd.connectReorder(self._on_pos_write)

    def _on_pos_write(self):
        self._pos_touched = True

    def getFCM_impl(self):
        return self.fcm

    def forwardTime(self):
        pd = self.pd
        self.steps += 1
        for it in self.interactors:
            it.updateSimulationTime(self.steps * self.dt)
        if self.steps == 1:
            for it in self.interactors:
                it.updateTimeStep(self.dt)
                it.updateTemperature(self.temperature)
                it.updateBox(self.fcm.getBox())
        pd.getForce("write").zero_()
        if pd.isDirAllocated():            # computeCurrentForces, BDHI_FCM.cu:50-57
            pd.getTorque("write").zero_()
        for it in self.interactors:
            it.sum(force=True)
        torque = pd.getTorqueIfAllocated("read")
        dirs = pd.getDirIfAllocated("readwrite")
        w = None
        if torque is None and dirs is None:   # no rotation: the update rides in the solver's interpolation kernel
            kept = not self._pos_touched      # (getPosWriteRequestedSignal: somebody may have moved the particles since our last step)
            pos = pd.getPos("readwrite")
            self.fcm.stepEulerMaruyama(pos, pd.getForce("read"), pd.N, self.temperature, 1.0 / math.sqrt(self.dt), self.dt,
                                       out=self._v, positions_kept=kept)
            self._pos_touched = False
            return
        if torque is not None:
            v, w = self.fcm.computeHydrodynamicDisplacementsTorque(pd.getPos("read"), pd.getForce("read"), torque, pd.N,
                                                                   self.temperature, 1.0 / math.sqrt(self.dt))
        else:
            v = self.fcm.computeHydrodynamicDisplacements(pd.getPos("read"), pd.getForce("read"), pd.N, self.temperature,
                                                          1.0 / math.sqrt(self.dt), out=self._v)
        if dirs is None:
            check(self.lib.uammd_fcm_euler_maruyama(_ptr(pd.getPos("readwrite")), None, _ptr(v), pd.N, self.dt,
                                                    current_stream()))
        else:
            check(self.lib.uammd_fcm_euler_maruyama_dir(_ptr(pd.getPos("readwrite")), _ptr(dirs), None, _ptr(v),
                                                        _ptr(w) if w is not None else None, pd.N, self.dt,
                                                        current_stream()))


class _PSEParameters(_Parameters):
    """pse_ns::Parameters (PSE/utils.cuh:17-24)."""

    def __init__(self, psi=0.5, shearStrain=0.0, **kw):
        super().__init__(**kw)
        self.psi, self.shearStrain = psi, shearStrain


class PSE:
    """BDHI::PSE — Positively Split Ewald RPY, the Method concept of BDHI::EulerMaruyama (BDHI_PSE.cuh:79-176)."""
    Parameters = _PSEParameters

    def __init__(self, pd, par):
        self.lib = _lib.load()
        self.pd = pd
        self.hydrodynamicRadius, self.temperature, self.dt = par.hydrodynamicRadius, par.temperature, par.dt
        L = [float(x) for x in par.box.boxSize]
        self.M0 = float(self.lib.uammd_fcm_self_mobility(par.hydrodynamicRadius, par.viscosity, L[0]))  # initialization.cu:31-47
        # pse_ns::checkInputValidity, PSE/initialization.cu:11-29
        if L[0] == 0 and L[1] == 0 and L[2] == 0:
            raise ValueError("Box of size zero detected")
        if par.tolerance > 0.1:
            raise ValueError("Tolerance too high")
        # NearField ctor draws its seed first, then FarField (initialization.cu:57-59)
        seed_near = pd.rng.next32()
        h, rc, npts = C.c_void_p(), C.c_float(0), C.c_int(0)
        check(self.lib.uammd_pse_near_create(f3(L), float(par.viscosity), float(par.hydrodynamicRadius), float(par.tolerance),
                                             float(par.psi), float(par.shearStrain), seed_near, C.byref(h), C.byref(rc),
                                             C.byref(npts)))
        self.near, self.rcut, self.nPointsTable = h, float(rc.value), int(npts.value)
        # CellList::update rebuilds only after a position write or a reorder (CellList.cuh:94-98,134-136): one build per step, not one
        # per near-field call
        check(self.lib.uammd_pse_near_set_option(self.near, b"lazy_list", 1))
        pd.connectPosWrite(self._positions_changed)
        pd.connectReorder(self._positions_changed)
        seed_far = pd.rng.next32()
        raw = i3(0)
        check(self.lib.uammd_pse_far_raw_cells(f3(L), float(par.psi), float(par.tolerance), raw))
        self.cells = nextFFTWiseSize3D(list(raw))
        hf, sup, eta = C.c_void_p(), C.c_int(0), C.c_float(0)
        check(self.lib.uammd_pse_far_create(f3(L), i3(self.cells), float(par.viscosity), float(par.hydrodynamicRadius),
                                            float(par.tolerance), float(par.psi), float(par.shearStrain), seed_far, C.byref(hf),
                                            C.byref(sup), C.byref(eta)))
        self.far, self.support, self.eta = hf, int(sup.value), float(eta.value)
        self.lastLanczosIterations = 0

    def __del__(self):
        try:
            if getattr(self, "near", None):
                self.lib.uammd_pse_near_destroy(self.near)
                self.near = None
            if getattr(self, "far", None):
                self.lib.uammd_fcm_destroy(self.far)
                self.far = None
        except Exception:
            pass

    def _positions_changed(self, *a):
        if getattr(self, "near", None):
            self.lib.uammd_pse_near_positions_changed(self.near)

    def setup_step(self):
        pass

    def finish_step(self):
        pass

    def setShearStrain(self, g):
        check(self.lib.uammd_pse_near_set_shear_strain(self.near, float(g)))
        check(self.lib.uammd_pse_far_set_shear_strain(self.far, float(g)))

    def getHydrodynamicRadius(self):
        return self.hydrodynamicRadius

    def getSelfMobility(self):
        return self.M0

    def _far(self, force, MF, temperature, prefactor):
        pd = self.pd
        seed2 = pd.rng.next32() if temperature > 0 else 0        # FarField.cuh:499: drawn only when T > 0
        check(self.lib.uammd_pse_far_displacements(self.far, _ptr(pd.getPos("read")), _ptr(force), pd.N, float(temperature),
                                                   float(prefactor), seed2, _ptr(MF), current_stream()))

    def _near_stochastic(self, BdW, temperature, prefactor):
        pd = self.pd
        if temperature == 0:
            return
        seed2 = pd.rng.next32()                                   # NearField.cuh:276
        it = C.c_int(0)
        check(self.lib.uammd_pse_near_stochastic(self.near, _ptr(pd.getPos("read")), pd.N, float(temperature), float(prefactor),
                                                 seed2, _ptr(BdW), current_stream(), C.byref(it)))
        self.lastLanczosIterations = int(it.value)

    def computeMF(self, MF):
        """MF = M_far F + far noise (prefactor 1/sqrt(dt)) + M_near F   (BDHI_PSE.cuh:92-120)."""
        pd = self.pd
        MF.zero_()
        force = pd.getForce("read")
        # (the near field's list and pair records are queued first: their one host read then happens while the far field runs)
        check(self.lib.uammd_pse_near_prepare(self.near, _ptr(pd.getPos("read")), pd.N, current_stream()))
        self._far(force, MF, self.temperature, 1.0 / math.sqrt(self.dt))
        check(self.lib.uammd_pse_near_mdot(self.near, _ptr(pd.getPos("read")), _ptr(force), pd.N, _ptr(MF), current_stream()))

    def computeBdW(self, BdW):
        self._near_stochastic(BdW, self.temperature, 1.0)       # BDHI_PSE.cuh:122-126

    def computeMFandBdW(self, MF, BdW):
        """computeMF and computeBdW (BDHI_PSE.cuh:92-126) as EulerMaruyama runs them when T > 0, queued so that the step's one wait for the
        GPU — the Lanczos solve's convergence check — has work behind it: near-field list and pair records, the solve, and the FAR FIELD
        from inside the solve (uammd_pse_near_set_interleave: behind the check's kernels, while the host is busy with the check), then the
        near-field M F.  Run one after the other, the host came back from the check to an empty stream and the GPU idled through the
        step's last launches and the next step's first (~50 us of 0.6 ms).  The same calls with the same arguments; the two draws of
        System::rng() keep the reference's order (far field: FarField.cuh:499, then near field: NearField.cuh:276)."""
        pd = self.pd
        MF.zero_()
        force = pd.getForce("read")
        pos = pd.getPos("read")
        st = current_stream()
        seed_far = pd.rng.next32()
        seed_near = pd.rng.next32()
        # (no uammd_pse_near_prepare here: the solve below starts with the same list and record launches, and leaves the copy of the records'
        # counters to its own noise kernel)
        failed = []

        def far_half(half):
            def queue(_ctx, _stream):
                try:
                    check(self.lib.uammd_pse_far_displacements_half(self.far, _ptr(pos), _ptr(force), pd.N, float(self.temperature),
                                                                    1.0 / math.sqrt(self.dt), seed_far, _ptr(MF), half, st))
                    return 0
                except Exception as e:      # (an exception must not cross the C frames)
                    failed.append(e)
                    return -1
            return _lib.INTERLEAVE_FN(queue)
        # the far field in two halves around the check: spreading and forward transforms while the host answers it (the GPU idled in
        # the kernel that waits for the answer), the rest while the host reacts to its outcome
        self._interleave_cb = (far_half(1), far_half(2))
        check(self.lib.uammd_pse_near_set_interleave_early(self.near, C.cast(self._interleave_cb[0], C.c_void_p), None))
        check(self.lib.uammd_pse_near_set_interleave(self.near, C.cast(self._interleave_cb[1], C.c_void_p), None))
        it = C.c_int(0)
        # (the near field's M F rides on the solve's first product: the pair records are streamed once for the noise vector and F)
        check(self.lib.uammd_pse_near_set_mdot_rider(self.near, _ptr(force), _ptr(MF)))
        rc = self.lib.uammd_pse_near_stochastic(self.near, _ptr(pos), pd.N, float(self.temperature), 1.0, seed_near, _ptr(BdW), st,
                                                C.byref(it))
        self._interleave_cb = None      # (one-shot; the closure holds self)
        if failed:
            raise failed[0]
        check(rc)
        self.lastLanczosIterations = int(it.value)

    def computeHydrodynamicDisplacements(self, force, MF, temperature, noise_prefactor):
        """BDHI_PSE.cuh:135-155, statement for statement: with forces AND T > 0 the Lanczos result overwrites the near
        deterministic term (reference behaviour, kept)."""
        pd = self.pd
        MF.zero_()
        check(self.lib.uammd_pse_near_mdot(self.near, _ptr(pd.getPos("read")), _ptr(force), pd.N, _ptr(MF), current_stream()))
        self._near_stochastic(MF, temperature, noise_prefactor)
        self._far(force, MF, temperature, noise_prefactor)


class _BDHI2D(Integrator):
    """BDHI2D<HydroKernel> (Integrator/Hydro/BDHI_quasi2D.cuh:155-257): forwardTime = reset, interactors, spread, Fourier-space
    Green's function and noise, gather, Euler update — for particles confined to z = 0."""
    kernel = None

    class Parameters(_Parameters):
        def __init__(self, cells=(-1, -1), **kw):
            super().__init__(**kw)
            self.cells = list(cells)

    def __init__(self, pd, par):
        super().__init__(pd)
        from ._lib import BDHI2DParameters
        p = BDHI2DParameters()
        L = par.box.boxSize
        p.boxSize[0], p.boxSize[1] = float(L[0]), float(L[1])
        p.hydrodynamicRadius, p.viscosity, p.temperature, p.dt = (float(par.hydrodynamicRadius), float(par.viscosity),
                                                                  float(par.temperature), float(par.dt))
        p.cells[0], p.cells[1] = int(par.cells[0]), int(par.cells[1])
        self.seed = par.seed if par.seed else pd.rng.next32()          # seed = sys->rng().next32(), .cu:28
        p.seed, p.kernel = int(self.seed) & 0xFFFFFFFF, int(self.kernel)
        h, cells, sup = C.c_void_p(), (C.c_int * 2)(), C.c_int(0)
        try:
            check(self.lib.uammd_bdhi2d_create(C.byref(p), C.byref(h), C.byref(cells), C.byref(sup)))
        except _lib.UammdHipError as e:
            if "Invalid" in str(e):       # std::runtime_error("Invalid box" / "Invalid hydrodynamic radius"), .cu:46-57
                raise RuntimeError(str(e)) from e
            raise
        self.h, self.cells, self.support = h, [cells[0], cells[1]], int(sup.value)
        self.par = par
        self.box = Box([float(L[0]), float(L[1]), 0.0])
        self._vel = torch.zeros((pd.N, 2), dtype=torch.float32, device=pd.device)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.uammd_bdhi2d_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def forwardTime(self):
        pd, par = self.pd, self.par
        for it in self.interactors:
            it.updateSimulationTime(self.steps * par.dt)
        self.steps += 1
        if self.steps == 1:
            for it in self.interactors:
                it.updateTemperature(par.temperature)
                it.updateBox(self.box)
                it.updateTimeStep(par.dt)
        pd.getForce("write").zero_()
        for it in self.interactors:
            it.sum(force=True)
        force = _ptr(pd.getForce("read")) if self.interactors else None
        check(self.lib.uammd_bdhi2d_velocities(self.h, _ptr(pd.getPos("read")), force, pd.N, _ptr(self._vel), current_stream()))
        check(self.lib.uammd_bdhi2d_update_positions(_ptr(pd.getPos("readwrite")), _ptr(self._vel), pd.N, float(par.dt), current_stream()))


class True2D(_BDHI2D):
    """BDHI::True2D = BDHI2D<BDHI2D_ns::True2D> (BDHI_quasi2D.cuh:252)."""
    kernel = 0


class Quasi2D(_BDHI2D):
    """BDHI::Quasi2D = BDHI2D<BDHI2D_ns::Quasi2D> (BDHI_quasi2D.cuh:253)."""
    kernel = 1


class FIB(Integrator):
    """BDHI::FIB (Integrator/BDHI/FIB/FIB.cuh:131-236): fluctuating Stokes on a staggered grid, Peskin 3-point window, midpoint
    scheme.  As in the reference, Scheme.IMPROVED_MIDPOINT runs the simple midpoint scheme too (FIB.cu:1072-1079)."""
    MIDPOINT, IMPROVED_MIDPOINT = 0, 1

    class Parameters:
        def __init__(self, temperature=0.0, viscosity=1.0, hydrodynamicRadius=-1.0, dt=0.0, box=None, cells=(-1, -1, -1), scheme=1,
                     tolerance=1e-5, seed=0):
            self.temperature, self.viscosity, self.hydrodynamicRadius, self.dt = temperature, viscosity, hydrodynamicRadius, dt
            self.box, self.cells, self.scheme, self.tolerance, self.seed = box, list(cells), scheme, tolerance, seed

    def __init__(self, pd, par):
        super().__init__(pd)
        from ._lib import FIBParameters
        p = FIBParameters()
        for k in range(3):
            p.boxSize[k] = float(par.box.boxSize[k])
            p.cells[k] = int(par.cells[k])
        p.temperature, p.viscosity, p.hydrodynamicRadius, p.dt = (float(par.temperature), float(par.viscosity),
                                                                  float(par.hydrodynamicRadius), float(par.dt))
        p.scheme = int(par.scheme)
        p.seed = int(par.seed if par.seed else pd.rng.next32()) & 0xFFFFFFFF
        h, cells, rh = C.c_void_p(), (C.c_int * 3)(), C.c_float(0)
        try:
            check(self.lib.uammd_fib_create(C.byref(p), C.byref(h), C.byref(cells), C.byref(rh)))
        except _lib.UammdHipError as e:
            if "FIB]" in str(e):      # System::CRITICAL in the reference (FIB.cu:95-103)
                raise RuntimeError(str(e)) from e
            raise
        self.h, self.cells, self.hydrodynamicRadius = h, [int(c) for c in cells], float(rh.value)
        self.par, self.box = par, par.box

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.uammd_fib_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def getHydrodynamicRadius(self):
        return self.hydrodynamicRadius

    def getCellSize(self):
        return float(np.float32(self.box.boxSize[0]) / np.float32(self.cells[0]))

    def getSelfMobility(self):
        return float(self.lib.uammd_fib_self_mobility(self.hydrodynamicRadius, float(self.par.viscosity), float(self.box.boxSize[0])))

    def set_noise(self, random):
        """Test hook: device float[6, ncells] used as the fluid random numbers of the following steps (None: Saru)."""
        self._noise = random
        check(self.lib.uammd_fib_set_noise(self.h, _ptr(random) if random is not None else None))

    def forwardTime(self):
        pd, par = self.pd, self.par
        self.steps += 1
        if self.steps == 1:
            for it in self.interactors:
                it.updateSimulationTime(0)
                it.updateTimeStep(par.dt)
                it.updateTemperature(par.temperature)
                it.updateBox(self.box)
        pd.getForce("write").zero_()
        for it in self.interactors:
            it.sum(force=True)
        check(self.lib.uammd_fib_forward(self.h, _ptr(pd.getPos("readwrite")), _ptr(pd.getForce("read")), pd.N, current_stream()))
        for it in self.interactors:
            it.updateSimulationTime(self.steps * par.dt)


class Cholesky:
    """BDHI::Cholesky — the Method concept of BDHI::EulerMaruyama with a dense mobility matrix and its Cholesky factor
    (Integrator/BDHI/BDHI_Cholesky.cuh:37-80, .cu:83-262).  noise_fn() -> float[3N] N(0,1) replaces the reference's cuRAND
    generator (third party, stream unpinned); default torch.randn on the particles' device."""
    Parameters = _Parameters

    def __init__(self, pd, par, noise_fn=None):
        self.lib = _lib.load()
        self.pd, self.par = pd, par
        if par.hydrodynamicRadius < 0 and not pd.isAllocated("radius"):
            raise RuntimeError("[BDHI::Cholesky] You need to provide Cholesky with either an hydrodynamic radius or via the "
                               "individual particle radius.")   # BDHI_Cholesky.cu:98-101
        h = C.c_void_p()
        check(self.lib.uammd_bdhi_cholesky_create(pd.N, float(par.viscosity), float(par.hydrodynamicRadius), C.byref(h)))
        self.h = h
        self.temperature, self.dt = par.temperature, par.dt
        self.noise_fn = noise_fn or (lambda: torch.randn(3 * pd.N, dtype=torch.float32, device=pd.device))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.uammd_bdhi_cholesky_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _radius(self):
        return _ptr(self.pd.getRadius("read")) if self.pd.isAllocated("radius") else None

    def setup_step(self):
        check(self.lib.uammd_bdhi_cholesky_setup_step(self.h, _ptr(self.pd.getPos("read")), None, self._radius(), current_stream()))

    def computeMF(self, MF):
        check(self.lib.uammd_bdhi_cholesky_mf(self.h, _ptr(self.pd.getPos("read")), _ptr(self.pd.getForce("read")), None,
                                              self._radius(), _ptr(MF), current_stream()))

    def computeBdW(self, BdW):
        BdW.view(-1)[:3 * self.pd.N].copy_(self.noise_fn())
        check(self.lib.uammd_bdhi_cholesky_bdw(self.h, _ptr(self.pd.getPos("read")), None, self._radius(), _ptr(BdW), current_stream()))

    def finish_step(self):
        pass

    def getHydrodynamicRadius(self):
        return self.par.hydrodynamicRadius

    def getSelfMobility(self):
        rh = self.par.hydrodynamicRadius
        return -1.0 if rh < 0 else 1.0 / (6.0 * math.pi * self.par.viscosity * rh)


class EulerMaruyama(Integrator):
    """BDHI::EulerMaruyama<Method> (BDHI_EulerMaruyama.cu:125-166): dR = dt (K R + M F) + sqrt(2 T dt) B dW."""

    def __init__(self, pd, par, method=None, Method=None):
        super().__init__(pd)
        self.par = par
        self.bdhi = method if method is not None else Method(pd, par)
        self.MF = torch.zeros((pd.N, 3), dtype=torch.float32, device=pd.device)
        self.BdW = torch.zeros((pd.N + 1, 3), dtype=torch.float32, device=pd.device)
        K = getattr(par, "K", None)
        self.K = None if not K else (C.c_float * 9)(*[float(x) for row in K for x in row])
        self.is2D = bool(getattr(par, "is2D", False))

    def forwardTime(self):
        pd, par = self.pd, self.par
        self.steps += 1
        for it in self.interactors:
            it.updateSimulationTime(self.steps * par.dt)
        if self.steps == 1:
            for it in self.interactors:
                it.updateTimeStep(par.dt)
                it.updateTemperature(par.temperature)
                it.updateBox(par.box)
        pd.getForce("write").zero_()
        for it in self.interactors:
            it.sum(force=True)
        self.bdhi.setup_step()
        if par.temperature > 0 and hasattr(self.bdhi, "computeMFandBdW"):   # (a method that interleaves the two: see PSE.computeMFandBdW)
            self.bdhi.computeMFandBdW(self.MF, self.BdW)
        else:
            self.bdhi.computeMF(self.MF)
            if par.temperature > 0:
                self.bdhi.computeBdW(self.BdW)
        sqrt2Tdt = math.sqrt(2 * par.dt * par.temperature)
        self.bdhi.finish_step()
        check(self.lib.uammd_bdhi_euler_maruyama(_ptr(pd.getPos("readwrite")), None, _ptr(self.MF),
                                                 _ptr(self.BdW) if par.temperature > 0 else None, self.K, pd.N, sqrt2Tdt,
                                                 float(par.dt), int(self.is2D), current_stream()))


class LanczosSolver:
    """lanczos::Solver (misc/LanczosAlgorithm.cuh:32-83).  `dot(v, Mv)` is a Python callable on torch tensors that
    writes Mv = M v (the MatrixDot concept, LanczosAlgorithm/MatrixDot.h)."""

    def __init__(self):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.uammd_lanczos_create(C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.uammd_lanczos_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def setIterationHardLimit(self, n):
        check(self.lib.uammd_lanczos_set_iteration_hard_limit(self.h, int(n)))

    def setOption(self, name, value):
        """"defer_checks" (1): the convergence checks before the iteration the previous run stopped at are evaluated together."""
        check(self.lib.uammd_lanczos_set_option(self.h, name.encode(), int(value)))

    def setAllReduce(self, group=None, owns_first_element=None, enabled=True):
        """Sharded vectors (one slice per rank): the dot products of the recurrence are summed over `group` with
        torch.distributed.all_reduce (RCCL on the GPUs; host staging under gloo, which cannot move device memory)."""
        import torch.distributed as dist
        if not enabled:
            check(self.lib.uammd_lanczos_set_allreduce(self.h, None, None, 1))
            self._reduce_cb = None
            return
        staged = dist.get_backend(group) == "gloo"

        def _wrap(ptr, count):
            class _Raw:
                pass
            r = _Raw()
            r.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
            return torch.as_tensor(r, device="cuda")

        def cb(ctx, d_values, count, stream):
            try:
                t = _wrap(d_values, count)
                if staged:
                    h = t.cpu()
                    dist.all_reduce(h, group=group)
                    t.copy_(h)
                else:
                    dist.all_reduce(t, group=group)
                return 0
            except Exception:  # surfaced as a library error
                return -98
        self._reduce_cb = _lib.ALLREDUCE_FN(cb)     # keep the trampoline alive
        first = (dist.get_rank(group) == 0) if owns_first_element is None else bool(owns_first_element)
        check(self.lib.uammd_lanczos_set_allreduce(self.h, C.cast(self._reduce_cb, C.c_void_p), None, int(first)))

    def getLastRunRequiredSteps(self):
        v = C.c_int(0)
        check(self.lib.uammd_lanczos_get_last_run_required_steps(self.h, C.byref(v)))
        return int(v.value)

    def run(self, dot, Bv, v, tolerance, N=None):
        n = int(N if N is not None else v.numel())
        dev = v.device

        def _wrap(ptr, count):
            class _Raw:
                pass
            r = _Raw()
            r.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
            return torch.as_tensor(r, device=dev)
        err = []

        def cb(ctx, d_v, d_Mv, nn, stream):
            try:
                dot(_wrap(d_v, nn), _wrap(d_Mv, nn))
                return 0
            except Exception as e:  # surfaced after the C call returns
                err.append(e)
                return -99
        fn = _lib.MATVEC_FN(cb)
        it = C.c_int(0)
        rc = self.lib.uammd_lanczos_run(self.h, C.cast(fn, C.c_void_p), None, _ptr(Bv), _ptr(v), float(tolerance), n,
                                        current_stream(), C.byref(it))
        if err:
            raise err[0]
        check(rc)
        return int(it.value)


class Lanczos:
    """BDHI::Lanczos (Integrator/BDHI/BDHI_Lanczos.cuh:20-67, .cu): open-boundary RPY mobility applied matrix free as an
    all-pairs product, noise by the Lanczos iteration.  The Method concept of BDHI::EulerMaruyama."""
    Parameters = _Parameters

    def __init__(self, pd, par):
        self.lib = _lib.load()
        self.pd, self.par = pd, par
        self.hydrodynamicRadius, self.temperature, self.tolerance = par.hydrodynamicRadius, par.temperature, par.tolerance
        if par.hydrodynamicRadius < 0 and not pd.isAllocated("radius"):
            raise RuntimeError("[BDHI::Lanczos] You need to provide Lanczos with either an hydrodynamic radius or via the "
                               "individual particle radius.")
        self.solver = LanczosSolver()
        # the reference seeds cuRAND from System::rng().next() here (BDHI_Lanczos.cu:45-46); torch's generator stands in
        self.gen = torch.Generator(device=pd.device)
        self.gen.manual_seed(pd.rng.next() & 0x7FFFFFFFFFFFFFFF)
        self.lastIterations = 0

    def _radius(self):
        return None if self.hydrodynamicRadius > 0 else self.pd.getRadius("read")

    def setup_step(self):
        pass

    def finish_step(self):
        pass

    def getHydrodynamicRadius(self):
        return self.par.hydrodynamicRadius

    def getSelfMobility(self):
        rh = self.par.hydrodynamicRadius
        return -1.0 if rh < 0 else 1.0 / (6.0 * math.pi * self.par.viscosity * rh)

    def computeMF(self, MF):
        pd = self.pd
        check(self.lib.uammd_rpy_nbody_mdot(_ptr(pd.getPos("read")), _ptr(pd.getForce("read")), 4, _ptr(self._radius()),
                                            float(self.hydrodynamicRadius), float(self.par.viscosity), pd.N, _ptr(MF),
                                            current_stream()))

    def computeBdW(self, BdW, noise=None):
        if not self.temperature > 0:
            return
        pd = self.pd
        if noise is None:
            noise = torch.randn((pd.N, 3), dtype=torch.float32, device=pd.device, generator=self.gen)
        it = C.c_int(0)
        check(self.lib.uammd_rpy_lanczos_bdw(self.solver.h, _ptr(pd.getPos("read")), _ptr(self._radius()),
                                             float(self.hydrodynamicRadius), float(self.par.viscosity), pd.N, _ptr(noise),
                                             float(self.tolerance), _ptr(BdW), current_stream(), C.byref(it)))
        self.lastIterations = int(it.value)


class BDHI:
    LanczosSolver = LanczosSolver
    FCM = FCM
    PSE = PSE
    Lanczos = Lanczos
    Cholesky = Cholesky
    True2D = True2D
    FIB = FIB
    Quasi2D = Quasi2D
    EulerMaruyama = EulerMaruyama
    FCMIntegrator = FCMIntegrator
    FCM_impl = FCM_impl
    Kernels = Kernels
