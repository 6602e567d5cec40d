"""Host-side mirror of the reference's Hydro namespace (Integrator/Hydro/ICM.cuh): class names, parameters, error behaviour.
Every call goes through the C ABI (uammd_icm_*); no CPU fallback."""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from ._lib import ICMParameters, check
from .md import Integrator, _ptr, current_stream


class ICM(Integrator):
    """Hydro::ICM(pd, par) — ICM.cuh:123-231: inertial coupling of the particles to an incompressible fluctuating fluid."""

    class Parameters:
        def __init__(self, temperature=0.0, viscosity=-1.0, density=-1.0, hydrodynamicRadius=-1.0, dt=0.0, box=None, cells=(-1, -1, -1),
                     sumThermalDrift=False, removeTotalMomentum=True, seed=0):
            self.temperature, self.viscosity, self.density, self.hydrodynamicRadius = temperature, viscosity, density, hydrodynamicRadius
            self.dt, self.box, self.cells = dt, box, list(cells)
            self.sumThermalDrift, self.removeTotalMomentum, self.seed = sumThermalDrift, removeTotalMomentum, seed

    def __init__(self, pd, par):
        super().__init__(pd)
        p = ICMParameters()
        for k in range(3):
            p.boxSize[k] = float(par.box.boxSize[k])
            p.cells[k] = int(par.cells[k])
        p.temperature, p.viscosity, p.density, p.hydrodynamicRadius, p.dt = (float(par.temperature), float(par.viscosity),
                                                                             float(par.density), float(par.hydrodynamicRadius), float(par.dt))
        p.sumThermalDrift, p.removeTotalMomentum = int(bool(par.sumThermalDrift)), int(bool(par.removeTotalMomentum))
        p.seed = int(par.seed if par.seed else pd.rng.next32()) & 0xFFFFFFFF       # seed = sys->rng().next32(), ICM.cu:831
        h, cells, rh = C.c_void_p(), (C.c_int * 3)(), C.c_float(0)
        try:
            check(self.lib.uammd_icm_create(C.byref(p), C.byref(h), C.byref(cells), C.byref(rh)))
        except _lib.UammdHipError as e:
            if "ICM]" in str(e):          # System::CRITICAL in the reference (ICM.cu:833-839, :869-872)
                raise RuntimeError(str(e)) from e
            raise
        self.h, self.cells, self.hydrodynamicRadius = h, [int(c) for c in cells], float(rh.value)
        self.par, self.box = par, par.box

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.uammd_icm_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def getHydrodynamicRadius(self):
        return self.hydrodynamicRadius

    def getSelfMobility(self):       # ICM.cuh:164-168
        rh = self.hydrodynamicRadius
        return 1.0 / (6 * math.pi * self.par.viscosity * rh) * (1 - 2.837297 * rh / float(self.box.boxSize[0]))

    def getNumberFluidCells(self):
        return list(self.cells)

    def getFluidVelocities(self, collocated=True):
        """Cell-centred fluid velocities real3[nz, ny, nx] (ICM::getFluidVelocities); collocated=False: the staggered field."""
        nx, ny, nz = self.cells
        out = torch.empty((nz, ny, nx, 3), dtype=torch.float32, device=self.pd.device)
        check(self.lib.uammd_icm_get_fluid_velocity(self.h, _ptr(out), int(collocated), current_stream()))
        return out

    def setFluidVelocities(self, v):
        check(self.lib.uammd_icm_set_fluid_velocity(self.h, _ptr(v.contiguous()), current_stream()))

    def set_noise(self, random):
        self._noise = random
        check(self.lib.uammd_icm_set_noise(self.h, _ptr(random) if random is not None else None))

    def forwardTime(self):
        pd, par = self.pd, self.par
        self.steps += 1
        if self.steps == 1:
            for it in self.interactors:
                it.updateTemperature(par.temperature)
                it.updateTimeStep(par.dt)
                it.updateBox(self.box)
                it.updateSimulationTime(0)
            for it in self.interactors:      # ICM.cu:1201-1203 (these forces are overwritten before they are used)
                it.sum(force=True)
        check(self.lib.uammd_icm_predictor(self.h, _ptr(pd.getPos("readwrite")), pd.N, current_stream()))
        for it in self.interactors:
            it.updateSimulationTime((self.steps - 0.5) * par.dt)
        force = None
        if self.interactors:                 # spreadParticleForces, :1040-1066: forces at q^{n+1/2}
            pd.getForce("write").zero_()
            for it in self.interactors:
                it.sum(force=True)
            force = _ptr(pd.getForce("read"))
        check(self.lib.uammd_icm_fluid_and_corrector(self.h, _ptr(pd.getPos("readwrite")), force, pd.N, current_stream()))
        pd.getForce("write").zero_()         # correctorStep resets the forces (:1176-1181)
        for it in self.interactors:
            it.updateSimulationTime(self.steps * par.dt)


class Hydro:
    ICM = ICM
