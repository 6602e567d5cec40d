"""Host-side mirror of the reference's triply periodic electrostatics Interactor (Interactor/SpectralEwaldPoisson.cuh):
class names, parameters, error behaviour.  Every call goes through the C ABI (uammd_poisson_*); no CPU fallback."""
import ctypes as C

import torch

from . import _lib
from ._lib import PoissonInfo, PoissonParameters, check
from .md import Interactor, _ptr, current_stream


class Poisson(Interactor):
    """Poisson(pd, par) — SpectralEwaldPoisson.cuh:83-136."""

    class Parameters:
        """Poisson::Parameters (SpectralEwaldPoisson.cuh:94-103).  cells and support exist in the reference's struct but its
        constructor never reads them; they are accepted and ignored here too."""

        def __init__(self, box=None, epsilon=-1.0, tolerance=1e-5, gw=-1.0, split=-1.0, upsampling=-1.0, cells=(-1, -1, -1),
                     support=-1):
            self.box, self.epsilon, self.tolerance, self.gw, self.split = box, epsilon, tolerance, gw, split
            self.upsampling, self.cells, self.support = upsampling, cells, support

    def __init__(self, pd, par):
        self.lib = _lib.load()
        self.pd = pd
        p = PoissonParameters()
        for k in range(3):
            p.boxSize[k] = float(par.box.boxSize[k])
        p.epsilon, p.tolerance, p.gw, p.split, p.upsampling = (float(par.epsilon), float(par.tolerance), float(par.gw),
                                                                float(par.split), float(par.upsampling))
        h, info = C.c_void_p(), PoissonInfo()
        try:
            check(self.lib.uammd_poisson_create(C.byref(p), C.byref(h), C.byref(info)))
        except _lib.UammdHipError as e:
            if "[Poisson]" in str(e):   # std::invalid_argument in the reference (.cu:95-102, :111-116)
                raise ValueError(str(e)) from e
            raise
        self.h = h
        self.box, self.epsilon, self.split, self.gw, self.tolerance = par.box, par.epsilon, par.split, par.gw, par.tolerance
        self.cells, self.support = [int(c) for c in info.cells], int(info.support)
        self.nearFieldCutOff, self.nTable, self.cellSize = float(info.nearFieldCutOff), int(info.nTable), float(info.h)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.uammd_poisson_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_option(self, name, value):
        check(self.lib.uammd_poisson_set_option(self.h, name.encode(), int(value)))

    def sum(self, force=False, energy=False, virial=False):
        """Poisson::sum (SpectralEwaldPoisson.cuh:110-123): far field (adds to force AND energy), then the near-field passes."""
        if virial:
            raise RuntimeError("[Poisson] not implemented")
        pd = self.pd
        check(self.lib.uammd_poisson_sum(self.h, _ptr(pd.getPos("read")), _ptr(pd.getCharge("read")), pd.N,
                                         _ptr(pd.getForce("readwrite")), _ptr(pd.getEnergy("readwrite")), int(bool(force)),
                                         int(bool(energy)), current_stream()))

    def computeFieldPotentialAtParticles(self):
        """-> real4[N] (Ex, Ey, Ez, phi); like the reference's call it also adds the far field to pd force/energy."""
        pd = self.pd
        out = torch.zeros((pd.N, 4), dtype=torch.float32, device=pd.device)
        check(self.lib.uammd_poisson_field_potential(self.h, _ptr(pd.getPos("read")), _ptr(pd.getCharge("read")), pd.N, _ptr(out),
                                                     _ptr(pd.getForce("readwrite")), _ptr(pd.getEnergy("readwrite")),
                                                     current_stream()))
        return out
