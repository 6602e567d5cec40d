"""Host-side mirror of the UAMMD interfaces on path A (short-range pair forces + integrators).

Same names, argument meaning and error behaviour as the reference classes; all compute happens in
libuammd_hip.so.  Citations (relative to /root/reference/src):
  Box                          utils/Box.cuh:16-58
  ParticleData                 ParticleData/ParticleData.cuh:161-465
  ParticleGroup                ParticleData/ParticleGroup.cuh:60-379
  CellList                     Interactor/NeighbourList/CellList.cuh:83-205
  VerletList                   Interactor/NeighbourList/VerletList.cuh:83-201
  Potential.LJ                 Interactor/Potential/Potential.cuh:25-85, RadialPotential.cuh:49-154
  PairForces                   Interactor/PairForces.cuh:23-64, PairForces.cu:43-78
  Integrator / Interactor      Integrator/Integrator.cuh:33-125, Interactor/Interactor.cuh:56-119
  VerletNVT.{Basic,GronbechJensen}  Integrator/VerletNVT.cuh, VerletNVT/Basic.cu, GronbechJensen.cu
  BD.EulerMaruyama, MidPoint, AdamsBashforth, Leimkuhler   Integrator/BrownianDynamics.cuh/.cu
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from ._lib import CellListData, LJPairParameters, VerletListData, check, f3, i3


def current_stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class Xorshift128plus:
    """System::rng() (utils/utils.h:38-115): seeds for the integrators come from next32()."""

    def __init__(self, s0=12679825035178159220, s1=15438657923749336752):
        self.s = [s0 & 0xFFFFFFFFFFFFFFFF, s1 & 0xFFFFFFFFFFFFFFFF]

    def set_seed(self, s0):
        m = 0xFFFFFFFFFFFFFFFF
        self.s = [s0 & m, ((s0 + 15438657923749336752) & m) % m]      # uint64 wrap first (utils/utils.h:110-113)

    def next(self):
        m = 0xFFFFFFFFFFFFFFFF
        x, y = self.s
        self.s[0] = y
        x ^= (x << 23) & m
        x ^= x >> 17
        x ^= y ^ (y >> 26)
        self.s[1] = x
        return (x + y) & m

    def next32(self):
        return self.next() % 0xFFFFFFFF


class Box:
    def __init__(self, L, periodic=(True, True, True)):
        L = np.broadcast_to(np.asarray(L, dtype=np.float32), (3,)).copy()
        self.boxSize = L
        self.periodic = [bool(p) and not (l == 0 or np.isinf(l)) for p, l in zip(np.broadcast_to(periodic, (3,)), L)]

    def setPeriodicity(self, x, y, z):
        self.periodic = [self.periodic[0] and bool(x), self.periodic[1] and bool(y), self.periodic[2] and bool(z)]

    def __eq__(self, o):
        return isinstance(o, Box) and np.array_equal(self.boxSize, o.boxSize) and list(self.periodic) == list(o.periodic)


class ParticleData:
    """Particle properties in HBM.  pos/force are real4 (xyz + type / unused), vel is real3."""

    def __init__(self, numberParticles, device="cuda", seed=0xf31337Bada55D00d):
        self.N = int(numberParticles)
        self.device = torch.device(device)
        self._props = {}
        self.rng = Xorshift128plus()
        self.rng.set_seed(seed)
        self._pos_write_callbacks = []
        self._reorder_callbacks = []
        self.id = torch.arange(self.N, dtype=torch.int32, device=self.device)

    def getNumParticles(self):
        return self.N

    def _get(self, name, width, dtype=torch.float32):
        if name not in self._props:
            shape = (self.N, width) if width > 1 else (self.N,)
            self._props[name] = torch.zeros(shape, dtype=dtype, device=self.device)
        return self._props[name]

    def isAllocated(self, name):
        return name in self._props

    def getPos(self, mode="read"):
        if mode != "read":
            self._emit(self._pos_write_callbacks)  # ParticleData::getPosWriteRequestedSignal
        return self._get("pos", 4)

    def getForce(self, mode="read"):
        return self._get("force", 4)

    def getVel(self, mode="read"):
        return self._get("vel", 3)

    def getTorque(self, mode="read"):
        return self._get("torque", 4)

    def getDir(self, mode="read"):
        """Orientation quaternions real4 (n, vx, vy, vz), initialised to the identity (ParticleData.cuh: dir)."""
        if "dir" not in self._props:
            d = self._get("dir", 4)
            d[:, 0] = 1.0
        return self._props["dir"]

    def getTorqueIfAllocated(self, mode="read"):
        return self._props.get("torque")

    def getDirIfAllocated(self, mode="read"):
        return self._props.get("dir")

    def isDirAllocated(self):
        return "dir" in self._props

    def isTorqueAllocated(self):
        return "torque" in self._props

    def getEnergy(self, mode="read"):
        return self._get("energy", 1)

    def getVirial(self, mode="read"):
        return self._get("virial", 1)

    def getMass(self, mode="read"):
        return self._get("mass", 1)

    def getCharge(self, mode="read"):
        return self._get("charge", 1)

    def getRadius(self, mode="read"):
        return self._get("radius", 1)

    def setPos(self, pos4):
        p = self.getPos("write")
        p.copy_(torch.as_tensor(np.ascontiguousarray(pos4, dtype=np.float32)).to(self.device))

    @staticmethod
    def _slot(cb):
        """A connection must not keep its receiver alive (the reference's receivers hold connection objects that disconnect with them):
        a bound method is held weakly — ParticleData -> callback -> solver -> ParticleData would otherwise be a cycle, and the solver's
        handle (device memory, streams, events) would live until the cyclic collector happens to run."""
        import types
        import weakref
        return weakref.WeakMethod(cb) if isinstance(cb, types.MethodType) else (lambda: cb)

    @staticmethod
    def _emit(slots):
        dead = False
        for s in list(slots):
            cb = s()
            if cb is None:
                dead = True
            else:
                cb()
        if dead:
            slots[:] = [s for s in slots if s() is not None]

    def connectPosWrite(self, cb):
        self._pos_write_callbacks.append(self._slot(cb))

    def connectReorder(self, cb):
        """ParticleData::getReorderSignal (ParticleData.cuh:196-208)."""
        self._reorder_callbacks.append(self._slot(cb))

    def hintSortByHash(self, hash_box, hash_cutOff):
        """ParticleData::hintSortByHash (ParticleData.cuh:389-394)."""
        self._hint_box, self._hint_cutoff = hash_box, np.broadcast_to(np.asarray(hash_cutOff, np.float32), (3,))

    def sortParticles(self):
        """ParticleData::sortParticles (ParticleData.cuh:492-522): reorder every allocated property by the Morton
        hash of the hint grid (default Box(128), cutOff 10: ParticleData.cuh:164-169) and emit the reorder signal."""
        lib = _lib.load()
        pos = self._get("pos", 4)
        box = getattr(self, "_hint_box", None) or Box(128.0)
        rc = getattr(self, "_hint_cutoff", np.full(3, 10.0, np.float32))
        cd = [int(np.float32(l) / np.float32(c)) for l, c in zip(box.boxSize, rc)]
        if cd[2] == 0:
            cd[2] = 1
        cl = getattr(self, "_sorter", None)   # kept: a fresh cell list per call costs ~10 ms of hipMalloc / hipFree
        if cl is None:
            cl = self._sorter = CellList()
        cl.update_grid(pos, box, cd)
        idx = cl.group_index()
        for name, t in list(self._props.items()):
            out = torch.empty_like(t)
            eb = t.element_size() * (t.shape[1] if t.dim() > 1 else 1)
            check(lib.uammd_gather(_ptr(t), _ptr(idx), _ptr(out), self.N, eb, current_stream()))
            self._props[name] = out
        out = torch.empty_like(self.id)
        check(lib.uammd_gather(_ptr(self.id), _ptr(idx), _ptr(out), self.N, 4, current_stream()))
        self.id = out
        self._emit(self._pos_write_callbacks)
        self._emit(self._reorder_callbacks)


class ParticleGroup:
    """ParticleGroup (ParticleData/ParticleGroup.cuh:170-379): a subset of the particles tracked by ID, so that it survives
    ParticleData::sortParticles.  selector: None ("All"), a callable (index, pos4 row, id) -> bool evaluated on the host at
    construction (particle_selector concept, :60-135), or an iterable of particle ids."""

    def __init__(self, pd, selector=None, name="noName"):
        self.pd, self.name = pd, name
        self.allParticles = selector is None
        if self.allParticles:
            self.ids = None
            self.numberParticles = pd.N
        else:
            if callable(selector):
                pos = pd.getPos("read").cpu().numpy()
                ids = pd.id.cpu().numpy()
                sel = [int(ids[i]) for i in range(pd.N) if selector(i, pos[i], int(ids[i]))]
            else:
                sel = [int(x) for x in selector]
            self.ids = torch.tensor(sorted(sel), dtype=torch.int64, device=pd.device)   # the reference keeps them id-ordered
            self.numberParticles = len(sel)
        self._index = None
        self._stale = True
        pd.connectReorder(self._handle_reorder)

    @staticmethod
    def IDRange(first, last):            # particle_selector::IDRange: ids in [first, last]
        return lambda i, p, pid: first <= pid <= last

    @staticmethod
    def Type(*types):                    # particle_selector::Type (pos.w)
        return lambda i, p, pid: int(p[3]) in types

    def _handle_reorder(self):
        self._stale = True

    def getNumberParticles(self):
        return self.numberParticles

    def getParticleData(self):
        return self.pd

    def getIndexIterator(self):
        """Current ParticleData indices of the members (int32[n] on the device); None for the "All" group = identity, as
        the C ABI takes it (d_index / d_globalIndex nullable)."""
        if self.allParticles:
            return None
        if self._stale:   # ParticleGroup_ns::updateGroupIndices (:140-153): index = id2index[id]
            id2index = torch.empty(self.pd.N, dtype=torch.int64, device=self.pd.device)
            id2index[self.pd.id.long()] = torch.arange(self.pd.N, device=self.pd.device)
            self._index = id2index[self.ids].to(torch.int32).contiguous()
            self._stale = False
        return self._index

    def getPropertyIterator(self, prop):
        """property[index[i]] for the members (a gathered copy; the "All" group returns the array itself)."""
        idx = self.getIndexIterator()
        return prop if idx is None else prop.index_select(0, idx.long())


class CellList:
    """NeighbourList concept: update(box, cutoff) + transverseList (LJ fast path) + getCellList."""

    def __init__(self, pd=None):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.uammd_celllist_create(C.byref(h)))
        self.h = h
        self.pd = pd
        self.force_next_update = True
        self.currentCutOff = None
        self.currentBox = None
        self.N = 0
        if pd is not None:
            pd.connectPosWrite(self._handle_pos_write)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.uammd_celllist_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _handle_pos_write(self):
        self.force_next_update = True

    def set_option(self, name, value):
        check(self.lib.uammd_celllist_set_option(self.h, name.encode(), int(value)))

    def check_errors(self):
        """uammd_celllist_check_errors: synchronises the stream and raises what the list's kernels flagged (NaN positions, particles
        outside a non-periodic box, a parameter table rewritten behind the list's cached copy) instead of waiting for the next update."""
        check(self.lib.uammd_celllist_check_errors(self.h, current_stream()))

    @staticmethod
    def create_update_grid(box, cutoff):
        lib = _lib.load()
        cd, Lo, po = i3(0), f3(0), i3(0)
        check(lib.uammd_celllist_create_grid(f3(box.boxSize), i3([int(p) for p in box.periodic]), f3(cutoff), cd, Lo, po))
        return list(cd), Box(list(Lo), [bool(p) for p in po])

    def needsRebuild(self, box, cutoff):
        if self.force_next_update:
            return True
        if self.currentCutOff is None or not np.array_equal(np.broadcast_to(cutoff, (3,)), self.currentCutOff):
            return True
        return not (box == self.currentBox)

    def update(self, box, cutoff, pos=None):
        """CellList::update(box, cutOff, st) (CellList.cuh:149-163)."""
        if pos is None:
            pos = self.pd.getPos("read")
        if self.needsRebuild(box, cutoff):
            self.currentBox = box
            self.currentCutOff = np.broadcast_to(np.asarray(cutoff, dtype=np.float32), (3,)).copy()
            cd, ubox = self.create_update_grid(box, cutoff)
            self.update_grid(pos, ubox, cd)
            self.force_next_update = False

    def fused_built(self, box, cutoff, pos):
        """The fused MD step built this list on `pos` as update(box, cutoff) would have: the lazy-update bookkeeping follows."""
        self.currentBox = box
        self.currentCutOff = np.broadcast_to(np.asarray(cutoff, dtype=np.float32), (3,)).copy()
        self.N = pos.shape[0]
        self._pos_ref = pos
        self.force_next_update = False

    def update_grid(self, pos, box, cellDim):
        """CellListBase::update(pos, N, grid, st) (CellListBase.cuh:124-140)."""
        assert pos.dtype == torch.float32 and pos.is_contiguous() and pos.shape[1] == 4
        self.N = pos.shape[0]
        self._pos_ref = pos
        check(self.lib.uammd_celllist_update(self.h, _ptr(pos), self.N, f3(box.boxSize),
                                             i3([int(p) for p in box.periodic]), i3(cellDim), current_stream()))

    def update_grid_gj1(self, pos, box, cellDim, vel, force, keys, skip, n_owned, dt, friction, is2D, noise, step_num, seed, mass=None,
                        default_mass=1.0):
        """uammd_celllist_update_gj1: update_grid over owned + ghost rows with GronbechJensen's first half step of the owned rows that
        are not marked in `skip` folded into the build's first kernel (keys: the global ids the thermostat's stream is keyed by)."""
        assert pos.dtype == torch.float32 and pos.is_contiguous() and pos.shape[1] == 4
        self.N = pos.shape[0]
        self._pos_ref = pos
        check(self.lib.uammd_celllist_update_gj1(self.h, _ptr(pos), self.N, f3(box.boxSize), i3([int(p) for p in box.periodic]),
                                                 i3(cellDim), _ptr(vel), _ptr(force), None if mass is None else _ptr(mass),
                                                 default_mass, _ptr(keys), None if skip is None else _ptr(skip), n_owned, dt, friction,
                                                 int(is2D), noise, step_num, seed, current_stream()))

    def getCellList(self):
        d = CellListData()
        check(self.lib.uammd_celllist_get(self.h, C.byref(d)))
        return d

    def _wrap(self, ptr, shape, dtype):
        n = int(np.prod(shape))
        if n == 0:
            return torch.empty(shape, dtype=dtype, device="cuda")
        # zero-copy view of library-owned device memory through __cuda_array_interface__
        class _Raw:
            pass
        r = _Raw()
        typestr = {torch.float32: "<f4", torch.int32: "<i4", torch.uint8: "|u1"}[dtype]
        r.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(r, device="cuda")

    def group_index(self):
        d = self.getCellList()
        return self._wrap(d.d_groupIndex, (d.numberParticles,), torch.int32)

    def to_host(self):
        """Copies the list to numpy (tests): hash, index, sortPos, cellStart, cellEnd, VALID_CELL."""
        d = self.getCellList()
        n = d.numberParticles
        nc = d.cellDim[0] * d.cellDim[1] * d.cellDim[2]
        torch.cuda.synchronize()
        out = dict(
            index=self._wrap(d.d_groupIndex, (n,), torch.int32).cpu().numpy(),
            hash=self._wrap(d.d_sortHash, (n,), torch.int32).cpu().numpy().view(np.uint32),
            sortPos=self._wrap(d.d_sortPos, (n, 4), torch.float32).cpu().numpy(),
            cellStart=self._wrap(d.d_cellStart, (nc,), torch.int32).cpu().numpy().view(np.uint32),
            cellEnd=self._wrap(d.d_cellEnd, (nc,), torch.int32).cpu().numpy(),
            validCell=int(d.VALID_CELL), cellDim=np.array(list(d.cellDim), np.int32),
            L=np.array(list(d.boxSize), np.float32), periodic=np.array(list(d.periodic), np.int32))
        return out

    def profile_enable(self, on=True):
        """Measurement hook: per-launch kernel time of the LJ traversal (uammd_lj_profile_enable, include/uammd_hip.h)."""
        check(self.lib.uammd_lj_profile_enable(self.h, int(bool(on))))

    def tile_stats(self, enable=True):
        """Measurement hook (uammd_lj_tile_stats): {'fallback_bricks', 'bricks'} counted since the last call; then on / off."""
        out = (C.c_uint * 16)()
        check(self.lib.uammd_lj_tile_stats(self.h, int(bool(enable)), out, current_stream()))
        return {"fallback_bricks": int(out[0]), "bricks": int(out[1]), "timeline": [int(x) for x in out[4:]]}

    def profile_read(self):
        """(summed kernel ms, launches) since profile_enable; waits for launches in flight."""
        ms, n = C.c_double(0.0), C.c_longlong(0)
        check(self.lib.uammd_lj_profile_read(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def transverse_lj(self, param_table, ntypes, box, force=None, energy=None, virial=None, global_index=None,
                      algo=0):
        check(self.lib.uammd_lj_transverse_celllist(self.h, _ptr(param_table), int(ntypes), f3(box.boxSize),
                                                    i3([int(p) for p in box.periodic]), _ptr(force), _ptr(energy),
                                                    _ptr(virial), _ptr(global_index), int(algo), current_stream()))


    def transverse_lj_gj2(self, param_table, ntypes, box, force, vel, dt, mass=None, default_mass=1.0, is2D=False, algo=0):
        """The LJ forces of the owned rows (option num_owned) with VerletNVT::GronbechJensen's second half step applied in the traversal's
        store (uammd_lj_transverse_celllist_gj2): `force` must be zero on the owned rows on entry."""
        check(self.lib.uammd_lj_transverse_celllist_gj2(self.h, _ptr(param_table), int(ntypes), f3(box.boxSize),
                                                        i3([int(p) for p in box.periodic]), _ptr(force), _ptr(vel), _ptr(mass),
                                                        float(0.0 if mass is not None else default_mass), float(dt), int(bool(is2D)), int(algo),
                                                        current_stream()))


class VerletList:
    """VerletList (Interactor/NeighbourList/VerletList.cuh:83-201): the NeighbourList concept on an explicit list that is
    only rebuilt when a particle has drifted (1.08 rc - rc)/2 from where it was at the last build."""

    def __init__(self, pd=None):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.uammd_verletlist_create(C.byref(h)))
        self.h = h
        self.pd = pd
        self.force_next_update = True
        self.currentCutOff = None
        self.currentBox = None
        self.rebuilds = 0
        if pd is not None:
            pd.connectPosWrite(self._handle_pos_write)
            pd.connectReorder(self._handle_reorder)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.uammd_verletlist_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _handle_pos_write(self):            # VerletList.cuh:171-175
        self.force_next_update = True

    def _handle_reorder(self):              # VerletList.cuh:177-182
        self.force_next_update = True
        check(self.lib.uammd_verletlist_force_next_update(self.h))

    def setCutOffMultiplier(self, m):
        self.force_next_update = True
        check(self.lib.uammd_verletlist_set_cutoff_multiplier(self.h, float(m)))

    def getNumberOfStepsSinceLastUpdate(self):
        v = C.c_int(0)
        check(self.lib.uammd_verletlist_get_steps_since_last_update(self.h, C.byref(v)))
        return int(v.value)

    def update(self, box, cutoff, pos=None):
        """VerletList::update(box, cutOff, st) (VerletList.cuh:112-124)."""
        c = np.broadcast_to(np.asarray(cutoff, dtype=np.float32), (3,))
        if c[0] != c[1] or c[0] != c[2]:
            raise RuntimeError("[VerletList] Invalid argument")      # VerletList.cuh:130-136
        rc = float(c[0])
        need = self.force_next_update or self.currentBox is None or not (box == self.currentBox) or rc != self.currentCutOff
        self.force_next_update = False
        if not need:
            return
        if pos is None:
            self.pd.hintSortByHash(box, [rc * 0.5] * 3)
            pos = self.pd.getPos("read")
        self.currentBox, self.currentCutOff = box, rc
        assert pos.dtype == torch.float32 and pos.is_contiguous() and pos.shape[1] == 4
        rebuilt = C.c_int(0)
        check(self.lib.uammd_verletlist_update(self.h, _ptr(pos), pos.shape[0], f3(box.boxSize),
                                               i3([int(p) for p in box.periodic]), rc, current_stream(), C.byref(rebuilt)))
        self.rebuilds += int(rebuilt.value)

    def getVerletList(self):
        d = VerletListData()
        check(self.lib.uammd_verletlist_get(self.h, C.byref(d)))
        return d

    def to_host(self):
        d = self.getVerletList()
        n, m = d.numberParticles, d.maxNeighboursPerParticle
        torch.cuda.synchronize()
        w = CellList._wrap
        return dict(neighbourList=w(None, d.d_neighbourList, ((m + 1) * n,), torch.int32).cpu().numpy(),
                    numberNeighbours=w(None, d.d_numberNeighbours, (n,), torch.int32).cpu().numpy(),
                    sortPos=w(None, d.d_sortPos, (n, 4), torch.float32).cpu().numpy(),
                    index=w(None, d.d_groupIndex, (n,), torch.int32).cpu().numpy(), stride=d.particleStride,
                    maxNeighboursPerParticle=m)

    def transverse_lj(self, param_table, ntypes, box, force=None, energy=None, virial=None, global_index=None, algo=0):
        check(self.lib.uammd_lj_transverse_verletlist(self.h, _ptr(param_table), int(ntypes), f3(box.boxSize),
                                                      i3([int(p) for p in box.periodic]), _ptr(force), _ptr(energy),
                                                      _ptr(virial), _ptr(global_index), current_stream()))


class Interactor:
    """Interactor::sum(Computables, stream) (Interactor/Interactor.cuh:94-114)."""

    def sum(self, force=True, energy=False, virial=False):
        raise NotImplementedError

    def updateSimulationTime(self, t):
        pass

    def updateTimeStep(self, dt):
        pass

    def updateTemperature(self, T):
        pass

    def updateBox(self, box):
        pass


class _LJ:
    """Potential::LJ = Radial<LJFunctor> with BasicParameterHandler (RadialPotential.cuh, ParameterHandler.cuh)."""

    class InputPairParameters:
        def __init__(self, cutOff=2.5, sigma=1.0, epsilon=1.0, shift=False):
            self.cutOff, self.sigma, self.epsilon, self.shift = cutOff, sigma, epsilon, shift

    def __init__(self):
        self.lib = _lib.load()
        self.ntypes = 1
        self.cutOff = 0.0
        self.table = np.zeros((1, 4), np.float32)
        self._dev = None

    def setPotParameters(self, ti, tj, p):
        self.cutOff = max(float(p.cutOff), self.cutOff)
        new = max(self.ntypes, ti + 1, tj + 1)
        if new != self.ntypes:
            tmp = np.zeros((new * new, 4), np.float32)
            for i in range(self.ntypes):
                for j in range(self.ntypes):
                    tmp[i + new * j] = self.table[i + self.ntypes * j]
            self.table, self.ntypes = tmp, new
        out = LJPairParameters()
        check(self.lib.uammd_lj_process_pair_parameters(p.cutOff, p.sigma, p.epsilon, int(bool(p.shift)), C.byref(out)))
        row = np.array([out.cutOff2, out.sigma2, out.epsilonDivSigma2, out.shift], np.float32)
        self.table[ti + self.ntypes * tj] = row
        if ti != tj:
            self.table[tj + self.ntypes * ti] = row
        self._dev = None

    def getCutOff(self):
        return self.cutOff

    def device_table(self):
        if self._dev is None:
            self._dev = torch.from_numpy(self.table.copy()).cuda()
            check(self.lib.uammd_lj_table_changed())  # (the allocator may hand out the address of the previous table)
        return self._dev


class Potential:
    LJ = _LJ


class PairForces(Interactor):
    """PairForces<Potential::LJ, CellList> (PairForces.cu:43-78): neighbour list unless the box is <= 3 rc in
    every direction, then all pairs."""
    _said_verlet = False

    def __init__(self, pd, box, pot, nl=None, algo=0, pg=None):
        self.lib = _lib.load()
        if isinstance(pd, ParticleGroup):          # PairForces(pg, par, pot) constructor of the reference
            pg, pd = pd, pd.getParticleData()
        self.pd, self.box, self.pot, self.nl, self.algo, self.pg = pd, box, pot, nl, algo, pg
        if isinstance(nl, VerletList) and not PairForces._said_verlet:
            # (as the C++ class: the cell list rebuilt every step is the faster neighbour list on this GPU)
            PairForces._said_verlet = True
            import logging
            logging.getLogger("uammd_amd").info("[PairForces] VerletList chosen: on MI355X PairForces<LJ, CellList> is ~1.9x faster per "
                                                "step for dense liquids (DESIGN.md 5.2b)")

    def fused_gj_arguments(self):
        """What uammd_verletnvt_gj_lj_step needs from this interactor, or None when it is not the plain PairForces<LJ, CellList> on all
        particles with a neighbour list (group, Verlet list, all-pairs box: the integrator then runs the plain sequence)."""
        if self.pg is not None or not isinstance(self.pot, _LJ):
            return None
        rc = np.float32(self.pot.getCutOff())
        L = self.box.boxSize
        if L[0] <= 3 * rc and L[1] <= 3 * rc and L[2] <= 3 * rc:
            return None
        if self.nl is None:
            self.nl = CellList(self.pd)
        if type(self.nl) is not CellList:
            return None
        cd, ubox = CellList.create_update_grid(self.box, rc)
        return self.nl, self.box, ubox, cd, self.pot.device_table(), self.pot.ntypes, self.algo

    def sum(self, force=True, energy=False, virial=False):
        pd = self.pd
        f = pd.getForce("readwrite") if force else None
        e = pd.getEnergy("readwrite") if energy else None
        v = pd.getVirial("readwrite") if virial else None
        rc = np.float32(self.pot.getCutOff())
        L = self.box.boxSize
        useNL = not (L[0] <= 3 * rc and L[1] <= 3 * rc and L[2] <= 3 * rc)
        tbl = self.pot.device_table()
        gidx = self.pg.getIndexIterator() if self.pg is not None else None
        if gidx is not None:
            # a group: the list is built on the members' positions (pg->getPropertyIterator(pos)) and the results go to
            # force[globalIndex[...]] (NeighbourList/common.cuh:17,30)
            gpos = self.pg.getPropertyIterator(pd.getPos("read")).contiguous()
            if useNL:
                if self.nl is None:
                    self.nl = CellList()
                self.nl.force_next_update = True
                self.nl.update(self.box, rc, gpos)
                self.nl.transverse_lj(tbl, self.pot.ntypes, self.box, f, e, v, gidx, self.algo)
            else:
                tmp = [None if t is None else torch.zeros((gpos.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
                       for t in (f, e, v)]
                check(self.lib.uammd_lj_transverse_nbody(_ptr(gpos), gpos.shape[0], _ptr(tbl), self.pot.ntypes, f3(L),
                                                         i3([int(p) for p in self.box.periodic]), _ptr(tmp[0]), _ptr(tmp[1]),
                                                         _ptr(tmp[2]), None, current_stream()))
                for t, out in zip(tmp, (f, e, v)):
                    if t is not None:
                        out.index_add_(0, gidx.long(), t)
            return
        if useNL:
            if self.nl is None:
                self.nl = CellList(pd)
            self.nl.update(self.box, rc)
            self.nl.transverse_lj(tbl, self.pot.ntypes, self.box, f, e, v, None, self.algo)
        else:
            check(self.lib.uammd_lj_transverse_nbody(_ptr(pd.getPos("read")), pd.N, _ptr(tbl), self.pot.ntypes,
                                                     f3(L), i3([int(p) for p in self.box.periodic]), _ptr(f), _ptr(e),
                                                     _ptr(v), None, current_stream()))


class Integrator:
    """Integrator::forwardTime / addInteractor (Integrator/Integrator.cuh:72-125)."""

    def __init__(self, pd):
        self.pd = pd
        self.lib = _lib.load()
        self.interactors = []
        self.steps = 0

    def addInteractor(self, it):
        self.interactors.append(it)

    def getInteractors(self):
        return self.interactors

    def forwardTime(self):
        raise NotImplementedError

    def sumEnergy(self):                      # Integrator.cuh:81
        return 0.0


class _VerletNVTBasic(Integrator):
    kind = "basic"

    class Parameters:
        def __init__(self, temperature=0.0, dt=0.0, friction=1.0, is2D=False, initVelocities=True, mass=-1.0):
            self.temperature, self.dt, self.friction = temperature, dt, friction
            self.is2D, self.initVelocities, self.mass = is2D, initVelocities, mass

    def __init__(self, pd, par):
        self.pg = None
        if isinstance(pd, ParticleGroup):
            self.pg, pd = pd, pd.getParticleData()
        super().__init__(pd)
        rng = pd.rng
        rng.next32(); rng.next32()                       # Basic.cu:36-38
        self.seed = rng.next32()
        self.dt, self.temperature, self.friction, self.is2D = par.dt, par.temperature, par.friction, par.is2D
        self.noiseAmplitude = math.sqrt(2 * par.dt * par.friction * par.temperature)
        useDefaultMass = not pd.isAllocated("mass")
        self.defaultMass = par.mass
        if useDefaultMass and self.defaultMass < 0:
            self.defaultMass = 1.0
        if par.initVelocities:
            self.initVelocities()

    def initVelocities(self):
        vel = self.pd.getVel("write")
        vamp = math.sqrt(3.0 * self.temperature)
        idx = self.pg.getIndexIterator() if self.pg is not None else None
        n = self.pg.getNumberParticles() if self.pg is not None else self.pd.N
        check(self.lib.uammd_verletnvt_initial_velocities(_ptr(vel), _ptr(idx), vamp, int(self.is2D), n,
                                                          self.pd.rng.next32(), current_stream()))

    def _mass(self):
        return None if self.defaultMass > 0 else self.pd.getMass("read")

    def _integrate(self, step):
        pd = self.pd
        fn = self.lib.uammd_verletnvt_gj if self.kind == "gj" else self.lib.uammd_verletnvt_basic
        idx = self.pg.getIndexIterator() if self.pg is not None else None
        n = self.pg.getNumberParticles() if self.pg is not None else pd.N
        check(fn(step, _ptr(pd.getPos("readwrite")), _ptr(pd.getVel("readwrite")), _ptr(pd.getForce("read")),
                 _ptr(self._mass()), self.defaultMass, _ptr(idx), n, self.dt, self.friction, int(self.is2D),
                 self.noiseAmplitude, self.steps, self.seed, current_stream()))

    def forwardTime(self):
        for it in self.interactors:
            it.updateSimulationTime(self.steps * self.dt)
        self.steps += 1
        if self.steps == 1:
            if self.pg is not None and self.pg.getIndexIterator() is not None:   # resetForces on the group only (Basic.cu:108-115)
                f = self.pd.getForce("write")
                check(self.lib.uammd_fill_zero_indexed(_ptr(f), _ptr(self.pg.getIndexIterator()), self.pg.getNumberParticles(), 16,
                                                       current_stream()))
            else:
                self.pd.getForce("write").zero_()
            for it in self.interactors:
                it.updateTemperature(self.temperature)
                it.updateTimeStep(self.dt)
            for it in self.interactors:
                it.sum(force=True)
        if self._fused_step():
            return
        self._integrate(1)
        for it in self.interactors:
            it.sum(force=True)
        self._integrate(2)

    fuse = True   # GronbechJensen + one PairForces<LJ, CellList> on every particle: uammd_verletnvt_gj_lj_step (bit-identical, five launches)
    fused_steps = 0   # how many forwardTime() calls went through it

    def _fused_step(self):
        if not (self.fuse and self.kind == "gj" and self.pg is None and len(self.interactors) == 1):
            return False
        it = self.interactors[0]
        args = getattr(it, "fused_gj_arguments", None)
        args = args() if args is not None else None
        if args is None:
            return False
        nl, box, ubox, cd, tbl, ntypes, algo = args
        pd = self.pd
        pos = pd.getPos("readwrite")          # (signals the position write: every list on this ParticleData rebuilds next time)
        check(self.lib.uammd_verletnvt_gj_lj_step(nl.h, _ptr(pos), _ptr(pd.getVel("readwrite")), _ptr(pd.getForce("readwrite")),
                                                  _ptr(self._mass()), self.defaultMass, pd.N, f3(box.boxSize),
                                                  i3([int(p) for p in box.periodic]), f3(ubox.boxSize), i3([int(p) for p in ubox.periodic]),
                                                  i3(cd), _ptr(tbl), ntypes, self.dt, self.friction, int(self.is2D), self.noiseAmplitude,
                                                  self.steps, self.seed, algo, current_stream()))
        nl.fused_built(box, it.pot.getCutOff(), pos)
        self.fused_steps += 1
        return True

    def sumEnergy(self):
        """sumKineticEnergy (VerletNVT/Basic.cu:186-207): energy[i] += m |v|^2 / 2; returns 0 like the reference."""
        pd = self.pd
        idx = self.pg.getIndexIterator() if self.pg is not None else None
        n = self.pg.getNumberParticles() if self.pg is not None else pd.N
        check(self.lib.uammd_sum_kinetic_energy(_ptr(pd.getVel("read")), _ptr(pd.getEnergy("readwrite")), _ptr(self._mass()),
                                                self.defaultMass, _ptr(idx), n, current_stream()))
        return 0.0


class _VerletNVTGJ(_VerletNVTBasic):
    kind = "gj"


class VerletNVT:
    Basic = _VerletNVTBasic
    GronbechJensen = _VerletNVTGJ


class _BDEulerMaruyama(Integrator):
    class Parameters:
        def __init__(self, temperature=0.0, viscosity=1.0, hydrodynamicRadius=-1.0, dt=0.0, is2D=False, K=None):
            self.temperature, self.viscosity, self.hydrodynamicRadius = temperature, viscosity, hydrodynamicRadius
            self.dt, self.is2D, self.K = dt, is2D, K

    def __init__(self, pd, par):
        super().__init__(pd)
        self.par = par
        self.seed = pd.rng.next32()
        self.selfMobility = 1.0 / (6.0 * math.pi * par.viscosity)   # BrownianDynamics.cu:12-20
        self.radius_from_pd = False
        if par.hydrodynamicRadius != -1.0:
            self.selfMobility /= par.hydrodynamicRadius
        elif pd.isAllocated("radius"):
            self.radius_from_pd = True

    def forwardTime(self):
        pd, par = self.pd, self.par
        self.steps += 1
        pd.getForce("write").zero_()
        for it in self.interactors:
            it.sum(force=True)
        K = None
        if par.K is not None:
            K = (C.c_float * 9)(*[float(x) for x in np.asarray(par.K, dtype=np.float32).reshape(9)])
        radius = pd.getRadius("read") if self.radius_from_pd else None
        check(self.lib.uammd_bd_euler_maruyama(_ptr(pd.getPos("readwrite")), None, _ptr(pd.getForce("read")), K,
                                               self.selfMobility, _ptr(radius), par.dt, int(par.is2D),
                                               par.temperature, pd.N, self.steps, self.seed, current_stream()))


class _BDOtherScheme(_BDEulerMaruyama):
    """BD::MidPoint / AdamsBashforth / Leimkuhler (Integrator/BrownianDynamics.cuh:121-183, .cu:160-387): the same parameters and force
    loop as EulerMaruyama, the position update through uammd_bd_scheme_step."""
    scheme = 0

    def __init__(self, pd, par):
        super().__init__(pd, par)
        self.aux = None

    def _forces(self):
        self.pd.getForce("write").zero_()
        for it in self.interactors:
            it.sum(force=True)

    def _advance(self, substep):
        pd, par = self.pd, self.par
        K = None
        if par.K is not None:
            K = (C.c_float * 9)(*[float(x) for x in np.asarray(par.K, dtype=np.float32).reshape(9)])
        radius = pd.getRadius("read") if self.radius_from_pd else None
        check(self.lib.uammd_bd_scheme_step(self.scheme, substep, _ptr(pd.getPos("readwrite")), _ptr(self.aux), None, None, _ptr(pd.getForce("read")),
                                            K, self.selfMobility, _ptr(radius), par.dt, int(par.is2D), par.temperature, pd.N, self.steps, self.seed,
                                            current_stream()))


class _BDMidPoint(_BDOtherScheme):
    scheme = 1

    def forwardTime(self):
        self.steps += 1
        if self.aux is None:
            self.aux = torch.empty_like(self.pd.getPos("read"))
        self._forces()
        self._advance(0)
        self._forces()
        self._advance(1)


class _BDAdamsBashforth(_BDOtherScheme):
    scheme = 2

    def forwardTime(self):
        self.steps += 1
        if self.steps == 1:
            self._forces()
        self.aux = self.pd.getForce("read").clone()   # storeCurrentForces
        self._forces()
        self._advance(0)


class _BDLeimkuhler(_BDOtherScheme):
    scheme = 3

    def forwardTime(self):
        self.steps += 1
        self._forces()
        self._advance(0)


class BD:
    EulerMaruyama = _BDEulerMaruyama
    MidPoint = _BDMidPoint
    AdamsBashforth = _BDAdamsBashforth
    Leimkuhler = _BDLeimkuhler
