"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product path).

ctypes front-end of the C restatement in ``oracle/src`` (built into ``oracle/_build``), one
instance per precision: ``get("f32")`` mirrors UAMMD's default ``real=float`` build, ``get("f64")``
the ``-DDOUBLE_PRECISION`` build the reference's own GTest suite uses (test/CMakeLists.txt:4).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")


def build(force=False):
    """Compile liboracle_f32.so / liboracle_f64.so with gcc (Makefile in this directory)."""
    targets = [os.path.join(_BUILD, "liboracle_f32.so"), os.path.join(_BUILD, "liboracle_f64.so")]
    srcs = [os.path.join(_HERE, "src", f) for f in os.listdir(os.path.join(_HERE, "src"))]
    newest = max(os.path.getmtime(s) for s in srcs)
    if not force and all(os.path.exists(t) and os.path.getmtime(t) >= newest for t in targets):
        return targets
    subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)
    return targets


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, precision="f32"):
        assert precision in ("f32", "f64")
        build()
        self.precision = precision
        self.real = np.float32 if precision == "f32" else np.float64
        self.creal = C.c_float if precision == "f32" else C.c_double
        self.lib = C.CDLL(os.path.join(_BUILD, f"liboracle_{precision}.so"))
        self.lib.oracle_morton_hash.restype = C.c_uint
        self.lib.oracle_celllist_next_valid_cell.restype = C.c_uint

    # ---- helpers -------------------------------------------------------------------------------
    def r(self, a):
        return np.ascontiguousarray(a, dtype=self.real)

    def _box(self, L, periodic):
        L = self.r(np.broadcast_to(np.asarray(L, dtype=np.float64), (3,)))
        per = np.ascontiguousarray(np.broadcast_to(np.asarray(periodic), (3,)), dtype=np.int32)
        return L, per

    # ---- path A: neighbour search ----------------------------------------------------------------
    def morton_hash(self, cx, cy, cz):
        return int(self.lib.oracle_morton_hash(int(cx), int(cy), int(cz)))

    def sort_end_bit(self, max_hash):
        return int(self.lib.oracle_sort_end_bit(C.c_uint(max_hash)))

    def celllist_create_grid(self, L, periodic, cutoff):
        L, per = self._box(L, periodic)
        rc = self.r(np.broadcast_to(np.asarray(cutoff, dtype=np.float64), (3,)))
        cd = np.zeros(3, np.int32)
        Lo = np.zeros(3, self.real)
        po = np.zeros(3, np.int32)
        self.lib.oracle_celllist_create_grid(_p(L), _p(per), _p(rc), _p(cd), _p(Lo), _p(po))
        return cd, Lo, po

    def next_valid_cell(self, n, state):
        """state: np.int64[2] = [-1, -1] initially.  Returns (validCell, needs_clear)."""
        clear = C.c_int(0)
        v = self.lib.oracle_celllist_next_valid_cell(int(n), _p(state), C.byref(clear))
        return int(v), bool(clear.value)

    def stable_sort_pairs(self, keys, vals, end_bit):
        keys = np.ascontiguousarray(keys, dtype=np.uint32).copy()
        vals = np.ascontiguousarray(vals, dtype=np.int32).copy()
        self.lib.oracle_stable_sort_pairs(_p(keys), _p(vals), len(keys), int(end_bit))
        return keys, vals

    def cell_of(self, pos4, L, periodic, cell_dim):
        pos4 = self.r(pos4)
        L, per = self._box(L, periodic)
        cd = np.ascontiguousarray(cell_dim, dtype=np.int32)
        out = np.zeros(len(pos4), np.int32)
        self.lib.oracle_cell_of(_p(pos4), len(pos4), _p(L), _p(per), _p(cd), _p(out))
        return out

    def celllist_build(self, pos4, L, periodic, cell_dim, valid_cell=None, cell_start=None, cell_end=None):
        pos4 = self.r(pos4)
        n = len(pos4)
        L, per = self._box(L, periodic)
        cd = np.ascontiguousarray(cell_dim, dtype=np.int32)
        ncells = int(cd[0]) * int(cd[1]) * int(cd[2])
        if valid_cell is None:
            valid_cell = n
        if cell_start is None:
            cell_start = np.zeros(ncells, np.uint32)
        if cell_end is None:
            cell_end = np.zeros(ncells, np.int32)
        h = np.zeros(n, np.uint32)
        idx = np.zeros(n, np.int32)
        sp = np.zeros((n, 4), self.real)
        err = self.lib.oracle_celllist_build(_p(pos4), n, _p(L), _p(per), _p(cd), C.c_uint(valid_cell), _p(h),
                                             _p(idx), _p(sp), _p(cell_start), _p(cell_end))
        return dict(hash=h, index=idx, sortPos=sp, cellStart=cell_start, cellEnd=cell_end, validCell=valid_cell,
                    error=int(err), cellDim=cd, L=L, periodic=per)

    # ---- path A: LJ ------------------------------------------------------------------------------
    def lj_params(self, cutoff, sigma, epsilon, shift=False):
        out = np.zeros(4, self.real)
        self.lib.oracle_lj_process_pair_parameters(self.creal(cutoff), self.creal(sigma), self.creal(epsilon),
                                                   int(bool(shift)), _p(out))
        return out

    def lj_transverse_celllist(self, cl, box_L, box_periodic, param_table, ntypes, n, want_force=True,
                               want_energy=False, want_virial=False, global_index=None, force=None):
        bl, bp = self._box(box_L, box_periodic)
        tbl = self.r(param_table)
        f = (np.zeros((n, 4), self.real) if force is None else force) if want_force else None
        e = np.zeros(n, self.real) if want_energy else None
        v = np.zeros(n, self.real) if want_virial else None
        gi = None if global_index is None else np.ascontiguousarray(global_index, dtype=np.int32)
        self.lib.oracle_lj_transverse_celllist(_p(cl["sortPos"]), _p(cl["index"]), _p(gi), n, _p(cl["cellStart"]),
                                               _p(cl["cellEnd"]), C.c_uint(cl["validCell"]), _p(cl["L"]),
                                               _p(cl["periodic"]), _p(cl["cellDim"]), _p(bl), _p(bp), _p(tbl),
                                               int(ntypes), _p(f), _p(e), _p(v))
        return f, e, v

    def lj_transverse_nbody(self, pos4, box_L, box_periodic, param_table, ntypes, want_force=True, want_energy=False,
                            want_virial=False):
        pos4 = self.r(pos4)
        n = len(pos4)
        bl, bp = self._box(box_L, box_periodic)
        tbl = self.r(param_table)
        f = np.zeros((n, 4), self.real) if want_force else None
        e = np.zeros(n, self.real) if want_energy else None
        v = np.zeros(n, self.real) if want_virial else None
        self.lib.oracle_lj_transverse_nbody(_p(pos4), None, n, _p(bl), _p(bp), _p(tbl), int(ntypes), _p(f), _p(e), _p(v))
        return f, e, v

    def lj_nbody_f64(self, pos4, L, periodic, cutoff, sigma, epsilon):
        p = np.ascontiguousarray(pos4, dtype=np.float64)
        n = len(p)
        Ld = np.ascontiguousarray(np.broadcast_to(np.asarray(L, dtype=np.float64), (3,)))
        per = np.ascontiguousarray(np.broadcast_to(np.asarray(periodic), (3,)), dtype=np.int32)
        f = np.zeros((n, 3), np.float64)
        self.lib.oracle_lj_nbody_f64(_p(p), n, _p(Ld), _p(per), C.c_double(cutoff), C.c_double(sigma),
                                     C.c_double(epsilon), _p(f))
        return f

    # ---- integrators -----------------------------------------------------------------------------
    def verletnvt_gj(self, step, pos4, vel3, force4, dt, friction, noise_amplitude, step_num, seed, default_mass=1.0,
                     mass=None, index=None, is2D=False):
        n = len(pos4)
        self.lib.oracle_verletnvt_gj(int(step), _p(pos4), _p(vel3), _p(force4), _p(mass), self.creal(default_mass),
                                     _p(index), n, self.creal(dt), self.creal(friction), int(is2D),
                                     self.creal(noise_amplitude), C.c_uint(step_num), C.c_uint(seed))

    def verletnvt_basic(self, step, pos4, vel3, force4, dt, friction, noise_amplitude, step_num, seed,
                        default_mass=1.0, mass=None, index=None, is2D=False):
        n = len(pos4)
        self.lib.oracle_verletnvt_basic(int(step), _p(pos4), _p(vel3), _p(force4), _p(mass), self.creal(default_mass),
                                        _p(index), n, self.creal(dt), self.creal(friction), int(is2D),
                                        self.creal(noise_amplitude), C.c_uint(step_num), C.c_uint(seed))

    def verletnvt_initial_velocities(self, n, vamp, seed, is2D=False, index=None):
        v = np.zeros((n, 3), self.real)
        self.lib.oracle_verletnvt_initial_velocities(_p(v), _p(index), self.creal(vamp), int(is2D), n, C.c_uint(seed))
        return v

    def bd_euler_maruyama(self, pos4, force4, self_mobility, dt, temperature, step_num, seed, K=None, radius=None,
                          index=None, is2D=False):
        n = len(pos4)
        K9 = None if K is None else self.r(K).reshape(9)
        self.lib.oracle_bd_euler_maruyama(_p(pos4), _p(index), _p(force4), _p(K9), self.creal(self_mobility),
                                          _p(radius), self.creal(dt), int(is2D), self.creal(temperature), n,
                                          C.c_uint(step_num), C.c_uint(seed))

    def fcm_euler_maruyama(self, pos4, linear_v3, dt, index=None):
        self.lib.oracle_fcm_euler_maruyama(_p(pos4), _p(index), _p(self.r(linear_v3)), len(pos4), self.creal(dt))

    # ---- Saru ------------------------------------------------------------------------------------
    def saru_u32(self, seeds, n):
        s = list(seeds) + [0, 0, 0]
        out = np.zeros(n, np.uint32)
        self.lib.oracle_saru_u32(len(seeds), C.c_uint(s[0]), C.c_uint(s[1]), C.c_uint(s[2]), n, _p(out))
        return out

    def saru_f_range(self, seed, low, high, n):
        out = np.zeros(n, np.float32)
        self.lib.oracle_saru_f_range(C.c_uint(seed), C.c_float(low), C.c_float(high), n, _p(out))
        return out

    def saru_gf(self, seeds3, mean, std, npairs):
        out = np.zeros(2 * npairs, np.float32)
        self.lib.oracle_saru_gf(C.c_uint(seeds3[0]), C.c_uint(seeds3[1]), C.c_uint(seeds3[2]), C.c_float(mean),
                                C.c_float(std), npairs, _p(out))
        return out


_CACHE = {}


def get(precision="f32"):
    if precision not in _CACHE:
        _CACHE[precision] = Oracle(precision)
    return _CACHE[precision]
