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

    def set_parallel(self, on):
        """All-cores mode of the cell-list build and the IBM spreading (bench.py's cpu_baseline)."""
        self.lib.oracle_set_parallel(int(bool(on)))

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


    # ---- path A: VerletList ----------------------------------------------------------------------
    def verletlist_fill(self, cl, box_L, box_periodic, cutoff2, max_neighbours, n):
        """fillBasicNeighbourList on a built cell list.  Returns (tooManyFlag, neighbourList[(max+1)*n], numberNeighbours[n])."""
        bl, bp = self._box(box_L, box_periodic)
        nl = np.full((max_neighbours + 1) * n, -1, np.int32)
        nn = np.zeros(n, np.int32)
        self.lib.oracle_verletlist_fill.restype = C.c_int
        flag = self.lib.oracle_verletlist_fill(_p(cl["sortPos"]), n, _p(cl["cellStart"]), _p(cl["cellEnd"]),
                                               C.c_uint(cl["validCell"]), _p(cl["L"]), _p(cl["periodic"]), _p(cl["cellDim"]),
                                               _p(bl), _p(bp), self.creal(cutoff2), int(max_neighbours), _p(nl), _p(nn))
        return int(flag), nl, nn

    def verletlist_check_drift(self, current, stored, max_dist, box_L, box_periodic):
        bl, bp = self._box(box_L, box_periodic)
        cur, sto = self.r(current), self.r(stored)
        self.lib.oracle_verletlist_check_drift.restype = C.c_uint
        return int(self.lib.oracle_verletlist_check_drift(_p(cur), _p(sto), len(cur), self.creal(max_dist), _p(bl), _p(bp)))

    def lj_transverse_verletlist(self, sort_pos, group_index, nl, nn, box_L, box_periodic, param_table, ntypes,
                                 want_force=True, want_energy=False, want_virial=False, global_index=None):
        n = len(sort_pos)
        bl, bp = self._box(box_L, box_periodic)
        tbl = self.r(param_table)
        sp = self.r(sort_pos)
        f = np.zeros((n, 4), self.real) if want_force else None
        e = np.zeros(n, self.real) if want_energy else None
        v = np.zeros(n, self.real) if want_virial else None
        gi = None if global_index is None else np.ascontiguousarray(global_index, dtype=np.int32)
        self.lib.oracle_lj_transverse_verletlist(_p(sp), _p(np.ascontiguousarray(group_index, dtype=np.int32)), _p(gi), n,
                                                 _p(np.ascontiguousarray(nl, dtype=np.int32)),
                                                 _p(np.ascontiguousarray(nn, dtype=np.int32)), _p(bl), _p(bp), _p(tbl),
                                                 int(ntypes), _p(f), _p(e), _p(v))
        return f, e, v


    # ---- path B: IBM -----------------------------------------------------------------------------
    def ibm_kernel(self, kind, support, prefactor=0.0, tau=0.0, rmax=np.inf, invh=(0, 0, 0)):
        """kind: 'gaussian' | 'peskin3' | 'peskin4' | 'constant' | 'barnett_magland' | 'sixpoint' (oracle/src/ibm.c)."""
        kinds = {"gaussian": 0, "peskin3": 1, "peskin4": 2, "constant": 3, "barnett_magland": 4, "sixpoint": 5, "gauss2d": 6,
                 "gauss2d_drift_x": 7, "gauss2d_drift_y": 8}
        creal = self.creal

        class K(C.Structure):
            _fields_ = [("kind", C.c_int), ("support", C.c_int * 3), ("prefactor", creal), ("tau", creal),
                        ("rmax", creal), ("invh", creal * 3)]
        sup = np.broadcast_to(np.asarray(support), (3,))
        ih = np.broadcast_to(np.asarray(invh, dtype=np.float64), (3,))
        k = K(kinds[kind], (C.c_int * 3)(int(sup[0]), int(sup[1]), int(sup[2])), prefactor, tau, rmax,
              (creal * 3)(float(ih[0]), float(ih[1]), float(ih[2])))
        return k

    def bm_kernel(self, alpha, beta, support, length_unit=1.0):
        """IBM_kernels::BarnettMagland(alpha, beta) with the reference's Simpson-rule norm."""
        self.lib.oracle_bm_norm.restype = self.creal
        self.lib.oracle_bm_norm.argtypes = [self.creal, self.creal]
        norm = self.real(self.lib.oracle_bm_norm(alpha, beta))
        return self.ibm_kernel("barnett_magland", support, prefactor=float(self.real(1.0 / float(norm))), tau=beta, rmax=alpha, invh=[length_unit] * 3)

    def phi_sixpoint(self, h, r):
        self.lib.oracle_phi_sixpoint.restype = self.creal
        self.lib.oracle_phi_sixpoint.argtypes = [self.creal, self.creal]
        return np.array([self.lib.oracle_phi_sixpoint(1.0 / h, float(x)) for x in np.atleast_1d(r)])

    def fcm_gaussian(self, h, tolerance):
        """FCM_ns::Kernels::Gaussian(h, tol) -> dict(support, upsampling, width, prefactor, tau, rmax, a_eff, kernel)."""
        out = np.zeros(6, self.real)
        self.lib.oracle_fcm_gaussian_init.restype = C.c_int
        sup = self.lib.oracle_fcm_gaussian_init(self.creal(h), self.creal(tolerance), _p(out))
        d = dict(support=int(sup), upsampling=float(out[0]), width=float(out[1]), prefactor=float(out[2]),
                 tau=float(out[3]), rmax=float(out[4]), a_eff=float(out[5]))
        d["kernel"] = self.ibm_kernel("gaussian", sup, out[2], out[3], out[4])
        return d

    def fcm_advise_grid_size(self, a, tolerance):
        self.lib.oracle_fcm_advise_grid_size.restype = self.creal
        return float(self.lib.oracle_fcm_advise_grid_size(self.creal(a), self.creal(tolerance)))

    def ibm_spread(self, pos, v, L, periodic, cell_dim, kernel, grid=None, nx_stride=None):
        pos = self.r(pos)
        v = self.r(v)
        n = len(pos)
        ncomp = 1 if v.ndim == 1 else v.shape[1]
        L, per = self._box(L, periodic)
        cd = np.ascontiguousarray(cell_dim, dtype=np.int32)
        nxs = int(cd[0]) if nx_stride is None else int(nx_stride)
        if grid is None:
            grid = np.zeros((int(cd[2]), int(cd[1]), nxs, ncomp), self.real)
        self.lib.oracle_ibm_spread(_p(pos), pos.shape[1], _p(v), ncomp, n, _p(L), _p(per), _p(cd), nxs,
                                   C.byref(kernel), _p(grid))
        return grid

    def ibm_gather(self, pos, grid, L, periodic, cell_dim, kernel, out=None, nx_stride=None):
        pos = self.r(pos)
        grid = self.r(grid)
        n = len(pos)
        ncomp = grid.shape[-1] if grid.ndim == 4 else 1
        L, per = self._box(L, periodic)
        cd = np.ascontiguousarray(cell_dim, dtype=np.int32)
        nxs = int(cd[0]) if nx_stride is None else int(nx_stride)
        if out is None:
            out = np.zeros((n, ncomp), self.real)
        self.lib.oracle_ibm_gather(_p(pos), pos.shape[1], _p(out), ncomp, n, _p(L), _p(per), _p(cd), nxs,
                                   C.byref(kernel), _p(grid))
        return out

    def ibm_stencil(self, pos3, L, periodic, cell_dim, kernel):
        L, per = self._box(L, periodic)
        cd = np.ascontiguousarray(cell_dim, dtype=np.int32)
        p = self.r(pos3)
        ci, P, sup = np.zeros(3, np.int32), np.zeros(3, np.int32), np.zeros(3, np.int32)
        wx, wy, wz = np.zeros(64, self.real), np.zeros(64, self.real), np.zeros(64, self.real)
        self.lib.oracle_ibm_stencil(_p(p), _p(L), _p(per), _p(cd), C.byref(kernel), _p(ci), _p(P), _p(sup), _p(wx),
                                    _p(wy), _p(wz))
        return ci, P, sup, wx[:sup[0]], wy[:sup[1]], wz[:sup[2]]

    # ---- path B: FCM k-space ------------------------------------------------------------------------
    def fcm_force_fourier_to_vel(self, grid_k, viscosity, L, cell_dim):
        """grid_k: complex array [nz, ny, nx/2+1, 3] of this precision; in place."""
        L, _ = self._box(L, 1)
        cd = np.ascontiguousarray(cell_dim, dtype=np.int32)
        assert grid_k.flags.c_contiguous
        self.lib.oracle_fcm_force_fourier_to_vel(_p(grid_k), self.creal(viscosity), _p(L), _p(cd))
        return grid_k

    def fcm_fourier_brownian_noise(self, grid_k, L, cell_dim, noise_prefactor, viscosity, seed1, seed2):
        L, _ = self._box(L, 1)
        cd = np.ascontiguousarray(cell_dim, dtype=np.int32)
        assert grid_k.flags.c_contiguous
        self.lib.oracle_fcm_fourier_brownian_noise(_p(grid_k), _p(L), _p(cd), self.creal(noise_prefactor),
                                                   self.creal(viscosity), C.c_uint(seed1), C.c_uint(seed2))
        return grid_k

    def fcm_noise_prefactor(self, prefactor, temperature, L, cell_dim):
        L, _ = self._box(L, 1)
        cd = np.ascontiguousarray(cell_dim, dtype=np.int32)
        self.lib.oracle_fcm_noise_prefactor.restype = self.creal
        return float(self.lib.oracle_fcm_noise_prefactor(self.creal(prefactor), self.creal(temperature), _p(L), _p(cd)))

    def fcm_self_mobility(self, a, viscosity, Lx):
        self.lib.oracle_fcm_self_mobility.restype = C.c_double
        return float(self.lib.oracle_fcm_self_mobility(C.c_double(a), C.c_double(viscosity), C.c_double(Lx)))

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
        n = len(pos4) if index is None else len(index)
        K9 = None if K is None else self.r(K).reshape(9)
        self.lib.oracle_bd_euler_maruyama(_p(pos4), _p(index), _p(force4), _p(K9), self.creal(self_mobility),
                                          _p(radius), self.creal(dt), int(is2D), self.creal(temperature), n,
                                          C.c_uint(step_num), C.c_uint(seed))

    def bd_midpoint(self, step, pos4, initial4, force4, self_mobility, dt, temperature, step_num, seed, K=None, radius=None, index=None, is2D=False):
        K9 = None if K is None else self.r(K).reshape(9)
        n = len(pos4) if index is None else len(index)
        self.lib.oracle_bd_midpoint(int(step), _p(pos4), _p(initial4), _p(index), _p(force4), _p(K9), self.creal(self_mobility), _p(radius),
                                    self.creal(dt), int(is2D), self.creal(temperature), n, C.c_uint(step_num), C.c_uint(seed))

    def bd_adams_bashforth(self, pos4, previous4, force4, self_mobility, dt, temperature, step_num, seed, K=None, radius=None, index=None, is2D=False):
        K9 = None if K is None else self.r(K).reshape(9)
        n = len(pos4) if index is None else len(index)
        self.lib.oracle_bd_adams_bashforth(_p(pos4), _p(previous4), _p(index), _p(force4), _p(K9), self.creal(self_mobility), _p(radius),
                                           self.creal(dt), int(is2D), self.creal(temperature), n, C.c_uint(step_num), C.c_uint(seed))

    def bd_leimkuhler(self, pos4, force4, self_mobility, dt, temperature, step_num, seed, K=None, radius=None, index=None, original_index=None,
                      is2D=False):
        K9 = None if K is None else self.r(K).reshape(9)
        n = len(pos4) if index is None else len(index)
        self.lib.oracle_bd_leimkuhler(_p(pos4), _p(index), _p(original_index), _p(force4), _p(K9), self.creal(self_mobility), _p(radius),
                                      self.creal(dt), int(is2D), self.creal(temperature), n, C.c_uint(step_num), C.c_uint(seed))

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
