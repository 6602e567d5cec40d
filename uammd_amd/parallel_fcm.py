"""Multi-GPU path B: z-slab decomposition of the FCM Stokes solver (BASELINE configs[4]).

The reference is single GPU; this is new design.  One process per GPU (`torch.distributed`, backend "nccl" = RCCL over
xGMI; "gloo" in the CPU tests).  Rank r owns
  * the particles with z in its slab (the same `SlabDecomposition` as path A: local frame, migration),
  * nzLocal = nz/P xy-planes of the real grid, kept in a WINDOW of nzLocal + 2*halo planes so that every stencil of an
    owned particle lands inside it,
  * after the transpose, nyLocal = ny/P y-rows of the Fourier grid with all of z.
Per call of `displacements` (= FCM_impl::computeHydrodynamicDisplacements, FCM_impl.cuh:652-693):

    spread (local)  ->  halo planes SENT to the two neighbours and ADDED           2 P2P messages of He planes
    2-D R2C (x,y) of the owned planes  ->  all-to-all transpose                    1 all-to-all, 1/P of the grid per peer
    1-D FFT (z)  ->  Stokes + Fourier noise on the y-rows  ->  1-D inverse FFT (z)
    all-to-all transpose back  ->  2-D C2R into the owned planes                   1 all-to-all
    halo planes COPIED from the neighbours  ->  gather (local)                     2 P2P messages

No grid data goes through a ring all-reduce.  The Fourier noise needs NO communication: node `id` draws from
Saru(id, seed, seed2) and the k-space kernel regenerates the draw of a node's conjugate partner locally (gather form,
uammd_amd/csrc/fcm.hip), so every rank only needs the same (seed, seed2), which advance in lock step.

The per-rank compute stages are a backend: `HipSlabBackend` (libuammd_hip.so, uammd_fcm_slab_* in include/uammd_hip.h) is
the product; the CPU tests plug a numpy backend built on the oracle to check the exchange logic under gloo.  `ranks` may
hold ALL P rank states in one process (exchanges become tensor copies): that is how the decomposition is verified
against the single-GPU solver on one MI355X.
"""
import ctypes as C

import torch
import torch.distributed as dist

from .parallel import SlabDecomposition


class SlabGeometry:
    """Who owns what.  cells = (nx, ny, nz) of the global grid, `support` of the spreading kernel along z."""

    def __init__(self, cells, L, world, support_z, tile=8):
        self.cells = [int(c) for c in cells]
        self.L = [float(x) for x in L]
        self.world = int(world)
        nx, ny, nz = self.cells
        if nz % world or ny % world:
            raise ValueError(f"grid {self.cells} does not split into {world} z-slabs and y-row blocks")
        self.nzl, self.nyl = nz // world, ny // world
        self.nkx = nx // 2 + 1
        self.nxpad = 2 * self.nkx
        # planes a stencil of an owned particle can reach outside the slab: support/2 (+1 for the even-support shift,
        # IBM.cu:10-31) +1 for a particle whose local cell rounds one plane outside the slab
        self.he = int(support_z) // 2 + 2
        # the window keeps whole tiles when the tile-owned spread kernel can be used (cells % tile == 0)
        tiled = nx % tile == 0 and ny % tile == 0 and self.nzl % tile == 0
        self.halo = -(-self.he // tile) * tile if tiled else self.he
        # planes actually exchanged: the whole halo.  The spare planes beyond the stencil reach (3 when the halo is rounded up
        # to a tile) let an owned particle sit that far outside its slab, so migration need not run every step.
        self.drift_planes = self.halo - self.he
        self.he = self.halo
        if self.nzl < self.he:
            raise ValueError("slab thinner than the spreading stencil: halo would need second neighbours")
        self.nzw = self.nzl + 2 * self.halo


# ---------------------------------------------------------------------------------------------------------------------
class HipSlabBackend:
    """The compute stages of one rank on its GPU (include/uammd_hip.h, uammd_fcm_slab_*).  No CPU fallback."""

    def __init__(self, geom, rank, kernel, viscosity, seed, device=None):
        from . import _lib
        from ._lib import FCMParameters, check
        self._check = check
        self.lib = _lib.load()
        self.g, self.rank = geom, rank
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        p = FCMParameters()
        for k in range(3):
            p.boxSize[k] = geom.L[k]
            p.cells[k] = geom.cells[k]
        p.viscosity, p.seed, p.kernel, p.hydrodynamicRadius = float(viscosity), int(seed) & 0xFFFFFFFF, kernel, 0.0
        h = C.c_void_p()
        check(self.lib.uammd_fcm_slab_create(C.byref(p), geom.nzl, rank * geom.nzl, geom.halo, geom.nyl, rank * geom.nyl,
                                             C.byref(h)))
        self.h = h
        self.grid = torch.zeros((geom.nzw, 3, geom.cells[1], geom.nxpad), dtype=torch.float32, device=self.device)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.uammd_fcm_slab_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @staticmethod
    def _p(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    @staticmethod
    def _st():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def set_option(self, name, value):
        self._check(self.lib.uammd_fcm_slab_set_option(self.h, name.encode(), int(value)))

    def spread(self, pos_local, force):
        self._check(self.lib.uammd_fcm_slab_spread(self.h, self._p(pos_local), self._p(force), pos_local.shape[0],
                                                   self._p(self.grid), self._st()))
        return self.grid

    def forward_xy(self, grid):
        """In place; returns the owned planes viewed as complex [zl][c][y][kx][re,im]."""
        g = self.g
        self._check(self.lib.uammd_fcm_slab_forward_xy(self.h, self._p(grid), self._st()))
        return grid[g.halo:g.halo + g.nzl].view(g.nzl, 3, g.cells[1], g.nkx, 2)

    def forward_xy_fold(self, grid, from_down, from_up):
        """forward_xy with the neighbours' halo planes added to the first / last owned planes on the way (one launch less); None if
        the library's own FFT does not serve this grid or the planes do not qualify (the caller folds them itself)."""
        g = self.g
        ts = (from_down, from_up)
        if g.he == 0 or not all(t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and t.shape == (g.he,) + tuple(grid.shape[1:]) for t in ts):
            return None
        rc = self.lib.uammd_fcm_slab_forward_xy_fold(self.h, self._p(grid), self._p(from_down), self._p(from_up), g.he, self._st())
        if rc == 1:
            return None
        self._check(rc)
        return grid[g.halo:g.halo + g.nzl].view(g.nzl, 3, g.cells[1], g.nkx, 2)

    def spectrum_view(self, grid):
        g = self.g
        return grid[g.halo:g.halo + g.nzl].view(g.nzl, 3, g.cells[1], g.nkx, 2)

    def fft_z(self, buf, inverse):
        self._check(self.lib.uammd_fcm_slab_fft_z(self.h, self._p(buf), int(inverse), self._st()))

    def kspace(self, buf, have_force, temperature, prefactor, seed2):
        self._check(self.lib.uammd_fcm_slab_kspace(self.h, self._p(buf), int(have_force), float(temperature), float(prefactor),
                                                   int(seed2) & 0xFFFFFFFF, self._st()))

    def z_fused(self, buf, have_force, temperature, prefactor, seed2):
        """forward z + operator + inverse z in one pass; False when the grid does not allow it (the caller takes the three calls)."""
        rc = self.lib.uammd_fcm_slab_z_fused(self.h, self._p(buf), int(have_force), float(temperature), float(prefactor),
                                             int(seed2) & 0xFFFFFFFF, self._st())
        if rc == 1:
            return False
        self._check(rc)
        return True

    def inverse_xy(self, grid):
        self._check(self.lib.uammd_fcm_slab_inverse_xy(self.h, self._p(grid), self._st()))

    def inverse_xy_inter(self, wrap=False):
        """The inverse writing the owned planes of the gather's float4 window (self.inter) directly; None when the grid does not take the
        library's own FFT or the spread is not the tile-owned one (the caller then uses inverse_xy + gather).  wrap (world size 1, the
        rank is its own neighbour): the halo planes are written by the same pass, no exchange is needed afterwards."""
        if getattr(self, "_no_inter", False):
            return None
        g = self.g
        if getattr(self, "inter", None) is None:
            self.inter = torch.zeros((self.grid.shape[0], g.cells[1], g.cells[0], 4), dtype=torch.float32, device=self.device)
        if wrap:
            rc = self.lib.uammd_fcm_slab_inverse_xy_inter_wrap(self.h, self._p(self.grid), self._p(self.inter), g.he, self._st())
        else:
            rc = self.lib.uammd_fcm_slab_inverse_xy_inter(self.h, self._p(self.grid), self._p(self.inter), self._st())
        if rc == 1:
            self._no_inter = True
            return None
        self._check(rc)
        return self.inter

    def gather_inter(self, pos_local, inter):
        out = torch.empty((pos_local.shape[0], 3), dtype=torch.float32, device=self.device)
        self._check(self.lib.uammd_fcm_slab_gather_inter(self.h, self._p(pos_local), pos_local.shape[0], self._p(inter), self._p(out),
                                                         self._st()))
        return out

    def gather(self, pos_local, grid):
        out = torch.empty((pos_local.shape[0], 3), dtype=torch.float32, device=self.device)
        self._check(self.lib.uammd_fcm_slab_gather(self.h, self._p(pos_local), pos_local.shape[0], self._p(grid), self._p(out),
                                                   self._st()))
        return out

    def pair(self, add, dst0, src0, dst1, src1):
        """dst0 (+)= src0 and dst1 (+)= src1 in one launch (uammd_slab_add2 / uammd_slab_copy2); False if the views do not qualify."""
        ts = (dst0, src0, dst1, src1)
        if not all(t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 for t in ts) or len({t.numel() for t in ts}) != 1:
            return False
        fn = self.lib.uammd_slab_add2 if add else self.lib.uammd_slab_copy2
        self._check(fn(self._p(dst0), self._p(src0), self._p(dst1), self._p(src1), dst0.numel(), self._st()))
        return True

    def euler_maruyama(self, pos_local, v, dt):
        """pos += v dt (uammd_fcm_euler_maruyama); False if the arrays do not qualify."""
        if not (pos_local.is_cuda and pos_local.is_contiguous() and v.is_contiguous() and pos_local.dtype == torch.float32):
            return False
        self._check(self.lib.uammd_fcm_euler_maruyama(self._p(pos_local), None, self._p(v), pos_local.shape[0], float(dt), self._st()))
        return True

    def new_zbuffer(self):
        g = self.g
        return torch.empty((g.cells[2], 3, g.nyl, g.nkx, 2), dtype=torch.float32, device=self.device)


# ---------------------------------------------------------------------------------------------------------------------
class _Exchange:
    """The three communication patterns, for either ONE local rank under torch.distributed or ALL ranks in-process."""

    def __init__(self, world, local_ranks, group=None, comm=None):
        """comm: an uammd_amd.comm.AbiComm (uammd_comm_*: RCCL behind the C ABI) for the one-rank-per-process form — also at world 1,
        where every message then makes the round trip through RCCL; without it torch.distributed (the test double) carries them."""
        self.world, self.local, self.group, self.comm = world, list(local_ranks), group, comm
        self.in_process = len(self.local) == world
        if not self.in_process and len(self.local) != 1:
            raise ValueError("hold either one rank (torch.distributed) or all of them (in-process)")

    def neighbours(self, to_up, to_down, into_down=None, into_up=None):
        """to_up[i] goes to rank+1, to_down[i] to rank-1 (periodic).  Returns (from_down, from_up) lists.  into_down / into_up: where the
        caller wants the messages (contiguous tensors of the messages' shape): under torch.distributed they are received there and
        returned, so the caller has nothing to copy; the in-process and single-rank forms return the senders' tensors as before."""
        P = self.world
        if self.comm is not None and len(self.local) == 1:   # one rank per process through uammd_comm_* (also the world of one)
            su, sd = to_up[0].contiguous(), to_down[0].contiguous()
            ok = lambda into, like: into is not None and into[0].is_contiguous() and into[0].shape == like.shape and into[0].dtype == like.dtype
            from_down = into_down[0] if ok(into_down, su) else torch.empty_like(su)
            from_up = into_up[0] if ok(into_up, sd) else torch.empty_like(sd)
            flat = lambda t: t.view(torch.float32).reshape(1, -1) if t.numel() else t.view(torch.float32).reshape(0, 1)
            self.comm.halo_exchange(flat(su), flat(sd), flat(from_down), flat(from_up))
            return [from_down], [from_up]
        if self.in_process:
            return [to_up[(r - 1) % P] for r in range(P)], [to_down[(r + 1) % P] for r in range(P)]
        r = self.local[0]
        up, down = (r + 1) % P, (r - 1) % P
        if P == 1:
            return [to_up[0]], [to_down[0]]

        def landing(into, like):
            ok = into is not None and into[0].is_contiguous() and into[0].shape == like.shape and into[0].dtype == like.dtype
            return into[0] if ok else torch.empty_like(like)
        from_down, from_up = landing(into_down, to_up[0]), landing(into_up, to_down[0])
        # with 2 ranks both neighbours are the same peer: the messages are told apart by tag (gloo) / issue order (RCCL)
        t1, t2 = (1, 2) if P == 2 else (0, 0)
        su, sd = to_up[0].contiguous(), to_down[0].contiguous()
        staged = self._staged(su)
        if staged:      # gloo moves host memory only (several ranks on ONE GPU: tests/test_gpu_world2.py, bench.py UAMMD_BENCH_BACKEND=gloo)
            su, sd = su.cpu(), sd.cpu()
            rd, ru = torch.empty(from_down.shape, dtype=from_down.dtype), torch.empty(from_up.shape, dtype=from_up.dtype)
        else:
            rd, ru = from_down, from_up
        ops = [dist.P2POp(dist.isend, su, up, self.group, tag=t1),
               dist.P2POp(dist.irecv, rd, down, self.group, tag=t1),
               dist.P2POp(dist.isend, sd, down, self.group, tag=t2),
               dist.P2POp(dist.irecv, ru, up, self.group, tag=t2)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if staged:
            from_down.copy_(rd)
            from_up.copy_(ru)
        return [from_down], [from_up]

    def _staged(self, t):
        return t.is_cuda and dist.get_backend(self.group) == "gloo"

    def all_to_all(self, blocks):
        """blocks[i]: tensor [P, ...], slice d goes to rank d.  Returns tensors [P, ...] with slice s received from rank s."""
        P = self.world
        if self.comm is not None and len(self.local) == 1:
            src = blocks[0].contiguous()
            out = torch.empty_like(src)
            self.comm.alltoall(src, out)
            return [out]
        if P == 1:
            return [blocks[0]]          # the only rank keeps its block: nothing moves, nothing is copied
        if self.in_process:
            return [torch.stack([blocks[s][r] for s in range(P)], dim=0) for r in range(P)]
        out = torch.empty_like(blocks[0])
        if self._staged(out):   # gloo has no all-to-all on device memory: P - 1 host-staged point-to-point pairs + the own block
            src, r = blocks[0].contiguous().cpu(), self.local[0]
            host = torch.empty(src.shape, dtype=src.dtype)
            host[r] = src[r]
            ops = []
            for k in range(1, P):
                to, frm = (r + k) % P, (r - k) % P
                ops += [dist.P2POp(dist.isend, src[to], to, self.group, tag=k), dist.P2POp(dist.irecv, host[frm], frm, self.group, tag=k)]
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            out.copy_(host)
            return [out]
        dist.all_to_all_single(out, blocks[0].contiguous(), group=self.group)
        return [out]


class DistributedFCM:
    """FCM_impl over z-slabs.  `backends`: one per local rank (see _Exchange)."""

    def __init__(self, geom, backends, local_ranks, seed2=0, group=None, comm=None):
        self.g, self.b = geom, list(backends)
        self.x = _Exchange(geom.world, local_ranks, group, comm)
        self.seed2 = int(seed2)  # the reference's `static uint seed2` (FCM_impl.cuh:517)
        self._z = [None] * len(self.b)

    def displacements(self, pos_locals, forces, temperature, prefactor):
        """pos_locals[i]: real4[N_i] of local rank i in its window frame; forces[i]: real4[N_i] or None (all ranks alike).
        Returns the list of real3[N_i] velocities  M F + prefactor sqrt(2T) M^(1/2) dW."""
        g, P = self.g, self.g.world
        H, He, nzl, nyl = g.halo, g.he, g.nzl, g.nyl
        have_force = forces[0] is not None
        if temperature > 0:
            self.seed2 += 1
        n = len(self.b)
        grids = [self.b[i].spread(pos_locals[i], forces[i]) for i in range(n)]
        zbufs = []
        if have_force:
            fd, fu = self.x.neighbours([gr[H + nzl:H + nzl + He] for gr in grids], [gr[H - He:H] for gr in grids])
            xy = []
            for i, gr in enumerate(grids):
                fold = getattr(self.b[i], "forward_xy_fold", None)
                done = fold(gr, fd[i], fu[i]) if fold is not None else None
                if done is None:
                    pair = getattr(self.b[i], "pair", None)
                    if pair is None or not pair(True, gr[H:H + He], fd[i], gr[H + nzl - He:H + nzl], fu[i]):
                        gr[H:H + He] += fd[i]
                        gr[H + nzl - He:H + nzl] += fu[i]
                    done = self.b[i].forward_xy(gr)
                xy.append(done)
            send = [a.view(nzl, 3, P, nyl, g.nkx, 2).permute(2, 0, 1, 3, 4, 5).contiguous() for a in xy]
            recv = self.x.all_to_all(send)      # [src][zl][c][yl][kx] == [z][c][yl][kx]
            for i in range(n):
                zbufs.append(recv[i].reshape(g.cells[2], 3, nyl, g.nkx, 2))
        else:
            for i in range(n):
                if self._z[i] is None:
                    self._z[i] = self.b[i].new_zbuffer()
                zbufs.append(self._z[i])
        for i in range(n):
            fused = getattr(self.b[i], "z_fused", None)
            if fused is not None and fused(zbufs[i], have_force, temperature, prefactor, self.seed2):
                continue
            if have_force:
                self.b[i].fft_z(zbufs[i], False)
            self.b[i].kspace(zbufs[i], have_force, temperature, prefactor, self.seed2)
            self.b[i].fft_z(zbufs[i], True)
        back = self.x.all_to_all([zb.view(P, nzl, 3, nyl, g.nkx, 2) for zb in zbufs])   # [src(y block)][zl][c][yl][kx]
        out, fields = [], []
        for i in range(n):
            # [src][zl][c][yl][kx] -> [zl][c][y = (src, yl)][kx], written straight into the owned planes of the window
            dst = self.b[i].spectrum_view(grids[i]).view(nzl, 3, P, nyl, g.nkx, 2)
            if not (P == 1 and back[i].data_ptr() == dst.data_ptr()):   # (one rank: the z pass ran in place on the window's own spectrum)
                dst.copy_(back[i].permute(1, 2, 0, 3, 4, 5))
            inv = getattr(self.b[i], "inverse_xy_inter", None)
            it = inv(wrap=(P == 1)) if inv is not None else None     # (a property of the grid and the library: every rank decides alike)
            if it is None:
                self.b[i].inverse_xy(grids[i])
            fields.append(it)
        if any(f is None for f in fields):            # (all ranks decide alike: same grid, same library)
            for i, f in enumerate(fields):
                if f is not None:
                    raise RuntimeError("ranks disagree on the gather layout")
            fields = grids
        wrapped = P == 1 and fields is not grids     # (one rank: the inverse x pass has stored the halo planes of the float4 window itself)
        if not wrapped:
            fd, fu = self.x.neighbours([gr[H + nzl - He:H + nzl] for gr in fields], [gr[H:H + He] for gr in fields],
                                       into_down=[gr[H - He:H] for gr in fields], into_up=[gr[H + nzl:H + nzl + He] for gr in fields])
        for i, gr in enumerate(fields):
            landed = not wrapped and fd[i].data_ptr() == gr[H - He:H].data_ptr() and fu[i].data_ptr() == gr[H + nzl:H + nzl + He].data_ptr()
            if not wrapped and not landed:   # (under torch.distributed the planes were received in place)
                pair = getattr(self.b[i], "pair", None)
                if pair is None or not pair(False, gr[H - He:H], fd[i], gr[H + nzl:H + nzl + He], fu[i]):
                    gr[H - He:H] = fd[i]
                    gr[H + nzl:H + nzl + He] = fu[i]
            out.append(self.b[i].gather(pos_locals[i], gr) if fields is grids else self.b[i].gather_inter(pos_locals[i], gr))
        return out


class DistributedFCMIntegrator:
    """BDHI::FCMIntegrator::forwardTime (BDHI_FCM.cu:95-119) on slabs, one rank per process: forces -> displacements ->
    pos += v dt -> migrate.  `forces_fn(pos_local, *carried) -> real4[N]` is the Interactor stack (None: noise only);
    `carried` are per-particle arrays (ids, fixed forces, ...) that migrate with the particles."""

    def __init__(self, fcm, decomp, temperature, dt, forces_fn, migrate_every=1):
        self.fcm, self.d = fcm, decomp
        self.temperature, self.dt, self.forces_fn = float(temperature), float(dt), forces_fn
        self.steps = 0
        # with spare halo planes the owned particles may overshoot their slab for a while: migrate (one host sync, a repack
        # of every per-particle array) only every `migrate_every` steps; check_drift() verifies that it was enough
        self.migrate_every = int(migrate_every) if fcm.g.drift_planes > 0 else 1
        self.max_drift = None

    def forward_time(self, pos_local, *carried):
        self.steps += 1
        force = self.forces_fn(pos_local, *carried)
        v = self.fcm.displacements([pos_local], [force], self.temperature, 1.0 / self.dt ** 0.5)[0]
        em = getattr(self.fcm.b[0], "euler_maruyama", None)  # integrateEulerMaruyamaD, BDHI_FCM.cu:67-92
        if em is None or not em(pos_local, v, self.dt):
            pos_local[:, :3] += v * self.dt
        if self.steps % self.migrate_every == 0:
            if self.migrate_every > 1:
                self.max_drift = (pos_local[:, 2].abs().max() - 0.5 * self.d.width).clamp(min=0.0)
            return self.d.migrate(pos_local, *carried)
        return (pos_local,) + tuple(carried)

    def check_drift(self):
        """Host check (synchronises): no particle may have left its slab by more than the spare halo planes allow."""
        g = self.fcm.g
        allowed = (g.drift_planes - 0.5) * g.L[2] / g.cells[2]
        if self.max_drift is not None and float(self.max_drift) > allowed:
            raise RuntimeError(f"a particle moved {float(self.max_drift):.3f} outside its slab between migrations, more than the "
                               f"{allowed:.3f} the spare halo planes allow: lower migrate_every")


def make_decomposition(geom, rank, group=None, comm=None):
    """Particle ownership that matches the grid slabs: z in [-Lz/2 + r Lz/P, ...), cut-off = the stencil reach."""
    hz = geom.L[2] / geom.cells[2]
    return SlabDecomposition(geom.L, geom.he * hz, rank, geom.world, group, comm=comm)
