"""Multi-GPU path A: z-slab domain decomposition of the short-range pair-force path.

The reference is single-GPU (no NCCL/MPI anywhere), so this is new design, not a mirror:
  * one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on MI355X, "gloo" in the CPU tests);
  * rank r owns the particles with z in [zlo_r, zhi_r) (equal slabs of the periodic global box);
  * the reference computes FULL per-particle forces (no Newton-3 halving, common.cuh:10-34), so only POSITIONS of a
    one-cutoff halo cross the slab faces and forces need no reduction: per force evaluation each rank sends the
    particles within rc of its two faces to its two neighbours (point-to-point, one xGMI link each) and receives
    their ghosts — no ring all-reduce, no all-gather of the whole system;
  * after the integration step the particles that left the slab migrate to the neighbour rank.
Everything here is device-agnostic torch (boolean masks, cat, P2P), so the same code runs on CPU tensors under gloo;
the force evaluation itself is a callback (`forces_fn`): the HIP cell list + LJ traversal on the GPU, the oracle in
the CPU tests.

Local frame: rank r works in coordinates z' = z - zc_r (zc_r = slab centre, periodic images chosen next to the
slab), with a local box (Lx, Ly, slab + 2 rc) that is periodic in x, y and NOT periodic in z.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist


class SlabDecomposition:
    def __init__(self, L, rc, rank=None, world=None, group=None, skin=0.0, comm=None):
        """comm: an uammd_amd.comm.AbiComm — every message then goes through uammd_comm_* (RCCL behind the C ABI, the stack of the C++
        drivers), also at world 1 where the ring closes on the rank itself; without it the torch.distributed test double carries them.
        skin = delta > 0 turns on the cached exchange: ownership and the halo membership lists are refreshed only every
        few steps (DistributedLJ.exchange_every), in between a rank re-sends the CURRENT positions of the listed particles —
        no size messages, no host synchronisation.  Valid while no particle moves more than delta between refreshes: the
        lists then hold everything within rc + 3 delta of a face (owners may sit up to delta outside their slab, so may the
        particle that needs them, and a listed particle may drift delta itself), and the local box is rc + 4 delta thick."""
        self.group = group
        self.comm = comm
        self.skin = float(skin)
        self._halo_cache = None
        self.rank = (comm.rank if comm is not None else dist.get_rank(group)) if rank is None else rank
        self.world = (comm.world if comm is not None else dist.get_world_size(group)) if world is None else world
        if comm is not None and (comm.rank != self.rank or comm.world != self.world):
            raise ValueError("the communicator's rank / world differ from the decomposition's")
        self.L = [float(x) for x in L]
        self.rc = float(rc)
        self.width = self.L[2] / self.world
        if self.world > 1 and self.width < self.rc + 3.0 * self.skin:
            raise ValueError("slab thinner than the cut-off (+ skin): halo would need second neighbours")
        self.zlo = -self.L[2] / 2 + self.rank * self.width
        self.zhi = self.zlo + self.width
        self.zc = 0.5 * (self.zlo + self.zhi)
        self.up = (self.rank + 1) % self.world
        self.down = (self.rank - 1) % self.world

    # ---- helpers ---------------------------------------------------------------------------------------
    def fold_z(self, z):
        """z folded into the global box [-Lz/2, Lz/2)."""
        Lz = self.L[2]
        return z - torch.floor(z / Lz + 0.5) * Lz

    def owner_of(self, z):
        r = torch.floor((self.fold_z(z) + self.L[2] / 2) / self.width).to(torch.int64)
        return torch.clamp(r, 0, self.world - 1)

    def local_box(self):
        """(L, periodic) of the local frame."""
        # 1 % slack so that a ghost sitting exactly on the halo face (or rounded one ulp past it) is still inside
        return [self.L[0], self.L[1], self.width + 2.02 * (self.rc + 4.0 * self.skin)], [True, True, False]

    def to_local(self, pos):
        """Global coordinates -> local frame (z relative to the slab centre, image nearest to the slab)."""
        out = pos.clone()
        dz = pos[:, 2] - self.zc
        Lz = self.L[2]
        out[:, 2] = dz - torch.floor(dz / Lz + 0.5) * Lz
        return out

    # Boolean-mask indexing and an all_gather of message sizes cost one host sync EACH (a handful per exchange).  Instead the
    # two message sizes go to the two neighbours only, and ONE host read returns all four numbers an exchange needs; the
    # rows are then selected with nonzero_static (static size, no sync).
    def _counts(self, mask_up, mask_down):
        """-> (n_to_up, n_to_down, n_from_down, n_from_up) as host ints, one device->host sync."""
        return self._exchange_counts(torch.stack([mask_up.sum(), mask_down.sum()]))

    def _exchange_counts(self, mine):
        """`mine` = tensor [n_to_up, n_to_down] (any device).  -> the same two numbers and what the neighbours send, as host ints."""
        if self.comm is not None:
            if mine.is_cuda and mine.dtype == torch.int32 and mine.is_contiguous():
                return self.comm.exchange_counts_device(mine)      # sizes straight from device memory: one host read for all four
            a, b = mine.tolist()
            c, d = self.comm.exchange_counts(a, b)
            return a, b, c, d
        if self.world == 1:  # the only rank is its own neighbour on both sides (periodic images through the loop-back)
            a, b = mine.tolist()
            return a, b, a, b
        # whole tensors, not views, as message buffers (a backend that stages through the host may not write a view back)
        to_up, to_down = self._wire(mine[0:1].clone()), self._wire(mine[1:2].clone())
        from_down, from_up = torch.zeros_like(to_up), torch.zeros_like(to_up)
        t1, t2 = (1, 2) if self.world == 2 else (0, 0)
        ops = [dist.P2POp(dist.isend, to_up, self.up, self.group, tag=t1),
               dist.P2POp(dist.irecv, from_down, self.down, self.group, tag=t1),
               dist.P2POp(dist.isend, to_down, self.down, self.group, tag=t2),
               dist.P2POp(dist.irecv, from_up, self.up, self.group, tag=t2)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        a, b, c, d = torch.cat([to_up, to_down, from_down, from_up]).tolist()
        return a, b, c, d

    def _wire(self, t):
        """Message buffers live where the backend can reach them: gloo moves host memory only (its use with device tensors is
        a single-GPU debugging aid, bench.py UAMMD_BENCH_BACKEND=gloo); RCCL sends device memory directly."""
        if t.is_cuda and dist.get_backend(self.group) == "gloo":
            return t.cpu()
        return t

    @staticmethod
    def _index(mask, count):
        """Indices of the set entries of `mask` when their number is already known on the host (no sync)."""
        if count == 0:
            return torch.zeros(0, dtype=torch.int64, device=mask.device)
        try:
            return torch.nonzero_static(mask, size=count).flatten()
        except (RuntimeError, NotImplementedError):
            return torch.nonzero(mask).flatten()

    @staticmethod
    def _select(rows, mask, count):
        """rows[mask] when the number of hits is already known on the host (no sync)."""
        return rows.index_select(0, SlabDecomposition._index(mask, count))

    def _exchange(self, send_up, send_down, n_from_down, n_from_up):
        """Sends `send_up` to rank+1 and `send_down` to rank-1, returns (from_down, from_up).  Rows are float32; the
        receive sizes come from _counts."""
        ncol = send_up.shape[1]
        dev = send_up.device
        if self.comm is not None:
            from_down = torch.empty((n_from_down, ncol), dtype=torch.float32, device=dev)
            from_up = torch.empty((n_from_up, ncol), dtype=torch.float32, device=dev)
            self.comm.halo_exchange(send_up.contiguous(), send_down.contiguous(), from_down, from_up)
            return from_down, from_up
        if self.world == 1:
            return send_up, send_down
        send_up, send_down = self._wire(send_up), self._wire(send_down)
        from_down = torch.empty((n_from_down, ncol), dtype=send_up.dtype, device=send_up.device)
        from_up = torch.empty((n_from_up, ncol), dtype=send_up.dtype, device=send_up.device)
        # zero-row messages are skipped on both sides (each side knows the size of what it sends and receives); with 2 ranks
        # both neighbours are the same peer and the two messages are told apart by tag (gloo) / by issue order (RCCL)
        t1, t2 = (1, 2) if self.world == 2 else (0, 0)
        ops = []
        if send_up.shape[0] > 0:
            ops.append(dist.P2POp(dist.isend, send_up.contiguous(), self.up, self.group, tag=t1))
        if n_from_down > 0:
            ops.append(dist.P2POp(dist.irecv, from_down, self.down, self.group, tag=t1))
        if send_down.shape[0] > 0:
            ops.append(dist.P2POp(dist.isend, send_down.contiguous(), self.down, self.group, tag=t2))
        if n_from_up > 0:
            ops.append(dist.P2POp(dist.irecv, from_up, self.up, self.group, tag=t2))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return from_down.to(dev), from_up.to(dev)

    # ---- halo ---------------------------------------------------------------------------------------------
    def halo_exchange(self, pos_local, refresh=True):
        """pos_local: real4[N] of the OWNED particles in the local frame.  Returns real4[N+G]: owned particles followed by
        the ghosts received from below and above, already shifted into this frame.  refresh=False re-uses the membership
        lists (and the message sizes) of the last refresh: nothing but the payload moves, and nothing synchronises."""
        half = 0.5 * self.width
        reach = self.rc + 3.0 * self.skin
        if refresh or self._halo_cache is None:
            z = pos_local[:, 2]
            up_mask = z >= half - reach
            down_mask = z < -half + reach
            n_up, n_down, n_from_down, n_from_up = self._counts(up_mask, down_mask)
            idx_up = self._index(up_mask, n_up)
            idx_down = self._index(down_mask, n_down)
            self._halo_cache = (idx_up, idx_down, n_from_down, n_from_up)
        idx_up, idx_down, n_from_down, n_from_up = self._halo_cache
        send_up = pos_local.index_select(0, idx_up)
        send_down = pos_local.index_select(0, idx_down)
        # into the receiver's frame: its centre is one slab width above / below mine
        send_up[:, 2] -= self.width
        send_down[:, 2] += self.width
        from_down, from_up = self._exchange(send_up, send_down, n_from_down, n_from_up)
        return torch.cat([pos_local, from_down, from_up], dim=0), n_from_down + n_from_up

    def halo_refresh_into(self, allpos, n_owned):
        """Refresh of the membership lists on a persistent buffer: rows [0, n_owned) of `allpos` are the owned particles; the
        ghosts are written behind them.  Returns the number of ghosts.  One host read (the four message sizes)."""
        half = 0.5 * self.width
        reach = self.rc + 3.0 * self.skin
        z = allpos[:n_owned, 2]
        up_mask = z >= half - reach
        down_mask = z < -half + reach
        n_up, n_down, n_from_down, n_from_up = self._counts(up_mask, down_mask)
        assert n_owned + n_from_down + n_from_up <= allpos.shape[0], "the halo overflows the position buffer"
        self._halo_cache = (self._index(up_mask, n_up), self._index(down_mask, n_down), n_from_down, n_from_up)
        self._idx32 = None
        self.halo_refill(allpos, n_owned)
        return n_from_down + n_from_up

    def halo_refill(self, allpos, n_owned, pack=None):
        """Between refreshes: rows [0, n_owned) of `allpos` are the owned particles (integrated in place), rows [n_owned, ...)
        the ghosts of the last refresh in the order (from below, from above).  Re-sends the current positions of the listed
        particles and lets the ghosts land in the tail of `allpos` itself: no concatenation, no allocation, nothing
        synchronises.  On the GPU the two gathers + frame shifts are one kernel (uammd_halo_pack)."""
        idx_up, idx_down, n_from_down, n_from_up = self._halo_cache
        n_up, n_down = idx_up.shape[0], idx_down.shape[0]
        pos = allpos[:n_owned]
        tail_down = allpos[n_owned:n_owned + n_from_down]
        tail_up = allpos[n_owned + n_from_down:n_owned + n_from_down + n_from_up]
        loop = self.world == 1 and self.comm is None   # the only rank is its own neighbour: what goes up arrives from below
        if allpos.is_cuda:
            from . import _lib
            lib = _lib.load()
            if loop:
                out_up, out_down = tail_down, tail_up
            else:
                if getattr(self, "_send", None) is None or self._send[0].shape[0] < n_up or self._send[1].shape[0] < n_down:
                    self._send = (torch.empty((max(n_up, 1), 4), dtype=torch.float32, device=allpos.device),
                                  torch.empty((max(n_down, 1), 4), dtype=torch.float32, device=allpos.device))
                out_up, out_down = self._send[0][:n_up], self._send[1][:n_down]
            if getattr(self, "_idx32", None) is None or self._idx32[2] is not idx_up:  # (int32 copies of the lists, once per refresh)
                self._idx32 = (idx_up.to(torch.int32), idx_down.to(torch.int32), idx_up)
            if pack is not None:   # the caller's pack (uammd_halo_pack_gj1: the half step of the listed rows on the way)
                pack(self._idx32[0], n_up, self._idx32[1], n_down, -self.width, self.width, out_up, out_down)
            else:
                _lib.check(lib.uammd_halo_pack(pos.data_ptr(), self._idx32[0].data_ptr(), n_up, self._idx32[1].data_ptr(), n_down,
                                               -self.width, self.width, out_up.data_ptr(), out_down.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream))
            if loop:
                return
            send_up, send_down = out_up, out_down
            if self.comm is not None:   # straight into the ghost tail, on the stream: nothing is staged, nothing waits
                self.comm.halo_exchange(send_up, send_down, tail_down, tail_up)
                return
        else:
            send_up = pos.index_select(0, idx_up)
            send_down = pos.index_select(0, idx_down)
            send_up[:, 2] -= self.width
            send_down[:, 2] += self.width
            if loop:
                tail_down.copy_(send_up)
                tail_up.copy_(send_down)
                return
        send_up, send_down = self._wire(send_up), self._wire(send_down)
        staged = send_up.device != allpos.device      # gloo with device tensors (debugging aid): receive on the host, copy back
        rdown = torch.empty((n_from_down, 4), dtype=torch.float32, device=send_up.device) if staged else tail_down
        rup = torch.empty((n_from_up, 4), dtype=torch.float32, device=send_up.device) if staged else tail_up
        t1, t2 = (1, 2) if self.world == 2 else (0, 0)
        ops = []
        if n_up > 0:
            ops.append(dist.P2POp(dist.isend, send_up.contiguous(), self.up, self.group, tag=t1))
        if n_from_down > 0:
            ops.append(dist.P2POp(dist.irecv, rdown, self.down, self.group, tag=t1))
        if n_down > 0:
            ops.append(dist.P2POp(dist.isend, send_down.contiguous(), self.down, self.group, tag=t2))
        if n_from_up > 0:
            ops.append(dist.P2POp(dist.irecv, rup, self.up, self.group, tag=t2))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if staged:
            tail_down.copy_(rdown)
            tail_up.copy_(rup)

    # ---- migration ------------------------------------------------------------------------------------------
    def migrate(self, pos_local, *others):
        """Moves the particles that left the slab (|z'| > width/2 after integration) to the neighbour ranks together
        with the per-particle arrays in `others` (vel, id, ...).  Returns the new (pos_local, *others)."""
        z = pos_local[:, 2]
        half = 0.5 * self.width
        go_up = z >= half
        go_down = z < -half
        n_up, n_down, n_from_down, n_from_up = self._counts(go_up, go_down)
        self._halo_cache = None  # ownership changes: the membership lists are stale
        # int32 arrays (particle ids) travel bit-cast to float32 in the same message
        cols = [pos_local] + [(o.to(torch.int32).view(torch.float32) if o.dtype != torch.float32 else o).reshape(o.shape[0], -1)
                              for o in others]
        widths = [c.shape[1] for c in cols]
        packed = torch.cat(cols, dim=1)
        up_rows = self._select(packed, go_up, n_up)
        down_rows = self._select(packed, go_down, n_down)
        up_rows[:, 2] -= self.width
        down_rows[:, 2] += self.width
        from_down, from_up = self._exchange(up_rows, down_rows, n_from_down, n_from_up)
        kept = self._select(packed, ~(go_up | go_down), packed.shape[0] - n_up - n_down) if n_up + n_down > 0 else packed
        new = torch.cat([kept, from_down, from_up], dim=0) if n_from_down + n_from_up > 0 else kept
        out, c0 = [], 0
        for w, ref in zip(widths, [pos_local] + list(others)):
            block = new[:, c0:c0 + w]
            c0 += w
            block = block.contiguous()
            if ref.dtype != torch.float32:
                block = block.view(torch.int32).to(ref.dtype)
            out.append(block.reshape((new.shape[0],) + tuple(ref.shape[1:])).contiguous())
        return tuple(out)

    def migrate_inplace(self, bufs, n):
        """The same migration on persistent buffers, touching only the rows that change.  `bufs` = per-particle arrays with spare
        capacity (rows beyond n are scratch: the ghost tail of the position buffer, ...), the first one the positions in the
        local frame.  Leavers are packed and sent; arrivals fill the holes they leave; if more left than arrived the remaining
        holes are filled from the tail, if more arrived they are appended.  Returns the new number of owned rows.  The order of
        the owned particles changes (ids travel with them).  One host read (the four message sizes); everything else is
        asynchronous work on index lists of a few thousand rows instead of gathers of the whole arrays."""
        pos = bufs[0]
        z = pos[:n, 2]
        half = 0.5 * self.width
        go_up = z >= half
        go_down = z < -half
        n_up, n_down, n_from_down, n_from_up = self._counts(go_up, go_down)
        self._halo_cache = None
        n_leave, n_arrive = n_up + n_down, n_from_down + n_from_up
        if n_leave == 0 and n_arrive == 0:
            return n
        idx_up, idx_down = self._index(go_up, n_up), self._index(go_down, n_down)
        cols = [(b.to(torch.int32).view(torch.float32) if b.dtype != torch.float32 else b) for b in bufs]
        widths = [c.reshape(c.shape[0], -1).shape[1] for c in cols]
        up_rows = torch.cat([c.reshape(c.shape[0], -1).index_select(0, idx_up) for c in cols], dim=1)
        down_rows = torch.cat([c.reshape(c.shape[0], -1).index_select(0, idx_down) for c in cols], dim=1)
        up_rows[:, 2] -= self.width
        down_rows[:, 2] += self.width
        from_down, from_up = self._exchange(up_rows, down_rows, n_from_down, n_from_up)
        arrivals = torch.cat([from_down, from_up], dim=0)
        holes = torch.cat([idx_up, idx_down]).sort().values          # ascending
        new_n = n - n_leave + n_arrive
        assert new_n <= min(b.shape[0] for b in bufs), "migration overflows the particle buffers"
        if n_arrive >= n_leave:
            dest = torch.cat([holes, torch.arange(n, new_n, device=pos.device, dtype=holes.dtype)])
            src_rows = arrivals
        else:
            k = n_leave - n_arrive
            dest_a = holes[:n_arrive]
            rest = holes[n_arrive:]                                   # k holes still open
            # pair the open holes below new_n with the rows of the tail [new_n, n) that stay: both lists have the same (host-unknown)
            # length m <= k; sorted so that they come first, the other k - m pairs copy dropped rows onto dropped rows
            leaving = go_up | go_down
            tail = torch.arange(new_n, n, device=pos.device, dtype=holes.dtype)
            tail = tail[torch.argsort(leaving[new_n:n].to(torch.int8), stable=True)]
            rest = rest[torch.argsort((rest >= new_n).to(torch.int8), stable=True)]
            packed_tail = torch.cat([c.reshape(c.shape[0], -1).index_select(0, tail) for c in cols], dim=1)
            dest = torch.cat([dest_a, rest])
            src_rows = torch.cat([arrivals, packed_tail], dim=0)
        c0 = 0
        for b, c, w in zip(bufs, cols, widths):
            block = src_rows[:, c0:c0 + w].contiguous()
            c0 += w
            if b.dtype != torch.float32:
                block = block.view(torch.int32).to(b.dtype)
            b.reshape(b.shape[0], -1).index_copy_(0, dest, block.reshape(block.shape[0], -1))
        return new_n

    # ---- distribution of an initial configuration ------------------------------------------------------------
    def scatter_initial(self, pos_global):
        """Every rank passes the same global configuration; returns the owned subset in the local frame + global ids."""
        own = self.owner_of(pos_global[:, 2]) == self.rank
        ids = torch.nonzero(own, as_tuple=False).flatten().to(torch.int32)
        return self.to_local(pos_global[own]), ids


class DistributedLJ:
    """VerletNVT::GronbechJensen + PairForces<LJ> on a z-slab decomposition.  `forces_fn(pos_all, box_L, box_periodic)`
    returns real4 forces for every row of pos_all (owned + ghosts); `integrate_fn(step, pos, vel, force, step_num)` is
    the GJ kernel on the owned particles.

    With `forces_into(pos_all, box_L, box_periodic, force_all)` (accumulates into the given buffer) the step works on
    PERSISTENT buffers with spare capacity: the owned positions are integrated in place at the head of one buffer whose tail
    receives the ghosts, forces accumulate into the rows GJ step 1 has just zeroed (GronbechJensen.cu:55-57), a refresh moves
    only the particles that change rank (SlabDecomposition.migrate_inplace).  Nothing is allocated or concatenated per step;
    a refresh costs two host reads of message sizes."""

    def __init__(self, decomp, forces_fn, integrate_fn, exchange_every=1, forces_into=None, capacity_factor=1.25, forces_step2_into=None,
                 integrate_rows_fn=None, step1_fused=None):
        self.d, self.forces_fn, self.integrate_fn, self.forces_into = decomp, forces_fn, integrate_fn, forces_into
        # optional: forces_step2_into(allpos, box_L, periodic, fall, vel) = the forces AND the integrator's second half step of the owned
        # rows in one call (uammd_lj_transverse_celllist_gj2: the half step rides in the traversal's store); persistent mode only
        self.forces_step2_into = forces_step2_into
        # optional: integrate_rows_fn(step, pos, vel, force, rows, keys, step_num) = the integrator's half step on the rows `rows` (int32; None =
        # rows 0 .. len(keys) - 1) with the thermostat keyed by `keys` (their global ids).  When given it is the ONLY integrator the persistent
        # path calls (split and unsplit steps alike).  With it (persistent mode, GPU, a communicator, cached lists) the halo exchange
        # of a step between refreshes OVERLAPS the first half step: the listed particles (the ones the neighbours need) are integrated
        # first, their positions packed and sent on a side stream while the main stream integrates everybody else; the list build waits
        # for the ghosts.  Same arithmetic per particle: same bits as the unsplit step.
        self.integrate_rows_fn = integrate_rows_fn
        # optional: step1_fused = (pack_step1(pos, vel, force, keys, idx_up, n_up, idx_down, n_down, dz_up, dz_down, out_up, out_down, step_num),
        #                         forces_step12_into(allpos, box_L, periodic, fall, vel, keys, skip, step_num)):
        # between refreshes the first half step runs inside the halo pack (the listed rows: uammd_halo_pack_gj1) and inside the list build
        # (everybody else: uammd_celllist_update_gj1, `skip` = the listed rows' byte mask), the second one in the traversal's store — a
        # step is pack, exchange, build, traversal on ONE stream.  Needs a slab wider than two reaches (disjoint lists) and the library
        # refresh (it writes the mask); otherwise the step runs unfused with integrate_fn, which must then be the same arithmetic.
        self.step1_fused = step1_fused
        self._listed_mask = None
        self._split = None      # (listed rows, their keys, the other rows, their keys) of the current membership lists
        self._side = None       # side stream + events of the overlapped exchange
        self.steps = 0
        self.current_ids = None   # the owned rows' global ids, set before every integrate_fn call
        self.exchange_every = int(exchange_every) if decomp.skin > 0 else 1
        self.capacity_factor = capacity_factor
        self.max_drift = None   # device scalar: largest displacement of an owned particle between two refreshes (skin check)
        self._allpos = None     # owned + ghost positions of the current membership lists
        self._fall = None       # forces for the same rows
        self._ref = None        # owned positions at the last refresh
        self._bufs = None       # persistent mode: [pos (owned + ghosts), vel, ids, force] with spare rows
        self._nall = 0

    # ---- generic mode (any backend, allocates per step) ---------------------------------------------------------------------
    def compute_forces(self, pos_local, refresh=True):
        L, per = self.d.local_box()
        n = pos_local.shape[0]
        self.n_owned = n  # rows [n_owned, ...) are ghosts: the force callbacks may skip their forces
        if refresh or self._allpos is None:
            self._allpos, _ = self.d.halo_exchange(pos_local, True)
        else:
            self.d.halo_refill(self._allpos, n)
        return self.forces_fn(self._allpos, L, per)[:n]

    def _track_drift(self, pos):
        if self.d.skin > 0 and self._ref is not None and self._ref.shape[0] == pos.shape[0]:
            # the cached exchange is exact while NO particle has moved more than the skin since the membership lists were made
            # (an in-slab particle that crosses the skin towards a face is missing from the lists just as well)
            moved = (pos[:, :3] - self._ref[:, :3]).norm(dim=1).max()
            self.max_drift = moved if self.max_drift is None else torch.maximum(self.max_drift, moved)

    def forward_time(self, pos, vel, force, ids):
        """One step; returns the (possibly re-sized) owned arrays (views of the step's own buffers in persistent mode: pass
        back what the previous call returned)."""
        if self.forces_into is not None:
            return self._forward_persistent(pos, vel, force, ids)
        self.steps += 1
        if self.steps == 1:
            force = self.compute_forces(pos)
            pos = self._allpos[:pos.shape[0]]
            if self.d.skin > 0:
                self._ref = pos.clone()
        self.current_ids = ids     # (for callbacks that key the thermostat's stream on the global id: uammd_verletnvt_gj_keyed)
        self.integrate_fn(1, pos, vel, force, self.steps)
        refresh = (self.steps - 1) % self.exchange_every == 0 or self.d._halo_cache is None or self._allpos is None
        if refresh:
            self._track_drift(pos)
            pos, vel, ids = self.d.migrate(pos, vel, ids)
        force = self.compute_forces(pos, refresh)
        if refresh:
            pos = self._allpos[:pos.shape[0]]     # from here on the owned particles live at the head of the exchange buffer
            if self.d.skin > 0:
                self._ref = pos.clone()
        self.current_ids = ids
        self.integrate_fn(2, pos, vel, force, self.steps)
        return pos, vel, force, ids

    # ---- persistent mode ----------------------------------------------------------------------------------------------------------
    def _adopt(self, pos, vel, ids):
        """(Re)builds the persistent buffers from plain arrays (first step, after a sort of the owned particles)."""
        n = pos.shape[0]
        reach = self.d.rc + 3.0 * self.d.skin
        ghosts = int(2.2 * n * reach / self.d.width) + 4096                      # both halos of a uniform density, with slack
        cap = int(self.capacity_factor * n) + ghosts
        dev = pos.device
        bp = torch.zeros((cap, 4), dtype=torch.float32, device=dev)
        bv = torch.zeros((cap,) + tuple(vel.shape[1:]), dtype=vel.dtype, device=dev)
        bi = torch.zeros((cap,) + tuple(ids.shape[1:]), dtype=ids.dtype, device=dev)
        bf = torch.zeros((cap, 4), dtype=torch.float32, device=dev)
        bp[:n], bv[:n], bi[:n] = pos, vel, ids
        self._bufs = [bp, bv, bi, bf]
        self._ref = None
        if getattr(self, "_slab_ws", None) is not None:
            self._slab_ws["ref_n"] = -1      # the order changed: the stored reference positions are another permutation
        return n

    def _refresh_persistent(self, n):
        bp, bv, bi, bf = self._bufs
        self._split = None
        self._listed_mask = None
        if bp.is_cuda and bv.dtype == torch.float32 and bv.dim() == 2 and bv.shape[1] == 3 and bi.dtype == torch.int32 and bi.dim() == 1:
            return self._refresh_fused(n)
        self._track_drift(bp[:n])
        n = self.d.migrate_inplace([bp, bv, bi], n)
        bf[:n].zero_()                                # arrivals and moved rows: GJ step 1 zeroed the forces of the old layout
        g = self.d.halo_refresh_into(bp, n)
        self._nall = n + g
        if self.d.skin > 0:
            self._ref = bp[:n].clone()
        return n

    def _refresh_fused(self, n):
        """The same refresh on the GPU through the library's slab kernels (csrc/slab.hip): ordered selection of leavers and of
        halo members, packed migration rows, arrivals into the holes, the skin check's displacement — a dozen launches and two
        host reads (the message sizes) instead of ~60 element-wise / compaction / concatenation launches.  Lists are ascending
        as torch.nonzero's, so owned rows and ghosts come out in the same order as from the generic path."""
        from . import _lib
        lib = _lib.load()
        d = self.d
        bp, bv, bi, bf = self._bufs
        cap, dev = bp.shape[0], bp.device
        st = torch.cuda.current_stream().cuda_stream
        ws = getattr(self, "_slab_ws", None)
        if ws is None or ws["cap"] != cap:
            nbytes = C.c_size_t(0)
            _lib.check(lib.uammd_slab_select_workspace(cap, C.byref(nbytes)))
            ws = dict(cap=cap, idx=torch.empty((4, cap), dtype=torch.int32, device=dev), holes=torch.empty(cap, dtype=torch.int32, device=dev),
                      counts=torch.zeros(4, dtype=torch.int32, device=dev), tiles=torch.empty(int(nbytes.value), dtype=torch.uint8, device=dev),
                      rows=torch.empty((cap, 8), dtype=torch.float32, device=dev), ref=torch.empty((cap, 4), dtype=torch.float32, device=dev),
                      maxd=torch.zeros(1, dtype=torch.float32, device=dev), ref_n=-1)
            self._slab_ws = ws
        idx, counts = ws["idx"], ws["counts"]
        p = lambda t: C.c_void_p(t.data_ptr())
        if (d.comm is not None or d.world == 1) and os.environ.get("UAMMD_SLAB_REFRESH", "c") == "c" and self.integrate_rows_fn is None:
            # the whole refresh as ONE library call (uammd_slab_refresh_lj: the same kernels and messages in the same order, two host
            # reads, no interpreter between the launches)
            if "arrivals" not in ws:
                ws["arrivals"] = torch.empty((cap, 8), dtype=torch.float32, device=dev)
                ws["send"] = torch.empty((cap, 4), dtype=torch.float32, device=dev)
            out = (C.c_int * 10)()
            use_ref = d.skin > 0
            want_mask = self.step1_fused is not None and d.width > 2.0 * (d.rc + 3.0 * d.skin)
            if want_mask and "listed" not in ws:
                ws["listed"] = torch.zeros(cap, dtype=torch.uint8, device=dev)
            had_ref = use_ref and ws["ref_n"] == n
            _lib.check(lib.uammd_slab_refresh_lj(d.comm.h if d.comm is not None else None, p(bp), p(bv), p(bi), p(bf), n, cap, d.width,
                                                 d.rc + 3.0 * d.skin, p(idx), p(ws["holes"]), p(counts), p(ws["tiles"]), p(ws["rows"]),
                                                 p(ws["arrivals"]), p(ws["send"]), p(ws["ref"]) if use_ref else None, ws["ref_n"],
                                                 p(ws["maxd"]) if use_ref else None, p(ws["listed"]) if want_mask else None, out, st))
            self._listed_mask = ws["listed"] if want_mask else None
            if had_ref:
                self.max_drift = ws["maxd"][0]
            n, nall, h_up, h_down, g_from_down, g_from_up = (int(x) for x in out[:6])
            iu, idn = idx[2][:h_up], idx[3][:h_down]
            d._halo_cache = (iu, idn, g_from_down, g_from_up)
            d._idx32 = (iu, idn, iu)
            self._nall = nall
            self._split = None
            if use_ref:
                ws["ref_n"] = n
            return n
        if d.skin > 0 and ws["ref_n"] == n:
            _lib.check(lib.uammd_slab_max_displacement(p(bp), p(ws["ref"]), n, p(ws["maxd"]), st))
            self.max_drift = ws["maxd"][0]
        half = 0.5 * d.width
        # ---- who leaves ----
        _lib.check(lib.uammd_slab_select(p(bp), n, half, -half, p(idx[0]), p(idx[1]), p(counts), p(ws["tiles"]), st))
        n_up, n_down, n_from_down, n_from_up = d._exchange_counts(counts[:2])
        d._halo_cache = None
        n_leave, n_arrive = n_up + n_down, n_from_down + n_from_up
        if n_leave or n_arrive:
            rows = ws["rows"]
            assert n_leave <= cap and n_arrive <= cap, "migration overflows the exchange buffer"
            send_up, send_down = rows[:n_up], rows[n_up:n_leave]
            _lib.check(lib.uammd_slab_pack_rows(p(bp), p(bv), p(bi), p(idx[0]), n_up, p(idx[1]), n_down, -d.width, d.width,
                                                p(send_up) if n_up else None, p(send_down) if n_down else None, st))
            if d.world == 1 and d.comm is None:
                arrivals = rows            # what goes up arrives from below: [from_down | from_up] = [send_up | send_down]
            else:
                from_down, from_up = d._exchange(send_up, send_down, n_from_down, n_from_up)
                arrivals = torch.cat([from_down, from_up], dim=0).contiguous()
            new_n = n - n_leave + n_arrive
            assert new_n <= cap, "migration overflows the particle buffers"
            _lib.check(lib.uammd_slab_unpack_rows(p(bp), p(bv), p(bi), n, p(idx[0]), n_up, p(idx[1]), n_down, p(arrivals), n_arrive,
                                                  p(ws["holes"]), st))
            n = new_n
        bf[:n].zero_()                                # arrivals and moved rows: GJ step 1 zeroed the forces of the old layout
        # ---- who is in the halo ----
        reach = d.rc + 3.0 * d.skin
        _lib.check(lib.uammd_slab_select(p(bp), n, half - reach, -half + reach, p(idx[2]), p(idx[3]), p(counts[2:]), p(ws["tiles"]), st))
        h_up, h_down, g_from_down, g_from_up = d._exchange_counts(counts[2:])
        assert n + g_from_down + g_from_up <= cap, "the halo overflows the position buffer"
        iu, idn = idx[2][:h_up], idx[3][:h_down]
        d._halo_cache = (iu, idn, g_from_down, g_from_up)
        d._idx32 = (iu, idn, iu)
        d.halo_refill(bp, n)
        self._nall = n + g_from_down + g_from_up
        self._split = None
        if self.step1_fused is not None and d.width > 2.0 * reach:   # the listed rows' byte mask (the library refresh writes it itself)
            if "listed" not in ws:
                ws["listed"] = torch.zeros(cap, dtype=torch.uint8, device=dev)
            m = ws["listed"]
            m[:n].zero_()
            m[iu.long()] = 1
            m[idn.long()] = 1
            self._listed_mask = m
        if self.integrate_rows_fn is not None and d.comm is not None and d.width > 2.0 * reach and 0 < h_up + h_down < n:
            # (up and down lists are disjoint when the slab is wider than two reaches: their concatenation lists every row once)
            listed = torch.cat([iu, idn])
            mask = torch.ones(n, dtype=torch.bool, device=dev)
            mask[listed.long()] = False
            rest = torch.nonzero_static(mask, size=n - (h_up + h_down)).flatten().to(torch.int32)
            self._split = (listed, bi.index_select(0, listed.long()), rest, bi.index_select(0, rest.long()))
        if d.skin > 0:
            ws["ref"][:n].copy_(bp[:n])
            ws["ref_n"] = n
        return n

    def _forces_persistent(self, n):
        bp, bv, bi, bf = self._bufs
        L, per = self.d.local_box()
        self.n_owned = n
        self.forces_into(bp[:self._nall], L, per, bf[:self._nall])

    def _forward_persistent(self, pos, vel, force, ids):
        self.steps += 1
        n = pos.shape[0]
        if self._bufs is None or pos.data_ptr() != self._bufs[0].data_ptr():   # plain arrays handed in: first step, or after a sort
            n = self._adopt(pos, vel, ids)
            n = self._refresh_persistent(n)
            self._forces_persistent(n)
        bp, bv, bi, bf = self._bufs
        self.current_ids = bi[:n]
        refresh = (self.steps - 1) % self.exchange_every == 0 or self.d._halo_cache is None
        if not refresh and self.step1_fused is not None and self._listed_mask is not None:
            pack_step1, forces_step12_into = self.step1_fused
            step = self.steps
            self.d.halo_refill(bp, n, pack=lambda iu, nu, idn, nd, dzu, dzd, ou, od: pack_step1(bp, bv, bf, bi, iu, nu, idn, nd, dzu, dzd,
                                                                                                ou, od, step))
            L, per = self.d.local_box()
            self.n_owned = n
            forces_step12_into(bp[:self._nall], L, per, bf[:self._nall], bv[:n], bi[:n], self._listed_mask, step)
            return bp[:n], bv[:n], bf[:n], bi[:n]
        if not refresh and self._split is not None:
            listed, lkeys, rest, rkeys = self._split
            if self._side is None:
                self._side = (torch.cuda.Stream(), torch.cuda.Event(), torch.cuda.Event())
            side, ev_listed, ev_ghosts = self._side
            main = torch.cuda.current_stream()
            self.integrate_rows_fn(1, bp, bv, bf, listed, lkeys, self.steps)
            ev_listed.record(main)
            self.integrate_rows_fn(1, bp, bv, bf, rest, rkeys, self.steps)   # everybody else takes the half step, while ...
            with torch.cuda.stream(side):
                side.wait_event(ev_listed)
                self.d.halo_refill(bp, n)          # ... the listed positions are packed, sent, and the ghosts land in the tail
                ev_ghosts.record(side)
            main.wait_event(ev_ghosts)
        else:
            # (ONE callback defines the arithmetic of the half step when the overlapped form is armed: the unsplit steps call it too —
            # rows None = all n owned rows, keyed by their global ids — so that a caller's integrate_fn keyed any other way cannot give the
            # refresh steps a different noise stream than the split steps)
            if self.integrate_rows_fn is not None:
                self.integrate_rows_fn(1, bp[:n], bv[:n], bf[:n], None, bi[:n], self.steps)
            else:
                self.integrate_fn(1, bp[:n], bv[:n], bf[:n], self.steps)
            if refresh:
                n = self._refresh_persistent(n)
            else:
                self.d.halo_refill(bp, n)
        self.current_ids = bi[:n]
        if self.forces_step2_into is not None:
            L, per = self.d.local_box()
            self.n_owned = n
            self.forces_step2_into(bp[:self._nall], L, per, bf[:self._nall], bv[:n])
        else:
            self._forces_persistent(n)
            if self.integrate_rows_fn is not None:
                self.integrate_rows_fn(2, bp[:n], bv[:n], bf[:n], None, bi[:n], self.steps)
            else:
                self.integrate_fn(2, bp[:n], bv[:n], bf[:n], self.steps)
        return bp[:n], bv[:n], bf[:n], bi[:n]

    def reordered(self):
        """Call after permuting the owned arrays (a sort): the cached membership lists index the old order."""
        self.d._halo_cache = None
        self._allpos = None
        self._ref = None
        self._bufs = None
        self._split = None
        self._listed_mask = None

    def check_skin(self):
        """Host check (synchronises): the cached exchange is exact only if nobody out-ran the skin."""
        if self.max_drift is not None and float(self.max_drift) > self.d.skin:
            raise RuntimeError(f"a particle moved {float(self.max_drift):.3f} between refreshes of the halo membership lists, more "
                               f"than the skin {self.d.skin}: lower exchange_every or raise the skin")
