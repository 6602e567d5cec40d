"""The communication stack of the multi-GPU drivers: uammd_comm_* (RCCL over xGMI behind the C ABI, csrc/comm.hip) through ctypes — the
SAME entry points include/uammd/Distributed.h (uammd::Comm) drives from C++.

`AbiComm` is what bench.py's N > 1 path and `--force-distributed` use: SlabDecomposition (uammd_amd/parallel.py) and the FCM slab
exchanges (uammd_amd/parallel_fcm.py) hand it device pointers; nothing is staged through the host and torch.distributed carries no
payload (it is only the bootstrap that hands rank 0's 128-byte RCCL id to the other processes, as MPI or a file would — the library
leaves that to the program: include/uammd_hip.h "Set-up").  The torch.distributed transport those modules fall back to when no `comm`
is given is a TEST DOUBLE: gloo on the CPU (tests/test_distributed*_cpu.py) and several ranks on one GPU with host-staged messages
(tests/test_gpu_world2.py; RCCL refuses two ranks on one device)."""
import contextlib
import ctypes as C
import os
import sys

import torch

from . import _lib


@contextlib.contextmanager
def _stdout_to_stderr():
    """librccl prints a version banner on file descriptor 1 when a communicator is made; a program whose stdout is its result (bench.py:
    one JSON line) sends it to stderr instead."""
    libc = C.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        libc.fflush(None)     # (the banner sits in the C library's stdout buffer until flushed: push it out while 1 still points at stderr)
        os.dup2(saved, 1)
        os.close(saved)


class AbiComm:
    def __init__(self, rank, world, unique_id):
        self.lib = _lib.load()
        self.rank, self.world = int(rank), int(world)
        self.h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), 128)
        with _stdout_to_stderr():
            _lib.check(self.lib.uammd_comm_init(C.byref(self.h), self.rank, self.world, buf))
            torch.cuda.synchronize()
        self.up, self.down = (self.rank + 1) % self.world, (self.rank - 1) % self.world
        self._scratch = None

    # ---- set-up ------------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def unique_id():
        lib = _lib.load()
        buf = C.create_string_buffer(128)
        with _stdout_to_stderr():
            _lib.check(lib.uammd_comm_unique_id(buf))
        return bytes(buf.raw)

    @staticmethod
    def rccl_version():
        """'major.minor.patch' of the librccl behind uammd_comm_* (None when it cannot be loaded or does not say)."""
        lib = _lib.load()
        v = C.c_int(0)
        with _stdout_to_stderr():
            if lib.uammd_comm_rccl_version(C.byref(v)) != 0:
                return None
        return "%d.%d.%d" % (v.value // 10000, (v.value // 100) % 100, v.value % 100)

    @classmethod
    def from_torch_distributed(cls, dist, group=None):
        """Bootstrap over an initialised torch.distributed process group: rank 0 makes the RCCL id, everybody gets it."""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        return cls(rank, world, box[0])

    def close(self):
        if self.h:
            self.lib.uammd_comm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _st():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else None

    def describe(self):
        return f"uammd_comm_* (include/uammd_hip.h): RCCL send/recv, all-to-all and all-reduce behind the C ABI, world {self.world}"

    # ---- the patterns ----------------------------------------------------------------------------------------------------------------------
    def exchange_counts(self, to_up, to_down):
        """Two message sizes to the two ring neighbours; returns (from_down, from_up).  Synchronises the stream."""
        to = (C.c_int * 2)(int(to_up), int(to_down))
        frm = (C.c_int * 2)(0, 0)
        _lib.check(self.lib.uammd_comm_exchange_counts(self.h, to, frm, self._st()))
        return int(frm[0]), int(frm[1])

    def exchange_counts_device(self, mine):
        """`mine` = int32 device tensor [to_up, to_down] (as uammd_slab_select leaves it); returns (to_up, to_down, from_down, from_up) as host
        ints with ONE stream synchronisation (the sizes travel from device memory; the host reads all four together)."""
        assert mine.is_cuda and mine.dtype == torch.int32 and mine.numel() == 2 and mine.is_contiguous()
        out = (C.c_int * 4)(0, 0, 0, 0)
        _lib.check(self.lib.uammd_comm_exchange_counts_device(self.h, self._p(mine), out, self._st()))
        return int(out[0]), int(out[1]), int(out[2]), int(out[3])

    def halo_exchange(self, send_up, send_down, from_down, from_up):
        """float32 rows: send_up -> rank + 1, send_down -> rank - 1; from_down / from_up are the landing tensors (their row counts are the
        message sizes).  Asynchronous on the current stream."""
        w = None
        for t in (send_up, send_down, from_down, from_up):
            if t is not None and t.numel() > 0:
                assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
                cols = t.numel() // t.shape[0]
                assert w is None or w == cols
                w = cols
        if w is None:
            return
        n = lambda t: 0 if t is None else int(t.shape[0])
        _lib.check(self.lib.uammd_comm_halo_exchange(self.h, self._p(send_up), n(send_up), self._p(send_down), n(send_down), self._p(from_down),
                                                     n(from_down), self._p(from_up), n(from_up), w, self._st()))

    def alltoall(self, send, recv):
        """send / recv: contiguous tensors [world, ...]; slice p of send goes to rank p, slice p of recv comes from rank p."""
        assert send.is_contiguous() and recv.is_contiguous() and send.shape[0] == self.world and send.numel() == recv.numel()
        per = send.numel() * send.element_size() // self.world
        _lib.check(self.lib.uammd_comm_alltoall(self.h, self._p(send), self._p(recv), per, self._st()))

    def allreduce_sum_(self, t):
        """In-place float32 sum over the ranks (device tensor)."""
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        _lib.check(self.lib.uammd_comm_allreduce_sum(self.h, self._p(t), t.numel(), self._st()))
        return t

    # ---- what a timing harness needs: barrier, host reductions (all through the all-reduce) -----------------------------------------------
    def _slots(self):
        if self._scratch is None:
            self._scratch = torch.zeros(2 * max(self.world, 1), dtype=torch.float32, device="cuda")
        return self._scratch

    def barrier(self):
        s = self._slots()
        s.zero_()
        self.allreduce_sum_(s)
        torch.cuda.current_stream().synchronize()

    def gather_host(self, value):
        """Every rank's number, on every rank.  One pair of float32 slots per rank (only its owner writes them, the sum over the ranks is
        a gather): the value's float32 rounding and the float32 rounding of what that left — 48 bits of it survive the float32 wire, so
        counts above 2^24 (particle-steps, records) and millisecond timings keep their digits."""
        import numpy as np
        s = self._slots()
        s.zero_()
        v = float(value)
        hi = float(np.float32(v))
        lo = float(np.float32(v - hi)) if np.isfinite(hi) else 0.0
        s[2 * self.rank] = hi
        s[2 * self.rank + 1] = lo
        self.allreduce_sum_(s)
        flat = s.cpu().tolist()
        return [float(flat[2 * r]) + float(flat[2 * r + 1]) for r in range(self.world)]

    def reduce_host(self, values, op):
        """values: list of floats; op: "MAX" or "SUM" over the ranks, element-wise, accumulated in double on the host (gather_host)."""
        cols = [self.gather_host(v) for v in values]
        return [max(c) if op == "MAX" else sum(c) for c in cols]
