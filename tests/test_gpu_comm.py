"""GPU test of the RCCL-backed communication entry points (uammd_comm_*, include/uammd_hip.h) with a world of ONE rank: the only
configuration a single-GPU box can run.  The ring closes on the rank itself — what it sends up arrives from below — which exercises
the library loading, communicator set-up, grouped send / recv, the size exchange, the all-to-all and the all-reduce exactly as an
N-GPU run issues them; the N > 1 message pattern itself is covered by the gloo tests of uammd_amd/parallel*.py."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_comm_world_of_one(hip):
    from uammd_amd._lib import check, load
    lib = load()
    uid = C.create_string_buffer(128)
    check(lib.uammd_comm_unique_id(uid))
    comm = C.c_void_p()
    check(lib.uammd_comm_init(C.byref(comm), 0, 1, uid))
    try:
        assert lib.uammd_comm_rank(comm) == 0 and lib.uammd_comm_world(comm) == 1
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        # sizes: up -> from below, down -> from above
        to = (C.c_int * 2)(37, 5)
        frm = (C.c_int * 2)(0, 0)
        check(lib.uammd_comm_exchange_counts(comm, to, frm, st))
        assert list(frm) == [37, 5]
        # the same with the sizes in device memory: all four numbers from one read
        dev2 = torch.tensor([41, 9], dtype=torch.int32, device="cuda")
        all4 = (C.c_int * 4)(0, 0, 0, 0)
        check(lib.uammd_comm_exchange_counts_device(comm, C.c_void_p(dev2.data_ptr()), all4, st))
        assert list(all4) == [41, 9, 41, 9]
        # halo: pack two index lists with the frame shift, exchange, land in the tail of the position array
        n = 1000
        pos = torch.rand((n + 42, 4), dtype=torch.float32, device="cuda")
        idx_up = torch.arange(0, 37, dtype=torch.int32, device="cuda") * 3
        idx_down = torch.arange(0, 5, dtype=torch.int32, device="cuda") * 7 + 1
        send_up = torch.empty((37, 4), dtype=torch.float32, device="cuda")
        send_down = torch.empty((5, 4), dtype=torch.float32, device="cuda")
        check(lib.uammd_halo_pack(pos.data_ptr(), idx_up.data_ptr(), 37, idx_down.data_ptr(), 5, -10.0, 10.0, send_up.data_ptr(),
                                  send_down.data_ptr(), st))
        tail = pos[n:]
        check(lib.uammd_comm_halo_exchange(comm, send_up.data_ptr(), 37, send_down.data_ptr(), 5, tail[:37].data_ptr(), 37,
                                           tail[37:].data_ptr(), 5, 4, st))
        torch.cuda.synchronize()
        exp_up = pos[:n][idx_up.long()].clone(); exp_up[:, 2] -= 10.0
        exp_down = pos[:n][idx_down.long()].clone(); exp_down[:, 2] += 10.0
        assert torch.equal(tail[:37], exp_up) and torch.equal(tail[37:], exp_down)
        # all-to-all of one block = a copy; all-reduce over one rank = identity
        a = torch.arange(4096, dtype=torch.float32, device="cuda")
        b = torch.zeros_like(a)
        check(lib.uammd_comm_alltoall(comm, a.data_ptr(), b.data_ptr(), a.numel() * 4, st))
        check(lib.uammd_comm_allreduce_sum(comm, a.data_ptr(), 16, st))
        torch.cuda.synchronize()
        assert torch.equal(b, torch.arange(4096, dtype=torch.float32, device="cuda")) and torch.equal(a, b)
    finally:
        check(lib.uammd_comm_destroy(comm))
