"""GPU: lanczos::Solver on a vector sharded over 2 ranks (Lanczos row): every dot product of the recurrence is
completed by an all-reduce.  Two processes share cuda:0 and talk over gloo (RCCL refuses two ranks on one device); the result
must equal the single-rank solver on the whole vector."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N = 6000          # global length, two equal slices (gloo all_gather wants equal sizes)
SPLIT = 3000


def _matrix_apply(v):
    """SPD banded operator on the GLOBAL vector (torch, on the device): M = diag(d) + nearest and 7th-neighbour coupling."""
    n = v.shape[0]
    i = torch.arange(n, device=v.device, dtype=torch.float32)
    d = 2.5 + torch.sin(0.37 * i) ** 2
    out = d * v
    out[1:] += 0.6 * v[:-1]
    out[:-1] += 0.6 * v[1:]
    out[7:] += 0.3 * v[:-7]
    out[:-7] += 0.3 * v[7:]
    return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        import uammd_amd as hip
        lo, hi = (0, SPLIT) if rank == 0 else (SPLIT, N)
        z = torch.from_numpy(np.random.default_rng(3).normal(0, 1, N).astype(np.float32)).cuda()
        counts = [SPLIT, N - SPLIT]

        def dot(v_local, Mv_local):
            # the matvec needs the other rank's slice: all-gather through the host (gloo), then the local rows
            parts = [torch.empty(c, dtype=torch.float32) for c in counts]
            dist.all_gather(parts, v_local.cpu())
            full = torch.cat(parts).cuda()
            Mv_local.copy_(_matrix_apply(full)[lo:hi])
        solver = hip.BDHI.LanczosSolver()
        solver.setAllReduce()
        Bv = torch.zeros(hi - lo, dtype=torch.float32, device="cuda")
        it = solver.run(dot, Bv, z[lo:hi].contiguous(), 1e-4)
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), Bv=Bv.cpu().numpy(), it=it)
    finally:
        dist.destroy_process_group()


def test_sharded_lanczos_equals_single_rank(hip, tmp_path):
    z = torch.from_numpy(np.random.default_rng(3).normal(0, 1, N).astype(np.float32)).cuda()
    solver = hip.BDHI.LanczosSolver()
    ref = torch.zeros(N, dtype=torch.float32, device="cuda")
    it_ref = solver.run(lambda v, Mv: Mv.copy_(_matrix_apply(v)), ref, z, 1e-4)
    ref = ref.cpu().numpy()
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a = np.load(os.path.join(str(tmp_path), "rank0.npz"))
    b = np.load(os.path.join(str(tmp_path), "rank1.npz"))
    got = np.concatenate([a["Bv"], b["Bv"]])
    assert int(a["it"]) == int(b["it"]) == it_ref
    assert np.linalg.norm(got - ref) <= 1e-5 * np.linalg.norm(ref)
    # and it is the square root: B(Bz) = M z up to the tolerance
    solver2 = hip.BDHI.LanczosSolver()
    BB = torch.zeros(N, dtype=torch.float32, device="cuda")
    solver2.run(lambda v, Mv: Mv.copy_(_matrix_apply(v)), BB, torch.from_numpy(got).cuda(), 1e-5)
    Mz = _matrix_apply(z).cpu().numpy()
    assert np.linalg.norm(BB.cpu().numpy() - Mz) <= 5e-4 * np.linalg.norm(Mz)
