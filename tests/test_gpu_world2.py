"""PROCESS-level multi-rank runs of both decomposed paths through the real HIP kernels, on the one GPU a test box has.

Two (and four) ranks are started as separate processes (`torch.multiprocessing.spawn`), every rank binds cuda:0, the process group
is gloo and the messages are staged through the host (RCCL refuses two ranks on one device; `SlabDecomposition._wire`,
`parallel_fcm._Exchange._staged`).  What is exercised is everything a real N-GPU run does except the wire: one library handle set
per process, the halo / migration / transpose message schedule between processes, the lock-step noise seeds.

Checked against the single-GPU product path computed by the parent process on the same inputs:
  (i)   LJ forces of every particle (same pairs, another summation order: the tile kernel's tolerance, 1e-5 of max|F|) and a short
        deterministic trajectory matched by particle id;
  (ii)  FCM velocities, T = 0 and T > 0 (identical noise field), <= 1e-5 relative L2;
  (iii) 200 thermal LJ steps with cached halo lists and migration: the ids stay a partition of 0..N-1 (nobody lost or duplicated).
"""
import ctypes as C
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

LJ_L = (24.0, 24.0, 48.0)
LJ_CELLS = (22, 22, 44)                                # lattice sites (spacing 1.09)
LJ_N = LJ_CELLS[0] * LJ_CELLS[1] * LJ_CELLS[2]          # 21296 particles, rho* = 0.77
RC, DT = 2.5, 0.005


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import uammd_amd
    uammd_amd.load()          # fails loudly without the HIP library
    return uammd_amd


def _lj_sim(hip, d, T, exchange_every, seed):
    from uammd_amd._lib import check, load
    from uammd_amd.parallel import DistributedLJ
    lib = load()
    noise = math.sqrt(2 * DT * T)
    pot = hip.Potential.LJ()
    pot.setPotParameters(0, 0, pot.InputPairParameters(RC, 1.0, 1.0, False))
    cl = hip.CellList()
    cache = {}
    holder = {}

    def forces_into(allpos, box_L, periodic, fall):
        key = (tuple(box_L), tuple(periodic))
        if key not in cache:
            box = hip.Box(box_L, periodic)
            cache[key] = (box,) + tuple(hip.CellList.create_update_grid(box, RC))
        box, cd, ubox = cache[key]
        cl.update_grid(allpos, ubox, cd)
        cl.set_option("num_owned", holder["sim"].n_owned)
        cl.transverse_lj(pot.device_table(), 1, box, fall, None, None, None, 0)

    def forces_fn(allpos, box_L, periodic):
        f = torch.zeros((allpos.shape[0], 4), dtype=torch.float32, device=allpos.device)
        holder["sim"].n_owned = getattr(holder["sim"], "n_owned", allpos.shape[0])
        forces_into(allpos.contiguous(), box_L, periodic, f)
        return f

    def integrate_fn(step, p, v, f, step_num):
        check(lib.uammd_verletnvt_gj(step, C.c_void_p(p.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(f.data_ptr()), None,
                                     1.0, None, p.shape[0], DT, 1.0, 0, noise, step_num, seed,
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    sim = DistributedLJ(d, forces_fn, integrate_fn, exchange_every=exchange_every, forces_into=forces_into)
    holder["sim"] = sim
    return sim


def _lj_inputs():
    rng = np.random.default_rng(23)
    g = np.stack(np.meshgrid(*[np.arange(m) for m in LJ_CELLS], indexing="ij"), -1).reshape(-1, 3)
    p = (g + 0.5) / np.asarray(LJ_CELLS) * np.asarray(LJ_L) - np.asarray(LJ_L) / 2 + rng.uniform(-0.1, 0.1, (LJ_N, 3))
    pos = np.zeros((LJ_N, 4), np.float32)
    pos[:, :3] = p[rng.permutation(LJ_N)]
    vel = np.random.default_rng(3).normal(0, 1.0, (LJ_N, 3)).astype(np.float32)
    return pos, vel


def _lj_worker(rank, world, port, out_dir):
    hip = _init(rank, world, port)
    try:
        from uammd_amd.parallel import SlabDecomposition
        pos, vel = _lj_inputs()
        # (i) forces + a deterministic trajectory, ownership refreshed every step
        d = SlabDecomposition(LJ_L, RC, rank, world, skin=0.0)
        lpos, ids = d.scatter_initial(torch.from_numpy(pos).cuda())
        lvel = torch.from_numpy(vel).cuda()[ids.long()].contiguous()
        sim = _lj_sim(hip, d, 0.0, 1, 4242)
        sim.n_owned = lpos.shape[0]
        f0 = sim.compute_forces(lpos.contiguous()).clone()
        ids0 = ids.clone()
        p, v, f = lpos.clone().contiguous(), lvel, torch.zeros_like(lpos)
        for _ in range(10):
            p, v, f, ids = sim.forward_time(p, v, f, ids)
        gp = p.clone()
        gp[:, 2] += d.zc
        # (iii) 200 thermal steps, cached halo membership (skin 0.3, refresh every 5 steps), migration
        d2 = SlabDecomposition(LJ_L, RC, rank, world, skin=0.3)
        lp2, id2 = d2.scatter_initial(torch.from_numpy(pos).cuda())
        lv2 = torch.zeros((lp2.shape[0], 3), dtype=torch.float32, device="cuda")
        sim2 = _lj_sim(hip, d2, 1.0, 5, 99 + rank)
        p2, v2, f2 = lp2.contiguous(), lv2, torch.zeros_like(lp2)
        start = set(id2.tolist())
        for _ in range(200):
            p2, v2, f2, id2 = sim2.forward_time(p2, v2, f2, id2)
        torch.cuda.synchronize()
        sim2.check_skin()
        assert torch.isfinite(p2).all()
        np.savez(os.path.join(out_dir, f"lj{rank}.npz"), ids0=ids0.cpu().numpy(), f0=f0.cpu().numpy(), ids=ids.cpu().numpy(),
                 pos=gp.cpu().numpy(), ids200=id2.cpu().numpy(), changed=len(set(id2.tolist()) - start))
    finally:
        dist.destroy_process_group()


def _single_gpu_lj(hip):
    pos, vel = _lj_inputs()
    pd = hip.ParticleData(LJ_N, seed=1)
    pd.setPos(pos)
    pd.getVel("write").copy_(torch.from_numpy(vel).cuda())
    box = hip.Box(list(LJ_L))
    pot = hip.Potential.LJ()
    pot.setPotParameters(0, 0, pot.InputPairParameters(RC, 1.0, 1.0, False))
    pf = hip.PairForces(pd, box, pot)
    pd.getForce("write").zero_()
    pf.sum(force=True)
    f0 = pd.getForce("read").cpu().numpy().copy()
    par = hip.VerletNVT.GronbechJensen.Parameters(temperature=0.0, dt=DT, friction=1.0, initVelocities=False)
    integ = hip.VerletNVT.GronbechJensen(pd, par)
    integ.addInteractor(pf)
    pd.getForce("write").zero_()
    for _ in range(10):
        integ.forwardTime()
    return f0, pd.getPos("read").cpu().numpy().copy()


@pytest.mark.parametrize("world", [2, 4])
def test_lj_slab_processes_match_single_gpu(hip, world, tmp_path):
    f_ref, p_ref = _single_gpu_lj(hip)
    mp.spawn(_lj_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    fmax = np.abs(f_ref[:, :3]).max()
    seen0, seen, seen200, changed = [], [], [], 0
    for r in range(world):
        g = np.load(os.path.join(str(tmp_path), f"lj{r}.npz"))
        seen0 += g["ids0"].tolist()
        seen += g["ids"].tolist()
        seen200 += g["ids200"].tolist()
        changed += int(g["changed"])
        # (i) forces: same pairs, the tile kernel's summation-order tolerance — plus what a change of FRAME costs in float: the slab
        # stores z relative to its centre (z -/+ 12 here), which rounds a coordinate by up to ulp(12)/2 = 5e-7, and a pair across the
        # periodic face is subtracted as small numbers instead of (zj - zi ~ Lz) - Lz (Box::apply_pbc, rounding 2e-6 at Lz = 48).
        # dF = F' dr ~ (13 F / r) x 1e-6: 2e-3 for the system's steepest pair (F = 157), i.e. 1.3e-5 of max|F| — the typical
        # particle (median) agrees to 1e-7 of max|F|.
        df = np.abs(g["f0"][:, :3] - f_ref[g["ids0"], :3]).max(axis=1)
        assert df.max() <= 5e-5 * fmax and np.median(df) <= 2e-7 * fmax
        d = g["pos"][:, :3] - p_ref[g["ids"], :3]
        d -= np.round(d / np.asarray(LJ_L)) * np.asarray(LJ_L)
        assert np.abs(d).max() <= 2e-4                                                      # (i) 10 deterministic steps
    assert sorted(seen0) == list(range(LJ_N)) and sorted(seen) == list(range(LJ_N))
    assert sorted(seen200) == list(range(LJ_N))                                             # (iii) a partition after 200 steps
    assert changed > 0                                                                      # ... and particles did migrate


FCM_CELLS, FCM_L, FCM_N = [64, 64, 64], [64.0] * 3, 20000


def _fcm_inputs():
    rng = np.random.default_rng(5)
    pos = np.zeros((FCM_N, 4), np.float32)
    pos[:, :3] = rng.uniform(-0.5, 0.5, (FCM_N, 3)) * np.asarray(FCM_L, np.float32)
    force = np.zeros((FCM_N, 4), np.float32)
    force[:, :3] = rng.normal(0, 1, (FCM_N, 3))
    return pos, force


def _fcm_worker(rank, world, port, out_dir):
    hip = _init(rank, world, port)
    try:
        from uammd_amd.parallel_fcm import (DistributedFCM, DistributedFCMIntegrator, HipSlabBackend, SlabGeometry,
                                            make_decomposition)
        pos, force = _fcm_inputs()
        kernel, a = hip.Kernels.Gaussian(1.0, 1e-3)
        geom = SlabGeometry(FCM_CELLS, FCM_L, world, kernel.support[2])
        back = HipSlabBackend(geom, rank, kernel, 1.3, 1234)
        d = make_decomposition(geom, rank)
        lpos, ids = d.scatter_initial(torch.from_numpy(pos).cuda())
        lpos = lpos.contiguous()
        lforce = torch.from_numpy(force).cuda()[ids.long()].contiguous()
        fcm = DistributedFCM(geom, [back], [rank])
        v0 = fcm.displacements([lpos], [lforce], 0.0, 0.0)[0].cpu().numpy().copy()
        v1 = fcm.displacements([lpos], [lforce], 0.7, 3.0)[0].cpu().numpy().copy()
        v2 = fcm.displacements([lpos], [lforce], 0.7, 3.0)[0].cpu().numpy().copy()
        # Euler-Maruyama with a large step: particles cross the slab faces and migrate with their forces
        integ = DistributedFCMIntegrator(DistributedFCM(geom, [HipSlabBackend(geom, rank, kernel, 1.3, 1234)], [rank]), d, 0.5, 0.5,
                                         lambda p, i, f: f, migrate_every=2)
        p, i, f = lpos.clone(), ids.clone(), lforce.clone()
        start = set(i.tolist())
        for _ in range(40):
            p, i, f = integ.forward_time(p, i, f)
        torch.cuda.synchronize()
        integ.check_drift()
        np.savez(os.path.join(out_dir, f"fcm{rank}.npz"), ids=ids.cpu().numpy(), v0=v0, v1=v1, v2=v2, ids_end=i.cpu().numpy(),
                 changed=len(set(i.tolist()) - start), seed2=fcm.seed2)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_fcm_slab_processes_match_single_gpu(hip, world, tmp_path):
    pos, force = _fcm_inputs()
    kernel, a = hip.Kernels.Gaussian(1.0, 1e-3)
    ref = hip.BDHI.FCM_impl(hip.Box(FCM_L), FCM_CELLS, kernel, 1.3, 1234, a)
    dp, df = torch.from_numpy(pos).cuda(), torch.from_numpy(force).cuda()
    v_ref = [ref.computeHydrodynamicDisplacements(dp, df, FCM_N, 0.0, 0.0).cpu().numpy(),
             ref.computeHydrodynamicDisplacements(dp, df, FCM_N, 0.7, 3.0).cpu().numpy(),
             ref.computeHydrodynamicDisplacements(dp, df, FCM_N, 0.7, 3.0).cpu().numpy()]
    mp.spawn(_fcm_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    seen, seen_end, changed = [], [], 0
    got = [np.zeros((FCM_N, 3), np.float32) for _ in range(3)]
    for r in range(world):
        g = np.load(os.path.join(str(tmp_path), f"fcm{r}.npz"))
        seen += g["ids"].tolist()
        seen_end += g["ids_end"].tolist()
        changed += int(g["changed"])
        assert int(g["seed2"]) == 2
        for k, name in enumerate(("v0", "v1", "v2")):
            got[k][g["ids"]] = g[name]
    assert sorted(seen) == list(range(FCM_N)) and sorted(seen_end) == list(range(FCM_N))
    assert changed > 0
    for k in range(3):
        err = np.linalg.norm(got[k] - v_ref[k]) / np.linalg.norm(v_ref[k])
        assert err <= 1e-5, (k, err)                                                       # (ii)


@pytest.mark.parametrize("world", [2, 4])
def test_bench_strong_scaling_line_on_one_device(world):
    """bench.py --gpus N with every rank on cuda:0 (gloo, host-staged messages): the launcher, the preflight and BOTH LJ lines — the weak
    headline (one slab of --particles per rank) and the strong one (the ONE box of --particles cut into N slabs: BASELINE's "LJ 1e6 ...
    1/2/4/8 GPU" read literally) — run to the end, conserve the particles (asserted inside bench.py) and report consistent sizes."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UAMMD_BENCH_SAME_DEVICE="1", UAMMD_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    n = 131072
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--workload", "lj", "--particles", str(n), "--steps", "40",
                        "--warmup", "10", "--equilibrate", "60", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert "error" not in line, line
    assert line["n_gpus"] == world and line["scaling"] == "weak" and line["config"]["particles_per_gpu"] == n
    strong = line["lj_strong"]
    assert strong["scaling"] == "strong" and strong["particles_total"] == n and strong["n_gpus"] == world
    L = 107.7217345 * (n / 1e6) ** (1 / 3)
    assert abs(strong["slab_width"] - L / world) < 1e-6 and strong["value"] > 0 and strong["ms_per_step"] > 0
