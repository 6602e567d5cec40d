#!/usr/bin/env python
"""bench.py — headline benchmark of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload lj|fcm]

Workload "lj" (default; BASELINE.json configs[2], the configuration the metric is quoted on):
  1e6 Lennard-Jones particles, rho* = 0.8 (L = 107.7217345 -> 43^3 cells), r_c = 2.5 sigma,
  VerletNVT::GronbechJensen (T = 1, friction 1, dt = 0.005), CellList rebuilt every step,
  PairForces<LJ>.  One "step" = one integrator->forwardTime() = cell-list build + traversal +
  two integrate kernels, exactly the reference's step.  Inputs are synthetic (jittered
  simple-cubic lattice, seed 1234), resident in HBM before the timed region.

One JSON line on stdout (rank 0).  `value` = particle-steps/s summed over all ranks.
The `roofline` object is for the dominant kernel (the LJ traversal): achieved = algorithmic flops
per launch / mean launch time measured with HIP events on the launch stream inside the timed loop.
`cpu_baseline` = the oracle (CPU port of the reference algorithm; the reference has no CPU path)
timed on this host's cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

# SURVEY §8(d) per-unit figures for the traversal kernel (stated in DESIGN.md §roofline)
FLOP_PER_PARTICLE = 1.0e4            # 27*12.58 ~ 340 candidate pairs x ~30 flop
BYTES_COMPULSORY_PER_PARTICLE = 52.6  # sortPos_i + groupIndex + force RMW + tables
PEAK_FP32_TFLOPS = 157.3             # MI355X_MICROARCH.md: FP32 vector = FP32 MFMA peak
PEAK_HBM_GBS = 8000.0


SLAB_STEP = None   # what the slab-decomposed LJ step was made of (set by run_lj_distributed, reported in config)
ABI_COMM = None   # uammd_amd.comm.AbiComm when the run's messages go through uammd_comm_* (the product's stack); None: torch.distributed
TRANSPORT = "none (single domain)"   # what actually carried the messages: goes into config.workload and comm.backend


def _allreduce(dist, values, op):
    """floats reduced over the ranks (through uammd_comm_allreduce_sum when the C-ABI stack carries the run; else host tensors under
    gloo, device tensors under torch's RCCL)."""
    if ABI_COMM is not None:
        return ABI_COMM.reduce_host(values, op)
    t = torch.tensor(values, dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
    dist.all_reduce(t, op=getattr(dist.ReduceOp, op))
    return t.tolist()


_RESULT_FD = None


def emit_result(obj):
    """the run's one line of stdout (main() moved file descriptor 1 itself to stderr for the libraries' chatter)"""
    line = (json.dumps(obj) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, line)


def _barrier(dist):
    if ABI_COMM is not None:
        ABI_COMM.barrier()
    elif dist is not None:
        dist.barrier()


def _reducing(dist):
    return ABI_COMM is not None or dist is not None


def read_traffic(name):
    """HBM bytes per launch from the committed PMC summary (tools/pmc_traffic.sh -> profiles/), or None."""
    tp = os.path.join(ROOT, "profiles", name)
    try:
        return json.load(open(tp)).get("hbm_bytes_per_launch")
    except Exception:
        return None


def read_kernel_fractions(name, bytes_of):
    """Per-kernel rooflines of a step from the committed rocprofv3 summary (tools/prof_any.sh -> profiles/<name>: lines
    `kernel | calls | total_us | avg_us | pct`): for every kernel `bytes_of` knows (prefix -> algorithmic HBM bytes per launch) its average
    duration, achieved GB/s and fraction of the HBM roof.  None if the file is not there."""
    tp = os.path.join(ROOT, "profiles", name)
    try:
        rows = [l.split(" | ") for l in open(tp) if " | " in l and not l.startswith("#")]
    except Exception:
        return None
    out = []
    for r in rows:
        kern = r[0].replace("void ", "").replace("uammd_hip::", "")
        for prefix, nbytes in bytes_of.items():
            if kern.startswith(prefix):
                us = float(r[3])
                gbs = nbytes / (us * 1e-6) / 1e9
                out.append({"kernel": kern.split("(")[0], "avg_us": us, "algorithmic_bytes": nbytes, "achieved_GBs": round(gbs, 1),
                            "frac": round(gbs / PEAK_HBM_GBS, 4)})
                break
    return {"source": "profiles/" + name, "kernels": out} if out else None


def lattice(n, L, seed, jitter=0.1):
    from util import lattice_positions
    return lattice_positions(n, L, seed=seed, jitter=jitter)


def lj_setup(hip, n, L, seed, T=1.0, dt=0.005, nl="cell", fcc=False):
    if fcc:   # the reference benchmark's own input: initLattice(box.boxSize, N, fcc) (examples/misc/benchmark.cu:63)
        from uammd_amd.initial_conditions import init_lattice
        pos = init_lattice([L] * 3, n, "fcc")
    else:
        pos = lattice(n, L, seed)
    pd = hip.ParticleData(n, seed=seed)
    pd.setPos(pos)
    box = hip.Box(L)
    pot = hip.Potential.LJ()
    pot.setPotParameters(0, 0, pot.InputPairParameters(2.5, 1.0, 1.0, False))
    par = hip.VerletNVT.GronbechJensen.Parameters(temperature=T, dt=dt, friction=1.0, initVelocities=True)
    verlet = hip.VerletNVT.GronbechJensen(pd, par)
    pf = hip.PairForces(pd, box, pot, nl=(hip.VerletList(pd) if nl == "verlet" else None))
    verlet.addInteractor(pf)
    if nl == "cell" and os.environ.get("UAMMD_BENCH_DEFAULT_HINT") != "1":
        # ParticleData::hintSortByHash is public API (ParticleData.cuh:389-394): sortParticles() then orders memory by the
        # Morton hash of the PairForces cell size instead of the 10-sigma default, i.e. exactly the order the cell list is built
        # in (the reference's VerletList gives the same hint with cutOff/2, VerletList.cuh:116)
        pd.hintSortByHash(box, [2.5] * 3)
    return pd, box, pot, verlet, pf, pos


def host_cores():
    """CPUs this process may actually use: the affinity mask, cut by the container's CPU quota (cgroup v2 cpu.max / v1 cfs quota).
    The GPU boxes show 256 hardware threads and grant 16 CPUs of time: threads beyond the quota only add throttling."""
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                cores = min(cores, max(1, q // per))
        except Exception:
            pass
    return cores


def _pin_threads():
    """One thread per usable core, close binding; must be in the environment before libgomp starts.
    Returns the number of threads (taken BEFORE OpenMP binds the calling thread to its place)."""
    cores = host_cores()
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    return int(os.environ["OMP_NUM_THREADS"])


def cpu_baseline_lj(n, L, seed, sample_steps):
    """Oracle ("port" of the reference algorithm, oracle/src/*.c) on ALL host cores: OpenMP over particles in the traversal and
    the integrator, chunked stable radix sort + parallel hash / reorder / cell tables in the build."""
    cores = _pin_threads()
    import oracle
    o = oracle.get("f32")
    o.set_parallel(True)
    pos = lattice(n, L, seed)
    vel = np.zeros((n, 3), np.float32)
    force = np.zeros((n, 4), np.float32)
    par = o.lj_params(2.5, 1.0, 1.0)
    dt, T = 0.005, 1.0
    noise = math.sqrt(2 * dt * T)

    def forces(p):
        cd, oL, oper = o.celllist_create_grid(L, 1, 2.5)
        cl = o.celllist_build(p, oL, oper, cd)
        f, _, _ = o.lj_transverse_celllist(cl, L, 1, par, 1, n)
        return f

    force = forces(pos)
    t0 = time.perf_counter()
    for s in range(1, sample_steps + 1):
        o.verletnvt_gj(1, pos, vel, force, dt, 1.0, noise, s, 1234)
        force = forces(pos)
        o.verletnvt_gj(2, pos, vel, force, dt, 1.0, noise, s, 1234)
    el = time.perf_counter() - t0
    return {"value": n * sample_steps / el, "unit": "particle-steps/s", "cores": cores, "kind": "port",
            "sample": f"{sample_steps} full NVT steps of the same 1e6-particle LJ box (oracle: cell-list build, traversal and "
                      f"integrator all OpenMP over {cores} pinned threads = the CPUs the container grants of {os.cpu_count()} hardware threads), {el:.1f} s"}



# FCM (BASELINE configs[3]): algorithmic HBM bytes per step = 10*G + 60*N with G = 12 * 2(nx/2+1)*ny*nz
def fcm_bytes_per_step(n, cells):
    G = 12 * 2 * (cells[0] // 2 + 1) * cells[1] * cells[2]
    return 10 * G + 60 * n


class FixedForce:
    """test/BDHI/FCM/FCM.cu:41-55 miniInteractor: an Interactor that applies a fixed force."""

    def __init__(self, pd, force):
        self.pd, self.force = pd, force

    def sum(self, force=True, energy=False, virial=False):
        self.pd.getForce("readwrite").add_(self.force)

    def updateSimulationTime(self, t):
        pass

    def updateTimeStep(self, dt):
        pass

    def updateTemperature(self, T):
        pass

    def updateBox(self, box):
        pass


def fcm_setup(hip, n, cells, L, seed, T=1.0, dt=0.01):
    rng = np.random.default_rng(seed)
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(-L / 2, L / 2, (n, 3))
    force = np.zeros((n, 4), np.float32)
    force[:, :3] = np.random.default_rng(4321).normal(0, 1, (n, 3))
    pd = hip.ParticleData(n, seed=seed)
    pd.setPos(pos)
    par = hip.BDHI.FCMIntegrator.Parameters(temperature=T, viscosity=1.0, tolerance=1e-3, dt=dt, box=hip.Box(L),
                                            cells=cells, seed=1234)
    integ = hip.BDHI.FCMIntegrator(pd, par)
    integ.addInteractor(FixedForce(pd, torch.from_numpy(force).cuda()))
    return pd, integ, pos, force


def run_fcm(hip, args, world, rank, dist):
    n, cells, L = 100_000, [128, 128, 128], 128.0
    pd, integ, pos, force = fcm_setup(hip, n, cells, L, seed=1234 + rank)
    if args.fcm_sort:
        pd.hintSortByHash(hip.Box(L), args.fcm_sort)
        pd.sortParticles()
        integ.interactors[0].force = pd._props.get("fixedforce", integ.interactors[0].force)
    for _ in range(args.fcm_warmup):
        integ.forwardTime()
    torch.cuda.synchronize()
    _barrier(dist)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.fcm_steps):
        integ.forwardTime()
    e1.record()
    torch.cuda.synchronize()
    _barrier(dist)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if _reducing(dist):
        el = _allreduce(dist, [el], "MAX")[0]
    assert np.isfinite(pd.getPos().cpu().numpy()).all()
    ms = el / args.fcm_steps * 1e3
    gbs = fcm_bytes_per_step(n, cells) / (ms * 1e-3) / 1e9
    out = {"metric": "FCM-BDHI steps/s @128^3 (1e5 particles, tol 1e-3, T=1)", "value": world * args.fcm_steps / el,
           "unit": "steps/s", "ms_per_step": ms, "steps": args.fcm_steps, "warmup": args.fcm_warmup,
           "config": {"workload": "BDHI::FCMIntegrator 1e5 particles, 128^3 grid, L=128, Gaussian support 6, "
                                  "fixed forces + Fourier-space noise (BASELINE configs[3])"},
           "roofline": {"bound": "hbm", "kernel": "whole FCM step (spread, 2x3 FFT, k-space, gather)",
                        "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                        "traffic": read_traffic("traffic_fcm_step.json"),
                        "algorithmic_bytes_per_step": fcm_bytes_per_step(n, cells)}}
    if world == 1:   # the deterministic step (T = 0: no Fourier-space noise) beside the thermal one
        pd0, integ0, _, _ = fcm_setup(hip, n, cells, L, seed=1234 + rank, T=0.0)
        for _ in range(args.fcm_warmup):
            integ0.forwardTime()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.fcm_steps):
            integ0.forwardTime()
        torch.cuda.synchronize()
        out["deterministic_T0"] = {"ms_per_step": (time.perf_counter() - t0) / args.fcm_steps * 1e3, "steps": args.fcm_steps}
        del integ0, pd0
    # every kernel of the step against the HBM roof on ITS OWN algorithmic bytes (G = one pass over the 3-component complex grid, 25.6 MB
    # at C4; the gather reads the float4 grid, 16 B per node; a particle's prepared stencil is 16 + 72 B), durations from the committed
    # rocprofv3 summary of tools/time_fcm.py
    G = 12 * 2 * (cells[0] // 2 + 1) * cells[1] * cells[2]
    nodes = cells[0] * cells[1] * cells[2]
    # (round 5: one preparation launch — position, velocity and entry in; position, stencil origin, 72 B of weights and the 8-byte record out)
    out["kernel_rooflines"] = read_kernel_fractions("r06_kernel_stats_fcm_c4.txt", {
        "k_fcm_spread_tile": G + 4.3 * n * (16 + 72), "k_fft_xy_r2c_plane": 2 * G, "k_fft_z_fused": 2 * G, "k_fft_lines": 2 * G,
        "k_fft_x_c2r": G + 16 * nodes, "k_fcm_gather_col": 16 * nodes + n * (16 + 72 + 12), "k_fcm_gather_half": 16 * nodes + n * (16 + 72 + 12),
        "k_fcm_step_prep": n * (16 + 12 + 16 + 16 + 16 + 72 + 8)})
    return out


def run_fcm_distributed(hip, args, world, rank, dist):
    """N > 1 (or --force-distributed): the z-slab decomposed solver, weak scaling — every GPU owns a 128 x 128 x 128 slab
    of a 128 x 128 x (128 N) grid and ~1e5 particles (at N = 8 the grid has 4x the nodes of BASELINE configs[4])."""
    from uammd_amd.parallel_fcm import (DistributedFCM, DistributedFCMIntegrator, HipSlabBackend, SlabGeometry,
                                        make_decomposition)
    n, T, dt = 100_000, 1.0, 0.01
    cells, L = [128, 128, 128 * world], [128.0, 128.0, 128.0 * world]
    kernel, a_eff = hip.Kernels.Gaussian(1.0, 1e-3)
    geom = SlabGeometry(cells, L, world, kernel.support[2])
    back = HipSlabBackend(geom, rank, kernel, 1.0, 1234)
    d = make_decomposition(geom, rank, comm=ABI_COMM)
    rng = np.random.default_rng(1234 + rank)
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(-64.0, 64.0, (n, 3))            # window frame: z relative to the slab centre
    force = np.zeros((n, 4), np.float32)
    force[:, :3] = np.random.default_rng(4321 + rank).normal(0, 1, (n, 3))
    pos, force = torch.from_numpy(pos).cuda(), torch.from_numpy(force).cuda()
    ids = torch.arange(n, dtype=torch.int32, device="cuda") + rank * n
    # thermal steps are ~0.03 h: with 3 spare halo planes the particles are re-assigned to their slabs every 20 steps
    integ = DistributedFCMIntegrator(DistributedFCM(geom, [back], [rank], comm=ABI_COMM), d, T, dt, lambda p, i, f: f, migrate_every=20)
    for _ in range(args.fcm_warmup):
        pos, ids, force = integ.forward_time(pos, ids, force)
    torch.cuda.synchronize()
    _barrier(dist)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.fcm_steps):
        pos, ids, force = integ.forward_time(pos, ids, force)
    torch.cuda.synchronize()
    _barrier(dist)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    cnt = float(pos.shape[0])
    if _reducing(dist):
        el = _allreduce(dist, [el], "MAX")[0]
        cnt = _allreduce(dist, [cnt], "SUM")[0]
    assert torch.isfinite(pos).all()
    assert abs(cnt - n * world) < 0.5, "particles were lost or duplicated in migration"
    integ.check_drift()
    ms = el / args.fcm_steps * 1e3
    nbytes = fcm_bytes_per_step(n * world, cells)
    gbs = nbytes / (ms * 1e-3) / 1e9
    # weak scaling: one 128^3-slab solve per GPU per step -> value counts slab-steps, = steps/s of the C4 problem at N = 1
    return {"metric": "FCM-BDHI slab-steps/s (128^3 grid slab + 1e5 particles per GPU, tol 1e-3, T=1)",
            "value": world * args.fcm_steps / el, "unit": "steps/s", "ms_per_step": ms, "steps": args.fcm_steps,
            "warmup": args.fcm_warmup,
            "config": {"workload": f"BDHI::FCMIntegrator, grid 128x128x{128 * world}, {n * world} particles, Gaussian support 6, "
                                   "fixed forces + Fourier-space noise; z-slab decomposition: halo planes by send/recv, "
                                   f"2 all-to-all transposes per step; messages through {TRANSPORT}",
                       "parallelism": f"slab{world}: 1 process per GPU"},
            "roofline": {"bound": "hbm", "kernel": "whole FCM step, all GPUs", "achieved": gbs, "peak": PEAK_HBM_GBS * world,
                         "unit": "GB/s", "frac": gbs / (PEAK_HBM_GBS * world), "traffic": None,
                         "algorithmic_bytes_per_step": nbytes}}


def run_fcm_c5(hip, args, world, rank, dist):
    """BASELINE configs[4] as it is named: BDHI::FCM, 2e5 particles, 256^3 grid, STRONG scaling — the same grid and particle count
    at every N, cut into N z-slabs (256 / N planes + 8 halo planes each; all-to-all transposes of the 3-D FFT).  N = 1 is the
    single-GPU solver on the whole grid."""
    n_total, nc, T, dt = 200_000, 256, 1.0, 0.01
    L, cells = float(nc), [nc] * 3
    steps, warm = max(10, args.fcm_steps // 4), max(3, args.fcm_warmup // 4)
    if world == 1 and not args.force_distributed:
        pd, integ, _, _ = fcm_setup(hip, n_total, cells, L, seed=1234)
        step = integ.forwardTime
        nloc = n_total
    else:
        from uammd_amd.parallel_fcm import (DistributedFCM, DistributedFCMIntegrator, HipSlabBackend, SlabGeometry,
                                            make_decomposition)
        if nc % world:
            return {"skipped": f"256 planes do not split into {world} slabs"}
        kernel, a_eff = hip.Kernels.Gaussian(1.0, 1e-3)
        geom = SlabGeometry(cells, [L] * 3, world, kernel.support[2])
        back = HipSlabBackend(geom, rank, kernel, 1.0, 1234)
        d = make_decomposition(geom, rank, comm=ABI_COMM)
        nloc = n_total // world
        rng = np.random.default_rng(1234 + rank)
        pos = np.zeros((nloc, 4), np.float32)
        pos[:, :2] = rng.uniform(-L / 2, L / 2, (nloc, 2))
        pos[:, 2] = rng.uniform(-L / (2 * world), L / (2 * world), nloc)       # window frame: z relative to the slab centre
        force = np.zeros((nloc, 4), np.float32)
        force[:, :3] = np.random.default_rng(4321 + rank).normal(0, 1, (nloc, 3))
        state = [torch.from_numpy(pos).cuda(), torch.arange(nloc, dtype=torch.int32, device="cuda") + rank * nloc,
                 torch.from_numpy(force).cuda()]
        integ = DistributedFCMIntegrator(DistributedFCM(geom, [back], [rank], comm=ABI_COMM), d, T, dt, lambda p, i, f: f, migrate_every=20)

        def step():
            state[0], state[1], state[2] = integ.forward_time(state[0], state[1], state[2])
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    _barrier(dist)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    _barrier(dist)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if _reducing(dist):
        el = _allreduce(dist, [el], "MAX")[0]
    ms = el / steps * 1e3
    nbytes = fcm_bytes_per_step(n_total, cells)
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"metric": "FCM-BDHI steps/s @256^3 (2e5 particles, tol 1e-3, T=1), the whole job", "value": steps / el, "unit": "steps/s",
            "ms_per_step": ms, "steps": steps, "warmup": warm, "scaling": "strong", "n_gpus": world,
            "config": {"workload": f"BDHI::FCMIntegrator, grid 256^3, {n_total} particles, Gaussian support 6, fixed forces + Fourier-space "
                                   f"noise (BASELINE configs[4]); " + ("single GPU" if world == 1 and not args.force_distributed else
                                   f"{world} z-slabs of {nc // world} planes, halo planes by send/recv, 2 all-to-all transposes per step (RCCL)")},
            "roofline": {"bound": "hbm", "kernel": "whole FCM step, all GPUs", "achieved": gbs, "peak": PEAK_HBM_GBS * world, "unit": "GB/s",
                         "frac": gbs / (PEAK_HBM_GBS * world), "traffic": None, "algorithmic_bytes_per_step": nbytes}}


def cpu_baseline_fcm(sample_steps):
    cores = _pin_threads()
    import oracle
    from oracle.fcm import FCMOracle
    o = oracle.get("f32")
    o.set_parallel(True)
    n, cells, L = 100_000, [128, 128, 128], 128.0
    rng = np.random.default_rng(1234)
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(-L / 2, L / 2, (n, 3))
    force = np.zeros((n, 4), np.float32)
    force[:, :3] = np.random.default_rng(4321).normal(0, 1, (n, 3))
    f = FCMOracle(o, L, cells, tolerance=1e-3, viscosity=1.0, seed=1234)
    t0 = time.perf_counter()
    for _ in range(sample_steps):
        v = f.displacements(pos, force, temperature=1.0, prefactor=10.0)
        o.fcm_euler_maruyama(pos, v, 0.01)
    el = time.perf_counter() - t0
    return {"value": sample_steps / el, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"{sample_steps} FCM steps at 128^3 / 1e5 particles (oracle: spread with atomic adds, gather, k-space and "
                      f"scipy pocketfft FFTs all on {cores} pinned threads), {el:.1f} s"}



# ---- PSE + Lanczos (SURVEY 8d "PSE+Lanczos variant": the spectral-Ewald half of the north star) -----------------------------------------
PSE_N, PSE_L, PSE_PSI, PSE_TOL = 100_000, 128.0, 0.5, 1e-3


def _pse_inputs():
    rng = np.random.default_rng(1234)
    pos = np.zeros((PSE_N, 4), np.float32)
    pos[:, :3] = rng.uniform(-PSE_L / 2, PSE_L / 2, (PSE_N, 3))
    force = np.zeros((PSE_N, 4), np.float32)
    force[:, :3] = np.random.default_rng(4321).normal(0, 1, (PSE_N, 3))
    return pos, force


def _settle():
    """Between the legs of a run: whatever the previous leg left for the cyclic collector (a solver handle frees its device buffers with a
    device-synchronising call each) goes NOW, not in the middle of the next leg's timed loop — that put 16 ms of frees into 50 timed PSE
    steps of the round-4 driver-like run (0.905 against 0.578 ms per step)."""
    import gc
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def _timed(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def run_pse(hip, args):
    """BDHI::EulerMaruyama<BDHI::PSE>: N = 1e5, L = 128, a = 1, psi = 0.5 (PSE/utils.cuh:17-24 default), tolerance 1e-3, T = 1, dt = 0.01,
    fixed N(0,1) forces.  One step = far field (spread, FFT, sheared Hasimoto-split operator + noise, FFT, gather) + near-field M F (one
    sparse product over the cell list, rc = sqrt(-ln tol) / psi = 5.26) + near-field noise by Lanczos (k products + the recurrence)
    + Euler update.  Parts are timed on their own as well; roofline per part."""
    import ctypes as C
    from uammd_amd._lib import check, load
    lib = load()
    pos, force = _pse_inputs()
    pd = hip.ParticleData(PSE_N, seed=1234)
    pd.setPos(pos)
    par = hip.BDHI.PSE.Parameters(temperature=1.0, viscosity=1.0, hydrodynamicRadius=1.0, tolerance=PSE_TOL, dt=0.01, box=hip.Box(PSE_L),
                                  psi=PSE_PSI)
    integ = hip.BDHI.EulerMaruyama(pd, par, Method=hip.BDHI.PSE)
    integ.addInteractor(FixedForce(pd, torch.from_numpy(force).cuda()))
    pse = integ.bdhi
    if os.environ.get("UAMMD_PSE_SKIN") is not None:   # (A/B: 0 = a list build per step)
        check(lib.uammd_pse_near_set_option(pse.near, b"list_skin_percent", int(os.environ["UAMMD_PSE_SKIN"])))
    if os.environ.get("UAMMD_PSE_FUSE") == "0":   # (A/B: the Lanczos recurrence as four launches per iteration instead of two)
        check(lib.uammd_pse_near_set_option(pse.near, b"fuse_recurrence", 0))
    steps, warm = args.pse_steps, 5
    for _ in range(warm):
        integ.forwardTime()
    torch.cuda.synchronize()
    # timed as blocks (the median block is the figure, as for the LJ line): one host hiccup in a 30 ms region — a driver-like run of round 4
    # came back with 0.705 ms where four repeats gave 0.555-0.560 — must not be the number
    its, block_ms = [], []
    nblocks = 5 if steps >= 25 else 1
    per_block = steps // nblocks
    for _ in range(nblocks):
        t0 = time.perf_counter()
        for _ in range(per_block):
            integ.forwardTime()
            its.append(pse.lastLanczosIterations)
        torch.cuda.synchronize()
        block_ms.append((time.perf_counter() - t0) / per_block * 1e3)
    ms = float(np.median(block_ms))
    steps = nblocks * per_block
    assert np.isfinite(pd.getPos().cpu().numpy()).all()
    stats = (C.c_longlong * 4)()
    check(lib.uammd_pse_near_list_stats(pse.near, stats))
    # the parts, each on its own (same state)
    MF = torch.zeros((PSE_N, 3), dtype=torch.float32, device="cuda")
    dforce = torch.from_numpy(force).cuda()
    dpos = pd.getPos("read")
    p = lambda t: C.c_void_p(t.data_ptr())
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    t_far = _timed(lambda: pse._far(dforce, MF, 1.0, 10.0), 50)
    t_mdot = _timed(lambda: check(lib.uammd_pse_near_mdot(pse.near, p(dpos), p(dforce), PSE_N, p(MF), st())), 50)
    v3 = torch.randn((PSE_N, 3), dtype=torch.float32, device="cuda")
    t_dot = _timed(lambda: check(lib.uammd_pse_near_dot(pse.near, p(dpos), p(v3), PSE_N, p(MF), st())), 50)
    t_stoch = _timed(lambda: pse._near_stochastic(MF, 1.0, 1.0), 20)
    k = float(np.mean(its))
    # near-field product: candidates per particle = 27 cells x particles per cell; ~30 flop per candidate test (sheared minimum image with
    # three roundf, dot, compare) + ~45 per pair inside rc (sqrt, two table interpolations, division, 3 x 3 product)
    ncell = int(PSE_L / pse.rcut)
    cand = 27.0 * PSE_N / ncell ** 3
    hits = 4.0 / 3.0 * math.pi * pse.rcut ** 3 * PSE_N / PSE_L ** 3
    flop_dot = PSE_N * (cand * 30.0 + hits * 45.0)
    G = 12 * 2 * (pse.cells[0] // 2 + 1) * pse.cells[1] * pse.cells[2]
    far_bytes = 10 * G + 60 * PSE_N
    # one Lanczos iteration besides the product (LanczosAlgorithm.cu:114-157): w -= beta v_{i-1}; alpha = w.v; w -= alpha v; beta = |w|;
    # v_{i+1} = w / beta  ->  ~8 passes over 3N floats; a convergence check adds the 3N x m gemv
    lanczos_bytes_iter = 8 * 12 * PSE_N
    return {"metric": "BDHI::PSE steps/s (1e5 particles, L=128, a=1, psi=0.5, tol 1e-3, T=1)", "value": 1e3 / ms, "unit": "steps/s",
            "ms_per_step": ms, "steps": steps, "timed_blocks": {"blocks": nblocks, "steps_each": per_block, "ms_per_step_min": float(min(block_ms)),
                                                                   "ms_per_step_max": float(max(block_ms))}, "lanczos_iterations_mean": k, "lanczos_iterations_minmax": [int(min(its)), int(max(its))],
            "near_list": {"builds_from_scratch": int(stats[0]), "record_builds_from_kept_candidates": int(stats[1]),
                          "repeated_builds": int(stats[2]), "kept_lists_on": bool(stats[3]), "steps": warm + steps},
            "config": {"workload": f"BDHI::EulerMaruyama<PSE>: near-field cut-off {pse.rcut:.3f} ({ncell}^3 cells, {cand:.0f} candidates and "
                                   f"{hits:.1f} neighbours per particle), RPY table {pse.nPointsTable} points, far-field grid "
                                   f"{list(pse.cells)}, Gaussian support {pse.support}; fixed forces"},
            "parts_ms": {"far_field": t_far, "near_MF (cell list + 1 product)": t_mdot, "near_product (cell list + memset + 1 product)": t_dot,
                         "near_noise (cell list + Saru vector + Lanczos)": t_stoch},
            "roofline_near_product": {"bound": "valu", "achieved": flop_dot / (t_dot * 1e-3) / 1e12, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                                      "frac": flop_dot / (t_dot * 1e-3) / 1e12 / PEAK_FP32_TFLOPS,
                                      "note": "time includes the per-call cell-list build; flops = N (27 Nppc x 30 + neighbours x 45)"},
            "roofline_lanczos": {"bound": "valu", "products": k + 1, "ms_per_iteration": t_stoch / (k + 1),
                                 "note": "near_noise / (iterations + 1): one product + ~8 vector passes (%.1f MB) + scalar recurrences per iteration" % (lanczos_bytes_iter / 1e6)},
            "roofline_far_field": {"bound": "hbm", "achieved": far_bytes / (t_far * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": far_bytes / (t_far * 1e-3) / 1e9 / PEAK_HBM_GBS, "algorithmic_bytes": far_bytes}}


def cpu_baseline_pse(sample_steps):
    cores = _pin_threads()
    import oracle
    from oracle.pse import PSEOracle
    o = oracle.get("f32")
    o.set_parallel(True)
    pos, force = _pse_inputs()
    ps = PSEOracle(o, PSE_L, 1.0, 1.0, PSE_TOL, PSE_PSI)
    t0 = time.perf_counter()
    for s in range(sample_steps):
        v = ps.computeHydrodynamicDisplacements(pos, force, 1.0, 10.0, seed2_near=s + 1, seed2_far=s + 1)
        pos[:, :3] += v * np.float32(0.01)
    el = time.perf_counter() - t0
    return {"value": sample_steps / el, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"{sample_steps} PSE steps (oracle: near product + Lanczos in numpy around the C product, far field with scipy FFTs) "
                      f"on {cores} threads, {el:.1f} s"}


# ---- N > 1: nothing may hang, and nothing is timed before the slab path has been shown to agree with the single-domain path ----------
STAGE = "start"            # where the run is (the watchdog's error line says it)
_WATCHDOG = {"deadline": None, "rank": 0, "world": 1}


def stage(name, seconds=None):
    """Name the phase the run is entering; `seconds` (optional) is the wall-clock budget of the phase: when it runs out the watchdog
    thread prints ONE JSON line with an `error` field through the result descriptor (rank 0) and ends the process — a collective that
    never completes (a rank that died, a communicator that did not form) must not leave the driver waiting on a silent program."""
    global STAGE
    STAGE = name
    _WATCHDOG["deadline"] = None if seconds is None else time.monotonic() + seconds
    print(f"[bench rank {_WATCHDOG['rank']}] {name}", file=sys.stderr, flush=True)


def _start_watchdog(rank, world, n_gpus):
    import threading
    _WATCHDOG["rank"], _WATCHDOG["world"] = rank, world

    def watch():
        while True:
            time.sleep(1.0)
            dl = _WATCHDOG["deadline"]
            if dl is not None and time.monotonic() > dl:
                msg = f"rank {rank} of {world}: stage '{STAGE}' did not finish within its wall-clock budget"
                print("bench.py: " + msg, file=sys.stderr, flush=True)
                fb = _WATCHDOG.get("fallback")
                if fb is not None:   # an OPTIONAL stage ran out of time: the run's result so far is complete — print it with the stage's failure noted
                    if rank == 0:
                        fb = dict(fb)
                        fb[_WATCHDOG.get("fallback_key", "optional_stage")] = {"error": msg, "stage": STAGE}
                        emit_result(fb)
                    os._exit(0)
                if rank == 0:
                    emit_result({"metric": "particle-steps/s (LJ 1e6, rho*=0.8) + FCM-BDHI steps/s @128^3, 1/2/4/8 GPU", "value": None,
                                 "unit": "particle-steps/s", "n_gpus": n_gpus, "error": msg, "stage": STAGE})
                os._exit(4)
    threading.Thread(target=watch, daemon=True).start()


def fail_all(dist, world, rank, n_gpus, local_error, what):
    """Every rank calls this with its own verdict (None = fine); if ANY rank failed, rank 0 prints the JSON error line and all ranks exit 3."""
    bad = 1.0 if local_error else 0.0
    if _reducing(dist) and world > 1:
        bad = _allreduce(dist, [bad], "MAX")[0]
    if bad == 0.0:
        return
    if local_error:
        print(f"bench.py: rank {rank}: {what}: {local_error}", file=sys.stderr, flush=True)
    if rank == 0:
        emit_result({"metric": "particle-steps/s (LJ 1e6, rho*=0.8) + FCM-BDHI steps/s @128^3, 1/2/4/8 GPU", "value": None,
                     "unit": "particle-steps/s", "n_gpus": n_gpus, "error": f"{what}: {local_error or 'failed on another rank (see stderr)'}",
                     "stage": STAGE})
    os._exit(3)


def preflight(hip, args, world, rank, dist):
    """Before anything is timed at N > 1 (or --force-distributed): a SMALL box of each path through the run's own communication stack,
    compared on every rank with the single-domain classes on the same global system — the slab LJ step (halo exchange, migration,
    thermostat keyed by the global particle id) after 12 steps, the slab FCM solve (halo fold, two all-to-all transposes, gather halo)
    at T = 0 and T > 0.  Returns a dict for the result line; raises RuntimeError with the figure that failed."""
    import ctypes as C
    from uammd_amd._lib import check, load
    from uammd_amd.parallel import DistributedLJ, SlabDecomposition
    from uammd_amd.parallel_fcm import DistributedFCM, HipSlabBackend, SlabGeometry, make_decomposition
    lib = load()
    report = {}
    # ---- path A ---------------------------------------------------------------------------------------------------------------------
    n, rc, dt, T, steps = 16384, 2.5, 0.005, 1.0, 12
    L1 = (n / 0.8) ** (1.0 / 3.0)
    noise = math.sqrt(2 * dt * T)
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def initial(r):   # rank r's slab in ITS frame, and its velocities: the same on every rank that asks
        p = torch.from_numpy(lattice(n, L1, 1234 + r)).cuda()
        v = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
        check(lib.uammd_verletnvt_initial_velocities(C.c_void_p(v.data_ptr()), None, 1.0, 0, n, 77 + r, None))
        return p, v
    d = SlabDecomposition([L1, L1, L1 * world], rc, rank, world, skin=0.5, comm=ABI_COMM)
    pos, vel = initial(rank)
    ids = torch.arange(n, dtype=torch.int32, device="cuda") + rank * n
    pot = hip.Potential.LJ()
    pot.setPotParameters(0, 0, pot.InputPairParameters(rc, 1.0, 1.0, False))
    cl = hip.CellList()
    grids = {}

    def grid_of(box_L, periodic):
        key = (tuple(box_L), tuple(periodic))
        if key not in grids:
            box = hip.Box(box_L, periodic)
            grids[key] = (box,) + tuple(hip.CellList.create_update_grid(box, rc))
        return grids[key]

    def forces_into(allpos, box_L, periodic, fall):
        box, cd, ubox = grid_of(box_L, periodic)
        cl.update_grid(allpos, ubox, cd)
        cl.set_option("num_owned", sim.n_owned)
        cl.transverse_lj(pot.device_table(), 1, box, fall, None, None, None, 0)

    def keyed(step, p, v, f, keys, step_num):
        check(lib.uammd_verletnvt_gj_keyed(step, C.c_void_p(p.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(f.data_ptr()), None, 1.0, None,
                                           C.c_void_p(keys.data_ptr()), p.shape[0], dt, 1.0, 0, noise, step_num, 4242, st()))
    # the step the timed run takes (run_lj_distributed): second half step in the traversal's store, first half step in the halo pack and
    # the list build between refreshes — the preflight must exercise THAT path through the run's communicator
    P = lambda t: C.c_void_p(t.data_ptr())

    def forces_step2_into(allpos, box_L, periodic, fall, v):
        box, cd, ubox = grid_of(box_L, periodic)
        cl.update_grid(allpos, ubox, cd)
        cl.set_option("num_owned", sim.n_owned)
        cl.transverse_lj_gj2(pot.device_table(), 1, box, fall, v, dt, None, 1.0, False, 0)

    def pack_step1(p, v, f, keys, iu, nu, idn, nd, dzu, dzd, ou, od, step_num):
        check(lib.uammd_halo_pack_gj1(P(p), P(v), P(f), None, 1.0, P(keys), P(iu), nu, P(idn), nd, dzu, dzd, P(ou), P(od), dt, 1.0, 0, noise,
                                      step_num, 4242, st()))

    def forces_step12_into(allpos, box_L, periodic, fall, v, keys, skip, step_num):
        box, cd, ubox = grid_of(box_L, periodic)
        cl.update_grid_gj1(allpos, ubox, cd, v, fall, keys, skip, sim.n_owned, dt, 1.0, 0, noise, step_num, 4242)
        cl.set_option("num_owned", sim.n_owned)
        cl.transverse_lj_gj2(pot.device_table(), 1, box, fall, v, dt, None, 1.0, False, 0)
        fused_steps.append(step_num)
    fused_steps = []
    timed_path = os.environ.get("UAMMD_BENCH_NO_GJ2") != "1" and os.environ.get("UAMMD_BENCH_OVERLAP") != "1"
    fuse1 = timed_path and os.environ.get("UAMMD_BENCH_NO_GJ1") != "1"
    sim = DistributedLJ(d, None, lambda step, p, v, f, k: keyed(step, p, v, f, sim.current_ids, k), exchange_every=4, forces_into=forces_into,
                        forces_step2_into=forces_step2_into if timed_path else None,
                        step1_fused=(pack_step1, forces_step12_into) if fuse1 else None)
    force = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    for _ in range(steps):
        pos, vel, force, ids = sim.forward_time(pos, vel, force, ids)
    torch.cuda.synchronize()
    sim.check_skin()
    # the same global system, no decomposition, on THIS rank: all the slabs' initial states stacked along z
    N = n * world
    gp = torch.empty((N, 4), dtype=torch.float32, device="cuda")
    gv = torch.empty((N, 3), dtype=torch.float32, device="cuda")
    for r in range(world):
        p, v = initial(r)
        p[:, 2] += (r - 0.5 * (world - 1)) * L1
        gp[r * n:(r + 1) * n], gv[r * n:(r + 1) * n] = p, v
    gkeys = torch.arange(N, dtype=torch.int32, device="cuda")
    gf = torch.zeros((N, 4), dtype=torch.float32, device="cuda")
    gcl = hip.CellList()
    gbox, gcd, gubox = grid_of([L1, L1, L1 * world], [1, 1, 1])

    def gforces():
        gcl.update_grid(gp, gubox, gcd)
        gcl.transverse_lj(pot.device_table(), 1, gbox, gf, None, None, None, 0)
    gforces()
    for k in range(1, steps + 1):
        keyed(1, gp, gv, gf, gkeys, k)
        gforces()
        keyed(2, gp, gv, gf, gkeys, k)
    torch.cuda.synchronize()
    mine = pos[:, :3].clone()
    mine[:, 2] += (rank - 0.5 * (world - 1)) * L1       # the slab's frame -> the global frame
    dx = mine - gp[ids.long(), :3]
    Lg = torch.tensor([L1, L1, L1 * world], device="cuda")
    dx -= torch.round(dx / Lg) * Lg
    err = float(dx.abs().max())
    report["lj"] = {"particles_per_rank": n, "steps": steps, "owned_after": int(pos.shape[0]), "max_dx_vs_single_domain": err, "bar": 2e-4,
                    "steps_with_the_first_half_step_in_pack_and_build": len(fused_steps)}
    if not (err <= 2e-4):
        raise RuntimeError(f"slab LJ differs from the single-domain run: max |dx| {err:.3e} after {steps} steps (bar 2e-4)")
    # ---- path B ---------------------------------------------------------------------------------------------------------------------
    m, nc = 6000, 32
    cells, Lf = [nc, nc, nc * world], [float(nc), float(nc), float(nc * world)]
    kernel, a_eff = hip.Kernels.Gaussian(1.0, 1e-3)
    geom = SlabGeometry(cells, Lf, world, kernel.support[2])
    back = HipSlabBackend(geom, rank, kernel, 1.0, 1234)
    dec = make_decomposition(geom, rank, comm=ABI_COMM)
    fcm = DistributedFCM(geom, [back], [rank], comm=ABI_COMM)

    def particles(r):   # window frame of slab r
        rng = np.random.default_rng(99 + r)
        p = np.zeros((m, 4), np.float32)
        p[:, :3] = rng.uniform(-nc / 2, nc / 2, (m, 3))
        f = np.zeros((m, 4), np.float32)
        f[:, :3] = np.random.default_rng(4321 + r).normal(0, 1, (m, 3))
        return p, f
    p, f = particles(rank)
    dp, df = torch.from_numpy(p).cuda(), torch.from_numpy(f).cuda()
    allp, allf = [], []
    for r in range(world):
        pr_, fr_ = particles(r)
        pr_[:, 2] += (r - 0.5 * (world - 1)) * nc
        allp.append(pr_), allf.append(fr_)
    gP, gF = torch.from_numpy(np.concatenate(allp)).cuda(), torch.from_numpy(np.concatenate(allf)).cuda()
    ref = hip.BDHI.FCM_impl(hip.Box(Lf), cells, kernel, 1.0, 1234, a_eff)
    errs = []
    for temperature, pref in ((0.0, 0.0), (1.0, 10.0)):
        v = fcm.displacements([dp], [df], temperature, pref)[0]
        vref = ref.computeHydrodynamicDisplacements(gP, gF, m * world, temperature, pref)[rank * m:(rank + 1) * m]
        torch.cuda.synchronize()
        errs.append(float((v - vref).norm() / vref.norm()))
    report["fcm"] = {"particles_per_rank": m, "grid": cells, "rel_l2_vs_single_domain": {"T=0": errs[0], "T=1": errs[1]}, "bar": 1e-5}
    if not (max(errs) <= 1e-5):
        raise RuntimeError(f"slab FCM differs from the single-GPU solver: rel L2 {errs[0]:.3e} (T = 0), {errs[1]:.3e} (T = 1), bar 1e-5")
    return report


def run_lj_distributed(hip, args, world, rank, dist, strong=False):
    """N > 1 (or --force-distributed): z-slab domain decomposition, halo exchange + migration through the run's communicator.
    weak (default): one slab of `--particles` particles per rank, the global box L x L x N L grows along z with N.
    strong: the ONE box of `--particles` particles (BASELINE's "LJ 1e6": L = 107.72, 43 planes of cells) cut into N slabs of L / N —
    5.4 planes of cells per rank at N = 8, of which 2 are halo on either side: the harder and the named case."""
    import ctypes as C
    from uammd_amd._lib import check, load
    from uammd_amd.parallel import DistributedLJ, SlabDecomposition
    lib = load()
    n_total = args.particles * (1 if strong else world)
    L1 = 107.7217345 * (args.particles / 1_000_000) ** (1.0 / 3.0)
    rc, dt, T = 2.5, 0.005, 1.0
    noise = math.sqrt(2 * dt * 1.0 * T)
    # cached exchange: skin 0.6 sigma, ownership + halo lists refreshed every 20 steps; DistributedLJ.check_skin() verifies after the
    # run that no particle out-ran the skin between two refreshes
    d = SlabDecomposition([L1, L1, L1 if strong else L1 * world], rc, rank, world, skin=args.skin, comm=ABI_COMM)
    if strong:
        # every rank makes the same lattice of the whole box and keeps the sites of its slab (global ids = lattice sites)
        allpos = lattice(args.particles, L1, 1234)
        mine = np.nonzero((allpos[:, 2] >= d.zlo) & (allpos[:, 2] < d.zhi))[0]
        local = allpos[mine].copy()
        local[:, 2] -= d.zc
        pos = torch.from_numpy(np.ascontiguousarray(local)).cuda()
        ids = torch.from_numpy(mine.astype(np.int32)).cuda()
        del allpos
    else:
        pos = torch.from_numpy(lattice(args.particles, L1, 1234 + rank)).cuda()          # local frame: z' in [-L1/2, L1/2)
        ids = torch.arange(args.particles, dtype=torch.int32, device="cuda") + rank * args.particles
    n = int(pos.shape[0])
    vel = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    check(lib.uammd_verletnvt_initial_velocities(C.c_void_p(vel.data_ptr()), None, math.sqrt(3 * T), 0, n, 77 + rank, None))
    pot = hip.Potential.LJ()
    pot.setPotParameters(0, 0, pot.InputPairParameters(rc, 1.0, 1.0, False))
    cl = hip.CellList()

    grid_cache = {}

    def forces_into(allpos, box_L, periodic, fall):
        """owned + ghost positions -> forces of the owned rows, ACCUMULATED into fall (GJ step 1 has zeroed the owned rows)."""
        key = (tuple(box_L), tuple(periodic))
        if key not in grid_cache:
            box = hip.Box(box_L, periodic)
            grid_cache[key] = (box,) + tuple(hip.CellList.create_update_grid(box, rc))
        box, cd, ubox = grid_cache[key]
        cl.update_grid(allpos, ubox, cd)
        cl.set_option("num_owned", sim.n_owned)
        cl.transverse_lj(pot.device_table(), 1, box, fall, None, None, None, args.algo)

    def forces_step2_into(allpos, box_L, periodic, fall, v):
        """the same with GronbechJensen's second half step of the owned rows in the traversal's store (uammd_lj_transverse_celllist_gj2)"""
        key = (tuple(box_L), tuple(periodic))
        if key not in grid_cache:
            box = hip.Box(box_L, periodic)
            grid_cache[key] = (box,) + tuple(hip.CellList.create_update_grid(box, rc))
        box, cd, ubox = grid_cache[key]
        cl.update_grid(allpos, ubox, cd)
        cl.set_option("num_owned", sim.n_owned)
        cl.transverse_lj_gj2(pot.device_table(), 1, box, fall, v, dt, None, 1.0, False, args.algo)

    def forces_fn(allpos, box_L, periodic):
        f = torch.zeros((allpos.shape[0], 4), dtype=torch.float32, device=allpos.device)
        forces_into(allpos.contiguous(), box_L, periodic, f)
        return f

    def integrate_fn(step, p, v, f, step_num):
        # the thermostat's stream keyed by the GLOBAL particle id, one seed for all ranks: kicks do not depend on the decomposition
        key = sim.current_ids
        check(lib.uammd_verletnvt_gj_keyed(step, C.c_void_p(p.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(f.data_ptr()), None,
                                           1.0, None, C.c_void_p(key.data_ptr()), p.shape[0], dt, 1.0, 0, noise, step_num, 4242,
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def integrate_rows_fn(step, p, v, f, rows, keys, step_num):
        check(lib.uammd_verletnvt_gj_keyed(step, C.c_void_p(p.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(f.data_ptr()), None,
                                           1.0, None if rows is None else C.c_void_p(rows.data_ptr()), C.c_void_p(keys.data_ptr()),
                                           keys.shape[0] if rows is None else rows.shape[0], dt, 1.0, 0, noise,
                                           step_num, 4242, C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def pack_step1(p, v, f, keys, iu, nu, idn, nd, dzu, dzd, ou, od, step_num):
        """the first half step of the listed rows on the way into the halo messages (uammd_halo_pack_gj1)"""
        P = lambda t: C.c_void_p(t.data_ptr())
        check(lib.uammd_halo_pack_gj1(P(p), P(v), P(f), None, 1.0, P(keys), P(iu), nu, P(idn), nd, dzu, dzd, P(ou), P(od), dt, 1.0, 0, noise,
                                      step_num, 4242, C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def forces_step12_into(allpos, box_L, periodic, fall, v, keys, skip, step_num):
        """... of everybody else inside the list build (uammd_celllist_update_gj1), then the traversal with the second half step"""
        key = (tuple(box_L), tuple(periodic))
        if key not in grid_cache:
            box = hip.Box(box_L, periodic)
            grid_cache[key] = (box,) + tuple(hip.CellList.create_update_grid(box, rc))
        box, cd, ubox = grid_cache[key]
        cl.update_grid_gj1(allpos, ubox, cd, v, fall, keys, skip, sim.n_owned, dt, 1.0, 0, noise, step_num, 4242)
        cl.set_option("num_owned", sim.n_owned)
        cl.transverse_lj_gj2(pot.device_table(), 1, box, fall, v, dt, None, 1.0, False, args.algo)

    no_gj2 = os.environ.get("UAMMD_BENCH_NO_GJ2") == "1"
    overlap = os.environ.get("UAMMD_BENCH_OVERLAP") == "1"
    fuse1 = not no_gj2 and not overlap and os.environ.get("UAMMD_BENCH_NO_GJ1") != "1"
    sim = DistributedLJ(d, forces_fn, integrate_fn, exchange_every=args.exchange_every, forces_into=forces_into,
                        forces_step2_into=None if no_gj2 else forces_step2_into,
                        integrate_rows_fn=integrate_rows_fn if overlap else None,
                        step1_fused=(pack_step1, forces_step12_into) if fuse1 else None)
    # (UAMMD_BENCH_OVERLAP=1: the halo exchange on a side stream behind the half step of the unlisted particles.  Bit-identical
    # (tests/test_gpu_slab_lj.py) and, at a world of one, SLOWER — 0.289 against 0.255 ms: the half step fills the chip, so only RCCL's
    # 12 us kernel can hide, and the two cross-stream waits cost 6 + 15 us; DESIGN 7.  Off by default.)
    force = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    sorter = hip.CellList()

    def sort_owned(pos, vel, force, ids):
        """ParticleData::sortParticles on the owned particles (examples/misc/benchmark.cu:154-156 does it every 500 steps):
        Morton order on a coarse grid of the local frame keeps the per-step cell-list build and the traversal coherent."""
        if os.environ.get("UAMMD_BENCH_NOSORT") == "1":
            return pos, vel, force, ids
        bl, per = d.local_box()
        cd = [max(1, int(x / 10.0)) for x in bl]
        sorter.update_grid(pos.contiguous(), hip.Box(bl, per), cd)
        order = sorter.group_index().long()
        sim.reordered()
        return pos[order].contiguous(), vel[order].contiguous(), force[order].contiguous(), ids[order].contiguous()

    pos, vel, force, ids = sort_owned(pos, vel, force, ids)
    # the same synthetic input as the single-GPU line: the lattice is melted first (untimed).  The reference's initial velocities
    # are sqrt(3) too hot (Basic.cu:12-29) and the first steps out-run a skin sized for the equilibrated liquid, so the melt
    # refreshes the membership lists every other step.
    sim.exchange_every = 2
    for _ in range(args.equilibrate):
        pos, vel, force, ids = sim.forward_time(pos, vel, force, ids)
    sim.exchange_every = args.exchange_every
    pos, vel, force, ids = sort_owned(pos, vel, force, ids)
    for _ in range(args.warmup):
        pos, vel, force, ids = sim.forward_time(pos, vel, force, ids)
    pos, vel, force, ids = sort_owned(pos, vel, force, ids)
    torch.cuda.synchronize()
    _barrier(dist)
    torch.cuda.synchronize()
    sim.max_drift = None  # the skin check below covers the timed region (the reference's initial velocities are sqrt(3) too hot,
                          # Basic.cu:12-29: the first steps of the warm-up out-run a skin sized for the equilibrated liquid)
    cl.profile_enable(True)
    global SLAB_STEP
    SLAB_STEP = {"first_half_step": "listed rows in the halo pack (uammd_halo_pack_gj1), the others in the list build's hash kernel "
                                    "(uammd_celllist_update_gj1); own kernel on the refresh steps" if sim.step1_fused is not None
                 else "own kernel (uammd_verletnvt_gj_keyed)",
                 "second_half_step": "in the traversal's store (uammd_lj_transverse_celllist_gj2)" if sim.forces_step2_into is not None
                 else "own kernel (uammd_verletnvt_gj_keyed)",
                 "halo_exchange": "on a side stream behind the half step of the unlisted particles" if sim.integrate_rows_fn is not None
                 else "on the step's stream, between the first half step and the list build",
                 "skin": args.skin, "membership_refresh_every": args.exchange_every}
    t0 = time.perf_counter()
    for j in range(args.steps):
        if j > 0 and j % 500 == 0:
            pos, vel, force, ids = sort_owned(pos, vel, force, ids)
        pos, vel, force, ids = sim.forward_time(pos, vel, force, ids)
    torch.cuda.synchronize()
    _barrier(dist)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    total = float(pos.shape[0])
    if _reducing(dist):
        el = _allreduce(dist, [el], "MAX")[0]
        total = _allreduce(dist, [total], "SUM")[0]
    assert torch.isfinite(pos).all()
    assert abs(total - n_total) < 0.5, "particles were lost or duplicated in migration"
    sim.check_skin()
    if sim.max_drift is not None:   # (how much of the skin the timed region used: the cached exchange is exact while this stays below it)
        SLAB_STEP["largest_displacement_between_refreshes"] = float(sim.max_drift)
    tot, cnt = cl.profile_read()
    k_ms = tot / max(cnt, 1)
    return n_total * args.steps / el, el / args.steps * 1e3, k_ms, L1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--min-timed-seconds", type=float, default=12.0, help="with --repeats 0: repeat the timed --steps block until the LJ headline has timed at least this much GPU work")
    ap.add_argument("--repeats", type=int, default=0, help="how many times the timed --steps block is run (0 = as many as --min-timed-seconds asks, at least 10 for blocks of <= 100 steps); "
                                                           "value / ms_per_step are the median block, the spread is reported")
    ap.add_argument("--equilibrate", type=int, default=2000, help="untimed steps that melt the lattice before warm-up (part of the synthetic input; 2000 = the warm-up of the reference's benchmark.cu:129-157, SURVEY 8d)")
    ap.add_argument("--workload", default="both", choices=["lj", "fcm", "both", "pse"])
    ap.add_argument("--pse-steps", type=int, default=50)
    ap.add_argument("--cpu-pse-steps", type=int, default=5)
    ap.add_argument("--fcm-steps", type=int, default=200)
    ap.add_argument("--fcm-warmup", type=int, default=20)
    ap.add_argument("--cpu-fcm-steps", type=int, default=40)
    ap.add_argument("--fcm-sort", type=float, default=0.0, help="sort particles on a grid of this cell size first")
    ap.add_argument("--particles", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c5", action="store_true", help="skip the extra FCM C5 (256^3) line: profiling runs that must see C4 kernels only")
    ap.add_argument("--cpu-sample-steps", type=int, default=20)
    ap.add_argument("--algo", type=int, default=0)
    ap.add_argument("--sort-every", type=int, default=500, help="ParticleData::sortParticles period of the LJ run (benchmark.cu: 500)")
    ap.add_argument("--nl", default="cell", choices=["cell", "verlet"],
                    help="neighbour list of PairForces: CellList (BASELINE configs[2], default) or VerletList (the "
                         "reference's examples/misc/benchmark.cu default)")
    # (skin 0.6 / refresh every 20 steps was marginal: the fastest of N x 1e6 thermal particles moves 0.55-0.62 sigma in 20 steps — the
    # skin check tripped at 2e6 particles once the thermostat's stream was keyed by particle id; 0.5 / 10 leaves a factor 1.6)
    ap.add_argument("--skin", type=float, default=0.5, help="slab decomposition: skin of the cached halo exchange (0 = exchange sizes every step)")
    ap.add_argument("--exchange-every", type=int, default=10, help="slab decomposition: steps between ownership / halo-list refreshes "
                                                                      "(measured at world = 1: 10 / 0.4 -> 0.335 ms per step, 20 / 0.6 -> 0.315)")
    ap.add_argument("--force-distributed", action="store_true", help="use the slab-decomposition code path at N=1 too")
    ap.add_argument("--no-preflight", action="store_true", help="N > 1 / --force-distributed: skip the small slab-vs-single-domain parity runs before the timed ones")
    ap.add_argument("--stage-seconds", type=float, default=300.0, help="N > 1: wall-clock budget of process-group set-up and of the preflight, each; "
                    "a stage that runs out prints a JSON line with an `error` field and exits instead of hanging")
    ap.add_argument("--max-seconds", type=float, default=1500.0, help="N > 1: wall-clock budget of the timed workloads")
    args = ap.parse_args()

    same_device = os.environ.get("UAMMD_BENCH_SAME_DEVICE") == "1"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` on its own: start the N ranks here (one process per GPU, RCCL), exactly what the driver's
        # `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` does.  Never a silent single-rank run.
        ndev = torch.cuda.device_count()
        if ndev < args.gpus and not same_device:
            print(f"bench.py: --gpus {args.gpus} asked for but this machine shows {ndev} GPU(s): refusing to run fewer ranks than asked.  "
                  "(Single-GPU debugging of the N > 1 code path: UAMMD_BENCH_SAME_DEVICE=1 UAMMD_BENCH_BACKEND=gloo puts all ranks on "
                  "cuda:0 and stages the messages through the host.)", file=sys.stderr)
            sys.exit(2)
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))
    # From here on this process is a worker.  Its stdout is ONE JSON line (rank 0): whatever libraries print on file descriptor 1 — gloo's
    # and librccl's greetings, rocFFT notes — is sent to stderr for the whole run, and the line goes out through the saved descriptor.
    global _RESULT_FD
    sys.stdout.flush()
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: the two must agree", file=sys.stderr)
        sys.exit(2)
    dist = None
    comm_info = {"backend": None, "rccl_version": None, "devices": [torch.cuda.get_device_name(0) + " (cuda:0)"] if torch.cuda.is_available() else []}
    if world > 1 or args.force_distributed:
        _start_watchdog(rank, world, args.gpus)
        stage("process group + communicator", args.stage_seconds)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # debugging hooks (single-GPU box): UAMMD_BENCH_SAME_DEVICE=1 puts every rank on cuda:0 and UAMMD_BENCH_BACKEND=gloo
        # replaces RCCL (which refuses two ranks on one device), so that the N > 1 code path can be exercised on one GPU
        if same_device:
            local_rank = 0
        elif local_rank >= torch.cuda.device_count():
            print(f"bench.py: rank {rank} has no GPU (LOCAL_RANK {local_rank}, {torch.cuda.device_count()} visible)", file=sys.stderr)
            sys.exit(2)
        backend = os.environ.get("UAMMD_BENCH_BACKEND", "nccl")
        torch.cuda.set_device(local_rank)
        from uammd_amd.comm import _stdout_to_stderr
        with _stdout_to_stderr():   # (gloo and librccl greet on stdout; this program's stdout is one JSON line)
            import datetime
            lim = datetime.timedelta(seconds=args.stage_seconds)   # (torch's own collectives give up too, not only the watchdog)
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=lim)
            else:
                dist.init_process_group(backend, timeout=lim)
        mine = f"rank {rank}: {torch.cuda.get_device_name(local_rank)} (cuda:{local_rank}, pid {os.getpid()})"
        devs = [None] * world
        dist.all_gather_object(devs, mine)
        try:
            rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            rccl = None
        comm_info = {"backend": backend + (" (RCCL)" if backend == "nccl" else " (host-staged; all ranks on one device)" if same_device else ""),
                     "rccl_version": rccl, "devices": devs}
    else:
        torch.cuda.set_device(0)

    import uammd_amd as hip
    from uammd_amd._lib import check, load

    # ONE communication stack: every message of the slab-decomposed runs (N > 1, or --force-distributed at N = 1 where the ring closes on
    # the rank itself) goes through uammd_comm_* — RCCL behind the C ABI, the entry points include/uammd/Distributed.h drives from C++.
    # torch.distributed is then only the bootstrap that hands rank 0's RCCL id around.  The gloo harness (all ranks on one device, host
    # staged) keeps torch.distributed as the transport: a test double, and the labels say so.
    global ABI_COMM, TRANSPORT
    if world > 1 or args.force_distributed:
        backend = os.environ.get("UAMMD_BENCH_BACKEND", "nccl") if world > 1 else "nccl"
        want_abi = backend == "nccl" and os.environ.get("UAMMD_BENCH_COMM", "abi") == "abi"
        err = None
        if want_abi:
            from uammd_amd.comm import AbiComm
            try:
                ABI_COMM = AbiComm.from_torch_distributed(dist) if world > 1 else AbiComm(0, 1, AbiComm.unique_id())
            except Exception as e:   # (a rank that cannot build the communicator must not leave the others waiting: agree below)
                err = f"{type(e).__name__}: {e}"
            if world > 1:
                ok = torch.tensor([0.0 if err else 1.0], device="cuda")
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if ok.item() == 0.0 and ABI_COMM is not None:
                    ABI_COMM.close()
                    ABI_COMM = None
                    err = err or "another rank failed to build the communicator"
        if ABI_COMM is not None:
            TRANSPORT = "uammd_comm_* (RCCL behind the C ABI: csrc/comm.hip)"
            comm_info["backend"] = TRANSPORT
            comm_info["rccl_version"] = AbiComm.rccl_version() or comm_info.get("rccl_version")   # (of the librccl the library loaded)
            comm_info["bootstrap"] = "torch.distributed (hands rank 0's RCCL id to the other ranks; carries no payload)" if world > 1 else "in-process"
        elif world > 1:
            TRANSPORT = ("torch.distributed nccl (RCCL)" if backend == "nccl" else
                         f"torch.distributed {backend} (host-staged test double" + ("; all ranks on one device)" if same_device else ")"))
            comm_info["backend"] = TRANSPORT
        else:
            TRANSPORT = "in-process loop-back (world of one, no messages)"
            comm_info["backend"] = TRANSPORT
        if err:
            comm_info["abi_comm_error"] = err
        print(f"[bench rank {rank}] {comm_info['devices'][rank] if world > 1 else comm_info['devices']}; RCCL {comm_info.get('rccl_version')}; "
              f"messages through {TRANSPORT}", file=sys.stderr, flush=True)
        if not args.no_preflight:
            stage("preflight: small slab runs against the single-domain classes", args.stage_seconds)
            perr, rep_ = None, None
            try:
                rep_ = preflight(hip, args, world, rank, dist)
            except Exception as e:
                perr = f"{type(e).__name__}: {e}"
            fail_all(dist, world, rank, args.gpus, perr, "preflight")
            comm_info["preflight"] = rep_
            _settle()
        stage("timed workloads", args.max_seconds)

    if args.workload == "pse":
        if world > 1:
            print("bench.py: the PSE line is single-GPU (the near field shards like path A, the far field like path B)", file=sys.stderr)
            sys.exit(2)
        out = run_pse(hip, args)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_pse(args.cpu_pse_steps)
        out.update({"n_gpus": 1, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic"})
        emit_result(out)
        return
    if args.workload == "fcm":
        out = (run_fcm_distributed if (world > 1 or args.force_distributed) else run_fcm)(hip, args, world, rank, dist)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_fcm(args.cpu_fcm_steps)
        if not args.no_c5:
            _settle()
            out["fcm_c5"] = run_fcm_c5(hip, args, world, rank, dist)
        out.update({"n_gpus": world, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                    "data": "synthetic"})
        if rank == 0:
            out["comm"] = comm_info
            emit_result(out)
        if dist is not None:
            dist.destroy_process_group()
        return
    n = args.particles
    L = 107.7217345 * (n / 1_000_000) ** (1.0 / 3.0)
    if world > 1 or args.force_distributed:
        value, ms_per_step, k_ms, L1 = run_lj_distributed(hip, args, world, rank, dist)
        ghost_frac = 2 * 2.5 / L1 if world > 1 else 0.0
        achieved_tflops = FLOP_PER_PARTICLE * n * (1 + ghost_frac) / (k_ms * 1e-3) / 1e12
        out = {
            "metric": "particle-steps/s (LJ 1e6, rho*=0.8) + FCM-BDHI steps/s @128^3, 1/2/4/8 GPU",
            "value": value, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "LJ NVT: 1e6 particles per GPU, rho*=0.8, rc=2.5, z-slab domain decomposition (one 43-cell "
                                   f"slab per GPU, global box L x L x N*L), halo positions + migration through {TRANSPORT}",
                       "particles_per_gpu": n, "box": [L1, L1, L1 * world],
                       "parallelism": f"slab{world}: 1 process per GPU, P2P halo exchange", "slab_step": SLAB_STEP},
            "pair_interactions_per_s": 52.36 * value,
            "roofline": {"bound": "valu", "kernel": "LJ traversal (k_lj_tile4), owned + ghost particles",
                         "achieved": achieved_tflops, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tflops / PEAK_FP32_TFLOPS, "traffic": None, "kernel_ms": k_ms}}
        if args.workload == "both":
            out["fcm"] = run_fcm_distributed(hip, args, world, rank, dist)
            out["fcm_c5"] = run_fcm_c5(hip, args, world, rank, dist)
        out["comm"] = comm_info
        if world > 1 and os.environ.get("UAMMD_BENCH_NO_STRONG") != "1":
            # the STRONG-scaling line beside the weak headline: BASELINE's one 1e6 box over the N GPUs (at N = 1 it IS the single-domain
            # line).  Run LAST and as an optional stage: everything above is complete, and whatever happens here — a rank that raises, a
            # collective that never returns — the line printed is the result so far with the failure noted under "lj_strong".
            _settle()
            _WATCHDOG["fallback"], _WATCHDOG["fallback_key"] = out, "lj_strong"
            stage("strong-scaling LJ line (optional)", min(args.max_seconds, 420.0))
            serr = None
            try:
                sv, sms, sk, _ = run_lj_distributed(hip, args, world, rank, dist, strong=True)
                width = L1 / world
                out["lj_strong"] = {"value": sv, "unit": "particle-steps/s", "ms_per_step": sms, "scaling": "strong", "n_gpus": world,
                                    "particles_total": n, "particles_per_gpu": n / world, "box": [L1, L1, L1], "slab_width": width,
                                    "cell_planes_per_gpu": width / 2.5, "halo_fraction": 2 * (2.5 + 3 * args.skin) / width,
                                    "kernel_ms": sk, "steps": args.steps,
                                    "workload": f"LJ NVT: the ONE box of {n} particles (rho*=0.8, L = {L1:.4f}) cut into {world} z slabs; "
                                                "same step, halo and migration code as the weak line"}
            except Exception as e:   # (the other ranks may be waiting for this one inside a collective: the stage's budget ends them)
                serr = f"{type(e).__name__}: {e}"
                print(f"bench.py: rank {rank}: strong-scaling line failed: {serr}", file=sys.stderr, flush=True)
                out["lj_strong"] = {"error": serr}
                if rank == 0:
                    emit_result(out)
                os._exit(0)   # (no orderly shutdown with peers that may be stuck in a collective: their own watchdogs end them)
            _WATCHDOG["fallback"] = None
            stage("done")
        if rank == 0:
            emit_result(out)
        if dist is not None:
            dist.destroy_process_group()
        return
    # Single GPU from here on (world > 1 took the z-slab domain decomposition above — DESIGN.md §7).
    pd, box, pot, verlet, pf, _ = lj_setup(hip, n, L, seed=1234 + rank, nl=args.nl)
    pf.algo = args.algo
    # Synthetic input = an LJ LIQUID: the jittered lattice is melted for --equilibrate steps (untimed, part of building the input;
    # the reference's benchmark relaxes 2000 steps) so that the driver's short --warmup still times the steady state.
    for _ in range(args.equilibrate):
        verlet.forwardTime()
    for _ in range(args.warmup):
        verlet.forwardTime()
    pd.sortParticles()  # examples/misc/benchmark.cu:154-156 sorts every 500 steps: the timed region starts from a sorted state ...
    torch.cuda.synchronize()
    _barrier(dist)
    torch.cuda.synchronize()
    profiled = args.nl == "cell" and os.environ.get("UAMMD_BENCH_NOTIMER") != "1"
    if profiled:
        pf.nl.profile_enable(True)   # start / stop events on the traversal kernel's own dispatch, every launch of the timed region
    # The timed region is EXACTLY --steps steps between barrier + synchronize on both sides.  A 20-step block is 4 ms of GPU time, so the
    # block is repeated (each repeat bracketed the same way) and the line reports the MEDIAN block with the spread next to it.
    # With --repeats 0 (the default) the number of blocks follows the first block's time so that the headline's timed region holds at least
    # --min-timed-seconds (12 s: two periods of the sampler the driver reads gpu_busy from) of GPU work whatever --steps is: the driver's --steps 20 is 4 ms per block, 39 ms in ten blocks — too short
    # for anything that samples the GPU from outside (its gpu_busy probe read 0 % in rounds 1-3).
    blocks, done = [], 0
    nrep = args.repeats if args.repeats > 0 else 1
    while len(blocks) < nrep:
        t0 = time.perf_counter()
        for j in range(args.steps):
            verlet.forwardTime()
            done += 1
            if done % args.sort_every == 0:
                pd.sortParticles()  # ... and sorts again, INSIDE the timed region, after every 500th timed step (a sort is ~1.5 ms: charging
                                    # one to a 20-step run would overstate its amortised cost of 0.003 ms per step 25-fold)
        torch.cuda.synchronize()
        _barrier(dist)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if _reducing(dist):
            el = _allreduce(dist, [el], "MAX")[0]
        blocks.append(el)
        if args.repeats <= 0 and len(blocks) == 1:
            nrep = int(min(5000, max(10 if args.steps <= 100 else 1, math.ceil(args.min_timed_seconds / max(el, 1e-6)))))
            if _reducing(dist):
                nrep = int(_allreduce(dist, [float(nrep)], "MAX")[0])
    el = float(np.median(blocks))
    k_ms, k_launches = float("nan"), 0
    if profiled:
        tot, k_launches = pf.nl.profile_read()
        pf.nl.profile_enable(False)
        k_ms = tot / max(k_launches, 1)

    pos = pd.getPos().cpu().numpy()
    assert np.isfinite(pos).all(), "non finite positions after the run"
    ms_per_step = el / args.steps * 1e3
    if os.environ.get("UAMMD_BENCH_DEBUG") == "1":
        print(f"[debug] LJ timed loop: {el:.4f} s for {args.steps} steps", file=sys.stderr, flush=True)
    value = n * world * args.steps / el
    achieved_tflops = FLOP_PER_PARTICLE * n / (k_ms * 1e-3) / 1e12
    traffic = read_traffic("traffic_lj_traversal.json")
    kern = {0: "k_lj_tile4", 8: "k_lj_tile4", 10: "k_lj_tile", 9: "k_lj_ringh + k_pack_half", 7: "k_lj_ringh + k_pack_half"}.get(args.algo, f"algo {args.algo}")
    out = {
        "metric": "particle-steps/s (LJ 1e6, rho*=0.8) + FCM-BDHI steps/s @128^3, 1/2/4/8 GPU",
        "value": value, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "timed_blocks": {"repeats": len(blocks), "steps_each": args.steps, "ms_per_step_median": ms_per_step,
                         "ms_per_step_min": min(blocks) / args.steps * 1e3, "ms_per_step_max": max(blocks) / args.steps * 1e3,
                         # (the first and the last tenth of the blocks: a liquid a few thousand steps after the lattice is not the equilibrated
                         # one — measured with --equilibrate 300 / 20000 and three blocks each: traversal kernel 135.3 / 143.5 us, step 0.1849
                         # / 0.1932 ms; on FIXED positions 40 000 launches in a row run as fast as 2000 (0.1304 / 0.1311 ms, tools/time_lj.py):
                         # it is the liquid that changes, not the clocks — so the first blocks of a 12 s region run ~4 % faster than its
                         # median, which is the equilibrated liquid's)
                         "ms_per_step_first_tenth": float(np.median(blocks[:max(1, len(blocks) // 10)])) / args.steps * 1e3,
                         "ms_per_step_last_tenth": float(np.median(blocks[-max(1, len(blocks) // 10):])) / args.steps * 1e3,
                         "timed_seconds_total": float(sum(blocks))},
        "config": {"workload": "LJ NVT: 1e6 particles per GPU, rho*=0.8, rc=2.5, " +
                               ("CellList rebuilt every step, sortParticles every 500 steps with hintSortByHash(box, rc), " if args.nl == "cell" else
                                f"VerletList (1.08 rc, {getattr(pf.nl, 'rebuilds', 0)} rebuilds in {args.equilibrate + args.warmup + args.steps + 1} steps), ") +
                               f"VerletNVT::GronbechJensen T=1 dt=0.005 (BASELINE configs[2]); input = jittered lattice melted for "
                               f"{args.equilibrate} untimed steps, then {args.warmup} warm-up and {args.steps} timed steps",
                   "particles_per_gpu": n, "box": L, "cellDim": 43 if n == 1_000_000 else None,
                   "parallelism": "single GPU"},
        "pair_interactions_per_s": 52.36 * value,
        "roofline": {"bound": "valu",
                     "kernel": (kern + " (LJ traversal: cell-pair tiles, f16-MFMA distance prefilter, exact f32 evaluation of the hits)") if args.nl == "cell" else "k_lj_verlet (list traversal)",
                     "achieved": achieved_tflops, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": achieved_tflops / PEAK_FP32_TFLOPS,
                     "traffic": traffic, "kernel_ms": k_ms, "kernel_launches_timed": k_launches,
                     "note": "the kernel is bound by f32 vector-instruction issue, not HBM; achieved = the reference walk's 1.0e4 flop/particle "
                             "(340 candidates x ~30 flop, SURVEY 8d) / mean kernel time of every launch of the timed region; peak = f32 vector peak, which is "
                             "quoted on the packed v_pk_*_f32 instructions: they issue at half rate here (the drain rewritten on them: 144 -> 166 us, "
                             "DESIGN 5.2), single-issue f32 peaks at 78.6 TFLOP/s, and the kernel's issue port is busy for its whole duration "
                             "(SQ_ACTIVE_INST_VALU x 4 = 3.9e5 cycles per SIMD against 3.5e5 cycles of run time, profiles/r03_pmc_lj_tile4.txt)",
                     "frac_of_single_issue_peak": achieved_tflops / (PEAK_FP32_TFLOPS / 2),
                     "hbm_frac_compulsory": BYTES_COMPULSORY_PER_PARTICLE * n / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS},
    }
    if args.workload == "both" and args.nl == "cell" and world == 1:
        # the same box with the reference benchmark's neighbour list (examples/misc/benchmark.cu:82-84): extra information,
        # `value` above stays the CellList configuration BASELINE.json names
        del pd, verlet, pf
        pd2, _, _, verlet2, pf2, _ = lj_setup(hip, n, L, seed=1234, nl="verlet")
        for _ in range(150):
            verlet2.forwardTime()
        pd2.sortParticles()
        torch.cuda.synchronize()
        r0 = pf2.nl.rebuilds
        t1 = time.perf_counter()
        for _ in range(300):
            verlet2.forwardTime()
        torch.cuda.synchronize()
        t1 = time.perf_counter() - t1
        out["lj_verletlist"] = {"ms_per_step": t1 / 300 * 1e3, "value": n * 300 / t1, "unit": "particle-steps/s", "steps": 300,
                                "list_rebuilds": pf2.nl.rebuilds - r0, "cutOffMultiplier": 1.08}
        del pd2, verlet2, pf2
        _settle()
        # The reference's one published benchmark (examples/misc/benchmark.cu:8 "~90 FPS" on a GTX 980; parameters :172-181):
        # N = 2^20 on an FCC lattice in a 128^3 box (rho* = 0.5), rc = 2.5, dt = 0.01, T = 1, friction 1, VerletList with
        # rcutmult 1.2, sortParticles every 500 steps, 500 warm-up + 500 timed steps.
        nb, Lb = 1 << 20, 128.0
        pd3, _, _, verlet3, pf3, _ = lj_setup(hip, nb, Lb, seed=1234, dt=0.01, nl="verlet", fcc=True)
        pf3.nl.setCutOffMultiplier(1.2)
        pd3.sortParticles()
        for _ in range(500):
            verlet3.forwardTime()
        torch.cuda.synchronize()
        r0 = pf3.nl.rebuilds
        t2 = time.perf_counter()
        for j in range(500):
            verlet3.forwardTime()
            if j % 500 == 0:
                pd3.sortParticles()
        torch.cuda.synchronize()
        t2 = time.perf_counter() - t2
        out["reference_benchmark"] = {"config": "examples/misc/benchmark.cu: 1048576 LJ particles started from initLattice(L, N, fcc) (the reference's "
                                                "own generator, pinned bit for bit: tests/test_initial_conditions.py), box 128^3, rc 2.5, dt 0.01, "
                                                "GronbechJensen, VerletList x1.2, sort every 500 steps", "steps_per_s": 500 / t2,
                                      "ms_per_step": t2 / 500 * 1e3, "list_rebuilds": pf3.nl.rebuilds - r0,
                                      "published": "~90 steps/s on a GTX 980 (benchmark.cu:8), other hardware: orientation only"}
        del pd3, verlet3, pf3
        _settle()
        # the same configuration with PairForces<LJ, CellList> (the list this library is fastest with: the fused step of DESIGN 5.3)
        pd4, _, _, verlet4, pf4, _ = lj_setup(hip, nb, Lb, seed=1234, dt=0.01, nl="cell", fcc=True)
        pd4.sortParticles()
        for _ in range(500):
            verlet4.forwardTime()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for j in range(500):
            verlet4.forwardTime()
            if j % 500 == 0:
                pd4.sortParticles()
        torch.cuda.synchronize()
        t3 = time.perf_counter() - t3
        out["reference_benchmark"]["with_celllist"] = {"steps_per_s": 500 / t3, "ms_per_step": t3 / 500 * 1e3,
                                                       "note": "same box, potential, integrator and sort period with PairForces<LJ, CellList>"}
        del pd4, verlet4, pf4
        _settle()
    if args.workload == "both":
        _settle()
        fcm = run_fcm(hip, args, world, rank, dist)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            fcm["cpu_baseline"] = cpu_baseline_fcm(args.cpu_fcm_steps)
        out["fcm"] = fcm
        if not args.no_c5:
            _settle()
            out["fcm_c5"] = run_fcm_c5(hip, args, world, rank, dist)
    if args.workload == "both" and world == 1:
        _settle()
        out["pse"] = run_pse(hip, args)
        if not args.no_cpu_baseline:
            out["pse"]["cpu_baseline"] = cpu_baseline_pse(args.cpu_pse_steps)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_lj(n, L, 1234, args.cpu_sample_steps)
    if rank == 0:
        out["comm"] = comm_info
        emit_result(out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
