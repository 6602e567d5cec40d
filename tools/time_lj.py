#!/usr/bin/env python
"""Traversal-only timing of the LJ kernels on a melted C3-like configuration (tools, not part of bench.py's contract).
usage: python tools/time_lj.py [N] [algo ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import uammd_amd as hip
from util import lattice_positions

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
algos = [int(a) for a in sys.argv[2:]] or [7, 8]
L = 107.7217345 * (n / 1e6) ** (1 / 3)
pd = hip.ParticleData(n, seed=1234)
pd.setPos(lattice_positions(n, L, seed=1234, jitter=0.1))
box = hip.Box(L)
pot = hip.Potential.LJ(); pot.setPotParameters(0, 0, pot.InputPairParameters(2.5, 1.0, 1.0, False))
par = hip.VerletNVT.GronbechJensen.Parameters(temperature=1.0, dt=0.005, friction=1.0)
integ = hip.VerletNVT.GronbechJensen(pd, par)
pf = hip.PairForces(pd, box, pot); pf.algo = 7
integ.addInteractor(pf)
pd.hintSortByHash(box, [2.5] * 3)
for _ in range(int(os.environ.get("MELT", "300"))):
    integ.forwardTime()
pd.sortParticles()
integ.forwardTime()
cl = pf.nl
if os.environ.get("OUTSIDE"):   # every particle stored in a random periodic image (what unwrapped trajectories become): all tiles take the
    g = torch.Generator(device="cuda").manual_seed(3)   # minimum-image variant
    pos = pd.getPos("readwrite")
    pos[:, :3] += L * torch.randint(-1, 2, (n, 3), generator=g, device="cuda").float()
    integ.forwardTime()
f = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
ref = None
if os.environ.get("TIMELINE"):   # a -DUAMMD_TILE_TIMELINE build (tools/variants_tile.sh): where a workgroup's lifetime goes
    cl.tile_stats(True)
    cl.transverse_lj(pot.device_table(), 1, box, f, None, None, None, 8)
    torch.cuda.synchronize()
    st = cl.tile_stats(False)
    t, nb = st["timeline"], st["bricks"]
    nw = max(t[4], 1)
    print(f"timeline (us, 100 MHz ticks / 100): bricks {nb}, waves with owners {t[4]}; per wave of a brick: ranges known {t[0] / (4 * nb) / 100:.2f}, "
          f"staging landed {t[1] / (4 * nb) / 100:.2f}, halo staged (barrier) {t[2] / (4 * nb) / 100:.2f}, wave done {t[3] / nw / 100:.2f}", flush=True)
for algo in algos:
    f.zero_()
    cl.transverse_lj(pot.device_table(), 1, box, f, None, None, None, algo)
    torch.cuda.synchronize()
    got = f.cpu().numpy()
    if ref is None:
        ref = got
    err = np.abs(got[:, :3] - ref[:, :3]).max() / np.abs(ref[:, :3]).max()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = int(os.environ.get("REPS", "50"))
    e0.record()
    for _ in range(reps):
        cl.transverse_lj(pot.device_table(), 1, box, f, None, None, None, algo)
    e1.record(); torch.cuda.synchronize()
    print(f"algo {algo}: {e0.elapsed_time(e1) / reps:.4f} ms per launch; max|dF|/max|F| vs first algo {err:.2e}", flush=True)
