#!/usr/bin/env python
"""Cell-list build time against the ORDER of the input (C3 size, uniform random positions): sorted on the cell grid itself, on a
coarse grid, unsorted.  usage: python tools/time_hash_order.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import uammd_amd as hip

n, L, rc = 1_000_000, 107.7217345, 2.5
rng = np.random.default_rng(1)
base = np.zeros((n, 4), np.float32)
base[:, :3] = rng.uniform(-L / 2, L / 2, (n, 3))
box = hip.Box(L)
cd, ubox = hip.CellList.create_update_grid(box, [rc] * 3)
cases = (("cell grid", rc), ("coarse 10 sigma", 10.0), ("unsorted", None), ("cell grid + 0.3 drift wrapped", rc), ("cell grid + 0.3 drift", rc), ("cell grid, rewritten before every build", rc))
only = os.environ.get("ORDER")   # e.g. ORDER="cell grid" to profile one case
for name, hint in [c for c in cases if only in (None, c[0])]:
    pd = hip.ParticleData(n, seed=1)
    pd.setPos(base)
    if hint is not None:
        pd.hintSortByHash(box, [hint] * 3)
        pd.sortParticles()
    pos = pd.getPos("read")
    if "drift" in name:   # the order a simulation has some steps after a sort
        g = torch.Generator(device="cuda").manual_seed(7)
        pos = pos.clone()
        pos[:, :3] += 0.3 * torch.randn((n, 3), generator=g, device="cuda")
        if "wrapped" in name:
            pos[:, :3] -= torch.floor(pos[:, :3] / L + 0.5) * L
    cl = hip.CellList()
    cl.update_grid(pos, ubox, cd)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        if "rewritten" in name:   # as in a simulation: another kernel has just written the positions
            pos.mul_(1.0)
        cl.update_grid(pos, ubox, cd)
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per build", flush=True)
