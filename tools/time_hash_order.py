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
for name, hint in (("cell grid", rc), ("coarse 10 sigma", 10.0), ("unsorted", None)):
    pd = hip.ParticleData(n, seed=1)
    pd.setPos(base)
    if hint is not None:
        pd.hintSortByHash(box, [hint] * 3)
        pd.sortParticles()
    pos = pd.getPos("read")
    cl = hip.CellList()
    cl.update_grid(pos, ubox, cd)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        cl.update_grid(pos, ubox, cd)
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per build", flush=True)
