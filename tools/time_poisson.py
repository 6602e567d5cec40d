#!/usr/bin/env python
"""Poisson::sum(force) timing, the configurations of tools/bench_widened.py (for rocprofv3).  usage: time_poisson.py [split]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import uammd_amd as hip

split = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
n, L = 1_000_000, 128.0
rng = np.random.default_rng(1)
pos = np.zeros((n, 4), np.float32); pos[:, :3] = rng.uniform(-L / 2, L / 2, (n, 3))
q = np.random.default_rng(2).normal(0, 1, n).astype(np.float32); q -= q.mean()
pd = hip.ParticleData(n); pd.setPos(pos); pd.getCharge("write").copy_(torch.from_numpy(q))
p = hip.Poisson(pd, hip.Poisson.Parameters(box=hip.Box(L), epsilon=1.0, gw=0.5, tolerance=1e-4, split=split))
for _ in range(3):
    p.sum(force=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    p.sum(force=True)
e1.record(); torch.cuda.synchronize()
print(f"Poisson split {split}: {e0.elapsed_time(e1) / 10:.3f} ms per sum(force), grid {p.cells}, support {p.support}, near cut-off {p.nearFieldCutOff:.3f}")
