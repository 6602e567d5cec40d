#!/usr/bin/env python
"""FCM step timing with library options (tools; not part of bench.py's contract).  usage: time_fcm.py [opt=value ...]"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import uammd_amd as hip
from uammd_amd._lib import check

n, nc = int(os.environ.get("N", 100000)), int(os.environ.get("NC", 128))
L, cells = float(nc), [nc] * 3
rng = np.random.default_rng(1234)
pos = np.zeros((n, 4), np.float32); pos[:, :3] = rng.uniform(-L / 2, L / 2, (n, 3))
force = np.zeros((n, 4), np.float32); force[:, :3] = rng.normal(0, 1, (n, 3))
k, a_eff = hip.Kernels.Gaussian(1.0, 1e-3)
fcm = hip.BDHI.FCM_impl(hip.Box(L), cells, k, 1.0, 1234, a_eff)
for a in sys.argv[1:]:
    name, v = a.split("=")
    check(fcm.lib.uammd_fcm_set_option(fcm.h, name.encode(), int(v)))
dp, df = torch.from_numpy(pos).cuda(), torch.from_numpy(force).cuda()
out = torch.empty((n, 3), dtype=torch.float32, device="cuda")
for _ in range(10):
    fcm.computeHydrodynamicDisplacements(dp, df, n, float(os.environ.get("T", 1.0)), 10.0, out=out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = int(os.environ.get("REPS", "100"))
e0.record()
for _ in range(reps):
    fcm.computeHydrodynamicDisplacements(dp, df, n, float(os.environ.get("T", 1.0)), 10.0, out=out)
e1.record(); torch.cuda.synchronize()
print(f"{' '.join(sys.argv[1:]) or 'default'}: {e0.elapsed_time(e1) / reps:.4f} ms per solve ({nc}^3, {n} particles)", flush=True)
