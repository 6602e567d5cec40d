#!/usr/bin/env python
"""BDHI::FCMIntegrator step timing with library options (tools; bench.py's FCM leg with switches).
usage: time_fcm_step.py [opt=value ...]   e.g. fused_update=0 bin_ahead=0   (N, NC from the environment)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import uammd_amd as hip
import bench

n, nc = int(os.environ.get("N", 100000)), int(os.environ.get("NC", 128))
pd, integ, _, _ = bench.fcm_setup(hip, n, [nc] * 3, float(nc), seed=1234)
for a in sys.argv[1:]:
    name, v = a.split("=")
    integ.fcm.set_option(name, int(v))
for _ in range(20):
    integ.forwardTime()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 200
e0.record()
for _ in range(reps):
    integ.forwardTime()
e1.record(); torch.cuda.synchronize()
print(f"{' '.join(sys.argv[1:]) or 'default'}: {e0.elapsed_time(e1) / reps:.4f} ms per step ({nc}^3, {n} particles)", flush=True)
