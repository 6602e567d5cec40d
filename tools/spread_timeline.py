#!/usr/bin/env python
"""Where a tile of k_fcm_spread_tile spends its life (a -DUAMMD_SPREAD_TIMELINE build: tools/variants_fcm.sh, VNAMES=timeline).
usage: UAMMD_HIP_LIB=tools/_build/libf_timeline.so python tools/spread_timeline.py   (N, NC from the environment)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import uammd_amd as hip
from uammd_amd._lib import load

n, nc = int(os.environ.get("N", 100000)), int(os.environ.get("NC", 128))
L = float(nc)
rng = np.random.default_rng(1234)
pos = np.zeros((n, 4), np.float32); pos[:, :3] = rng.uniform(-L / 2, L / 2, (n, 3))
force = np.zeros((n, 4), np.float32); force[:, :3] = rng.normal(0, 1, (n, 3))
k, a_eff = hip.Kernels.Gaussian(1.0, 1e-3)
fcm = hip.BDHI.FCM_impl(hip.Box(L), [nc] * 3, k, 1.0, 1234, a_eff)
dp, df = torch.from_numpy(pos).cuda(), torch.from_numpy(force).cuda()
out = torch.empty((n, 3), dtype=torch.float32, device="cuda")
lib = load()
fn = lib._lib.uammd_debug_spread_timeline if hasattr(lib, "_lib") else C.CDLL(os.environ["UAMMD_HIP_LIB"]).uammd_debug_spread_timeline
buf = (C.c_ulonglong * 8)()
for _ in range(5):
    fcm.computeHydrodynamicDisplacements(dp, df, n, 1.0, 10.0, out=out)
torch.cuda.synchronize()
fn(buf)
reps = 20
for _ in range(reps):
    fcm.computeHydrodynamicDisplacements(dp, df, n, 1.0, 10.0, out=out)
torch.cuda.synchronize()
fn(buf)
tiles = buf[7]
names = ["ranges known", "list complete", "weights in LDS", "matrix phase done", "stored"]
print(f"{nc}^3, {n} particles: {tiles // reps} tiles per launch; mean time since workgroup start (us):")
for i, nm in enumerate(names):
    print(f"  {nm:18s} {buf[i] / tiles / 100:.2f}")
