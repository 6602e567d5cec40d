#!/usr/bin/env python
"""Do the tile-owned spread (latency bound) and the x / y FFT passes (HBM bound) overlap when queued on two streams?  Two z-slabs of the
grid as two slab handles: spread of slab B on one stream, forward_xy of slab A on another; alone and together.  (NC, N from the environment)
Round 6: this probe said yes (C5 halves: 149 + 74 us alone, 162 together) and the solve built on it said no — the slab spread is four
dependent launches whose gaps the other stream filled; the tile kernel itself and the row pass slow each other down in proportion
(66 -> 101 us and 36 -> 67 us side by side: the spread's round trips lengthen under the streaming pass's load, and its resident workgroups
take the slots the pass needs).  DESIGN.md 9."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import uammd_amd as hip
from uammd_amd.parallel_fcm import SlabGeometry, HipSlabBackend

nc, n = int(os.environ.get("NC", 256)), int(os.environ.get("N", 200000))
L = float(nc)
k, a_eff = hip.Kernels.Gaussian(1.0, 1e-3)
geom = SlabGeometry([nc] * 3, [L] * 3, 2, k.support[2] if hasattr(k, "support") else 6)
A, B = HipSlabBackend(geom, 0, k, 1.0, 1234), HipSlabBackend(geom, 1, k, 1.0, 1234)
rng = np.random.default_rng(1)
nl = n // 2
pos = np.zeros((nl, 4), np.float32)
pos[:, :2] = rng.uniform(-L / 2, L / 2, (nl, 2)); pos[:, 2] = rng.uniform(0, geom.nzl * L / nc, nl) - 0.5 * geom.nzl * L / nc
frc = np.zeros((nl, 4), np.float32); frc[:, :3] = rng.normal(0, 1, (nl, 3))
dp, df = torch.from_numpy(pos).cuda(), torch.from_numpy(frc).cuda()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def spread():
    with torch.cuda.stream(s1): B.spread(dp, df)
def xy():
    with torch.cuda.stream(s2): A.forward_xy(A.grid)
def gather():
    with torch.cuda.stream(s1): B.gather(dp, B.grid)
def inv():
    with torch.cuda.stream(s2): A.inverse_xy(A.grid)

def timed(fs, reps=200):
    for f in fs: f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        for f in fs: f()
        # the two streams meet after every round, as the halves of a solve would
        e1, e2 = torch.cuda.Event(), torch.cuda.Event()
        e1.record(s1); e2.record(s2); s1.wait_event(e2); s2.wait_event(e1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6

for name, fs in (("spread(B)", [spread]), ("forward_xy(A)", [xy]), ("spread(B) || forward_xy(A)", [spread, xy]),
                 ("gather(B)", [gather]), ("inverse_xy(A)", [inv]), ("gather(B) || inverse_xy(A)", [gather, inv])):
    print(f"{nc}^3 half slabs, {nl} particles: {name:30s} {timed(fs):8.1f} us per round")
