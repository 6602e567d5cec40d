"""tools/time_pse.py — the PSE near-field product and the near-noise Lanczos on the bench's PSE workload, timed on their own."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

sys.argv = [sys.argv[0]]
import bench
import uammd_amd as hip
from uammd_amd._lib import check, load

lib = load()
pos, force = bench._pse_inputs()
pd = hip.ParticleData(bench.PSE_N, seed=1234)
pd.setPos(pos)
par = hip.BDHI.PSE.Parameters(temperature=1.0, viscosity=1.0, hydrodynamicRadius=1.0, tolerance=bench.PSE_TOL, dt=0.01,
                              box=hip.Box(bench.PSE_L), psi=bench.PSE_PSI)
pse = hip.BDHI.PSE(pd, par)
N = bench.PSE_N
p = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
dpos = pd.getPos("read")
v3 = torch.randn((N, 3), dtype=torch.float32, device="cuda")
MF = torch.zeros((N, 3), dtype=torch.float32, device="cuda")
dot = lambda: check(lib.uammd_pse_near_dot(pse.near, p(dpos), p(v3), N, p(MF), st()))
check(lib.uammd_pse_near_set_option(pse.near, b"near_kernel", 0))
print("near product, wave per cell (list reused): %.1f us" % (1e3 * bench._timed(dot, 200)))
ref = MF.clone()
check(lib.uammd_pse_near_set_option(pse.near, b"near_kernel", 1))
print("near product, 8 lanes per particle        : %.1f us" % (1e3 * bench._timed(dot, 200)))
print("max |8-lane - cell| / max|Mv| = %.2e" % float((MF - ref).abs().max() / ref.abs().max()))
check(lib.uammd_pse_near_set_option(pse.near, b"exact_order", 1))
print("near product, exact order : %.1f us" % (1e3 * bench._timed(dot, 50)))
print("max |8-lane - exact| / max|Mv| = %.2e" % float((MF - ref).abs().max() / ref.abs().max()))
check(lib.uammd_pse_near_set_option(pse.near, b"exact_order", 0))
print("near noise (Lanczos)      : %.1f us, %d iterations" % (1e3 * bench._timed(lambda: pse._near_stochastic(MF, 1.0, 1.0), 50), pse.lastLanczosIterations))
dforce = torch.from_numpy(force).cuda()
print("far field                 : %.1f us" % (1e3 * bench._timed(lambda: pse._far(dforce, MF, 1.0, 10.0), 50)))
