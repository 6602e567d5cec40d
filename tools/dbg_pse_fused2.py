import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import uammd_amd as hip
from uammd_amd._lib import check
from uammd_amd.md import _ptr, current_stream
n, L = 1000, 30.0
for fuse in (0, 1):
    rng = np.random.default_rng(7)
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(-L / 2, L / 2, (n, 3))
    pd = hip.ParticleData(n, seed=1)
    pd.setPos(pos)
    par = hip.BDHI.PSE.Parameters(psi=1.0, temperature=1.0, viscosity=1.0, hydrodynamicRadius=1.0, tolerance=1e-4, dt=1.0, box=hip.Box(L))
    pse = hip.BDHI.PSE(pd, par)
    check(pse.lib.uammd_pse_near_set_option(pse.near, b"fuse_recurrence", fuse))
    BdW = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    it = C.c_int(0)
    rc = pse.lib.uammd_pse_near_stochastic(pse.near, _ptr(pd.getPos()), n, 1.0, 1.0, 555, _ptr(BdW), current_stream(), C.byref(it))
    print("fuse", fuse, "rc", rc, "it", it.value, flush=True)
