import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import uammd_amd as hip
from uammd_amd._lib import check
from uammd_amd.md import _ptr, current_stream

def run(n, L, fuse, seed2=555, psi=1.0, tol=1e-4):
    rng = np.random.default_rng(7)
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(-L / 2, L / 2, (n, 3))
    pd = hip.ParticleData(n, seed=1)
    pd.setPos(pos)
    par = hip.BDHI.PSE.Parameters(psi=psi, temperature=1.0, viscosity=1.0, hydrodynamicRadius=1.0, tolerance=tol, dt=1.0, box=hip.Box(L))
    pse = hip.BDHI.PSE(pd, par)
    check(pse.lib.uammd_pse_near_set_option(pse.near, b"fuse_recurrence", int(fuse)))
    out = []
    for rep in range(3):
        BdW = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
        it = C.c_int(0)
        rc = pse.lib.uammd_pse_near_stochastic(pse.near, _ptr(pd.getPos()), n, 1.0, 1.0, seed2 + rep, _ptr(BdW), current_stream(), C.byref(it))
        torch.cuda.synchronize()
        print(f"n={n} fuse={fuse} rep={rep} rc={rc} it={it.value} |BdW|={float(BdW.norm()):.6e}", pse.lib.uammd_hip_last_error() if rc else "")
        out.append(BdW.cpu().numpy())
    return out

for n, L in ((1, 32.0), (2, 8.0), (1000, 30.0), (20000, 64.0)):
    a = run(n, L, 0)
    b = run(n, L, 1)
    for x, y in zip(a, b):
        print("   max|diff| / max|x| =", np.abs(x - y).max() / max(np.abs(x).max(), 1e-30))
