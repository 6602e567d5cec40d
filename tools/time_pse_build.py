"""tools/time_pse_build.py — the PSE near field's list + pair-record build on the bench's PSE workload (for rocprofv3: tools/variants_pse.sh)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
dforce = torch.from_numpy(force).cuda()
MF = torch.zeros((N, 3), dtype=torch.float32, device="cuda")
mdot = lambda: check(lib.uammd_pse_near_mdot(pse.near, p(dpos), p(dforce), N, p(MF), st()))
print("near M F (list + records + product): %.1f us" % (1e3 * bench._timed(mdot, 100)))
