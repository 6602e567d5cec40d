#!/usr/bin/env python
"""Cell-list build timing on a melted C3-like configuration; checks the build variants against each other.
usage: python tools/time_build.py [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import uammd_amd as hip
from util import lattice_positions

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = 107.7217345 * (n / 1e6) ** (1 / 3)
pd = hip.ParticleData(n, seed=1234)
pd.setPos(lattice_positions(n, L, seed=1234, jitter=0.1))
box = hip.Box(L)
pot = hip.Potential.LJ(); pot.setPotParameters(0, 0, pot.InputPairParameters(2.5, 1.0, 1.0, False))
par = hip.VerletNVT.GronbechJensen.Parameters(temperature=1.0, dt=0.005, friction=1.0)
integ = hip.VerletNVT.GronbechJensen(pd, par)
pf = hip.PairForces(pd, box, pot)
integ.addInteractor(pf)
pd.hintSortByHash(box, [2.5] * 3)
for _ in range(int(os.environ.get("MELT", "300"))):
    integ.forwardTime()
pd.sortParticles()
for _ in range(100):   # the bench's steady state: up to 500 steps since the last sort
    integ.forwardTime()
pos = pd.getPos("read")
cd, ubox = hip.CellList.create_update_grid(box, [2.5] * 3)
outs = {}
for name, opts in (("counting", {}), ("radix", {"force_radix": 1})):
    cl = hip.CellList()
    for k, v in opts.items():
        cl.set_option(k, v)
    cl.update_grid(pos, ubox, cd)
    outs[name] = cl.to_host()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        cl.update_grid(pos, ubox, cd)
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per build", flush=True)
for name in ("radix",):
    a, b = outs["counting"], outs[name]
    same = all(np.array_equal(a[k], b[k]) for k in ("index", "hash", "sortPos", "cellEnd"))
    va, vb = a["cellStart"] >= a["validCell"], b["cellStart"] >= b["validCell"]
    same = same and np.array_equal(va, vb) and np.array_equal((a["cellStart"] - a["validCell"])[va], (b["cellStart"] - b["validCell"])[vb])
    print(f"counting vs {name}: {'identical' if same else 'DIFFERENT'}")
