#!/usr/bin/env python
"""k_hash_agg takes 12 us in a loop of builds and 30 us inside the LJ step: which neighbour in the stream makes the difference?
usage (under tools/prof_any.sh): CASE=build|gj1+build|gj1+build+lj|step python tools/time_build_context.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import uammd_amd as hip
from util import lattice_positions

n, L = 1_000_000, 107.7217345
pd = hip.ParticleData(n, seed=1234)
pd.setPos(lattice_positions(n, L, seed=1234, jitter=0.1))
box = hip.Box(L)
pot = hip.Potential.LJ(); pot.setPotParameters(0, 0, pot.InputPairParameters(2.5, 1.0, 1.0, False))
par = hip.VerletNVT.GronbechJensen.Parameters(temperature=1.0, dt=0.005, friction=1.0)
integ = hip.VerletNVT.GronbechJensen(pd, par)
pf = hip.PairForces(pd, box, pot)
integ.addInteractor(pf)
pd.hintSortByHash(box, [2.5] * 3)
for _ in range(100):
    integ.forwardTime()
pd.sortParticles()
integ.forwardTime()
torch.cuda.synchronize()
case = os.environ.get("CASE", "build")
cd, ubox = hip.CellList.create_update_grid(box, [2.5] * 3)
cl = pf.nl
if "fresh" in case:      # another list object on the same positions
    cl = hip.CellList()
if "copy" in case:       # the same list, a copy of the positions in another allocation
    pos_copy = pd.getPos("read").clone()
f = pd.getForce("readwrite")
print("MARK start", case, flush=True)
for _ in range(int(os.environ.get("REPS", "2000"))):   # (many: the 100 builds of the melt are in the profile too)
    pos = pos_copy if "copy" in case else pd.getPos("read")
    if case == "step":
        integ.forwardTime()
        continue
    if "gj1" in case:
        integ._integrate(1)
    cl.update_grid(pos, ubox, cd)
    if "lj" in case:
        cl.transverse_lj(pot.device_table(), 1, box, f, None, None, None, 0)
torch.cuda.synchronize()
