#!/usr/bin/env python
"""AUTO traversal timing of the variants the headline does not exercise: two particle types (parameter lookup per pair), energy + virial
outputs, every particle stored in another periodic image.  usage: python tools/time_lj_variants.py [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import uammd_amd as hip
from util import lattice_positions

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = 107.7217345 * (n / 1e6) ** (1 / 3)
for ntypes in (1, 2):
    pd = hip.ParticleData(n, seed=1234)
    pos = lattice_positions(n, L, seed=1234, jitter=0.1, ntypes=ntypes)
    pd.setPos(pos)
    box = hip.Box(L)
    pot = hip.Potential.LJ()
    for a in range(ntypes):
        for b in range(a, ntypes):
            pot.setPotParameters(a, b, pot.InputPairParameters(2.5, 1.0, 1.0 if a == b else 0.9, False))
    par = hip.VerletNVT.GronbechJensen.Parameters(temperature=1.0, dt=0.005, friction=1.0)
    integ = hip.VerletNVT.GronbechJensen(pd, par)
    pf = hip.PairForces(pd, box, pot)
    integ.addInteractor(pf)
    pd.hintSortByHash(box, [2.5] * 3)
    for _ in range(200):
        integ.forwardTime()
    pd.sortParticles()
    integ.forwardTime()
    cl = pf.nl
    f = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    e = torch.zeros(n, dtype=torch.float32, device="cuda")
    v = torch.zeros(n, dtype=torch.float32, device="cuda")
    for name, (ee, vv) in (("force", (None, None)), ("force+energy+virial", (e, v))):
        for _ in range(3):
            cl.transverse_lj(pot.device_table(), ntypes, box, f, ee, vv, None, 0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            cl.transverse_lj(pot.device_table(), ntypes, box, f, ee, vv, None, 0)
        e1.record(); torch.cuda.synchronize()
        print(f"{ntypes} type(s), {name}: {e0.elapsed_time(e1) / 50:.4f} ms per traversal", flush=True)
    del integ, pf, pd
