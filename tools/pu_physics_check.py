#!/usr/bin/env python
"""Is the blow-up of the reference's examples/advanced/ParameterUpdatable.cu physics or a fault of the custom-Transverser path?  The same
system through the LIBRARY's Lennard-Jones path (Potential::LJ with epsilon(t) set each step, the fused kernels) and torch for the wall
and the gravity: box 32 x 32 x 40, 16384 particles on the fcc lattice, GronbechJensen T = 0.6, dt = 0.005, friction 1,
wall force -0.1 (z - (-20 + 3 sin(pi t))), gravity -1 while t < 10, epsilon(t) = 1 - 0.5 exp(-0.1 t), cut-off 2.5."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import uammd_amd as hip
from uammd_amd.initial_conditions import init_lattice

n, L = 16384, [32.0, 32.0, 40.0]
pos = init_lattice(L, n, "fcc")
pd = hip.ParticleData(n, seed=int(os.environ.get("SEED", 1234)))
pd.setPos(pos)
box = hip.Box(L)
pot = hip.Potential.LJ()
pot.setPotParameters(0, 0, pot.InputPairParameters(2.5, 1.0, 0.5, False))
par = hip.VerletNVT.GronbechJensen.Parameters(temperature=0.6, dt=0.005, friction=1.0, initVelocities=True)
verlet = hip.VerletNVT.GronbechJensen(pd, par)
pf = hip.PairForces(pd, box, pot)

class Extra:
    def __init__(self): self.t = 0.0
    def sum(self, force=True, energy=False, virial=False):
        f = pd.getForce("readwrite"); p = pd.getPos("read")
        f[:, 2] += -0.1 * (p[:, 2] - (-20.0 + 3.0 * math.sin(math.pi * self.t)))
        if self.t < 10: f[:, 2] -= 1.0
    def updateSimulationTime(self, t):
        self.t = t
        pot.setPotParameters(0, 0, pot.InputPairParameters(2.5, 1.0, 1 - 0.5 * math.exp(-0.1 * t), False))
    def updateTimeStep(self, dt): pass
    def updateTemperature(self, T): pass
    def updateBox(self, b): pass

verlet.addInteractor(pf); verlet.addInteractor(Extra())
steps = int(os.environ.get("STEPS", 20000))
for s in range(steps):
    verlet.forwardTime()
    if s % 1000 == 0 or s == steps - 1:
        p = pd.getPos("read"); v = pd.getVel("read")
        z = p[:, 2]
        print(f"step {s:6d} t = {s * 0.005:6.2f}: finite {bool(torch.isfinite(p).all())}, |v|max {float(v.norm(dim=1).max()):8.2f}, kT {float((v * v).mean()):6.3f}, z in [{float(z.min()):7.2f}, {float(z.max()):7.2f}], median z {float(z.median()):7.2f}", flush=True)
        if not torch.isfinite(p).all(): break
