"""Throughput of the §8(f) rows beyond the two headline paths (not part of bench.py's contract line): one JSON object per line.
usage: python tools/bench_widened.py [--quick]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import uammd_amd as hip  # noqa: E402


class Fixed:
    def __init__(self, pd, f):
        self.pd, self.f = pd, f

    def sum(self, force=False, energy=False, virial=False):
        self.pd.getForce("readwrite").add_(self.f)

    def updateSimulationTime(self, t): pass
    def updateTimeStep(self, dt): pass
    def updateTemperature(self, T): pass
    def updateBox(self, box): pass


def timed(step, warm, n):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def particles(n, L, seed, z=True):
    rng = np.random.default_rng(seed)
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3 if z else 2] = rng.uniform(-0.5, 0.5, (n, 3 if z else 2)) * L
    f = np.zeros((n, 4), np.float32)
    f[:, :3 if z else 2] = rng.normal(0, 1, (n, 3 if z else 2))
    return pos, torch.from_numpy(f).cuda()


def main():
    quick = "--quick" in sys.argv
    reps = 20 if quick else 100
    out = []
    # Poisson: 1e6 charges, tolerance 1e-4
    n, L = 1_000_000, 128.0
    pos, _ = particles(n, L, 1)
    q = np.random.default_rng(2).normal(0, 1, n).astype(np.float32)
    q -= q.mean()
    pd = hip.ParticleData(n)
    pd.setPos(pos)
    pd.getCharge("write").copy_(torch.from_numpy(q))
    for split in (0.5, 1.0):
        p = hip.Poisson(pd, hip.Poisson.Parameters(box=hip.Box(L), epsilon=1.0, gw=0.5, tolerance=1e-4, split=split))
        ms = timed(lambda: p.sum(force=True), 3, 10)
        out.append({"module": "Poisson::sum(force)", "particles": n, "grid": p.cells, "support": p.support, "split": split,
                    "near_cutoff": round(p.nearFieldCutOff, 3), "ms": ms})
        del p
    del pd
    # BDHI::Quasi2D / True2D: 1e5 particles in a 512 a box
    n, L = 100_000, 512.0
    pos, f = particles(n, L, 3, z=False)
    for Scheme in (hip.BDHI.Quasi2D, hip.BDHI.True2D):
        pd = hip.ParticleData(n)
        pd.setPos(pos)
        bd = Scheme(pd, Scheme.Parameters(temperature=1.0, viscosity=1.0, hydrodynamicRadius=1.0, dt=0.001, box=hip.Box([L, L, 0.0]), seed=5))
        bd.addInteractor(Fixed(pd, f))
        out.append({"module": f"BDHI::{Scheme.__name__}::forwardTime", "particles": n, "grid": bd.cells, "support": bd.support,
                    "ms": timed(bd.forwardTime, 5, reps)})
        del bd, pd
    # BDHI::FIB and Hydro::ICM: 1e5 particles, 128^3 staggered grid, T = 1
    n, L = 100_000, 128.0
    pos, f = particles(n, L, 4)
    pd = hip.ParticleData(n)
    pd.setPos(pos)
    fib = hip.BDHI.FIB(pd, hip.BDHI.FIB.Parameters(temperature=1.0, viscosity=1.0, dt=0.001, box=hip.Box(L), cells=[128, 128, 128], seed=6))
    fib.addInteractor(Fixed(pd, f))
    out.append({"module": "BDHI::FIB::forwardTime", "particles": n, "grid": fib.cells, "ms": timed(fib.forwardTime, 5, reps)})
    del fib, pd
    pd = hip.ParticleData(n)
    pd.setPos(pos)
    icm = hip.Hydro.ICM(pd, hip.Hydro.ICM.Parameters(temperature=1.0, viscosity=1.0, density=1.0, dt=0.001, box=hip.Box(L), cells=[128, 128, 128],
                                                     seed=7))
    icm.addInteractor(Fixed(pd, f))
    out.append({"module": "Hydro::ICM::forwardTime", "particles": n, "grid": icm.cells, "ms": timed(icm.forwardTime, 5, reps)})
    del icm, pd
    # BDHI::Cholesky and BDHI::Lanczos through EulerMaruyama: 4096 particles, open boundaries
    n = 4096
    pos, f = particles(n, 64.0, 8)
    for Method in (hip.BDHI.Cholesky, hip.BDHI.Lanczos):
        pd = hip.ParticleData(n)
        pd.setPos(pos)
        par = Method.Parameters(temperature=1.0, viscosity=1.0, hydrodynamicRadius=0.5, dt=0.001, tolerance=1e-3)
        em = hip.BDHI.EulerMaruyama(pd, par, Method=Method)
        em.addInteractor(Fixed(pd, f))
        out.append({"module": f"BDHI::EulerMaruyama<{Method.__name__}>::forwardTime", "particles": n, "ms": timed(em.forwardTime, 2, 10)})
        del em, pd
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
