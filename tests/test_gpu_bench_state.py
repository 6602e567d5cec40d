"""The state bench.py TIMES, pinned against the oracle ("pin the state you time").

Every other parity input is a jittered lattice.  bench.py's headline runs on an LJ LIQUID: bench.lj_setup()'s lattice melted for 2000
steps of VerletNVT::GronbechJensen (Integrator/VerletNVT/GronbechJensen.cu:28-62,88-115) with PairForces<LJ, CellList>.  A liquid has
occupancy fluctuations (they are what send a 2 x 2 x 2-cell brick to the tile kernel's dense fallback) and pairs at r ~ 0.9 sigma; here

* test_c3_melted_state_vs_oracle builds EXACTLY that state with the product (bench.lj_setup + 2000 fused steps, C3: 1e6 particles),
  pulls the positions, and compares the product's cell tables (word for word) and forces / energy / virial (<= 1e-5) with the
  oracle on those positions, for AUTO (the tile kernel bench.py times) and EXACT; and asserts through uammd_lj_tile_stats how many
  bricks took the dense fallback (none at rho* = 0.8);
* test_c2_fused_trajectory_vs_oracle runs the product's fused step (uammd_verletnvt_gj_lj_step: what forwardTime() launches) for 20
  steps at C2 (1e5 particles) against the ORACLE's loop — half step, list build, traversal, half step — not against the unfused HIP
  sequence (tests/test_gpu_fused_step.py does that).
"""
import math
import os
import sys

import numpy as np
import pytest
import torch

from util import canon_cell_tables

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def _compare_state(hip, o32, pd, box, pot, pf, n, rc, label, algos):
    """cell tables and forces of the product on pd's CURRENT positions vs the oracle on the same positions"""
    pos = pd.getPos().cpu().numpy().copy()
    cd, ubox = hip.CellList.create_update_grid(box, rc)
    ocd, oL, oper = o32.celllist_create_grid(box.boxSize, [1, 1, 1], rc)
    assert list(cd) == list(ocd)
    ref_cl = o32.celllist_build(pos, oL, oper, ocd)
    ref_f, ref_e, ref_v = o32.lj_transverse_celllist(ref_cl, box.boxSize, [1, 1, 1], pot.table, 1, n, True, True, True)
    cl = hip.CellList()
    cl.update_grid(torch.from_numpy(pos).cuda(), ubox, cd)
    got = cl.to_host()
    assert np.array_equal(got["hash"], ref_cl["hash"])
    assert np.array_equal(got["index"], ref_cl["index"])
    assert np.array_equal(got["sortPos"].view(np.uint32), ref_cl["sortPos"].view(np.uint32))
    gs, ge = canon_cell_tables(got)
    rs, re = canon_cell_tables(ref_cl)
    assert np.array_equal(gs, rs) and np.array_equal(ge, re)
    occ = (re - rs)[rs >= 0]
    fmax = np.abs(ref_f[:, :3]).max(axis=1) + 1e-30
    fmax_reordered = np.maximum(fmax, np.median(fmax))   # tile kernels: see test_gpu_lj._check_force(reordered=True)
    out = {}
    for name, algo, stats in algos:
        f = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
        e = torch.zeros(n, dtype=torch.float32, device="cuda")
        v = torch.zeros(n, dtype=torch.float32, device="cuda")
        if stats:
            cl.tile_stats(True)
        # forces only first: the instantiation bench.py's step launches (reduced units); then with energy and virial
        cl.transverse_lj(pot.device_table(), 1, box, f, None, None, None, algo)
        torch.cuda.synchronize()
        st = cl.tile_stats(False) if stats else None
        gf = f.cpu().numpy()
        tile = algo in (0, 10)
        err = (np.abs(gf[:, :3] - ref_f[:, :3]).max(axis=1) / (fmax_reordered if tile else fmax)).max()
        nbits = int((gf[:, :3].view(np.uint32) != ref_f[:, :3].view(np.uint32)).sum())
        f.zero_()
        cl.transverse_lj(pot.device_table(), 1, box, f, e, v, None, algo)
        torch.cuda.synchronize()
        gf2 = f.cpu().numpy()
        err2 = (np.abs(gf2[:, :3] - ref_f[:, :3]).max(axis=1) / (fmax_reordered if tile else fmax)).max()
        eerr = np.abs(e.cpu().numpy() - ref_e).max() / np.abs(ref_e).max()
        verr = np.abs(v.cpu().numpy() - ref_v).max() / np.abs(ref_v).max()
        print(f"[{label}, {name}] force err {err:.2e} ({nbits} words differ; with E/V {err2:.2e}), energy {eerr:.2e}, virial {verr:.2e}"
              + (f", bricks {st['bricks']} of which dense fallback {st['fallback_bricks']}" if st else ""))
        assert err <= 1e-5 and err2 <= 1e-5 and eerr <= 1e-5 and verr <= 1e-5, name
        assert nbits == 0 or tile, name      # only the tile kernels may differ from the oracle's bits
        if tile:   # the un-floored per-particle figure (tests/test_gpu_full_size.py states the bounds)
            from test_gpu_full_size import UNFLOORED_MAX, UNFLOORED_P999
            raw = np.abs(gf[:, :3] - ref_f[:, :3]).max(axis=1) / fmax
            p999, worst = np.quantile(raw, 0.999), raw.max()
            print(f"    un-floored per-particle |dF|/max|F_i|: 99.9th percentile {p999:.2e}, max {worst:.2e} "
                  f"(the worst particle's max|F_i| is {fmax[raw.argmax()] / np.median(fmax):.1e} of the median)")
            assert p999 <= UNFLOORED_P999 and worst <= UNFLOORED_MAX, name
        out[name] = st
    print(f"[{label}] cell occupancy: mean {occ.mean():.2f}, max {occ.max()}, empty cells {int((rs < 0).sum())}; max|F| {fmax.max():.1f}")
    return out, occ


def test_c3_melted_state_vs_oracle(hip, o32):
    bench = _bench()
    n, L, rc = 1_000_000, 107.7217345, 2.5
    pd, box, pot, verlet, pf, pos0 = bench.lj_setup(hip, n, L, seed=1234)   # bench.py's rank-0 input
    for _ in range(2000):                                                   # bench.py --equilibrate (default)
        verlet.forwardTime()
    torch.cuda.synchronize()
    # a liquid, not the lattice: the particles have moved by a good fraction of sigma
    moved = np.abs(pd.getPos().cpu().numpy()[:, :3] - pos0[:, :3])
    moved = np.minimum(moved, L - moved)
    assert np.median(moved.max(axis=1)) > 0.15
    pd.sortParticles()                                                      # bench.py sorts before the timed region
    verlet.forwardTime()
    torch.cuda.synchronize()
    stats, occ = _compare_state(hip, o32, pd, box, pot, pf, n, rc, "C3 melted 2000 steps",
                                [("auto = tile4", 0, True), ("exact", 9, False)])
    st = stats["auto = tile4"]
    assert st["bricks"] == 22 ** 3
    # the dense fallback (a brick whose 4 x 4 x 4-cell halo holds more than 1024 candidates, or a wave with more than 12 words) is for
    # densities above the liquid's: at rho* = 0.8 a halo holds 805 +- 30
    assert st["fallback_bricks"] == 0
    assert occ.max() >= 18 and occ.std() > 1.5   # the fluctuations a jittered lattice does not have (mean 12.6 per cell)


def test_c2_fused_trajectory_vs_oracle(hip, o32):
    bench = _bench()
    n, L, rc, dt, T = 100_000, 50.0, 2.5, 0.005, 1.0
    pd, box, pot, verlet, pf, pos0 = bench.lj_setup(hip, n, L, seed=77)
    assert pf.algo == 0
    torch.cuda.synchronize()
    rp = pos0.copy()
    rv = pd.getVel().cpu().numpy().copy()
    noise = math.sqrt(2 * dt * 1.0 * T)

    def forces(p):
        cd, oL, oper = o32.celllist_create_grid(L, 1, rc)
        cl = o32.celllist_build(p, oL, oper, cd)
        f, _, _ = o32.lj_transverse_celllist(cl, L, 1, pot.table, 1, n)
        return f

    rf = None
    nsteps = 20
    for s in range(1, nsteps + 1):
        verlet.forwardTime()
        if s == 1:
            rf = forces(rp)
        o32.verletnvt_gj(1, rp, rv, rf, dt, 1.0, noise, s, verlet.seed)
        rf = forces(rp)
        o32.verletnvt_gj(2, rp, rv, rf, dt, 1.0, noise, s, verlet.seed)
        if s in (1, 5, 10, 20):
            torch.cuda.synchronize()
            gp, gv = pd.getPos().cpu().numpy(), pd.getVel().cpu().numpy()
            dv = np.abs(gv - rv).max(axis=1)
            print(f"[C2 fused step {s}] max|dx| {np.abs(gp[:, :3] - rp[:, :3]).max():.2e}, |dv| max {dv.max():.2e} rms {np.sqrt((dv ** 2).mean()):.2e} "
                  f"99.9th percentile {np.quantile(dv, 0.999):.2e}; max|F| {np.abs(rf[:, :3]).max():.0f}")
    assert verlet.fused_steps == nsteps   # the fused entry point ran (the first step computes f(t) the plain way)
    gp, gv, gf = pd.getPos().cpu().numpy(), pd.getVel().cpu().numpy(), pd.getForce().cpu().numpy()
    # The tile kernel sums the same pairs in another order (3e-7 of max|F| per step, max|F| ~ 250 in this hot start), and the particles in a
    # hard collision (dF/dr ~ 5e3 at r = 0.85) amplify a 1e-6 position difference into 1e-4 of velocity within a few steps: the bulk of the
    # particles stays at rounding level, the worst of 1e5 is bounded two decades above it.
    dv = np.abs(gv - rv).max(axis=1)
    assert np.abs(gp[:, :3] - rp[:, :3]).max() <= 2e-5
    assert np.sqrt((dv ** 2).mean()) <= 2e-5 and np.quantile(dv, 0.999) <= 1e-4 and dv.max() <= 2e-3
    fmax = np.abs(rf[:, :3]).max(axis=1)
    assert (np.abs(gf[:, :3] - rf[:, :3]).max(axis=1) / np.maximum(fmax, np.median(fmax))).max() <= 1e-3
