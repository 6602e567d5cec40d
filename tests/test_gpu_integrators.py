"""GPU parity: integrator kernels vs the oracle.

Deterministic part (T = 0): bit exact.  With noise the Gaussians go through logf/sinf/cosf, which
differ between libm and the device by a few ulp: tolerance 2e-6 * noise amplitude (stated here).
The integer Saru stream itself is checked bit-exactly through the uniform path.
"""
import math

import numpy as np
import pytest
import torch

from util import lattice_positions

pytestmark = pytest.mark.gpu


def _state(n, seed=3):
    rng = np.random.default_rng(seed)
    pos = lattice_positions(n, 20.0, seed=seed)
    vel = rng.normal(0, 1, (n, 3)).astype(np.float32)
    force = np.zeros((n, 4), np.float32)
    force[:, :3] = rng.normal(0, 5, (n, 3))
    return pos, vel, force


@pytest.mark.parametrize("kind", ["gj", "basic"])
@pytest.mark.parametrize("T", [0.0, 1.3])
def test_verletnvt_steps(hip, o32, kind, T):
    import ctypes as C
    from uammd_amd._lib import check, load
    lib = load()
    n, dt, friction, seed = 10007, 0.005, 1.0, 0xC0FFEE
    noise = math.sqrt(2 * dt * friction * T)
    pos, vel, force = _state(n)
    rp, rv, rf = pos.copy(), vel.copy(), force.copy()
    dp, dv, df = (torch.from_numpy(a.copy()).cuda() for a in (pos, vel, force))
    fn_o = o32.verletnvt_gj if kind == "gj" else o32.verletnvt_basic
    fn_g = lib.uammd_verletnvt_gj if kind == "gj" else lib.uammd_verletnvt_basic
    for step in (1, 2):
        if step == 2:  # forces of the new configuration
            rf[:, :3] = force[:, :3][::-1]
            df.copy_(torch.from_numpy(rf))
        fn_o(step, rp, rv, rf, dt, friction, noise, 17, seed)
        check(fn_g(step, C.c_void_p(dp.data_ptr()), C.c_void_p(dv.data_ptr()), C.c_void_p(df.data_ptr()), None, 1.0,
                   None, n, dt, friction, 0, noise, 17, seed, None))
        torch.cuda.synchronize()
        gp, gv, gf = dp.cpu().numpy(), dv.cpu().numpy(), df.cpu().numpy()
        if T == 0.0:
            assert np.array_equal(gp, rp) and np.array_equal(gv, rv)
        else:
            tol = 2e-6 * max(noise, 1e-30) * 10 + 1e-7 * np.abs(rv).max()
            assert np.abs(gp - rp).max() <= tol and np.abs(gv - rv).max() <= tol
        if step == 1:
            assert np.all(gf == 0) and np.all(rf == 0)


def test_initial_velocities(hip, o32):
    import ctypes as C
    from uammd_amd._lib import check, load
    n, T = 5000, 2.0
    vamp = math.sqrt(3 * T)
    ref = o32.verletnvt_initial_velocities(n, vamp, 4242)
    v = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    check(load().uammd_verletnvt_initial_velocities(C.c_void_p(v.data_ptr()), None, vamp, 0, n, 4242, None))
    torch.cuda.synchronize()
    assert np.abs(v.cpu().numpy() - ref).max() <= 1e-5 * vamp
    assert abs(v.cpu().numpy().std() - vamp) < 0.05 * vamp


@pytest.mark.parametrize("T", [0.0, 1.0])
def test_bd_euler_maruyama(hip, o32, T):
    """README example shape (config C1): non interacting particles, dt = 0.1."""
    import ctypes as C
    from uammd_amd._lib import check, load
    n, dt, M, seed = 10000, 0.1, 1.0 / (6 * math.pi), 1234567
    rng = np.random.default_rng(1)
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(-0.5, 0.5, (n, 3))
    force = np.zeros((n, 4), np.float32)
    force[:, :3] = rng.normal(0, 1, (n, 3))
    K = np.array([[0, 0.1, 0], [0, 0, 0], [0.2, 0, 0]], np.float32)
    rp = pos.copy()
    dp = torch.from_numpy(pos.copy()).cuda()
    dforce = torch.from_numpy(force).cuda()
    Kc = (C.c_float * 9)(*[float(x) for x in K.reshape(9)])
    for step in range(1, 4):
        o32.bd_euler_maruyama(rp, force, M, dt, T, step, seed, K=K)
        check(load().uammd_bd_euler_maruyama(C.c_void_p(dp.data_ptr()), None, C.c_void_p(dforce.data_ptr()), Kc, M,
                                             None, dt, 0, T, n, step, seed, None))
        torch.cuda.synchronize()
        if T == 0:
            assert np.array_equal(dp.cpu().numpy(), rp)
        else:
            B = math.sqrt(2 * T * M * dt)
            assert np.abs(dp.cpu().numpy() - rp).max() <= 2e-5 * B * step


@pytest.mark.parametrize("T", [0.0, 1.0])
@pytest.mark.parametrize("scheme", ["MidPoint", "AdamsBashforth", "Leimkuhler"])
def test_bd_other_schemes(hip, o32, scheme, T):
    """BD::MidPoint / AdamsBashforth / Leimkuhler (Integrator/BrownianDynamics.cu:160-387) through uammd_bd_scheme_step against the oracle's
    restatement, three steps as forwardTime strings the calls: a proper subgroup, shear, per-particle radii, forces that change between
    the calls.  T = 0: the same bits; T > 0: the Gaussian draws differ by libm vs device logf / sinf (2e-5 of the noise amplitude)."""
    import ctypes as C
    from uammd_amd._lib import check, load
    lib = load()
    n, dt, M, seed = 6000, 0.1, 1.0 / (6 * math.pi), 24680
    rng = np.random.default_rng(3)
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(-0.5, 0.5, (n, 3))
    pos[:, 3] = rng.integers(0, 3, n)
    radius = rng.uniform(0.5, 1.5, n).astype(np.float32)
    index = np.sort(rng.choice(n, 4000, replace=False)).astype(np.int32)
    original = rng.permutation(n).astype(np.int32)
    K = np.array([[0, 0.1, 0], [0, 0, 0], [0.2, 0, 0]], np.float32)
    Kc = (C.c_float * 9)(*[float(x) for x in K.reshape(9)])
    code = {"MidPoint": 1, "AdamsBashforth": 2, "Leimkuhler": 3}[scheme]
    rp, dp = pos.copy(), torch.from_numpy(pos.copy()).cuda()
    raux, daux = np.zeros((len(index), 4), np.float32), torch.zeros((len(index), 4), dtype=torch.float32, device="cuda")
    dindex, dorig, dradius = torch.from_numpy(index).cuda(), torch.from_numpy(original).cuda(), torch.from_numpy(radius).cuda()
    P = lambda t: C.c_void_p(t.data_ptr())

    def forces(k):
        f = np.zeros((n, 4), np.float32)
        f[:, :3] = np.random.default_rng(100 + k).normal(0, 1, (n, 3))
        return f

    def gpu(sub, f, step):
        check(lib.uammd_bd_scheme_step(code, sub, P(dp), P(daux), P(dindex), P(dorig), P(f), Kc, M, P(dradius), dt, 0, T, len(index), step, seed, None))

    B = math.sqrt(2 * T * M * dt / 0.5)   # the largest noise amplitude among the members (radius >= 0.5)
    for step in range(1, 4):
        f0, f1 = forces(2 * step), forces(2 * step + 1)
        if scheme == "MidPoint":
            o32.bd_midpoint(0, rp, raux, f0, M, dt, T, step, seed, K=K, radius=radius, index=index)
            gpu(0, torch.from_numpy(f0).cuda(), step)
            o32.bd_midpoint(1, rp, raux, f1, M, dt, T, step, seed, K=K, radius=radius, index=index)
            gpu(1, torch.from_numpy(f1).cuda(), step)
        elif scheme == "AdamsBashforth":
            raux[:] = f0[index]
            daux.copy_(torch.from_numpy(f0[index]))
            o32.bd_adams_bashforth(rp, raux, f1, M, dt, T, step, seed, K=K, radius=radius, index=index)
            gpu(0, torch.from_numpy(f1).cuda(), step)
        else:
            o32.bd_leimkuhler(rp, f0, M, dt, T, step, seed, K=K, radius=radius, index=index, original_index=original)
            gpu(0, torch.from_numpy(f0).cuda(), step)
        torch.cuda.synchronize()
        got = dp.cpu().numpy()
        if T == 0:
            assert np.array_equal(got, rp), np.abs(got - rp).max()
        else:
            assert np.abs(got - rp).max() <= 4e-5 * B * step
            assert np.array_equal(got[:, 3], rp[:, 3])
        others = np.setdiff1d(np.arange(n), index)
        assert np.array_equal(got[others], pos[others])      # nobody outside the group moved


def test_md_steps_lj_nvt(hip, o32):
    """Whole path A: VerletNVT::GronbechJensen + PairForces<LJ> for a few steps vs the same loop on the oracle
    (cell list rebuilt every step as in the reference, CellList.cuh:134-136)."""
    n, L, rc, dt, T = 8000, 21.5, 2.5, 0.005, 1.0
    pos = lattice_positions(n, L, seed=21, jitter=0.05)
    pd = hip.ParticleData(n, seed=1234)
    pd.setPos(pos)
    box = hip.Box(L)
    pot = hip.Potential.LJ()
    pot.setPotParameters(0, 0, pot.InputPairParameters(rc, 1.0, 1.0, False))
    par = hip.VerletNVT.GronbechJensen.Parameters(temperature=T, dt=dt, friction=1.0, initVelocities=True)
    verlet = hip.VerletNVT.GronbechJensen(pd, par)
    verlet.addInteractor(hip.PairForces(pd, box, pot))
    torch.cuda.synchronize()
    # oracle replica (same seeds: same Xorshift stream)
    rp = pos.copy()
    rv = pd.getVel().cpu().numpy().copy()
    vref = o32.verletnvt_initial_velocities(n, math.sqrt(3 * T), 0)  # shape check only
    assert vref.shape == rv.shape
    rf = np.zeros((n, 4), np.float32)
    noise = math.sqrt(2 * dt * 1.0 * T)

    def forces(p):
        cd, oL, oper = o32.celllist_create_grid(L, 1, rc)
        cl = o32.celllist_build(p, oL, oper, cd)
        f, _, _ = o32.lj_transverse_celllist(cl, L, 1, pot.table, 1, n)
        return f

    nsteps = 3
    for s in range(1, nsteps + 1):
        verlet.forwardTime()
        if s == 1:
            rf = forces(rp)
        o32.verletnvt_gj(1, rp, rv, rf, dt, 1.0, noise, s, verlet.seed)
        rf = forces(rp)
        o32.verletnvt_gj(2, rp, rv, rf, dt, 1.0, noise, s, verlet.seed)
    torch.cuda.synchronize()
    gp, gv = pd.getPos().cpu().numpy(), pd.getVel().cpu().numpy()
    assert np.abs(gp - rp).max() <= 1e-5
    assert np.abs(gv - rv).max() <= 1e-4


def test_gj_noise_keyed_by_global_id(hip):
    """uammd_verletnvt_gj_keyed (the domain-decomposed drivers' thermostat): row t draws the stream of key[t], so a particle gets the same
    kicks whichever row (and rank) holds it — the rows of a shuffled copy, keyed by their original numbers, land exactly where the
    unshuffled ones do; and two "ranks" holding different ids with the same seed draw different, uncorrelated kicks (with the plain entry
    point and equal seeds their rows t would draw identical ones: ADVICE round 3)."""
    import ctypes as C
    from uammd_amd._lib import check, load
    lib = load()
    n, dt, T = 4096, 0.005, 1.3
    noise = math.sqrt(2 * dt * 1.0 * T)
    g = torch.Generator(device="cuda").manual_seed(11)
    pos = torch.rand((n, 4), generator=g, device="cuda") * 10
    vel = torch.randn((n, 3), generator=g, device="cuda")
    force = torch.randn((n, 4), generator=g, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())

    def step(pos, vel, force, key, seed=99):
        pp, vv, ff = pos.clone(), vel.clone(), force.clone()
        if key is None:
            check(lib.uammd_verletnvt_gj(1, p(pp), p(vv), p(ff), None, 1.0, None, n, dt, 1.0, 0, noise, 7, seed, None))
        else:
            check(lib.uammd_verletnvt_gj_keyed(1, p(pp), p(vv), p(ff), None, 1.0, None, p(key), n, dt, 1.0, 0, noise, 7, seed, None))
        torch.cuda.synchronize()
        return pp, vv
    ref_p, ref_v = step(pos, vel, force, None)
    ident = torch.arange(n, dtype=torch.int32, device="cuda")
    kp, kv = step(pos, vel, force, ident)
    assert torch.equal(kp, ref_p) and torch.equal(kv, ref_v)                      # key = row number: the plain entry point
    perm = torch.randperm(n, generator=g, device="cuda")
    sp, sv = step(pos[perm].contiguous(), vel[perm].contiguous(), force[perm].contiguous(), perm.to(torch.int32))
    assert torch.equal(sp, ref_p[perm]) and torch.equal(sv, ref_v[perm])          # the kicks travel with the id, not with the row
    # two slabs, same seed, ids 0..n-1 and n..2n-1, zero forces and velocities: the velocity after step 1 IS the kick
    zero_v, zero_f = torch.zeros_like(vel), torch.zeros_like(force)
    _, va = step(pos, zero_v, zero_f, ident)
    _, vb = step(pos, zero_v, zero_f, ident + n)
    assert not torch.equal(va, vb)
    c = torch.corrcoef(torch.stack([va.flatten(), vb.flatten()]))[0, 1].item()
    assert abs(c) < 5.0 / math.sqrt(3 * n)
