"""Path A's slab decomposition on the GPU at world = 1 (the rank is its own neighbour through the periodic z faces):
the persistent-buffer step with the library's slab kernels (uammd_slab_select / pack_rows / unpack_rows / max_displacement,
uammd_halo_pack) against (a) the same step with the generic torch refresh — the two must produce the same rows in the same order,
hence bit-identical trajectories — and (b) the single-domain integrator through the same C ABI (same particles by id, forces to the
tile kernel's tolerance).  The multi-rank exchange itself is covered under gloo in test_distributed_cpu.py."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from util import lattice_positions

pytestmark = pytest.mark.gpu
sim_box = []   # the last simulation _run built (for assertions on which path it took)


def _run(hip, n, L, steps, fused, exchange_every=5, skin=0.3, algo=0, T=1.0, comm=None, overlap=False, step2=False, fuse1=False, aggregate_hash=True, keyed_plain=False):
    from uammd_amd._lib import check, load
    from uammd_amd.parallel import DistributedLJ, SlabDecomposition
    lib = load()
    rc, dt = 2.5, 0.005
    noise = math.sqrt(2 * dt * T)
    d = SlabDecomposition([L, L, L], rc, 0, 1, skin=skin, comm=comm)
    pos = torch.from_numpy(lattice_positions(n, L, seed=11, jitter=0.1)).cuda()
    vel = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    check(lib.uammd_verletnvt_initial_velocities(C.c_void_p(vel.data_ptr()), None, 1.0, 0, n, 77, None))
    ids = torch.arange(n, dtype=torch.int32, device="cuda")
    pot = hip.Potential.LJ()
    pot.setPotParameters(0, 0, pot.InputPairParameters(rc, 1.0, 1.0, False))
    cl = hip.CellList()
    if not aggregate_hash:
        cl.set_option("aggregate_hash", 0)
    cache = {}

    def forces_into(allpos, box_L, periodic, fall):
        key = (tuple(box_L), tuple(periodic))
        if key not in cache:
            box = hip.Box(box_L, periodic)
            cache[key] = (box,) + tuple(hip.CellList.create_update_grid(box, rc))
        box, cd, ubox = cache[key]
        cl.update_grid(allpos, ubox, cd)
        cl.set_option("num_owned", sim.n_owned)
        cl.transverse_lj(pot.device_table(), 1, box, fall, None, None, None, algo)

    def integrate_fn(step, p, v, f, step_num):
        check(lib.uammd_verletnvt_gj(step, C.c_void_p(p.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(f.data_ptr()), None,
                                     1.0, None, p.shape[0], dt, 1.0, 0, noise, step_num, 4242,
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def keyed(step, p, v, f, rows, keys, count, step_num):
        check(lib.uammd_verletnvt_gj_keyed(step, C.c_void_p(p.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(f.data_ptr()), None, 1.0,
                                           None if rows is None else C.c_void_p(rows.data_ptr()), C.c_void_p(keys.data_ptr()), count, dt, 1.0, 0,
                                           noise, step_num, 4242, C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def forces_step2_into(allpos, box_L, periodic, fall, v):
        key = (tuple(box_L), tuple(periodic))
        if key not in cache:
            box = hip.Box(box_L, periodic)
            cache[key] = (box,) + tuple(hip.CellList.create_update_grid(box, rc))
        box, cd, ubox = cache[key]
        cl.update_grid(allpos, ubox, cd)
        cl.set_option("num_owned", sim.n_owned)
        cl.transverse_lj_gj2(pot.device_table(), 1, box, fall, v, dt, None, 1.0, False, algo)

    def pack_step1(p, v, f, keys, iu, nu, idn, nd, dzu, dzd, ou, od, step_num):
        P = lambda t: C.c_void_p(t.data_ptr())
        check(lib.uammd_halo_pack_gj1(P(p), P(v), P(f), None, 1.0, P(keys), P(iu), nu, P(idn), nd, dzu, dzd, P(ou), P(od), dt, 1.0, 0, noise,
                                      step_num, 4242, C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def forces_step12_into(allpos, box_L, periodic, fall, v, keys, skip, step_num):
        key = (tuple(box_L), tuple(periodic))
        if key not in cache:
            box = hip.Box(box_L, periodic)
            cache[key] = (box,) + tuple(hip.CellList.create_update_grid(box, rc))
        box, cd, ubox = cache[key]
        cl.update_grid_gj1(allpos, ubox, cd, v, fall, keys, skip, sim.n_owned, dt, 1.0, 0, noise, step_num, 4242)
        cl.set_option("num_owned", sim.n_owned)
        cl.transverse_lj_gj2(pot.device_table(), 1, box, fall, v, dt, None, 1.0, False, algo)
        sim.fused1_steps = getattr(sim, "fused1_steps", 0) + 1

    if (fuse1 or keyed_plain) and comm is None:   # (the noise keyed by the global id in the unfused steps too)
        integrate_fn = lambda step, p, v, f, step_num: keyed(step, p, v, f, None, sim.current_ids, p.shape[0], step_num)
    if comm is not None:   # the bench's configuration: thermostat keyed by the global id, every message through uammd_comm_*
        integrate_fn = lambda step, p, v, f, step_num: keyed(step, p, v, f, None, sim.current_ids, p.shape[0], step_num)
    sim = DistributedLJ(d, None, integrate_fn, exchange_every=exchange_every, forces_into=forces_into,
                        forces_step2_into=forces_step2_into if step2 else None,
                        integrate_rows_fn=(lambda step, p, v, f, rows, keys, step_num: keyed(step, p, v, f, rows, keys, keys.shape[0] if rows is None else rows.shape[0], step_num))
                        if overlap else None, step1_fused=(pack_step1, forces_step12_into) if fuse1 else None)
    sim_box.append(sim)
    if not fused:
        sim._refresh_fused = None   # _refresh_persistent falls through to the generic path
        orig = sim._refresh_persistent

        def generic(nn):
            bp, bv, bi, bf = sim._bufs
            sim._track_drift(bp[:nn])
            nn = sim.d.migrate_inplace([bp, bv, bi], nn)
            bf[:nn].zero_()
            g = sim.d.halo_refresh_into(bp, nn)
            sim._nall = nn + g
            if sim.d.skin > 0:
                sim._ref = bp[:nn].clone()
            return nn
        sim._refresh_persistent = generic
    force = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    for _ in range(steps):
        pos, vel, force, ids = sim.forward_time(pos, vel, force, ids)
    torch.cuda.synchronize()
    sim.check_skin()
    return pos.cpu().numpy().copy(), vel.cpu().numpy().copy(), ids.cpu().numpy().copy(), float(sim.max_drift) if sim.max_drift is not None else 0.0


def test_overlapped_exchange_and_fused_half_step_are_bit_identical(hip):
    """The bench's slab step — halo exchange through uammd_comm_* (RCCL, the ring closing on the rank itself) on a side stream while the
    main stream integrates the particles nobody else needs, second half step in the traversal's store — against the plain sequence
    (half step, exchange, build, traversal, half step) on one stream: same rows, same bits, with noise, across refreshes."""
    from uammd_amd.comm import AbiComm
    n, L = 30000, 33.5
    comm = AbiComm(0, 1, AbiComm.unique_id())
    try:
        a = _run(hip, n, L, 23, fused=True, comm=comm, overlap=False, step2=False)
        b = _run(hip, n, L, 23, fused=True, comm=comm, overlap=True, step2=True)
        assert sim_box[-1]._side is not None, "the overlapped path did not run"
    finally:
        comm.close()
    assert np.array_equal(a[2], b[2])
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    assert sorted(a[2].tolist()) == list(range(n))


@pytest.mark.parametrize("through_comm,aggregate_hash,algo", [(True, True, 0), (False, True, 0), (False, False, 0), (False, True, 9)])
def test_first_half_step_in_the_pack_and_the_list_build_is_bit_identical(hip, through_comm, aggregate_hash, algo):
    """The slab step with GronbechJensen's first half step folded into the halo pack (listed rows: uammd_halo_pack_gj1) and into the list
    build's hash kernel (everybody else: uammd_celllist_update_gj1 with the listed rows masked), second half step in the traversal's
    store — against the plain sequence of five calls: same rows, same bits, with noise, across membership refreshes.  Also where the
    build does not carry the half step in its hash kernel (aggregate_hash off: one masked launch first) and through the exact kernel."""
    from uammd_amd.comm import AbiComm
    n, L = 30000, 33.5
    comm = AbiComm(0, 1, AbiComm.unique_id()) if through_comm else None
    try:
        a = _run(hip, n, L, 23, fused=True, comm=comm, step2=False, fuse1=False, algo=algo)
        if comm is None:   # the plain run's noise keyed by the global id as well
            a = _run_keyed_plain(hip, n, L, 23, algo)
        b = _run(hip, n, L, 23, fused=True, comm=comm, step2=True, fuse1=True, aggregate_hash=aggregate_hash, algo=algo)
        assert getattr(sim_box[-1], "fused1_steps", 0) >= 15, "the fused first half step did not run"
    finally:
        if comm is not None:
            comm.close()
    assert np.array_equal(a[2], b[2])
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    assert sorted(a[2].tolist()) == list(range(n))


def _run_keyed_plain(hip, n, L, steps, algo):
    """_run's plain sequence with the thermostat keyed by the global id (what fuse1's unfused refresh steps use)."""
    return _run(hip, n, L, steps, fused=True, algo=algo, keyed_plain=True)


def test_fused_refresh_equals_generic_refresh(hip):
    n, L = 30000, 33.5          # rho = 0.8: particles cross the periodic z faces within a few steps
    a = _run(hip, n, L, 40, fused=True, algo=9)
    b = _run(hip, n, L, 40, fused=False, algo=9)
    assert np.array_equal(a[2], b[2]), "owned rows in another order"
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    assert sorted(a[2].tolist()) == list(range(n))                       # nobody lost or duplicated
    assert (np.abs(a[0][:, 2]) <= L / 2 + 0.3 + 1e-4).all()                # owners sit within the skin of their slab
    assert abs(a[3] - b[3]) <= 1e-6 and 0.0 < a[3] <= 0.3                   # the skin check saw the same displacement


def test_slab_world1_matches_single_domain(hip):
    """Same box integrated by PairForces + VerletNVT::GronbechJensen without decomposition (T = 0: the noise stream is keyed by the row
    index, which the migration permutes): particle by particle (ids), positions after 10 steps agree to the accumulated rounding of two
    summation orders (the trajectories are chaotic: short run, loose bar)."""
    n, L = 30000, 33.5
    pos0 = lattice_positions(n, L, seed=11, jitter=0.1)
    p, v, ids, _ = _run(hip, n, L, 10, fused=True, T=0.0)
    from uammd_amd._lib import check, load
    lib = load()
    pd = hip.ParticleData(n, seed=1)
    pd.setPos(pos0)
    vel = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    check(lib.uammd_verletnvt_initial_velocities(C.c_void_p(vel.data_ptr()), None, 1.0, 0, n, 77, None))
    pd.getVel("write").copy_(vel)
    box = hip.Box(L)
    pot = hip.Potential.LJ()
    pot.setPotParameters(0, 0, pot.InputPairParameters(2.5, 1.0, 1.0, False))
    par = hip.VerletNVT.GronbechJensen.Parameters(temperature=0.0, dt=0.005, friction=1.0, initVelocities=False)
    integ = hip.VerletNVT.GronbechJensen(pd, par)
    integ.addInteractor(hip.PairForces(pd, box, pot))
    for _ in range(10):
        integ.forwardTime()
    ref = pd.getPos("read").cpu().numpy()
    got = np.empty_like(ref)
    got[ids] = p
    dz = got[:, :3] - ref[:, :3]
    dz -= np.round(dz / L) * L          # the slab keeps z folded into its frame, the single domain does not fold
    assert np.abs(dz).max() <= 2e-4


def test_slab_step_through_comm_vs_oracle(hip, o32):
    """The bench's slab step (every message through uammd_comm_* = RCCL, membership refresh by the library, first half step in the halo
    pack and the list build, second in the traversal's store) against the ORACLE's loop — cell list, LJ traversal, GronbechJensen with the
    same thermostat stream (keyed by the particle's global id = its row in the oracle's arrays) — particle by particle after 12 thermal
    steps with refreshes and migration across the periodic z faces.  The chain to the oracle does not run through another HIP path."""
    from uammd_amd._lib import check, load
    from uammd_amd.comm import AbiComm
    lib = load()
    n, L, rc, dt, T, steps = 30000, 33.5, 2.5, 0.005, 1.0, 12
    comm = AbiComm(0, 1, AbiComm.unique_id())
    try:
        gp, gv, gids, _ = _run(hip, n, L, steps, fused=True, comm=comm, step2=True, fuse1=True, T=T, exchange_every=5)
        assert getattr(sim_box[-1], "fused1_steps", 0) >= 8
    finally:
        comm.close()
    rp = lattice_positions(n, L, seed=11, jitter=0.1).copy()
    vel = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    check(lib.uammd_verletnvt_initial_velocities(C.c_void_p(vel.data_ptr()), None, 1.0, 0, n, 77, None))
    rv = vel.cpu().numpy().copy()
    pot = hip.Potential.LJ()
    pot.setPotParameters(0, 0, pot.InputPairParameters(rc, 1.0, 1.0, False))
    noise = math.sqrt(2 * dt * T)

    def forces(p):
        cd, oL, oper = o32.celllist_create_grid(L, 1, rc)
        cl = o32.celllist_build(p, oL, oper, cd)
        f, _, _ = o32.lj_transverse_celllist(cl, L, 1, pot.table, 1, n)
        return f
    rf = forces(rp)
    for s in range(1, steps + 1):
        o32.verletnvt_gj(1, rp, rv, rf, dt, 1.0, noise, s, 4242)
        rf = forces(rp)
        o32.verletnvt_gj(2, rp, rv, rf, dt, 1.0, noise, s, 4242)
    order = np.argsort(gids)
    assert np.array_equal(gids[order], np.arange(n))
    dx = gp[order, :3] - rp[:, :3]
    dx -= np.round(dx / L) * L
    dv = np.abs(gv[order] - rv).max(axis=1)
    print(f"[slab vs oracle, {steps} steps] max|dx| {np.abs(dx).max():.2e}; |dv| max {dv.max():.2e}, 99.9th percentile {np.quantile(dv, 0.999):.2e}")
    assert np.abs(dx).max() <= 2e-5
    assert np.quantile(dv, 0.999) <= 1e-4 and dv.max() <= 2e-3
