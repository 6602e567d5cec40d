"""CPU tests of the multi-GPU path (world_size 2 and 3, gloo): the z-slab decomposition, halo exchange and migration
of uammd_amd/parallel.py, with the force evaluation done by the oracle.  Checks against the single-domain oracle:
forces of every particle (same pairs, different summation order -> 1e-5 of max|F|) and a few deterministic (T = 0)
integration steps with migration across slab faces, matched by global particle id."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import lattice_positions

L = (15.0, 15.4, 30.0)
RC = 2.5
N = 2600
DT = 0.004
NSTEPS = 6


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _config():
    pos = lattice_positions(N, L, seed=17, jitter=0.1)
    rng = np.random.default_rng(5)
    vel = rng.normal(0, 2.5, (N, 3)).astype(np.float32)
    # push some particles right next to the slab faces (z = 0 for 2 ranks, z = +-5 for 3 ranks, z = +-15 periodic
    # face) with a velocity that carries them across within the first steps -> migration is exercised
    for face, direction in ((0.0, +1), (-5.0, +1), (5.0, -1), (15.0, +1)):
        below = np.nonzero((pos[:, 2] < face) & (pos[:, 2] > face - 1.5))[0] if direction > 0 else \
            np.nonzero((pos[:, 2] > face) & (pos[:, 2] < face + 1.5))[0]
        sel = below[::7][:25]
        pos[sel, 2] = face - direction * 0.015
        vel[sel, 2] = direction * 3.0
    return pos, vel


def _oracle_forces(o, pos_np, box_L, periodic):
    cd, oL, oper = o.celllist_create_grid(box_L, periodic, RC)
    cl = o.celllist_build(pos_np, oL, oper, cd)
    assert cl["error"] == 0
    f, _, _ = o.lj_transverse_celllist(cl, box_L, periodic, o.lj_params(RC, 1.0, 1.0), 1, len(pos_np))
    return f


def _single_domain():
    import oracle
    o = oracle.get("f32")
    pos, vel = _config()
    f0 = _oracle_forces(o, pos, L, 1)
    p, v, f = pos.copy(), vel.copy(), f0.copy()
    for s in range(1, NSTEPS + 1):
        o.verletnvt_gj(1, p, v, f, DT, 0.0, 0.0, s, 1)
        f = _oracle_forces(o, p, L, 1)
        o.verletnvt_gj(2, p, v, f, DT, 0.0, 0.0, s, 1)
    return f0, p, v


def _worker(rank, world, port, out_dir, skin=0.0, every=1, persistent=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from uammd_amd.parallel import DistributedLJ, SlabDecomposition
        o = oracle.get("f32")
        pos, vel = _config()
        d = SlabDecomposition(L, RC, rank, world, skin=skin)
        gpos = torch.from_numpy(pos)
        lpos, ids = d.scatter_initial(gpos)
        lvel = torch.from_numpy(vel)[ids.long()].clone()

        def forces_fn(allpos, box_L, periodic):
            return torch.from_numpy(_oracle_forces(o, allpos.numpy(), box_L, [int(x) for x in periodic]))

        def integrate_fn(step, p, v, f, step_num):
            pn, vn, fn = p.numpy(), v.numpy(), f.numpy()
            o.verletnvt_gj(step, pn, vn, fn, DT, 0.0, 0.0, step_num, 1)

        def forces_into(allpos, box_L, periodic, fall):   # persistent-buffer mode: accumulate into the step's own force buffer
            fall += forces_fn(allpos, box_L, periodic)

        def forces_step2_into(allpos, box_L, periodic, fall, v):   # what uammd_lj_transverse_celllist_gj2 does in one launch on the GPU
            n_own = v.shape[0]
            fall[:n_own] += forces_fn(allpos, box_L, periodic)[:n_own]
            integrate_fn(2, allpos[:n_own], v, fall[:n_own], 0)

        sim = DistributedLJ(d, forces_fn, integrate_fn, exchange_every=every, forces_into=forces_into if persistent else None,
                            forces_step2_into=forces_step2_into if persistent == "step2" else None)
        f0 = sim.compute_forces(lpos).clone()
        ids0 = ids.clone()
        nmig = 0
        p, v, f = lpos.clone(), lvel, None
        for _ in range(NSTEPS):
            before = set(ids.tolist())
            p, v, f, ids = sim.forward_time(p, v, f if f is not None else torch.zeros_like(p), ids)
            nmig += len(set(ids.tolist()) - before)
        sim.check_skin()
        # back to global coordinates
        gp = p.clone()
        gp[:, 2] += d.zc
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ids0=ids0.numpy(), f0=f0.numpy(), ids=ids.numpy(), pos=gp.numpy(),
                 vel=v.numpy(), nmig=nmig)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,skin,every,persistent", [(2, 0.0, 1, False), (3, 0.0, 1, False), (2, 0.15, 3, False), (3, 0.15, 2, False),
                                                         (2, 0.15, 3, True), (3, 0.15, 2, True), (2, 0.0, 1, True), (2, 0.15, 3, "step2"),
                                                         (3, 0.0, 1, "step2")])
def test_slab_decomposition_matches_single_domain(world, skin, every, persistent, tmp_path):
    """skin > 0: ownership and halo membership refreshed every `every` steps, cached lists in between (no size messages)."""
    f0_ref, p_ref, v_ref = _single_domain()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), skin, every, persistent), nprocs=world, join=True)
    seen0, seen, nmig = [], [], 0
    fmax = np.abs(f0_ref).max()
    Lz = L[2]
    for r in range(world):
        g = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        seen0 += g["ids0"].tolist()
        seen += g["ids"].tolist()
        nmig += int(g["nmig"])
        assert np.abs(g["f0"][:, :3] - f0_ref[g["ids0"], :3]).max() <= 1e-5 * fmax
        dz = g["pos"][:, 2] - p_ref[g["ids"], 2]
        dz -= np.round(dz / Lz) * Lz                      # the slab frame and the single box may hold different images
        assert np.abs(g["pos"][:, :2] - p_ref[g["ids"], :2]).max() <= 2e-4 and np.abs(dz).max() <= 2e-4
        assert np.abs(g["vel"] - v_ref[g["ids"]]).max() <= 2e-2
    assert sorted(seen0) == list(range(N)) and sorted(seen) == list(range(N))   # a partition, before and after
    assert nmig > (3 if every == 1 else 0)                                     # migration really happened


def test_helpers_single_rank():
    from uammd_amd.parallel import SlabDecomposition
    d = SlabDecomposition(L, RC, rank=1, world=3)
    z = torch.tensor([-14.9, -5.1, 0.0, 4.9, 5.1, 14.9, 15.5])
    assert d.owner_of(z).tolist() == [0, 0, 1, 1, 2, 2, 0]
    box, per = d.local_box()
    assert box[2] == pytest.approx(10.0 + 5.05) and per == [True, True, False]
    with pytest.raises(ValueError):
        SlabDecomposition((10.0, 10.0, 10.0), 2.5, rank=0, world=5)


def test_single_rank_loops_back_on_itself():
    """world = 1: the rank is its own neighbour through the periodic z faces (ghosts are its own images, leavers re-enter on
    the other side), so the slab code path gives the single-domain result without any process group."""
    import oracle
    from uammd_amd.parallel import DistributedLJ, SlabDecomposition
    o = oracle.get("f32")
    pos, vel = _config()
    f0_ref, p_ref, v_ref = _single_domain()
    d = SlabDecomposition(L, RC, rank=0, world=1)
    lpos, ids = d.scatter_initial(torch.from_numpy(pos))
    lvel = torch.from_numpy(vel)[ids.long()].clone()

    def forces_fn(allpos, box_L, periodic):
        return torch.from_numpy(_oracle_forces(o, allpos.numpy(), box_L, [int(x) for x in periodic]))

    def integrate_fn(step, p, v, f, step_num):
        o.verletnvt_gj(step, p.numpy(), v.numpy(), f.numpy(), DT, 0.0, 0.0, step_num, 1)
    sim = DistributedLJ(d, forces_fn, integrate_fn)
    f0 = sim.compute_forces(lpos)
    assert np.abs(f0.numpy()[:, :3] - f0_ref[ids.numpy(), :3]).max() <= 1e-5 * np.abs(f0_ref).max()
    p, v, f = lpos.clone(), lvel, torch.zeros_like(lpos)
    for _ in range(NSTEPS):
        p, v, f, ids = sim.forward_time(p, v, f, ids)
    assert sorted(ids.tolist()) == list(range(N))
    dz = p.numpy()[:, 2] + d.zc - p_ref[ids.numpy(), 2]
    dz -= np.round(dz / L[2]) * L[2]
    assert np.abs(p.numpy()[:, :2] - p_ref[ids.numpy(), :2]).max() <= 2e-4 and np.abs(dz).max() <= 2e-4
