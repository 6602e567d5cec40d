"""CPU tests of the multi-GPU path B (uammd_amd/parallel_fcm.py) under gloo, world_size 2 and 3: halo add, the two
all-to-all transposes, halo fill, lock-step noise seeds and particle migration.  The per-rank compute stages are a
numpy backend built on the oracle (tests may use the oracle; the product backend is HipSlabBackend and is checked on
the GPU by tests/test_gpu_fcm_slab.py).  Reference: the single-domain FCMOracle on the same inputs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

CELLS = [12, 12, 24]
LBOX = [12.0, 12.0, 24.0]
NPART = 300
VISC = 0.9
SEED = 4321
TOL = 1e-3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _config():
    global CELLS, LBOX
    rng = np.random.default_rng(11)
    pos = np.zeros((NPART, 4), np.float32)
    pos[:, :3] = rng.uniform(-0.5, 0.5, (NPART, 3)) * np.asarray(LBOX, np.float32)
    pos[:6, 2] = [-12.0, -0.001, 0.0, 0.001, 3.999, 11.999]   # on / next to slab faces for 2 and 3 ranks
    force = np.zeros((NPART, 4), np.float32)
    force[:, :3] = rng.normal(0, 1, (NPART, 3))
    return pos, force


class NumpySlabBackend:
    """Same interface as uammd_amd.parallel_fcm.HipSlabBackend, computed with the oracle + numpy FFTs."""

    def __init__(self, o, geom, rank, kernel, viscosity, seed):
        self.o, self.g, self.rank, self.kernel, self.viscosity, self.seed = o, geom, rank, kernel, viscosity, seed
        hz = geom.L[2] / geom.cells[2]
        self.Lw = [geom.L[0], geom.L[1], hz * geom.nzw]
        self.cw = [geom.cells[0], geom.cells[1], geom.nzw]
        self._spec = torch.zeros((geom.nzl, 3, geom.cells[1], geom.nkx, 2), dtype=torch.float32)

    def spread(self, pos_local, force):
        g = self.g
        if force is None:
            return torch.zeros((g.nzw, 3, g.cells[1], g.nxpad), dtype=torch.float32)
        gr = self.o.ibm_spread(pos_local.numpy(), force.numpy()[:, :3].copy(), self.Lw, 1, self.cw, self.kernel, nx_stride=g.nxpad)
        return torch.from_numpy(np.ascontiguousarray(gr.transpose(0, 3, 1, 2)))      # [z][y][x][c] -> [z][c][y][x]

    def forward_xy(self, grid):
        g = self.g
        owned = grid[g.halo:g.halo + g.nzl, :, :, :g.cells[0]].numpy()
        spec = np.fft.rfft2(owned.astype(np.float64), axes=(2, 3)).astype(np.complex64)
        return torch.from_numpy(np.ascontiguousarray(spec.view(np.float32).reshape(g.nzl, 3, g.cells[1], g.nkx, 2)))

    def spectrum_view(self, grid):
        return self._spec

    def inverse_xy(self, grid):
        g = self.g
        nx, ny = g.cells[0], g.cells[1]
        spec = self._spec.numpy().reshape(g.nzl, 3, ny, g.nkx * 2).view(np.complex64)
        real = np.fft.irfft2(spec.astype(np.complex128), s=(ny, nx), axes=(2, 3)) * (nx * ny)   # unnormalised, as rocFFT
        grid[g.halo:g.halo + g.nzl, :, :, :nx] = torch.from_numpy(real.astype(np.float32))

    def new_zbuffer(self):
        g = self.g
        return torch.zeros((g.cells[2], 3, g.nyl, g.nkx, 2), dtype=torch.float32)

    def fft_z(self, buf, inverse):
        g = self.g
        c = buf.numpy().reshape(g.cells[2], 3, g.nyl, g.nkx * 2).view(np.complex64).astype(np.complex128)
        c = np.fft.ifft(c, axis=0) * g.cells[2] if inverse else np.fft.fft(c, axis=0)
        buf.copy_(torch.from_numpy(np.ascontiguousarray(c.astype(np.complex64)).view(np.float32).reshape(buf.shape)))

    def kspace(self, buf, have_force, temperature, prefactor, seed2):
        g, o = self.g, self.o
        nx, ny, nz = g.cells
        y0 = self.rank * g.nyl
        full = np.zeros((nz, ny, g.nkx, 3), np.complex64)       # the reference's complex3[nz][ny][nkx]
        if have_force:
            mine = buf.numpy().reshape(nz, 3, g.nyl, g.nkx * 2).view(np.complex64)     # [z][c][yl][kx]
            full[:, y0:y0 + g.nyl] = mine.transpose(0, 2, 3, 1)
            o.fcm_force_fourier_to_vel(full, self.viscosity, g.L, g.cells)
        if temperature > 0:
            npf = o.fcm_noise_prefactor(prefactor, temperature, g.L, g.cells)
            o.fcm_fourier_brownian_noise(full, g.L, g.cells, npf, self.viscosity, self.seed, seed2)
        out = np.ascontiguousarray(full[:, y0:y0 + g.nyl].transpose(0, 3, 1, 2))       # -> [z][c][yl][kx]
        buf.copy_(torch.from_numpy(out.view(np.float32).reshape(buf.shape)))

    def gather(self, pos_local, grid):
        g = self.g
        gv = np.ascontiguousarray(grid.numpy().transpose(0, 2, 3, 1))                  # -> [z][y][x][c]
        return torch.from_numpy(self.o.ibm_gather(pos_local.numpy(), gv, self.Lw, 1, self.cw, self.kernel, nx_stride=g.nxpad))


def _reference(temperature, prefactor, ncalls):
    import oracle
    from oracle.fcm import FCMOracle
    o = oracle.get("f32")
    pos, force = _config()
    f = FCMOracle(o, LBOX, CELLS, tolerance=TOL, viscosity=VISC, seed=SEED)
    return [f.displacements(pos, force, temperature, prefactor) for _ in range(ncalls)], f


def _worker(rank, world, port, out_dir, temperature, prefactor, ncalls, cells=None, every=1):
    global CELLS, LBOX
    if cells is not None:
        CELLS, LBOX = list(cells), [float(c) for c in cells]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from uammd_amd.parallel_fcm import DistributedFCM, DistributedFCMIntegrator, SlabGeometry, make_decomposition
        o = oracle.get("f32")
        pos, force = _config()
        kinfo = o.fcm_gaussian(1.0, TOL)
        geom = SlabGeometry(CELLS, LBOX, world, kinfo["support"])
        back = NumpySlabBackend(o, geom, rank, kinfo["kernel"], VISC, SEED)
        d = make_decomposition(geom, rank)
        lpos, ids = d.scatter_initial(torch.from_numpy(pos))
        lforce = torch.from_numpy(force)[ids.long()].clone()
        fcm = DistributedFCM(geom, [back], [rank])
        vs = [fcm.displacements([lpos], [lforce], temperature, prefactor)[0].numpy().copy() for _ in range(ncalls)]
        # a few deterministic Euler-Maruyama steps with a large dt: particles cross slab faces and migrate with their forces
        integ = DistributedFCMIntegrator(DistributedFCM(geom, [back], [rank]), d, 0.0, 2.0, lambda p, i, f: f, migrate_every=every)
        p, i, f = lpos.clone(), ids.clone(), lforce.clone()
        nmig = 0
        for _ in range(4 if every > 1 else 3):
            before = set(i.tolist())
            p, i, f = integ.forward_time(p, i, f)
            nmig += len(set(i.tolist()) - before)
        integ.check_drift()
        gp = p.clone()
        gp[:, 2] += d.zc
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ids=ids.numpy(), v=np.stack(vs), ids_end=i.numpy(), pos_end=gp.numpy(),
                 nmig=nmig, seed2=fcm.seed2)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,temperature,cells,every", [(2, 0.0, None, 1), (2, 0.8, None, 1), (3, 0.8, None, 1),
                                                            (2, 0.0, (16, 16, 32), 2)])
def test_slab_fcm_matches_single_domain(world, temperature, cells, every, tmp_path):
    """The last case has a tile-sized halo (3 spare planes): particles are re-assigned to their slabs every 2nd step only."""
    global CELLS, LBOX
    saved = (CELLS, LBOX)
    if cells is not None:
        CELLS, LBOX = list(cells), [float(c) for c in cells]
    try:
        _run_case(world, temperature, cells, every, tmp_path)
    finally:
        CELLS, LBOX = saved


def _run_case(world, temperature, cells, every, tmp_path):
    prefactor, ncalls = 2.0, 2
    v_ref, f = _reference(temperature, prefactor, ncalls)
    # single-domain Euler-Maruyama reference for the migration part (T = 0, dt = 2, 3 steps)
    import oracle
    from oracle.fcm import FCMOracle
    o = oracle.get("f32")
    pos, force = _config()
    fo = FCMOracle(o, LBOX, CELLS, tolerance=TOL, viscosity=VISC, seed=SEED)
    p_ref = pos.copy()
    for _ in range(4 if every > 1 else 3):
        v = fo.displacements(p_ref, force, 0.0, 0.0)
        p_ref[:, :3] += v * np.float32(2.0)
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), temperature, prefactor, ncalls, cells, every), nprocs=world, join=True)
    seen, seen_end, nmig = [], [], 0
    for r in range(world):
        g = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        seen += g["ids"].tolist()
        seen_end += g["ids_end"].tolist()
        nmig += int(g["nmig"])
        assert int(g["seed2"]) == (ncalls if temperature > 0 else 0)
        for c in range(ncalls):
            ref = v_ref[c][g["ids"]]
            assert np.linalg.norm(g["v"][c] - ref) <= 1e-5 * np.linalg.norm(v_ref[c]), (r, c)
        d = g["pos_end"][:, :3] - p_ref[g["ids_end"], :3]
        d[:, 2] -= np.round(d[:, 2] / LBOX[2]) * LBOX[2]
        assert np.abs(d).max() <= 1e-4
    assert sorted(seen) == list(range(NPART)) and sorted(seen_end) == list(range(NPART))
    assert nmig > 0


def test_geometry():
    from uammd_amd.parallel_fcm import SlabGeometry
    g = SlabGeometry([128, 128, 256], [128.0, 128.0, 256.0], 8, 6)
    assert (g.nzl, g.nyl, g.he, g.halo, g.nzw, g.drift_planes) == (32, 16, 8, 8, 48, 3)   # whole 8-node tiles in the window
    g = SlabGeometry([36, 36, 36], [36.0] * 3, 2, 6)
    assert (g.halo, g.nzw, g.drift_planes) == (5, 28, 0)                  # not tileable: halo = stencil reach
    with pytest.raises(ValueError):
        SlabGeometry([32, 32, 30], [1.0] * 3, 4, 6)
    with pytest.raises(ValueError):
        SlabGeometry([32, 32, 32], [1.0] * 3, 8, 6)                       # 4-plane slabs cannot hold a 5-plane halo
