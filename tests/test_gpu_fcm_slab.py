"""GPU parity of the z-slab decomposed FCM (uammd_fcm_slab_* + uammd_amd/parallel_fcm.py) against the single-GPU solver.

All P ranks of the decomposition run in ONE process on cuda:0 (the exchanges become tensor copies), so the test checks
exactly the arithmetic a P-GPU run performs: local windows, 2-D + 1-D FFTs with a transpose in between, the k-space
kernel in the y-pencil layout (noise ids are GLOBAL node indices, so the random field must be identical to the
single-GPU one).  Tolerance: 1e-5 relative L2 — the FFT factorisation and the window frame
(z relative to the slab centre) reorder float operations.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(hip, cells, L, n, seed, tol=1e-3):
    rng = np.random.default_rng(seed)
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(-0.5, 0.5, (n, 3)) * np.asarray(L, np.float32)
    force = np.zeros((n, 4), np.float32)
    force[:, :3] = rng.normal(0, 1, (n, 3))
    h = min(l / c for l, c in zip(L, cells))
    kernel, a = hip.Kernels.Gaussian(h, tol)
    return torch.from_numpy(pos).cuda(), torch.from_numpy(force).cuda(), kernel, a


def _slab_solver(hip, cells, L, P, kernel, viscosity, seed, atomic=False):
    from uammd_amd.parallel_fcm import DistributedFCM, HipSlabBackend, SlabGeometry, make_decomposition
    geom = SlabGeometry(cells, L, P, kernel.support[2])
    backs = [HipSlabBackend(geom, r, kernel, viscosity, seed) for r in range(P)]
    if atomic:
        for b in backs:
            b.set_option("atomic_spread", 1)
    decs = [make_decomposition(geom, r) for r in range(P)]
    return geom, DistributedFCM(geom, backs, list(range(P))), decs


def _scatter(decs, pos, force):
    pl, fl, idx = [], [], []
    for d in decs:
        own = d.owner_of(pos[:, 2]) == d.rank
        ids = torch.nonzero(own).flatten()
        pl.append(d.to_local(pos[ids]).contiguous())
        fl.append(force[ids].contiguous() if force is not None else None)
        idx.append(ids)
    return pl, fl, idx


def _assemble(vs, idx, n):
    out = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    for v, i in zip(vs, idx):
        out[i] = v
    return out


@pytest.mark.parametrize("cells,L,P,atomic", [([64, 64, 64], [64.0] * 3, 1, False), ([64, 64, 64], [64.0] * 3, 2, False),
                                               ([64, 64, 64], [64.0] * 3, 4, False), ([32, 64, 128], [32.0, 64.0, 128.0], 4, False),
                                               ([36, 36, 36], [40.0] * 3, 2, False), ([64, 64, 64], [64.0] * 3, 2, True),
                                               ([30, 36, 48], [30.0, 36.0, 48.0], 3, False)])
def test_slab_fcm_deterministic_matches_single_gpu(hip, cells, L, P, atomic):
    n, visc, seed = 4000, 1.3, 99
    pos, force, kernel, a = _setup(hip, cells, L, n, seed=5)
    ref = hip.BDHI.FCM_impl(hip.Box(L), cells, kernel, visc, seed, a)
    v_ref = ref.computeHydrodynamicDisplacements(pos, force, n, 0.0, 0.0)
    geom, fcm, decs = _slab_solver(hip, cells, L, P, kernel, visc, seed, atomic)
    pl, fl, idx = _scatter(decs, pos, force)
    assert sum(p.shape[0] for p in pl) == n
    v = _assemble(fcm.displacements(pl, fl, 0.0, 0.0), idx, n)
    torch.cuda.synchronize()
    err = (v - v_ref).norm() / v_ref.norm()
    assert err.item() <= 1e-5, err.item()


@pytest.mark.parametrize("P", [2, 4])
def test_slab_fcm_noise_field_is_the_single_gpu_one(hip, P):
    """T > 0: same (seed, seed2) -> the same Fourier noise on every node, wherever the node lives."""
    cells, L, n, visc, seed, T, pref = [64, 64, 64], [64.0] * 3, 3000, 1.0, 1234, 0.7, 3.0
    pos, force, kernel, a = _setup(hip, cells, L, n, seed=8)
    ref = hip.BDHI.FCM_impl(hip.Box(L), cells, kernel, visc, seed, a)
    geom, fcm, decs = _slab_solver(hip, cells, L, P, kernel, visc, seed)
    pl, fl, idx = _scatter(decs, pos, force)
    for call in range(2):  # seed2 advances in lock step
        v_ref = ref.computeHydrodynamicDisplacements(pos, force, n, T, pref)
        v = _assemble(fcm.displacements(pl, fl, T, pref), idx, n)
        err = (v - v_ref).norm() / v_ref.norm()
        assert err.item() <= 1e-5, (call, err.item())
    assert fcm.seed2 == ref.seed2() == 2
    # noise only (no forces): the forward transforms are skipped
    v_ref = ref.computeHydrodynamicDisplacements(pos, None, n, T, pref)
    v = _assemble(fcm.displacements(pl, [None] * P, T, pref), idx, n)
    err = (v - v_ref).norm() / v_ref.norm()
    assert err.item() <= 1e-5, err.item()


def test_slab_fcm_particles_on_slab_faces(hip):
    """Particles exactly on / next to the slab faces and the box faces: their stencils cross into the neighbour's planes."""
    cells, L, P, visc, seed = [64, 64, 64], [64.0] * 3, 4, 1.0, 7
    h = 1.0
    zs = []
    for face in (-32.0, -16.0, 0.0, 16.0, 31.999):
        zs += [face, face + 1e-4, face - 1e-4, face + 0.5 * h, face - 0.5 * h, face + 2.9 * h, face - 2.9 * h]
    zs = np.clip(np.asarray(zs, np.float32), -32.0, np.nextafter(np.float32(32.0), np.float32(0)))
    n = len(zs)
    rng = np.random.default_rng(3)
    pos = np.zeros((n, 4), np.float32)
    pos[:, :2] = rng.uniform(-32, 32, (n, 2))
    pos[:, 2] = zs
    force = np.zeros((n, 4), np.float32)
    force[:, :3] = rng.normal(0, 1, (n, 3))
    kernel, a = hip.Kernels.Gaussian(h, 1e-3)
    pos, force = torch.from_numpy(pos).cuda(), torch.from_numpy(force).cuda()
    ref = hip.BDHI.FCM_impl(hip.Box(L), cells, kernel, visc, seed, a)
    v_ref = ref.computeHydrodynamicDisplacements(pos, force, n, 0.0, 0.0)
    geom, fcm, decs = _slab_solver(hip, cells, L, P, kernel, visc, seed)
    pl, fl, idx = _scatter(decs, pos, force)
    v = _assemble(fcm.displacements(pl, fl, 0.0, 0.0), idx, n)
    err = (v - v_ref).abs().max() / v_ref.abs().max()
    assert err.item() <= 1e-5, err.item()


def test_slab_create_rejects_bad_arguments(hip):
    import ctypes as C
    from uammd_amd._lib import FCMParameters, load
    lib = load()
    kernel, a = hip.Kernels.Gaussian(1.0, 1e-3)
    p = FCMParameters()
    for k in range(3):
        p.boxSize[k], p.cells[k] = 32.0, 32
    p.viscosity, p.seed, p.kernel = 1.0, 1, kernel
    h = C.c_void_p()
    assert lib.uammd_fcm_slab_create(C.byref(p), 16, 24, 8, 16, 0, C.byref(h)) != 0      # slab outside the grid
    assert lib.uammd_fcm_slab_create(C.byref(p), 16, 0, 2, 16, 0, C.byref(h)) != 0       # halo thinner than the stencil
    assert b"halo" in lib.uammd_hip_last_error()
