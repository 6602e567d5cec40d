"""The slab decomposition's bookkeeping kernels (uammd_slab_*, csrc/slab.hip) against numpy: ordered selection, packed migration rows,
arrivals into holes + tail compaction (every relation between arrivals and leavers), largest displacement.  Integer work: exact."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else None


def _select(lib, check, pos, zup, zdown):
    n = pos.shape[0]
    nbytes = C.c_size_t(0)
    check(lib.uammd_slab_select_workspace(max(n, 1), C.byref(nbytes)))
    ws = torch.empty(int(nbytes.value), dtype=torch.uint8, device="cuda")
    iu = torch.full((max(n, 1),), -1, dtype=torch.int32, device="cuda")
    idn = torch.full((max(n, 1),), -1, dtype=torch.int32, device="cuda")
    cnt = torch.full((2,), -7, dtype=torch.int32, device="cuda")
    check(lib.uammd_slab_select(_p(pos), n, zup, zdown, C.c_void_p(iu.data_ptr()), C.c_void_p(idn.data_ptr()), C.c_void_p(cnt.data_ptr()),
                                C.c_void_p(ws.data_ptr()), None))
    a, b = cnt.tolist()
    return iu[:a].cpu().numpy(), idn[:b].cpu().numpy()


@pytest.mark.parametrize("n", [0, 1, 63, 1024, 1025, 100003])
def test_select_is_ordered_nonzero(hip, n):
    from uammd_amd._lib import check, load
    lib = load()
    rng = np.random.default_rng(n)
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(-10, 10, (n, 3))
    if n > 5:
        pos[3, 2] = 6.0       # exactly on the upper plane: z >= zUp selects it
        pos[4, 2] = -6.0      # exactly on the lower plane: z < zDown does not
    up, down = _select(lib, check, torch.from_numpy(pos).cuda(), 6.0, -6.0)
    assert np.array_equal(up, np.nonzero(pos[:, 2] >= 6.0)[0])
    assert np.array_equal(down, np.nonzero(pos[:, 2] < -6.0)[0])


@pytest.mark.parametrize("n,n_arrive_extra", [(5000, 0), (5000, 37), (5000, -40), (5000, -10**9), (300, 500), (64, -3)])
def test_pack_and_unpack_rows(hip, n, n_arrive_extra):
    """Leavers are packed, arrivals (another set of rows) go into the holes; the owned rows stay dense and nobody is lost."""
    from uammd_amd._lib import check, load
    lib = load()
    rng = np.random.default_rng(n + 7)
    cap = 2 * n + 1024
    pos = np.zeros((cap, 4), np.float32)
    pos[:n, :3] = rng.uniform(-10, 10, (n, 3))
    pos[:n, 3] = rng.integers(0, 3, n)
    vel = np.zeros((cap, 3), np.float32)
    vel[:n] = rng.normal(0, 1, (n, 3))
    ids = np.full(cap, -1, np.int32)
    ids[:n] = rng.permutation(n)
    dpos, dvel, dids = torch.from_numpy(pos).cuda(), torch.from_numpy(vel).cuda(), torch.from_numpy(ids).cuda()
    up, down = _select(lib, check, dpos[:n], 8.0, -8.0)
    n_up, n_down = len(up), len(down)
    n_leave = n_up + n_down
    iu, idn = torch.from_numpy(up).cuda(), torch.from_numpy(down).cuda()
    rows = torch.zeros((max(n_leave, 1), 8), dtype=torch.float32, device="cuda")
    check(lib.uammd_slab_pack_rows(_p(dpos), _p(dvel), _p(dids), _p(iu), n_up, _p(idn), n_down, -20.0, 20.0,
                                   _p(rows[:n_up]), _p(rows[n_up:n_leave]), None))
    got = rows.cpu().numpy()[:n_leave]
    exp = np.zeros((n_leave, 8), np.float32)
    order = np.concatenate([up, down])
    exp[:, :4] = pos[order]
    exp[:n_up, 2] += np.float32(-20.0)
    exp[n_up:, 2] += np.float32(20.0)
    exp[:, 4:7] = vel[order]
    exp[:, 7] = ids[order].view(np.float32)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    # arrivals: fresh particles with ids beyond n
    n_arrive = max(0, n_leave + n_arrive_extra)
    arr = np.zeros((max(n_arrive, 1), 8), np.float32)
    arr[:n_arrive, :3] = rng.uniform(-7, 7, (n_arrive, 3))
    arr[:n_arrive, 4:7] = rng.normal(0, 1, (n_arrive, 3))
    arr[:n_arrive, 7] = (n + np.arange(n_arrive, dtype=np.int32)).view(np.float32)
    darr = torch.from_numpy(arr).cuda()
    holes = torch.empty(max(n_leave, 1), dtype=torch.int32, device="cuda")
    check(lib.uammd_slab_unpack_rows(_p(dpos), _p(dvel), _p(dids), n, _p(iu), n_up, _p(idn), n_down, _p(darr), n_arrive,
                                     C.c_void_p(holes.data_ptr()), None))
    torch.cuda.synchronize()
    new_n = n - n_leave + n_arrive
    gp, gv, gi = dpos.cpu().numpy()[:new_n], dvel.cpu().numpy()[:new_n], dids.cpu().numpy()[:new_n]
    stay = np.setdiff1d(np.arange(n), order)
    want_ids = np.concatenate([ids[stay], n + np.arange(n_arrive, dtype=np.int32)])
    assert sorted(gi.tolist()) == sorted(want_ids.tolist()), "rows lost or duplicated"
    # every row carries its own particle's data
    ref_pos = {int(i): pos[k] for k, i in zip(stay, ids[stay])}
    ref_vel = {int(i): vel[k] for k, i in zip(stay, ids[stay])}
    for k in range(n_arrive):
        ref_pos[n + k] = arr[k, :4]
        ref_vel[n + k] = arr[k, 4:7]
    for r in range(new_n):
        assert np.array_equal(gp[r], ref_pos[int(gi[r])]) and np.array_equal(gv[r], ref_vel[int(gi[r])]), r
    # rows that did not have to move did not move
    untouched = stay[stay < min(new_n, n)]
    moved_targets = set(order.tolist())
    for k in untouched[:2000]:
        if k not in moved_targets:
            assert gi[k] == ids[k]


def test_max_displacement(hip):
    from uammd_amd._lib import check, load
    lib = load()
    rng = np.random.default_rng(3)
    n = 200001
    a = np.zeros((n, 4), np.float32)
    a[:, :3] = rng.uniform(-50, 50, (n, 3))
    b = a.copy()
    b[:, :3] += rng.normal(0, 0.05, (n, 3)).astype(np.float32)
    b[123456, :3] = a[123456, :3] + np.array([0.3, -0.4, 1.2], np.float32)
    out = torch.zeros(1, dtype=torch.float32, device="cuda")
    db, da = torch.from_numpy(b).cuda(), torch.from_numpy(a).cuda()
    check(lib.uammd_slab_max_displacement(_p(db), _p(da), n, C.c_void_p(out.data_ptr()), None))
    d = (b[:, :3].astype(np.float64) - a[:, :3]).astype(np.float32)
    exp = np.sqrt((d.astype(np.float64) ** 2).sum(axis=1)).max()
    assert abs(float(out.item()) - exp) <= 2e-6 * exp


@pytest.mark.parametrize("count,offset", [(0, 0), (1, 0), (1023, 0), (4096, 0), (4096, 1), (3 * 128 * 130 * 5, 0)])
def test_pair_add_and_copy(hip, count, offset):
    """uammd_slab_add2 / uammd_slab_copy2 against torch: two segments per launch, the 16-byte path (aligned, count % 4 == 0) and the
    scalar one (odd count or an offset view); bit-exact (one addition per element)."""
    from uammd_amd._lib import check, load
    lib = load()
    g = torch.Generator(device="cuda").manual_seed(count + offset)
    big = [torch.randn(count + 8, generator=g, device="cuda") for _ in range(4)]
    d0, s0, d1, s1 = (t[offset:offset + count] for t in big)
    want_add = (d0 + s0, d1 + s1)
    check(lib.uammd_slab_add2(_p(d0), _p(s0), _p(d1), _p(s1), count, None) if count else 0)
    assert torch.equal(d0, want_add[0]) and torch.equal(d1, want_add[1])
    check(lib.uammd_slab_copy2(_p(d0), _p(s0), _p(d1), _p(s1), count, None) if count else 0)
    assert torch.equal(d0, s0) and torch.equal(d1, s1)
    for t in big:  # nothing outside the segments moved
        assert torch.isfinite(t).all()
