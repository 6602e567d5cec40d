"""Shared synthetic inputs for the parity tests (numpy, seeded)."""
import numpy as np


def lattice_positions(n, L, seed=1234, jitter=0.1, ntypes=1, dtype=np.float32):
    """Simple cubic lattice filling the box [-L/2, L/2)^3 + uniform jitter; w = particle type."""
    L = np.broadcast_to(np.asarray(L, dtype=np.float64), (3,))
    m = int(np.ceil(n ** (1.0 / 3.0)))
    rng = np.random.default_rng(seed)
    g = np.stack(np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij"), -1).reshape(-1, 3)
    sel = rng.permutation(len(g))[:n]
    sel.sort()
    p = (g[sel] + 0.5) / m * L - L / 2
    p = p + rng.uniform(-jitter, jitter, p.shape)
    p = p[rng.permutation(n)]
    pos = np.zeros((n, 4), dtype)
    pos[:, :3] = p
    if ntypes > 1:
        pos[:, 3] = rng.integers(0, ntypes, n)
    return pos


def canon_cell_tables(cl):
    """(start, end) per cell with -1/-1 for empty cells: stale entries of empty cells are not part
    of the contract (the reference never reads cellEnd of an empty cell)."""
    cs = cl["cellStart"].astype(np.int64) - int(cl["validCell"])
    empty = cl["cellStart"].astype(np.int64) < int(cl["validCell"])
    start = np.where(empty, -1, cs)
    end = np.where(empty, -1, cl["cellEnd"].astype(np.int64))
    return start, end
