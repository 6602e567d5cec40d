"""initLattice (utils/InitialConditions.cuh:17-32): N positions on a Bravais lattice filling the box.

The reference delegates to its vendored generator (third_party/bravais/bravais.h: Bravais(), lattice types sc, bcc, fcc, dia, hcp, sq, tri)
and then shifts every particle by 0.56 (and zeroes z for a 2D box).  This is an own, vectorised statement of that construction with the
generator's arithmetic types kept (float products, the box offset and the node counts in double), so that it returns the SAME float32
positions — pinned by tests/golden/bravais_lattices.npz, which oracle/_ref/bravais_dump (the reference's own header, compiled by
oracle/ref.mk) produced.  The C++ twin is include/uammd/utils/InitialConditions.cuh.

Construction: ncells = ceil(N / basis size) unit cells are laid on an nx x ny x nz grid chosen from the box's aspect ratio
(n_x = ceil((ncells / V)^(1/3) Lx), ...), the lattice is stretched per axis to fill the box exactly, nodes are visited x-slowest /
basis-fastest and the first N are kept.  `dia` carries the generator's own basis table, including its seventh and eighth entries
(0.25, 0.25, 0.75) and (0, 0.75, 0.75).
"""
import math

import numpy as np

LATTICES = ("sc", "bcc", "fcc", "dia", "hcp", "sq", "tri")
_F = np.float32


def _lattice(kind):
    e = np.zeros((3, 3), _F)
    if kind in ("sc", "bcc", "fcc", "dia"):
        e[0, 0] = e[1, 1] = e[2, 2] = 1
    elif kind == "hcp":
        e[0, 0] = 1
        e[1, 0], e[1, 1] = 0.5, _F(math.sqrt(3) / 2)
        e[2, 2] = _F(2 * math.sqrt(6) / 3)
    elif kind == "sq":
        e[0, 0] = e[1, 1] = 1
    elif kind == "tri":
        e[0, 0] = 1
        e[1, 0], e[1, 1] = 0.5, _F(math.sqrt(3) / 2)
    else:
        raise ValueError(f"unknown lattice {kind!r} (one of {LATTICES})")
    basis = {
        "bcc": [(0, 0, 0), (.5, .5, .5)],
        "fcc": [(0, 0, 0), (.5, .5, 0), (.5, 0, .5), (0, .5, .5)],
        "dia": [(0, 0, 0), (.5, .5, 0), (.5, 0, .5), (0, .5, .5), (.25, .25, .25), (.75, .75, .25), (.25, .25, .75), (0, .75, .75)],
        "hcp": [(0, 0, 0), (.5, .25, math.sqrt(6.) / 3)],
    }.get(kind, [(0, 0, 0)])
    return e, np.asarray(basis, _F)


def init_lattice(L, N, kind="sc", shift=0.56):
    """(N, 4) float32 positions, w = 0.  L = (Lx, Ly, Lz); Lz = 0 means a 2D box (z = 0 everywhere).  shift: initLattice's offset of every
    coordinate (0.56; 0 gives the bare generator output)."""
    L3 = np.asarray(np.broadcast_to(np.asarray(L, _F), (3,)), _F).copy()
    N = int(N)
    e, basis = _lattice(kind)
    two_d = kind in ("sq", "tri")
    Lg = L3.copy()
    if two_d:
        Lg[2] = 1          # Lz does not contribute to the volume
    nb = len(basis)
    ncells = int(math.ceil(float(_F(N) / (_F(1.0) * _F(nb)))))
    V = _F(_F(Lg[0] * Lg[1]) * Lg[2])
    dens = float(_F(ncells) / V)                         # int / float: a float quotient, then double arithmetic
    if two_d:
        nx = int(math.ceil(math.sqrt(dens) * float(Lg[0])))
        ny = int(math.ceil(float(_F(ncells) / (_F(1.0) * _F(nx)))))
        nz = 1
    else:
        c = dens ** (1 / 3.)
        nx = int(math.ceil(c * float(Lg[0])))
        ny = int(math.ceil(c * float(Lg[1])))
        nz = int(math.ceil(float(_F(ncells) / _F(_F(_F(1.0) * _F(nx)) * _F(ny)))))
    with np.errstate(divide="ignore"):
        stretch = np.array([Lg[0] / (_F(nx) * e[0, 0]), Lg[1] / (_F(ny) * e[1, 1]), Lg[2] / (_F(nz) * e[2, 2])], _F)
    # the first N nodes in the generator's visiting order: i (x) slowest, then j, k, basis element fastest
    per_i = ny * nz * nb
    ni = min(nx, -(-N // per_i))
    i, j, k, l = np.meshgrid(np.arange(ni), np.arange(ny), np.arange(nz), np.arange(nb), indexing="ij")
    i, j, k, l = (a.reshape(-1)[:N] for a in (i, j, k, l))
    if len(i) < N:
        raise ValueError("initLattice: the lattice holds fewer nodes than particles")  # (cannot happen: nx ny nz nb >= N)
    fi, fj, fk = i.astype(_F), j.astype(_F), k.astype(_F)
    pos = np.zeros((N, 4), _F)
    for d in range(3):
        inner = ((fi * e[0, d] + fj * e[1, d]) + fk * e[2, d]) + basis[l, d]                   # float
        with np.errstate(invalid="ignore"):   # (2D lattices: the z stretch is 1 / 0; z is zeroed below)
            r = (-(Lg[d].astype(np.float64)) / 2. + (stretch[d] * inner).astype(np.float64)).astype(_F)  # the sum in double, stored as float
        if d == 0 and kind in ("tri", "hcp"):
            r = np.where(r > Lg[0] / _F(2), r - Lg[0], r).astype(_F)
        if d == 2 and two_d:
            r = np.zeros_like(r)
        pos[:, d] = r
    pos[:, :3] += _F(shift)
    if L3[2] == 0:
        pos[:, 2] = 0
    return pos
