#!/usr/bin/env python
"""Golden vectors for initLattice (utils/InitialConditions.cuh:17-32) from the REFERENCE's own generator.

Runs oracle/_ref/bravais_dump — oracle/ref_drivers/bravais_dump.c (a main() of ours) around /root/reference/src/third_party/bravais/
bravais.h, compiled where it lies by `make -C oracle -f ref.mk` — for a set of (lattice, N, box) cases and stores the float32 positions
BEFORE initLattice's + 0.56 shift (the shift is part of what the product is tested on: the test adds it in float32, as the reference
does).  The benchmark's own case (fcc, 2^20 particles, L = 128: examples/misc/benchmark.cu:21,63) is stored as a SHA-256 of the bytes.
Only runs in the build container (needs /root/reference)."""
import hashlib
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
EXE = os.path.join(ROOT, "oracle", "_ref", "bravais_dump")
TYPES = {"sc": 0, "bcc": 1, "fcc": 2, "dia": 3, "hcp": 4, "sq": 5, "tri": 6}
CASES = [("sc", 1000, (10, 10, 10)), ("sc", 777, (9.5, 12.25, 7.0)), ("bcc", 432, (8, 8, 8)), ("fcc", 500, (10, 10, 10)),
         ("fcc", 4001, (21.5, 17.0, 33.0)), ("dia", 512, (12, 12, 12)), ("hcp", 600, (11, 13, 9)), ("sq", 400, (20, 20, 0)),
         ("tri", 333, (16, 24, 0)), ("fcc", 16384, (27.36, 27.36, 27.36))]
BIG = [("fcc", 1 << 20, (128, 128, 128)), ("sc", 1_000_000, (107.7217345, 107.7217345, 107.7217345))]


def run(kind, n, L):
    out = subprocess.run([EXE, str(TYPES[kind]), str(n)] + [repr(float(np.float32(x))) for x in L], check=True, capture_output=True).stdout
    return np.frombuffer(out, np.float32).reshape(n, 4).copy()


if __name__ == "__main__":
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-f", "ref.mk"], check=True)
    data = {}
    for c, (kind, n, L) in enumerate(CASES):
        data[f"case{c}_kind"] = np.array(kind)
        data[f"case{c}_L"] = np.asarray(L, np.float32)
        data[f"case{c}_pos"] = run(kind, n, L)
    for c, (kind, n, L) in enumerate(BIG):
        data[f"big{c}_kind"] = np.array(kind)
        data[f"big{c}_n"] = np.array(n)
        data[f"big{c}_L"] = np.asarray(L, np.float32)
        data[f"big{c}_sha256"] = np.array(hashlib.sha256(run(kind, n, L).tobytes()).hexdigest())
    np.savez_compressed(os.path.join(HERE, "bravais_lattices.npz"), **data)
    print("wrote bravais_lattices.npz:", len(CASES), "cases +", len(BIG), "digests")
