"""Generates the golden fixtures under tests/golden/ from the ORACLE (oracle/src/*.c).

The reference (UAMMD, CUDA header-only) cannot be compiled or run in this image (no nvcc / CUDA headers /
GPU), so the vectors come from the oracle, which is itself pinned against every known-answer
test the reference ships for this path (tests/test_oracle_*.py).  Seeds are recorded in each file.
Re-run with `python tests/golden/make_golden.py`; tests/test_golden.py checks the oracle and (on the GPU
box) the HIP library against the committed files.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from oracle.fcm import FCMOracle  # noqa: E402
from util import lattice_positions  # noqa: E402


def lanczos_dense():
    """lanczos_dense: the size-64 dense SPD case of test/misc/lanczos/test_lanczos.cu:196-209,236-269 — M from
    std::mt19937{29374238} + uniform_real_distribution{0,1}, symmetrised + 5 size I; operator M M^T, vector from
    std::mt19937{1234567} in [-10, 10]; answer sqrt(M M^T) v = M v.  Bv and iteration count of the oracle in both precisions."""
    from oracle.lanczos import LanczosOracle, std_mt19937_uniform_real
    size = 64
    A = std_mt19937_uniform_real(29374238, size * size, 0.0, 1.0).reshape(size, size)
    M = 0.5 * (A + A.T) + 5 * size * np.eye(size)
    M2 = M @ M.T
    v = std_mt19937_uniform_real(1234567, size, -10.0, 10.0)
    Bv64, it64 = LanczosOracle(np.float64).run(lambda x: M2 @ x, v, 1e-7, return_all=True)
    M2f = M2.astype(np.float32)
    Bv32, it32 = LanczosOracle(np.float32).run(lambda x: M2f @ x, v.astype(np.float32), 1e-6, return_all=True)
    assert np.abs((Bv64 - M @ v) / (M @ v)).max() <= 1e-7          # the reference's own assertion (test_lanczos.cu:262-266)
    np.savez_compressed(os.path.join(HERE, "lanczos_dense.npz"), size=size, M=M, M2=M2, v=v, theory=M @ v, Bv_f64=Bv64, iterations_f64=it64,
                        tolerance_f64=1e-7, Bv_f32=Bv32, iterations_f32=it32, tolerance_f32=1e-6, seed_matrix=29374238, seed_vector=1234567)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "lanczos_dense":
        return lanczos_dense()
    lanczos_dense()
    o32, o64 = oracle.get("f32"), oracle.get("f64")
    # saru_u32: first 16 outputs for 8 seed triples (exact)
    seeds = [(0, 0, 0), (1, 2, 3), (1234, 0, 0), (0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF), (7, 8, 9), (12345, 678, 9),
             (99, 1, 0xDEADBEEF), (42, 42, 42)]
    saru = np.stack([o32.saru_u32(list(s), 16) for s in seeds])
    saru1 = np.stack([o32.saru_u32([s[0]], 16) for s in seeds])
    saru2 = np.stack([o32.saru_u32([s[0], s[1]], 16) for s in seeds])
    np.savez_compressed(os.path.join(HERE, "saru_u32.npz"), seeds=np.array(seeds, np.uint64), three=saru, one=saru1,
                        two=saru2, f_range_1234=o32.saru_f_range(1234, -0.5, 0.5, 60))
    # celllist_c2small + lj_forces: N=4096, L=16, rc=2.5, jittered lattice seed 1234
    n, L, rc = 4096, 16.0, 2.5
    pos = lattice_positions(n, L, seed=1234, jitter=0.12)
    cd, oL, oper = o32.celllist_create_grid(L, 1, rc)
    cl = o32.celllist_build(pos, oL, oper, cd)
    par = o32.lj_params(rc, 1.0, 1.0)
    f32, e32, v32 = o32.lj_transverse_celllist(cl, L, 1, par, 1, n, True, True, True)
    f64 = o32.lj_nbody_f64(pos, L, 1, rc, 1.0, 1.0)
    np.savez_compressed(os.path.join(HERE, "celllist_lj_c2small.npz"), pos=pos, L=L, rc=rc, cellDim=cd, seed=1234,
                        hash=cl["hash"], index=cl["index"], cellStartMinusValid=cl["cellStart"].astype(np.int64) - n,
                        cellEnd=cl["cellEnd"], force_f32=f32, energy_f32=e32, virial_f32=v32, force_f64=f64)
    # ibm_spread_gather: 128 particles, 32^3, Gaussian P=6 and Peskin 3pt
    rng = np.random.default_rng(123)
    cells, Lg = [32, 32, 32], 16.0
    p = np.zeros((128, 4), np.float32)
    p[:, :3] = rng.uniform(-Lg / 2, Lg / 2, (128, 3))
    q = rng.normal(0, 1, (128, 3)).astype(np.float32)
    field = rng.normal(0, 1, (32, 32, 32, 3)).astype(np.float32)
    out = {}
    for name, k in (("gaussian", o32.fcm_gaussian(Lg / 32, 1e-3)["kernel"]),
                    ("peskin3", o32.ibm_kernel("peskin3", 3, invh=[32 / Lg] * 3))):
        out[name + "_spread"] = o32.ibm_spread(p, q, Lg, 1, cells, k)
        out[name + "_gather"] = o32.ibm_gather(p, field, Lg, 1, cells, k)
    np.savez_compressed(os.path.join(HERE, "ibm_spread_gather.npz"), pos=p, q=q, field=field, L=Lg, cells=cells, seed=123, **out)
    # fcm_det + fcm_noise: N=256, 32^3, tol 1e-3
    n = 256
    rng = np.random.default_rng(1234)
    p = np.zeros((n, 4), np.float32)
    p[:, :3] = rng.uniform(-16, 16, (n, 3))
    f = np.zeros((n, 4), np.float32)
    f[:, :3] = np.random.default_rng(4321).normal(0, 1, (n, 3))
    fcm = FCMOracle(o32, 32.0, [32, 32, 32], tolerance=1e-3, viscosity=1.0, seed=1234)
    v = fcm.displacements(p, f)
    nk = np.zeros((32, 32, 17, 3), np.complex64)
    npf = o32.fcm_noise_prefactor(10.0, 1.0, fcm.L, fcm.cells)
    o32.fcm_fourier_brownian_noise(nk, fcm.L, fcm.cells, npf, 1.0, 1234, 1)
    np.savez_compressed(os.path.join(HERE, "fcm_32.npz"), pos=p, force=f, velocity=v, noise_first64=nk.reshape(-1, 3)[:64],
                        noise_prefactor=npf, seed=1234, seed2=1, support=fcm.kinfo["support"], a_eff=fcm.hydrodynamicRadius)
    # reference known answers reproduced by the oracle (documented numbers, see tests/test_oracle_fcm.py)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
