"""The oracle's all-cores mode (bench.py's cpu_baseline) must not change what the oracle computes."""
import numpy as np
def test_all_cores_build_is_identical(o32):
    """bench.py's cpu_baseline runs the oracle's build with OpenMP (oracle_set_parallel): chunked stable radix sort, parallel
    hash / reorder / cell tables.  Same tables, bit for bit."""
    rng = np.random.default_rng(5)
    n, L = 20000, 30.0
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(-L / 2, L / 2, (n, 3))
    cd, oL, oper = o32.celllist_create_grid(L, 1, 2.5)
    a = o32.celllist_build(pos, oL, oper, cd)
    o32.set_parallel(True)
    try:
        b = o32.celllist_build(pos, oL, oper, cd)
    finally:
        o32.set_parallel(False)
    for k in ("hash", "index", "sortPos", "cellStart", "cellEnd"):
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
