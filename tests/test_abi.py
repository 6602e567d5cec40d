"""CPU test: the C-ABI library loads without a GPU and exports every symbol include/uammd_hip.h declares (and the
ctypes table in uammd_amd/_lib.py lists exactly those).  No compute calls."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "uammd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(uammd_[a-z0-9_]+)\s*\(", src))


def test_header_library_bindings_agree():
    from uammd_amd import _lib
    from uammd_amd import build as hipbuild
    hipbuild.build()
    lib = _lib.load()
    declared = _declared()
    assert len(declared) >= 30
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (uammd_[a-z0-9_]+)", nm))
    assert declared <= exported, f"declared but not exported: {sorted(declared - exported)}"
    assert exported <= declared, f"exported but not declared in the header: {sorted(exported - declared)}"
    assert set(_lib.SIGNATURES) == declared, sorted(set(_lib.SIGNATURES) ^ declared)
    assert lib.uammd_hip_abi_version() == 1
    assert lib.uammd_hip_last_error() is not None


def test_host_only_entry_points():
    """Entry points that are pure host logic may be called without a GPU."""
    import ctypes as C
    from uammd_amd import _lib
    from uammd_amd._lib import IBMKernel, LJPairParameters, check, f3, i3
    lib = _lib.load()
    cd, Lo, po = i3(0), f3(0), i3(0)
    check(lib.uammd_celllist_create_grid(f3(107.7217345), i3(1), f3(2.5), cd, Lo, po))
    assert list(cd) == [43, 43, 43] and list(po) == [1, 1, 1]
    check(lib.uammd_celllist_create_grid(f3(50.0), i3(1), f3(2.5), cd, Lo, po))
    assert list(cd) == [20, 20, 20]
    p = LJPairParameters()
    check(lib.uammd_lj_process_pair_parameters(2.5, 1.0, 1.0, 0, C.byref(p)))
    assert (p.cutOff2, p.sigma2, p.epsilonDivSigma2, p.shift) == (6.25, 1.0, 1.0, 0.0)
    k, a = IBMKernel(), C.c_float()
    check(lib.uammd_fcm_gaussian_kernel(1.0, 1e-3, C.byref(k), C.byref(a)))
    assert list(k.support) == [6, 6, 6] and abs(a.value - 1.46674) < 1e-5
    assert abs(lib.uammd_fcm_self_mobility(1.0, 1.0, 1e9) - 1 / (6 * 3.141592653589793)) < 1e-9
    assert lib.uammd_hip_set_tunable(b"nope", 1) != 0 and b"unknown" in lib.uammd_hip_last_error()


def test_product_does_not_import_oracle():
    """The product path must never route through the oracle (or any CPU fallback)."""
    for d, _, files in os.walk(os.path.join(ROOT, "uammd_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, (d, f)
    for f in os.listdir(os.path.join(ROOT, "include")):
        p = os.path.join(ROOT, "include", f)
        if os.path.isfile(p):
            assert "oracle" not in open(p).read().lower()


def test_xorshift_seed_wraps_like_uint64():
    """Xorshift128plus::setSeed (utils/utils.h:110-113): s[1] = (s0 + K) % RANDOM_MAX in uint64 arithmetic — the sum wraps at
    2^64 BEFORE the modulo.  The default System seed overflows, so the Python mirror must wrap too."""
    import numpy as np
    from uammd_amd.md import Xorshift128plus
    K = 15438657923749336752
    for s0 in (0xf31337Bada55D00d, 1234, 2**64 - 1, 2**64 - K, 2**64 - K - 1):
        r = Xorshift128plus()
        r.set_seed(s0)
        with np.errstate(over="ignore"):
            want = int((np.uint64(s0) + np.uint64(K)) % np.uint64(0xFFFFFFFFFFFFFFFF))
        assert r.s == [s0, want], hex(s0)
