"""Builds libuammd_hip.so (hand-written HIP, gfx950) in-tree with hipcc.

`python -m uammd_amd.build` or `uammd_amd.build.build()`.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "lib", "libuammd_hip.so")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [
        os.path.join(_HERE, "..", "include", "uammd_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


def build(force=False, jobs=None, verbose=False):
    if not force and not _stale():
        return LIB
    jobs = jobs or min(8, os.cpu_count() or 1)
    cmd = ["make", "-C", CSRC, f"-j{jobs}"] + (["-B"] if force else [])
    res = subprocess.run(cmd, capture_output=not verbose, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libuammd_hip.so failed:\n" + (res.stdout or "") + (res.stderr or ""))
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
