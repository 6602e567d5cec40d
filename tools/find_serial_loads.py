#!/usr/bin/env python
"""Finds loops in which the compiler waits for every global load before it issues the next (`global_load; s_waitcnt vmcnt(0)` once per
iteration): a copy loop `lds[f(i)] = g[h(i)]` over a run-time count comes out that way, one dependent round trip per element.
usage: python tools/find_serial_loads.py [file.hip ...]   (default: every .hip under uammd_amd/csrc; needs hipcc, no GPU)"""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function --cuda-device-only -S".split()

for src in sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "uammd_amd", "csrc", "*.hip"))):
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, src, "-o", tmp.name], check=True, stderr=subprocess.DEVNULL)
        lines = open(tmp.name).read().split("\n")
    kernels = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^(_ZN9uammd_hip\w+):", l)] if m]
    for start, name in kernels:
        end = next((j for j in range(start, len(lines)) if "s_endpgm" in lines[j]), len(lines))
        in_loop, pending, singles, multi = False, 0, 0, 0
        for l in lines[start:end]:
            if re.match(r"^\.LBB", l):
                in_loop = "Loop" in l
            if re.search(r"\b(global_load|buffer_load)", l):
                pending += 1
            m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)
            if m and int(m.group(1)) == 0:
                if in_loop and pending == 1:
                    singles += 1
                elif in_loop and pending > 1:
                    multi += 1
                pending = 0
        if singles:
            demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            print(f"{os.path.basename(src)}: {demangled[:110]}: {singles} single-load waits inside loops ({multi} with several loads in flight)")
