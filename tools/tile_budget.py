#!/usr/bin/env python
"""Static instruction budget of k_lj_tile4<2, false, false> by section (CPU only; no GPU needed).
Builds uammd_amd/csrc/lj_tile.hip to assembly with -DUAMMD_TILE_MARKERS (comment marks at the section boundaries) and counts the
instructions between the marks in program order: vector ALU (v_*, MFMA apart), scalar (s_*), LDS (ds_*), global / scratch memory.
Loop bodies are counted once; the caller multiplies by the measured trip counts (12.6 words and 18.9 drain iterations per pass at C3).
Program order is not execution order across branches: the fallback paths (tile_solo) sit in their own blocks after the marks of the
main path and are reported under the mark that precedes them in the file ("after ..."), so read the loop bodies and the straight-line
sections, not the tails.  usage: python tools/tile_budget.py"""
import os, re, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "tools", "_build", "lj_tile_marked.s")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                "-fno-slp-vectorize", "-w", "-DUAMMD_TILE_MARKERS", "--cuda-device-only", "-S", "-x", "hip",
                os.path.join(ROOT, "uammd_amd", "csrc", "lj_tile.hip"), "-o", out], check=True)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"_ZN9uammd_hip10k_lj_tile4ILi2ELb0ELb0E\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
sec, counts, order = "entry", collections.OrderedDict(), []
def kind(op):
    if "mfma" in op: return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")): return "vmem"
    if op.startswith("s_"): return "salu"
    return "other"
for l in lines[start + 1:end + 1]:
    t = l.strip()
    m = re.match(r"; MARK (\S+) (\d)", t)
    if m:
        sec = m.group(1) + ("_pbc" if m.group(2) == "1" and not m.group(1).startswith("k4") else "")
        continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"): continue
    op = t.split()[0]
    counts.setdefault(sec, collections.Counter())[kind(op)] += 1
print(f"{'section':<22}{'valu':>6}{'mfma':>6}{'lds':>6}{'vmem':>6}{'salu':>6}")
for s, c in counts.items():
    print(f"{s:<22}{c['valu']:>6}{c['mfma']:>6}{c['lds']:>6}{c['vmem']:>6}{c['salu']:>6}")
