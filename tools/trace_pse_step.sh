#!/bin/bash
# timeline of ONE BDHI::PSE step of bench.py (between two builds of the pair records): kernel start / end times from the rocprofv3 kernel
# trace, the gaps between consecutive kernels and their sum -> gpurun_out/trace_pse_step.txt
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt_trs
(cd $R && rocprofv3 --kernel-trace -d /tmp/kt_trs -o kt -- python bench.py --workload pse --pse-steps 20 --no-cpu-baseline > /tmp/kt_trs.log 2>&1)
python3 - <<PY > $R/gpurun_out/trace_pse_step.txt
import sqlite3, glob
for f in glob.glob('/tmp/kt_trs/**/*.db', recursive=True):
    db = sqlite3.connect(f); c = db.cursor()
    rows = list(c.execute("select name, start, end from kernels order by start"))
    builds = [i for i, r in enumerate(rows) if 'k_pse_pairs_build' in r[0]]
    # a step in the middle of the timed loop: from the first kernel after the previous step's last one; take builds[12] .. builds[13]
    for b in (12, 13):
        i0, i1 = builds[b], builds[b + 1]
        # step start = the hash kernel before the build
        first = lambda n: 'k_hash_agg' in n or 'k_pse_refresh_sorted' in n
        while not first(rows[i0][0]): i0 -= 1
        while not first(rows[i1][0]): i1 -= 1
        t0 = rows[i0][1]; prev_end = t0; gaps = 0.0; busy = 0.0
        print(f"# step from build {b}: {len(rows[i0:i1])} kernels, wall {(rows[i1][1]-t0)/1e3:.1f} us")
        for r in rows[i0:i1]:
            gap = (r[1] - prev_end) / 1e3
            gaps += max(gap, 0.0); busy += (r[2] - r[1]) / 1e3
            if b == 12: print(f"{(r[1]-t0)/1e3:9.1f} us  +gap {gap:6.1f}  dur {(r[2]-r[1])/1e3:6.1f}  {r[0][:70]}")
            prev_end = max(prev_end, r[2])
        gap = (rows[i1][1] - prev_end) / 1e3
        print(f"# last gap to the next step {gap:.1f}; busy {busy:.1f} us, gaps {gaps + gap:.1f} us")
PY
tail -4 $R/gpurun_out/trace_pse_step.txt
