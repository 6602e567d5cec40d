#!/bin/bash
# kernel timeline of a window of a bench.py run: start, gap to the previous kernel's end, duration — and busy / gap totals
#   tools/trace_gaps.sh <tag> <first kernel index> <count> <bench.py args...>     -> gpurun_out/trace_<tag>.txt
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; FIRST=$2; COUNT=$3; shift 3
rm -rf /tmp/kt_g_$TAG
(cd $R && rocprofv3 --kernel-trace -d /tmp/kt_g_$TAG -o kt -- python bench.py "$@" > /tmp/kt_g_$TAG.log 2>&1)
python3 - <<PY > $R/gpurun_out/trace_$TAG.txt
import sqlite3, glob
for f in glob.glob('/tmp/kt_g_$TAG/**/*.db', recursive=True):
    db = sqlite3.connect(f); c = db.cursor()
    rows = list(c.execute("select name, start, end from kernels order by start"))
    first = $FIRST if $FIRST >= 0 else len(rows) + $FIRST
    win = rows[first:first + $COUNT]
    t0 = win[0][1]; prev = t0; gaps = busy = 0.0
    for r in win:
        gap = (r[1] - prev) / 1e3
        gaps += max(gap, 0.0); busy += (r[2] - r[1]) / 1e3
        print(f"{(r[1]-t0)/1e3:9.1f} us  +gap {gap:6.1f}  dur {(r[2]-r[1])/1e3:6.1f}  {r[0][:80]}")
        prev = max(prev, r[2])
    print(f"# {len(win)} kernels of {len(rows)}: wall {(prev-t0)/1e3:.1f} us, busy {busy:.1f}, gaps {gaps:.1f}")
PY
tail -1 $R/gpurun_out/trace_$TAG.txt
