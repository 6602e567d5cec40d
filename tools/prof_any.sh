#!/bin/bash
# usage (on the GPU box): tools/prof_any.sh <tag> <python script + args...>  -> gpurun_out/stats_<tag>.txt
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
rm -rf /tmp/kt_$TAG
(cd $R && rocprofv3 --kernel-trace --stats -d /tmp/kt_$TAG -o kt -- python "$@" > /tmp/kt_$TAG.log 2>&1)
python3 - <<PY > $R/gpurun_out/stats_$TAG.txt
import sqlite3, glob
print("# rocprofv3 --kernel-trace --stats -- python $*")
print("# name | calls | total_us | avg_us | pct")
for f in glob.glob('/tmp/kt_$TAG/**/*.db', recursive=True):
    db = sqlite3.connect(f); c = db.cursor()
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 25"):
        print(f"{r[0][:90]} | {r[1]} | {r[2]:.1f} | {r[3]:.2f} | {r[4]:.2f}")
PY
grep -v "^W2026\|^E2026" /tmp/kt_$TAG.log | tail -6
cat $R/gpurun_out/stats_$TAG.txt
