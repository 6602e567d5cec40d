#!/bin/bash
# (GPU box) per-kernel times of the C5 solve: tools/c5_prof.sh [ENV=VALUE ...]
cd /tmp; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rm -rf /tmp/kt5; N=${N:-200000} NC=${NC:-256} rocprofv3 --kernel-trace --stats -d /tmp/kt5 -o kt -- python $GRAFT_REPO_ROOT/tools/time_fcm.py > /tmp/kt5.log 2>&1; grep "ms per" /tmp/kt5.log
python3 - <<PY
import sqlite3, glob
for f in glob.glob("/tmp/kt5/**/*.db", recursive=True):
    db = sqlite3.connect(f); c = db.cursor()
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 9"):
        print(f"{r[0][:80]} | {r[1]} | {r[3]:.2f}")
PY
