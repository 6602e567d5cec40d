#!/bin/bash
# kernel timeline (start, end, duration) of the last solves of tools/time_fcm.py (N, NC from the environment; library options as arguments)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt_pipe
(cd $R && N=${N:-200000} NC=${NC:-256} REPS=20 rocprofv3 --kernel-trace -d /tmp/kt_pipe -o kt -- python tools/time_fcm.py "$@" > /tmp/kt_pipe.log 2>&1)
python3 - <<PY
import sqlite3, glob
for f in glob.glob('/tmp/kt_pipe/**/*.db', recursive=True):
    db = sqlite3.connect(f); c = db.cursor()
    rows = list(c.execute("select name, start, end from kernels order by start"))
    win = rows[-22:]
    t0 = win[0][1]
    for r in win:
        print(f"{(r[1]-t0)/1e3:9.1f} .. {(r[2]-t0)/1e3:9.1f} us  dur {(r[2]-r[1])/1e3:6.1f}  {r[0][:70]}")
PY
