#!/bin/bash
# timeline of ONE near-noise (Lanczos) call: kernel start / end times from the rocprofv3 kernel trace -> gpurun_out/trace_pse.txt
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt_tr
(cd $R && rocprofv3 --kernel-trace -d /tmp/kt_tr -o kt -- python tools/time_pse.py > /tmp/kt_tr.log 2>&1)
python3 - <<PY > $R/gpurun_out/trace_pse.txt
import sqlite3, glob
for f in glob.glob('/tmp/kt_tr/**/*.db', recursive=True):
    db = sqlite3.connect(f); c = db.cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if 'kernel_dispatch' in t and 'rocpd' in t][0] if any('kernel_dispatch' in t for t in tabs) else None
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    # views: kernels has name, start, end
    v = [t for t in tabs if t == 'kernels']
    rows = list(c.execute("select name, start, end from kernels order by start"))
    # find the last k_pse_noise_sorted and print the 60 kernels after it
    idx = max(i for i, r in enumerate(rows) if 'k_pse_noise_sorted' in r[0])
    t0 = rows[idx][1]
    prev_end = t0
    for r in rows[idx:idx + 60]:
        print(f"{(r[1]-t0)/1e3:9.1f} us  +gap {(r[1]-prev_end)/1e3:6.1f}  dur {(r[2]-r[1])/1e3:6.1f}  {r[0][:60]}")
        prev_end = r[2]
        if 'k_pse_unsort3' in r[0]: break
PY
head -70 $R/gpurun_out/trace_pse.txt
