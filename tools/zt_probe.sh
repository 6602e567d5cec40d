#!/bin/bash
# (GPU box) the fused z pass with and without the Fourier noise at C5 and C4: rocprofv3 average of k_fft_z_fused for T = 1 and T = 0
# measured r5: 146.8 / 31.8 us with the noise, 114.0 / 23.2 us without
cd /tmp; export TMPDIR=/tmp
for T in 1.0 0.0; do
for NC in 256 128; do
N=$((NC==256?200000:100000))
rm -rf /tmp/kt; T=$T N=$N NC=$NC rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/tools/time_fcm.py > /tmp/kt.log 2>&1; tail -1 /tmp/kt.log
python3 - <<PY
import sqlite3, glob
for f in glob.glob("/tmp/kt/**/*.db", recursive=True):
    db = sqlite3.connect(f); c = db.cursor()
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 8"):
        if "z_fused" in r[0]: print("T=$T NC=$NC", r[0][:50], f"{r[3]:.2f} us")
PY
done; done
