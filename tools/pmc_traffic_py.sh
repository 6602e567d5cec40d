#!/bin/bash
# usage (on the GPU box): tools/pmc_traffic_py.sh <tag> <python script + args...>  -> gpurun_out/traffic_py_<tag>.txt
# FETCH_SIZE / WRITE_SIZE per kernel (KiB per launch; FETCH x2 on gfx950: 128-B requests tallied at 64 B), each counter in its own pass
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
for CNT in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/trp_${TAG}_$CNT
  (cd $R && rocprofv3 --pmc $CNT -d /tmp/trp_${TAG}_$CNT -o pmc -- python "$@" > /tmp/trp_${TAG}_$CNT.log 2>&1)
done
python3 - <<PY > $R/gpurun_out/traffic_py_$TAG.txt
import sqlite3, glob
rows = {}
for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f'/tmp/trp_${TAG}_{cnt}/**/*.db', recursive=True):
        c = sqlite3.connect(f).cursor()
        for r in c.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name='%s' group by kernel_name" % cnt):
            rows.setdefault(r[0][:70], {})[cnt] = (r[1], r[2])
print("# kernel | launches | fetch MB per launch (x2 corrected) | write MB per launch")
for k, v in sorted(rows.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", (0, 0))[0])):
    f = v.get("FETCH_SIZE", (0, 0)); w = v.get("WRITE_SIZE", (0, 0))
    print(f"{k} | {f[1]} | {2 * f[0] * 1024 / 1e6:.1f} | {w[0] * 1024 / 1e6:.1f}")
PY
cat $R/gpurun_out/traffic_py_$TAG.txt
