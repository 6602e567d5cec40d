#!/bin/bash
# usage (GPU box): tools/pmc_traffic.sh <tag> <kernel-name-pattern> <bench args...>
# HBM traffic of one kernel from the TCC counters, FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit
# one pass).  Units: rocprofv3 reports both in KiB-like units of 1 KB; FETCH_SIZE on gfx950 counts 64 B per 128-B request
# (MI355X_MICROARCH.md, HBM section) -> doubled here.  Writes gpurun_out/traffic_<tag>.json
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; PAT=$2; shift; shift
for CNT in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tr_${TAG}_$CNT
  rocprofv3 --pmc $CNT -d /tmp/tr_${TAG}_$CNT -o pmc -- python $R/bench.py "$@" > /tmp/tr_${TAG}_$CNT.log 2>&1
done
python3 - <<PY
import sqlite3, glob, json
pats = "$PAT".split(",")   # several patterns: bytes summed over all of them, per launch of the FIRST one
out = {"kernel_patterns": pats, "command": "python bench.py $*", "raw": {}}
tot = {}
for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f'/tmp/tr_${TAG}_{cnt}/**/*.db', recursive=True):
        c = sqlite3.connect(f).cursor()
        launches = None
        s = 0.0
        for i, pat in enumerate(pats):
            q = "select kernel_name, sum(value), count(*) from counters_collection where kernel_name like '%%%s%%' and counter_name='%s' group by kernel_name" % (pat, cnt)
            for r in c.execute(q):
                out["raw"].setdefault(cnt, []).append({"kernel": r[0][:80], "sum_value": r[1], "launches": r[2]})
                s += r[1]
                if i == 0:   # (every instantiation of the first pattern counts: k_fcm_spread_tile<4, true> + <4, false>)
                    launches = (launches or 0) + r[2]
        if launches:
            tot[cnt] = s / launches
fs, ws = tot.get("FETCH_SIZE"), tot.get("WRITE_SIZE")
# counter unit is kilobytes (1024 B); gfx950 FETCH_SIZE tallies 128-B requests at 64 B -> x2
if fs is not None and ws is not None:
    out["fetch_bytes_per_launch"] = 2.0 * fs * 1024
    out["write_bytes_per_launch"] = ws * 1024
    out["hbm_bytes_per_launch"] = out["fetch_bytes_per_launch"] + out["write_bytes_per_launch"]
    out["correction"] = "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE uncalibrated, taken as is; unit KiB"
json.dump(out, open("$R/gpurun_out/traffic_$TAG.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
