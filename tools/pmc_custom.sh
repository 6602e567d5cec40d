#!/bin/bash
# usage (on the GPU box): tools/pmc_custom.sh <tag> <kernel-name-pattern> "<counters of one pass>" <python script + args...>
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; PAT=$2; CNT=$3; shift; shift; shift
rm -rf /tmp/pmcc_$TAG
(cd $R && rocprofv3 --pmc $CNT -d /tmp/pmcc_$TAG -o pmc -- python "$@" > /tmp/pmcc_$TAG.log 2>&1)
python3 - <<PY
import sqlite3, glob
for f in glob.glob('/tmp/pmcc_$TAG/**/*.db', recursive=True):
    db = sqlite3.connect(f); c = db.cursor()
    q = "select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%$PAT%' group by kernel_name, counter_name"
    for r in c.execute(q): print(r[0][:40], r[1], r[2], r[3])
PY
grep -i "error\|invalid\|not found" /tmp/pmcc_$TAG.log | head -3
