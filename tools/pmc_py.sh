#!/bin/bash
# usage (on the GPU box): tools/pmc_py.sh <tag> <kernel-name-pattern> <python script + args...>  -> gpurun_out/pmc_<tag>_<pass>.txt
# Counters in their own passes (no tracing domains beside --pmc).
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; PAT=$2; shift; shift
i=0
for CNT in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  rm -rf /tmp/pmc_${TAG}_$i
  (cd $R && rocprofv3 --pmc $CNT -d /tmp/pmc_${TAG}_$i -o pmc -- python "$@" > /tmp/pmc_${TAG}_$i.log 2>&1)
  python3 - <<PY > $R/gpurun_out/pmc_${TAG}_$i.txt
import sqlite3, glob
for f in glob.glob('/tmp/pmc_${TAG}_$i/**/*.db', recursive=True):
    db = sqlite3.connect(f); c = db.cursor()
    q = "select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%$PAT%' group by kernel_name, counter_name"
    for r in c.execute(q): print(r[0][:50], r[1], r[2], r[3])
PY
done
cat $R/gpurun_out/pmc_${TAG}_*.txt
