#!/bin/bash
# same-box A/B of the LJ step: UAMMD_AB unset vs set (a temporary getenv switch in the code under test)
for i in 1 2; do
  for ab in "" 1; do
    if [ -n "$ab" ]; then export UAMMD_AB=1; else unset UAMMD_AB; fi
    python bench.py --workload lj --steps 400 --warmup 200 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('AB=${ab:-0}', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"
  done
done
