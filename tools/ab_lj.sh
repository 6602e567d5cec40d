#!/bin/bash
# same-box comparison of LJ traversal kernels through bench.py --algo (0 AUTO = ring + half prefilter, 6 ring, 1 general)
for i in 1 2; do
  for algo in ${ALGOS:-0 6 1}; do
    python bench.py --workload lj --algo $algo --steps 400 --warmup 200 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('algo=$algo', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"
  done
done
