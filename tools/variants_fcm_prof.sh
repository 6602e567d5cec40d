#!/bin/bash
# on the GPU box: the spread kernel's time in each tools/_build/libf_<name>.so (rocprofv3 stats of tools/time_fcm.py)
cd "$(dirname "$0")/.."
for n in ${VNAMES:-base}; do
  UAMMD_HIP_LIB=$PWD/tools/_build/libf_$n.so tools/prof_any.sh v_$n tools/time_fcm.py > /dev/null 2>&1
  echo "$n: $(grep 'spread_tile<4, true>' gpurun_out/stats_v_$n.txt | cut -d'|' -f4)"
done
