#!/bin/bash
# per-phase counters of k_fcm_spread_tile at C4 by ablation (tools/variants_fcm.sh build with VNAMES="base ab4 ab6 ab7 ab15" first):
# base -> ab4 (no tile sum / store) -> ab6 (+ no matrix phase) -> ab7 (+ no weights copy) -> ab15 (+ nothing accepted: ranges and tests only)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for n in base ab4 ab6 ab7 ab15; do
  echo "== $n"; UAMMD_HIP_LIB=$R/tools/_build/libf_$n.so python tools/time_fcm.py | tail -1
  UAMMD_HIP_LIB=$R/tools/_build/libf_$n.so timeout 300 tools/pmc_py.sh spread_$n k_fcm_spread_tile tools/time_fcm.py > /dev/null 2>&1
  cat gpurun_out/pmc_spread_${n}_*.txt | grep "true" | awk '{print $(NF-2), $(NF-1)}' | sort | tr '\n' ';'; echo
done
