#!/bin/bash
# the reference's five GoogleTest programs (examples/_build/ref_gtest_*) on the GPU box
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for t in ParticleSorter test_ibm_regular test_lanczos fcm_test pse_test; do
  echo "=== $t"; ( time timeout ${GT_TIMEOUT:-900} examples/_build/ref_gtest_$t ) > gpurun_out/gtest_$t.log 2>&1; echo "rc=$?"; grep -E "^\[  (PASSED|FAILED)|^\[==========\]|Failure" gpurun_out/gtest_$t.log | head -20; grep real gpurun_out/gtest_$t.log
done
