#!/bin/bash
# the reference's GoogleTest programs inside SURVEY 8 (five pins of 8c + the three of the 8f.4 consumers) (examples/_build/ref_gtest_*) on the GPU box
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for t in ParticleSorter test_ibm_regular test_ibm test_lanczos fcm_test pse_test quasi2d_test test_poisson test_tp_quadrupole; do
  echo "=== $t"; ( time timeout ${GT_TIMEOUT:-900} examples/_build/ref_gtest_$t ) > gpurun_out/gtest_$t.log 2>&1; echo "rc=$?"; grep -E "^\[  (PASSED|FAILED)|^\[==========\]|Failure" gpurun_out/gtest_$t.log | head -20; grep real gpurun_out/gtest_$t.log
done
