cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_f64.py tests/test_abi.py -q -x -k "bd_schemes or abi or exports" 2>&1 | tail -3
examples/_build/dp_bd 2>&1 | tail -12
examples/_build/chebyshev_grid | tail -2
python -m pytest tests/test_cxx_interface.py -q -x 2>&1 | tail -3
for k in 1 2 3; do ( time timeout 900 examples/_build/ref_gtest_quasi2d_test ) > gpurun_out/gtest_quasi2d_test_try$k.log 2>&1; echo "try $k rc=$?"; grep -E "^\[  (PASSED|FAILED)" gpurun_out/gtest_quasi2d_test_try$k.log | head -3; done
