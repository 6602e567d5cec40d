#!/bin/bash
# round 3: single-launch key scan + hash kernel with one particle per thread in the fused step — parity, then timing
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_celllist.py tests/test_gpu_edge_cases.py tests/test_gpu_fused_step.py tests/test_gpu_lj_tile.py tests/test_gpu_lj.py tests/test_gpu_verletlist.py tests/test_gpu_slab_lj.py tests/test_gpu_integrators.py tests/test_gpu_ibm_fcm.py tests/test_gpu_pse.py -m gpu -x -q > gpurun_out/r3_build_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3_build_tests.log
MELT=100 timeout 600 python tools/time_build.py 2>&1 | tail -4
timeout 600 python bench.py --workload lj --steps 500 --no-cpu-baseline 2> gpurun_out/r3_build_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], d['timed_blocks'], 'kernel_ms', d['roofline']['kernel_ms'])"
