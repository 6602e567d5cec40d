#!/bin/bash
# round 3, first GPU call: the new tests, then bench.py's own N-rank launcher on a 1-GPU box
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_world2.py tests/test_gpu_lj_tile.py tests/test_golden.py tests/test_gpu_full_size.py tests/test_gpu_ibm_fcm.py tests/test_gpu_fcm_slab.py -m gpu -x -q > gpurun_out/r3_newtests.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r3_newtests.log
python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_refuse.out 2> gpurun_out/r3_refuse.err; echo "bench --gpus 2 on one GPU rc=$? (must be non-zero)"; cat gpurun_out/r3_refuse.err | tail -2
UAMMD_BENCH_SAME_DEVICE=1 UAMMD_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 100 --warmup 10 --fcm-steps 50 --particles 1000000 > gpurun_out/r3_bench_world2_samedevice.json 2> gpurun_out/r3_bench_world2_samedevice.err; echo "world2 rc=$?"; tail -3 gpurun_out/r3_bench_world2_samedevice.err; cat gpurun_out/r3_bench_world2_samedevice.json
