#!/bin/bash
# the round's GPU evidence in one call: parity tests, default bench, driver-like bench, kernel stats, HBM traffic, N > 1 on one device
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
T0=$(date +%s); timeout 900 python bench.py --gpus 1 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - T0 )) s"
cat gpurun_out/bench_default.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_driverlike.json 2>/dev/null
UAMMD_BENCH_SAME_DEVICE=1 UAMMD_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 100 --warmup 10 --fcm-steps 50 > gpurun_out/bench_world2_samedevice.json 2>/dev/null
timeout 300 tools/prof_stats.sh lj --workload lj --steps 500 --no-cpu-baseline > /dev/null 2>&1
timeout 300 tools/prof_stats.sh fcm --workload fcm --fcm-steps 50 --no-cpu-baseline > /dev/null 2>&1
timeout 300 tools/prof_stats.sh pse --workload pse --pse-steps 20 --no-cpu-baseline > /dev/null 2>&1
timeout 300 tools/prof_any.sh fcm_c4 tools/time_fcm.py > /dev/null 2>&1
N=200000 NC=256 timeout 300 tools/prof_any.sh fcm_c5 tools/time_fcm.py > /dev/null 2>&1
MELT=100 timeout 300 tools/prof_any.sh build tools/time_build.py > gpurun_out/time_build.log 2>&1
timeout 400 tools/pmc_traffic.sh lj_traversal k_lj_tile4 --workload lj --steps 20 --warmup 10 --equilibrate 100 --no-cpu-baseline > /dev/null 2>&1
timeout 400 tools/pmc_traffic.sh fcm_step k_fcm_spread,k_fcm_gather,k_fcm_step,k_fcm_prep,k_fcm_bin,k_fcm_tile,k_fcm_update,k_fcm_euler,k_fft_ --workload fcm --fcm-steps 20 --no-cpu-baseline --no-c5 > /dev/null 2>&1
timeout 400 tools/pmc_any.sh lj_tile k_lj_tile4 --workload lj --steps 20 --warmup 10 --equilibrate 100 --no-cpu-baseline > /dev/null 2>&1
timeout 500 python bench.py --force-distributed --workload lj --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_slab_world1_lj.json 2>/dev/null
timeout 500 python bench.py --force-distributed --workload fcm --fcm-steps 100 --no-cpu-baseline > gpurun_out/bench_slab_world1_fcm.json 2>/dev/null
timeout 400 tools/pmc_any.sh fcm_spread k_fcm_spread --workload fcm --fcm-steps 20 --no-cpu-baseline --no-c5 > /dev/null 2>&1
ls gpurun_out
