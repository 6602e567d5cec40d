#!/bin/bash
# world-1 overhead of the slab-decomposed drivers when every message goes through uammd_comm_* (RCCL, the ring closing on the rank itself)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python bench.py --gpus 1 --force-distributed --steps 200 --warmup 20 --fcm-steps 100 --no-cpu-baseline > gpurun_out/r4_bench_slab_world1.json 2> gpurun_out/r4_bench_slab_world1.err; echo "rc=$?"; tail -3 gpurun_out/r4_bench_slab_world1.err
timeout 900 python bench.py --gpus 1 --steps 200 --warmup 20 --fcm-steps 100 --no-cpu-baseline > gpurun_out/r4_bench_single.json 2>/dev/null
python - <<'PY'
import json
a = json.load(open("gpurun_out/r4_bench_slab_world1.json")); b = json.load(open("gpurun_out/r4_bench_single.json"))
print("comm:", a["comm"])
print("LJ   slab world 1 %.4f ms  single %.4f ms  (+%.1f %%)" % (a["ms_per_step"], b["ms_per_step"], 100 * (a["ms_per_step"] / b["ms_per_step"] - 1)))
print("FCM  slab world 1 %.4f ms  single %.4f ms  (+%.1f %%)" % (a["fcm"]["ms_per_step"], b["fcm"]["ms_per_step"], 100 * (a["fcm"]["ms_per_step"] / b["fcm"]["ms_per_step"] - 1)))
print("C5   slab world 1 %.4f ms  single %.4f ms  (+%.1f %%)" % (a["fcm_c5"]["ms_per_step"], b["fcm_c5"]["ms_per_step"], 100 * (a["fcm_c5"]["ms_per_step"] / b["fcm_c5"]["ms_per_step"] - 1)))
print(a["config"]["workload"])
PY
