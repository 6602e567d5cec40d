#!/bin/bash
# build lj.hip variants (-D flags) into alternative libraries, then (on the GPU box) time each: tools/variants_lj.sh build|run
cd "$(dirname "$0")/.."
VARS=("RING_CAP=32 RING_TAKE=8" "RING_CAP=32 RING_TAKE=4" "RING_CAP=32 RING_TAKE=12" "RING_CAP=32 RING_TAKE=16" "RING_CAP=16 RING_TAKE=4" "RING_CAP=64 RING_TAKE=16" "RING_CAP=64 RING_TAKE=8")
if [ "$1" = build ]; then
  mkdir -p tools/_build
  i=0
  for v in "${VARS[@]}"; do
    D=""; for kv in $v; do D="$D -D$kv"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -w $D -c uammd_amd/csrc/lj.hip -o tools/_build/lj_v$i.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls uammd_amd/lib/obj/*.o | grep -v "/lj.o") tools/_build/lj_v$i.o -o tools/_build/libv$i.so -L/opt/rocm/lib -lrocfft -ldl
    rm -f tools/_build/lj_v$i.o
    i=$((i+1))
  done
else
  cp uammd_amd/lib/libuammd_hip.so /tmp/orig.so
  i=0
  for v in "${VARS[@]}"; do
    cp tools/_build/libv$i.so uammd_amd/lib/libuammd_hip.so
    python bench.py --workload lj --steps 300 --warmup 150 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"
    i=$((i+1))
  done
  cp /tmp/orig.so uammd_amd/lib/libuammd_hip.so
fi
