#!/bin/bash
# A/B builds of pse.hip with -D flags into tools/_build/libp_<name>.so, timed with tools/time_pse_build.py through UAMMD_HIP_LIB:
#   VNAMES="base r4" tools/variants_pse.sh build   (here)      tools/variants_pse.sh run   (on the GPU box)
cd "$(dirname "$0")/.."
declare -A FLAGS=([base]="" [r3]="-DUAMMD_PSE_BUILD_ROWS=3" [r4]="-DUAMMD_PSE_BUILD_ROWS=4" [r6]="-DUAMMD_PSE_BUILD_ROWS=6" [w2]="-DUAMMD_PSE_BUILD_WROWS=2" [r4w2]="-DUAMMD_PSE_BUILD_ROWS=4 -DUAMMD_PSE_BUILD_WROWS=2")
NAMES=(${VNAMES:-base r3 r4 r6})
if [ "$1" = build ]; then
  mkdir -p tools/_build
  for n in "${NAMES[@]}"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -w ${FLAGS[$n]} -x hip -c uammd_amd/csrc/pse.hip -o tools/_build/pse_$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls uammd_amd/lib/obj/*.o | grep -v "/pse.o") tools/_build/pse_$n.o -o tools/_build/libp_$n.so -L/opt/rocm/lib -lrocfft -ldl
    rm -f tools/_build/pse_$n.o
  done
  ls tools/_build/libp_*.so
else
  for n in "${NAMES[@]}"; do
    UAMMD_HIP_LIB=$PWD/tools/_build/libp_$n.so tools/prof_any.sh pse_$n tools/time_pse_build.py > /dev/null 2>&1
    echo "$n: $(grep k_pse_pairs_build gpurun_out/stats_pse_$n.txt | cut -d'|' -f2-4)"
  done
fi
