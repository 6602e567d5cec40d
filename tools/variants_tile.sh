#!/bin/bash
# build lj_tile.hip variants (-D flags) into alternative libraries (tools/_build/libt_<name>.so), then on the GPU box time each with
# tools/time_lj.py through UAMMD_HIP_LIB: tools/variants_tile.sh build|run
cd "$(dirname "$0")/.."
NAMES=(${VNAMES:-base alignbit nounit timeline})
declare -A FLAGS=([base]="" [alignbit]="-DUAMMD_TILE_ALIGNBIT" [nounit]="-DUAMMD_TILE_NOUNIT" [timeline]="-DUAMMD_TILE_TIMELINE" [old]="-DUAMMD_TILE_ALIGNBIT -DUAMMD_TILE_NOUNIT" [pop3]="-DUAMMD_TILE_POP3" [reload]="-DUAMMD_TILE_RELOAD_NEXT" [nonewton]="-DUAMMD_TILE_NO_NEWTON" [both]="-DUAMMD_TILE_RELOAD_NEXT -DUAMMD_TILE_NO_NEWTON")
if [ "$1" = build ]; then
  mkdir -p tools/_build
  for n in "${NAMES[@]}"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -w ${FLAGS[$n]} $VEXTRA -x hip -c ${VSRC:-uammd_amd/csrc/lj_tile.hip} -o tools/_build/lj_tile_$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls uammd_amd/lib/obj/*.o | grep -v "/lj_tile.o") tools/_build/lj_tile_$n.o -o tools/_build/libt_$n.so -L/opt/rocm/lib -lrocfft -ldl
    rm -f tools/_build/lj_tile_$n.o
  done
  ls -la tools/_build/*.so
else
  for rep in 1 2; do
    for n in "${NAMES[@]}"; do
      echo -n "$n: "; UAMMD_HIP_LIB=$PWD/tools/_build/libt_$n.so python tools/time_lj.py ${TN:-1000000} 8 2>&1 | grep "algo 8"
    done
  done
fi
