#!/bin/bash
# A/B builds of fcm.hip with -D flags into tools/_build/libf_<name>.so, timed with tools/time_fcm.py through UAMMD_HIP_LIB:
#   tools/variants_fcm.sh build   (here)      tools/variants_fcm.sh run   (on the GPU box)
cd "$(dirname "$0")/.."
declare -A FLAGS=([base]="" [w6144p4]="-DUAMMD_SP_WORDS=6144 -DUAMMD_SP_PER_THREAD=4" [p4]="-DUAMMD_SP_PER_THREAD=4" [p2]="-DUAMMD_SP_PER_THREAD=2" [p1]="-DUAMMD_SP_PER_THREAD=1" [ab1]="-DUAMMD_SP_ABLATE=1" [ab2]="-DUAMMD_SP_ABLATE=2" [ab4]="-DUAMMD_SP_ABLATE=4" [ab8]="-DUAMMD_SP_ABLATE=8" [ab3]="-DUAMMD_SP_ABLATE=3" [ab6]="-DUAMMD_SP_ABLATE=6" [ab7]="-DUAMMD_SP_ABLATE=7" [ab15]="-DUAMMD_SP_ABLATE=15" [w6144]="-DUAMMD_SP_WORDS=6144" [timeline]="-DUAMMD_SPREAD_TIMELINE" [pl1]="-DUAMMD_PREP_LANES=1" [pl2]="-DUAMMD_PREP_LANES=2" [pl4]="-DUAMMD_PREP_LANES=4" [pl8]="-DUAMMD_PREP_LANES=8" [zp256]="-DUAMMD_ZP_THREADS=256" [zp512]="-DUAMMD_ZP_THREADS=512" [branchy]="-DUAMMD_SP_BRANCHY" [plane64]="-DUAMMD_PLANE_ATTR=__attribute__((amdgpu_waves_per_eu(8,8)))")
NAMES=(${VNAMES:-base w6144p4 p4 p2 w6144})
if [ "$1" = build ]; then
  mkdir -p tools/_build
  for n in "${NAMES[@]}"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -w ${FLAGS[$n]} -x hip -c uammd_amd/csrc/fcm.hip -o tools/_build/fcm_$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls uammd_amd/lib/obj/*.o | grep -v "/fcm.o") tools/_build/fcm_$n.o -o tools/_build/libf_$n.so -L/opt/rocm/lib -lrocfft -ldl
    rm -f tools/_build/fcm_$n.o
  done
  ls tools/_build/libf_*.so
else
  for rep in 1 2; do
    for n in "${NAMES[@]}"; do
      echo -n "$n: "; UAMMD_HIP_LIB=$PWD/tools/_build/libf_$n.so python tools/${VTOOL:-time_fcm.py} | tail -1
      echo -n "$n (108^3): "; NC=108 UAMMD_HIP_LIB=$PWD/tools/_build/libf_$n.so python tools/time_fcm.py | tail -1
    done
  done
fi
