#!/bin/bash
# A/B build of the library with extra -D flags on ONE source: tools/build_variant.sh NAME SRC.hip -DFOO=1 ...  -> scd_amd/lib/libscd_hip_NAME.so
# (the other objects come from the default build; load the result through SCD_HIP_LIB, e.g. tools/history/gpu_r04_ab.sh)
set -eu
name=$1; src=$2; shift 2
L=scd_amd/lib; python -m scd_amd.build > /dev/null
extra=""; [ "$src" = sim.hip ] && extra="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -fno-gpu-rdc $extra "$@" -c scd_amd/csrc/$src -o /tmp/variant_$name.o
objs=$(ls $L/obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libscd_hip_$name.so $objs /tmp/variant_$name.o -ldl
echo $L/libscd_hip_$name.so
