#!/bin/bash
# cache-policy A/B of the GEMM's ring fills: tools/gpu_r04_nt.sh LIB_B   (same box: bench A/B, then a FETCH_SIZE pass per build)
set -u
R=$PWD; out=$R/gpurun_out/r04; mkdir -p $out
B=$1
bash tools/gpu_r04_ab.sh $B || exit 1
cd /tmp && export TMPDIR=/tmp
for lib in default $B; do
  if [ $lib = default ]; then unset SCD_HIP_LIB; else export SCD_HIP_LIB=$R/$lib; fi
  tag=$(basename $lib .so)
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pf_$tag --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --images 7980 --no-cpu-baseline > $out/pf_$tag.log 2>&1
  rc=$?; echo "[pmc $tag] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  python3 $R/tools/pmc_fetch_variants.py $out/pf_$tag | cut -c1-160
  rm -rf $out/pf_$tag
done
