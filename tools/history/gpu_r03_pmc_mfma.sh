#!/bin/bash
# round-3 matrix-pipe utilisation (PMC, one pass per tool; kernel trace only besides)
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03; mkdir -p $out
CTR="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
one() { # tag filter -- script args
  local tag=$1 filt=$2; shift 2
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $CTR -d $out/pm_$tag --output-format csv -- python3 "$@" > $out/pm_$tag.log 2>&1
  local rc=$?; echo "[pmc $tag] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  { echo "# rocprofv3 --kernel-trace --pmc $CTR -- python3 ${*#$R/}"; python3 $R/tools/pmc_mfma_util.py $out/pm_$tag $filt; } > $out/r03_pmc_mfma_$tag.txt
  cat $out/r03_pmc_mfma_$tag.txt | cut -c1-140
  rm -rf $out/pm_$tag $out/pm_$tag.log
}
one sim "sim_topk" $R/tools/sim_bench.py 126976 3 && \
one km_c4 "estep_rb" $R/tools/kmeans_bench.py 512 0.8 160146 1000 && \
one encoder "gemm_w4 attention" $R/bench.py --steps 1 --warmup 0 --images 7980 --no-cpu-baseline
