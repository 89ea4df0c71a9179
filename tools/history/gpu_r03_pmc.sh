#!/bin/bash
# round-3 PMC traffic passes (FETCH_SIZE / WRITE_SIZE in separate runs, kernel trace only besides) for the new kernels
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03; mkdir -p $out
pass() { # tag counter script args...
  local tag=$1 ctr=$2; shift 2
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr -d $out/pmc_${tag}_$ctr --output-format csv -- python3 "$@" > $out/pmc_${tag}_$ctr.log 2>&1
  local rc=$?; echo "[pmc $tag $ctr] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
}
summ() { # tag filter...
  local tag=$1; shift
  { echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two runs) -- python3 $TAGCMD"; python3 $R/tools/pmc_kernels.py $out/pmc_${tag}_FETCH_SIZE $out/pmc_${tag}_WRITE_SIZE "$@"; } > $out/r03_pmc_$tag.txt
  cat $out/r03_pmc_$tag.txt | cut -c1-130
  rm -rf $out/pmc_${tag}_FETCH_SIZE $out/pmc_${tag}_WRITE_SIZE $out/pmc_${tag}_*.log
}
TAGCMD="tools/sim_bench.py 126976 3"
pass sim FETCH_SIZE $R/tools/sim_bench.py 126976 3 && pass sim WRITE_SIZE $R/tools/sim_bench.py 126976 3 && summ sim sim_ wmax
TAGCMD="tools/kmeans_bench.py 512 0.8 160146 1000"
pass km_c4 FETCH_SIZE $R/tools/kmeans_bench.py 512 0.8 160146 1000 && pass km_c4 WRITE_SIZE $R/tools/kmeans_bench.py 512 0.8 160146 1000 && summ km_c4 estep mstep finalize
TAGCMD="tools/sskm_phases.py 126976 512 100"
pass sskm FETCH_SIZE $R/tools/sskm_phases.py 126976 512 100 && pass sskm WRITE_SIZE $R/tools/sskm_phases.py 126976 512 100 && summ sskm estep mstep finalize muf_ minupd kpp_ labels_sync inertia
