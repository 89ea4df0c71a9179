#!/bin/bash
# PMC passes over tools/sim_bench.py for the similarity kernels (program directly after `--`; env vars set outside).
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/sim_pmc; mkdir -p $out
export SCD_SIM_RB=${SCD_SIM_RB:-16}
for x in 0 1 3; do
  export SCD_SIM_X=$x
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU \
     -d $out/x$x --output-format csv -- python3 $R/tools/sim_bench.py 126976 3 > $out/x$x.log 2>&1
  rc=$?; echo "[pmc x=$x] rc=$rc"; tail -n 3 $out/x$x.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  python3 $R/tools/pmc_summary.py $out/x$x sim_topk_rb > $out/summary_x$x.txt; cat $out/summary_x$x.txt
  python3 $R/tools/pmc_clock.py $out/x$x 2>/dev/null | grep sim_topk_rb | tail -n 6
  find $out/x$x -name "*.db" -delete; find $out/x$x -name "*agent_info*" -delete
done
