#!/bin/bash
# token assembly: images of one token position per wave (SCD_ASSEMBLE_ROWS = 4 / 8 / 16; 1 = the one-row-per-wave kernel) - kernel time
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rb in 1 2 3 4 6 8 16; do
  o=$R/gpurun_out/r06/asm_$rb; mkdir -p $o
  # (2, 3, 6 and 16 need their template instantiations added to scd_vit_encode_image's ASM_GO switch: the shipped library has 4 and 8)
  SCD_ASSEMBLE_ROWS=$rb timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $o --output-format csv -- python3 $R/tools/tower_bench.py 3 3990 clip > $o/run.log 2>&1
  f=$(find $o -name "*kernel_stats.csv" | head -n 1)
  grep -i "assemble" $f | awk -F, -v rb=$rb '{printf "rows per wave %2d: %s calls avg %.1f us\n", rb, $2, $4/1e3}' | cut -c1-120
  rm -rf $o
done
