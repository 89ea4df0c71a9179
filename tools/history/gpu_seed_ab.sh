#!/bin/bash
# seeding A/B: the rounds from Python (SCD_KPP_SEED_RUN=0), in C with the tile kernel (SCD_KPP_FILTER=0), in C with the filter
set -u
out=gpurun_out/km; mkdir -p $out
for cfg in "95000 768 100" "126976 512 100" "160146 512 1000"; do
  for mode in "0 1" "1 0" "1 1"; do
    set -- $mode
    SCD_KPP_SEED_RUN=$1 SCD_KPP_FILTER=$2 timeout -k 10 200 python tools/sskm_phases.py $cfg > $out/seed_ab.txt 2>&1
    echo "[$cfg] SEED_RUN=$1 FILTER=$2: $(tail -n 2 $out/seed_ab.txt | head -n 1 | cut -c1-120)"
  done
done
