#!/bin/bash
# round-6 evidence steps on the GPU box, selected by name: tools/gpu_r06.sh STEP [STEP ...]   (outputs under gpurun_out/r06)
#   dino_tests     the GELU / DINO parity tests
#   tower_ab LIB.. same-box tower_bench of the default library and the given builds (scd_amd/lib/libscd_hip_NAME.so)
#   dino_prof      rocprofv3 kernel stats of a DINO-only and a CLIP-only encode (3,990-image launches)
set -u
R=$PWD; out=$R/gpurun_out/r06; mkdir -p $out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$R}
step=$1; shift
case $step in
dino_tests)
  timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "gelu or dino or gemm or tower or encoder" > $out/dino_tests.txt 2>&1; rc=$?
  echo "[dino_tests] rc=$rc"; tail -n 5 $out/dino_tests.txt; [ $rc -eq 0 ] || exit 1 ;;
tower_ab)
  for lib in default "$@"; do
    if [ $lib = default ]; then unset SCD_HIP_LIB; else export SCD_HIP_LIB=$R/scd_amd/lib/libscd_hip_$lib.so; fi
    timeout -k 10 300 python tools/tower_bench.py 6 3990 >> $out/tower_ab.jsonl 2> $out/tower_ab.err || { tail -n 20 $out/tower_ab.err; exit 1; }
    tail -n 1 $out/tower_ab.jsonl
  done ;;
dino_prof)
  cd /tmp && export TMPDIR=/tmp
  for t in dino clip; do
    o=$out/prof_$t; mkdir -p $o
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $o --output-format csv -- python3 $R/tools/tower_bench.py 4 3990 $t > $o/run.log 2>&1
    rc=$?; echo "[prof $t] rc=$rc"; if [ $rc -ne 0 ]; then tail -n 20 $o/run.log; exit 1; fi
    f=$(find $o -name "*kernel_stats.csv" | head -n 1); cp $f $out/r06_${t}_kernel_stats.csv
    python3 - $out/r06_${t}_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print("  %-100s calls %5s avg %9.1f us  %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
    rm -rf $o
  done ;;
*) echo "unknown step $step"; exit 2 ;;
esac
