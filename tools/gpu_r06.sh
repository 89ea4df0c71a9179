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
c3_ab)     # bench.py --config c3 on the default library and the given builds, then once more with the ConSSKM phase profile
  for lib in default "$@"; do
    if [ $lib = default ]; then unset SCD_HIP_LIB; else export SCD_HIP_LIB=$R/scd_amd/lib/libscd_hip_$lib.so; fi
    timeout -k 10 400 python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline > $out/c3_$lib.json 2> $out/c3.err || { tail -n 20 $out/c3.err; exit 1; }
    python - <<PY
import json
d=json.load(open("$out/c3_$lib.json"))
print("lib %-10s %9.1f images/s  stages %s  fit %s ms" % ("$lib", d["value"], d["stage_ms_per_step"], d["consskm"]["fit_ms_per_step"]))
PY
  done
  unset SCD_HIP_LIB
  SCD_CONSSKM_PROFILE=1 timeout -k 10 400 python bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline > $out/c3_phases.json 2> $out/c3.err || { tail -n 20 $out/c3.err; exit 1; }
  python -c "import json; d=json.load(open('$out/c3_phases.json')); print('phases of the last fit', d['consskm']['phase_ms_last_fit'], 'fit', d['consskm']['fit_ms_per_step'])" ;;
c3_tests)
  timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "transport or consskm or constrained or c3 or ptsup or dist" > $out/c3_tests.txt 2>&1; rc=$?
  echo "[c3_tests] rc=$rc"; tail -n 3 $out/c3_tests.txt; [ $rc -eq 0 ] || { tail -n 40 $out/c3_tests.txt; exit 1; } ;;
bench_prof)   # kernel stats of the TIMED region of `bench.py ARGS` (marker kernels): tools/gpu_r06.sh bench_prof TAG ARGS...
  tag=$1; shift
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 500 rocprofv3 --kernel-trace -d $out/prof_$tag --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $out/prof_$tag.log 2>&1
  rc=$?; echo "[rocprof bench $tag] rc=$rc"; if [ $rc -ne 0 ]; then tail -n 20 $out/prof_$tag.log; exit 1; fi
  python3 $R/tools/trace_window_stats.py $out/prof_$tag $out/r06_bench_${tag}_kernel_stats.csv | cut -c1-170 | head -n 24
  rm -rf $out/prof_$tag ;;
pytest)    # tools/gpu_r06.sh pytest TAG -k EXPR
  tag=$1; shift
  timeout -k 10 1100 python -m pytest tests -x -q -m gpu "$@" > $out/pytest_$tag.txt 2>&1; rc=$?
  echo "[pytest $tag] rc=$rc"; tail -n 3 $out/pytest_$tag.txt; [ $rc -eq 0 ] || { tail -n 60 $out/pytest_$tag.txt; exit 1; } ;;
bench)     # tools/gpu_r06.sh bench TAG ARGS...  -> $out/bench_TAG.json
  tag=$1; shift
  timeout -k 10 900 python bench.py "$@" > $out/bench_$tag.json 2> $out/bench_$tag.err || { tail -n 30 $out/bench_$tag.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$out/bench_$tag.json"))
print("[$tag] %.1f images/s, %.2f ms/step, stages %s, sclk %s, cycles/img %s" % (d["value"], d["ms_per_step"], d["stage_ms_per_step"], d.get("sclk_mhz_median"), d.get("encode_cycles_per_image")))
print("   roofline", {k: v for k, v in d["roofline"].items() if k not in ("kernel", "note", "traffic_source")})
for s in d.get("secondary_rooflines", []): print("   ", {k: v for k, v in s.items() if k not in ("kernel", "note")}, s["kernel"][:50])
PY
  ;;
*) echo "unknown step $step"; exit 2 ;;
esac
