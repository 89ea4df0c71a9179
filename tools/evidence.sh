#!/bin/bash
# Evidence steps on the GPU box, one script for every round (round 6 folded the per-round tools/gpu_r0N_*.sh one-offs into it; those are
# kept under tools/history/ because committed profiles cite them):   tools/evidence.sh STEP [ARGS]     outputs under gpurun_out/$ROUND
#   pytest TAG [pytest args]     GPU suite (or a -k subset) -> pytest_TAG.txt
#   bench TAG [bench.py args]    one bench line -> bench_TAG.json, summary printed
#   bench_prof TAG [bench args]  rocprofv3 kernel stats of the TIMED region of the bench command (marker kernels) -> ${ROUND}_bench_TAG_kernel_stats.csv
#   prof TAG script.py [args]    rocprofv3 kernel stats of any python tool -> prof_TAG/kernel_stats.csv
#   ab LIB...                    same-box whole-bench comparison of the default library and builds scd_amd/lib/libscd_hip_LIB.so (two repetitions)
#   pmc_fc1                      HBM-side traffic of the dominant kernel: FETCH_SIZE and WRITE_SIZE in separate --pmc passes -> ${ROUND}_pmc_fc1.json
#   pmc_mfma                     matrix-pipe utilisation of the encoder kernels -> ${ROUND}_pmc_mfma_encoder.txt
#   tower_ab LIB... / dino_prof / dino_tests / c3_ab LIB... / c3_tests      round-6 steps (DINO tower, ConSSKM)
# Rules of the pool: one GPU process at a time, rocprofv3 takes the program itself after `--`, --pmc never together with other trace domains.
set -u
ROUND=${ROUND:-r06}
R=$PWD; out=$R/gpurun_out/$ROUND; mkdir -p $out
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
    f=$(find $o -name "*kernel_stats.csv" | head -n 1); cp $f $out/${ROUND}_${t}_kernel_stats.csv
    python3 - $out/${ROUND}_${t}_kernel_stats.csv <<'PY'
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
bench_prof)   # kernel stats of the TIMED region of `bench.py ARGS` (marker kernels): tools/evidence.sh bench_prof TAG ARGS...
  tag=$1; shift
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 500 rocprofv3 --kernel-trace -d $out/prof_$tag --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $out/prof_$tag.log 2>&1
  rc=$?; echo "[rocprof bench $tag] rc=$rc"; if [ $rc -ne 0 ]; then tail -n 20 $out/prof_$tag.log; exit 1; fi
  python3 $R/tools/trace_window_stats.py $out/prof_$tag $out/${ROUND}_bench_${tag}_kernel_stats.csv | cut -c1-170 | head -n 24
  rm -rf $out/prof_$tag ;;
pytest)    # tools/evidence.sh pytest TAG -k EXPR
  tag=$1; shift
  timeout -k 10 1100 python -m pytest tests -x -q -m gpu "$@" > $out/pytest_$tag.txt 2>&1; rc=$?
  echo "[pytest $tag] rc=$rc"; tail -n 3 $out/pytest_$tag.txt; [ $rc -eq 0 ] || { tail -n 60 $out/pytest_$tag.txt; exit 1; } ;;
bench)     # tools/evidence.sh bench TAG ARGS...  -> $out/bench_TAG.json
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
prof)
  tag=$1; shift
  cd /tmp && export TMPDIR=/tmp
  o=$out/prof_$tag; mkdir -p $o
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $o --output-format csv -- python3 $R/"$@" > $o/run.log 2>&1
  echo "[prof $tag] rc=$?"; tail -n 4 $o/run.log
  f=$(find $o -name "*kernel_stats.csv" | head -n 1)
  [ -n "$f" ] && cp $f $o/kernel_stats.csv && python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print("  %-84s calls %5s avg %9.1f us  min %9.1f  max %9.1f  %5.1f%%" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
PY
  find $o -name "*.db" -delete; find $o -name "*kernel_trace.csv" -delete; find $o -name "*agent_info*" -delete ;;
ab)
  for rep in 1 2; do
  for lib in default "$@"; do
    if [ $lib = default ]; then unset SCD_HIP_LIB; else export SCD_HIP_LIB=$R/scd_amd/lib/libscd_hip_$lib.so; fi
    timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/ab.json 2> $out/ab.err || { tail -n 20 $out/ab.err; exit 1; }
    python - <<PY | tee -a $out/ab.txt
import json
d=json.load(open("$out/ab.json"))
print("rep $rep lib %-24s %9.1f images/s  encode %.1f ms  fc1 frac %.4f  sclk %s MHz  %s cycles/image" % ("$lib", d["value"], d["stage_ms_per_step"]["encode"], d["roofline"]["frac"], d.get("sclk_mhz_median"), d.get("encode_cycles_per_image")))
PY
  done
  done ;;
pmc_fc1)
  cd /tmp && export TMPDIR=/tmp
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctr -d $out/pmc_fc1_$ctr --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --images 7980 --no-cpu-baseline > $out/pmc_fc1_$ctr.log 2>&1
    rc=$?; echo "[pmc fc1 $ctr] rc=$rc"; [ $rc -eq 0 ] || { tail -n 20 $out/pmc_fc1_$ctr.log; exit 1; }
  done
  python3 $R/tools/pmc_traffic.py $out/pmc_fc1_FETCH_SIZE $out/pmc_fc1_WRITE_SIZE 786432 3072 768 $out/${ROUND}_pmc_fc1.json 3990 | tail -n 12
  rm -rf $out/pmc_fc1_FETCH_SIZE $out/pmc_fc1_WRITE_SIZE ;;
pmc_mfma)
  cd /tmp && export TMPDIR=/tmp
  CTR="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $CTR -d $out/pm_enc --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --images 7980 --no-cpu-baseline > $out/pm_enc.log 2>&1
  rc=$?; echo "[pmc mfma] rc=$rc"; [ $rc -eq 0 ] || { tail -n 20 $out/pm_enc.log; exit 1; }
  { echo "# rocprofv3 --kernel-trace --pmc $CTR -- python3 bench.py --steps 1 --warmup 0 --images 7980 --no-cpu-baseline"; python3 $R/tools/pmc_mfma_util.py $out/pm_enc gemm_w4 attention sim_topk; } > $out/${ROUND}_pmc_mfma_encoder.txt
  cut -c1-140 $out/${ROUND}_pmc_mfma_encoder.txt; rm -rf $out/pm_enc ;;
front_end)   # round 6's encoder front end against the round-5 one: patch GEMM fed from the image / im2col, eight rows per wave / one row per
             # wave in the token assembly - bit-identity of both towers' features, then a same-box tower A/B of the four combinations
  python tools/patch_img_check.py save $out/feat_new.pt > /dev/null 2> $out/front_end.err || { tail -n 20 $out/front_end.err; exit 1; }
  SCD_PATCH_FROM_IMAGE=0 python tools/patch_img_check.py save $out/feat_im2col.pt > /dev/null 2>> $out/front_end.err || { tail -n 20 $out/front_end.err; exit 1; }
  SCD_ASSEMBLE_ROWS=1 python tools/patch_img_check.py save $out/feat_rows1.pt > /dev/null 2>> $out/front_end.err || { tail -n 20 $out/front_end.err; exit 1; }
  { python tools/patch_img_check.py cmp $out/feat_new.pt $out/feat_im2col.pt && python tools/patch_img_check.py cmp $out/feat_new.pt $out/feat_rows1.pt; } | tee $out/front_end_check.txt || exit 1
  rm -f $out/feat_new.pt $out/feat_im2col.pt $out/feat_rows1.pt
  for rep in 1 2; do
    for v in "default" "SCD_PATCH_FROM_IMAGE=0" "SCD_ASSEMBLE_ROWS=1" "SCD_PATCH_FROM_IMAGE=0 SCD_ASSEMBLE_ROWS=1"; do
      if [ "$v" = default ]; then python tools/tower_bench.py 6 3990 both > $out/fe.json; else env $v python tools/tower_bench.py 6 3990 both > $out/fe.json; fi
      python - "$v" $out/fe.json <<'PY' | tee -a $out/front_end_ab.txt
import json, sys
d = json.load(open(sys.argv[2]))
print("rep %-44s clip %8.1f %8.1f  dino %8.1f %8.1f images/s" % (sys.argv[1], d["clip_rep0"]["images_per_s"], d["clip_rep1"]["images_per_s"], d["dino_rep0"]["images_per_s"], d["dino_rep1"]["images_per_s"]))
PY
    done
  done ;;
*) echo "unknown step $step"; exit 2 ;;
esac
