#!/bin/bash
# round-4 evidence: rocprofv3 kernel stats of the bench command (SSKM default and --cluster KM), of a KM fit, PMC traffic of the
# dominant kernel (FETCH_SIZE / WRITE_SIZE in separate passes), matrix-pipe utilisation of the encoder kernels
set -u
R=$PWD; out=$R/gpurun_out/r04; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
stats() { # tag -- program args
  local tag=$1; shift
  timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $out/prof_$tag --output-format csv -- python3 "$@" > $out/prof_$tag.log 2>&1
  local rc=$?; echo "[rocprof $tag] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  local f=$(find $out/prof_$tag -name "*kernel_stats.csv" | head -n 1)
  [ -n "$f" ] && cp $f $out/r04_${tag}_kernel_stats.csv && head -n 6 $out/r04_${tag}_kernel_stats.csv | cut -c1-150
  rm -rf $out/prof_$tag
}
stats bench_default $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline && \
stats bench_km $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --cluster KM && \
stats km_fit $R/tools/km_fit_bench.py || exit 1
pass() { # ctr
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $1 -d $out/pmc_fc1_$1 --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --images 7980 --no-cpu-baseline > $out/pmc_fc1_$1.log 2>&1
  local rc=$?; echo "[pmc fc1 $1] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
}
pass FETCH_SIZE && pass WRITE_SIZE && python3 $R/tools/pmc_traffic.py $out/pmc_fc1_FETCH_SIZE $out/pmc_fc1_WRITE_SIZE 786432 3072 768 $out/r04_pmc_fc1.json 3990 | tail -n 12
rm -rf $out/pmc_fc1_FETCH_SIZE $out/pmc_fc1_WRITE_SIZE
CTR="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc $CTR -d $out/pm_enc --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --images 7980 --no-cpu-baseline > $out/pm_enc.log 2>&1
echo "[pmc mfma] rc=$?"
{ echo "# rocprofv3 --kernel-trace --pmc $CTR -- python3 bench.py --steps 1 --warmup 0 --images 7980 --no-cpu-baseline"; python3 $R/tools/pmc_mfma_util.py $out/pm_enc gemm_w4 attention sim_topk; } > $out/r04_pmc_mfma_encoder.txt
cat $out/r04_pmc_mfma_encoder.txt | cut -c1-140
rm -rf $out/pm_enc
