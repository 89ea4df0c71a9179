#!/bin/bash
# round-3 evidence, part B: C4-size bench line, similarity / k-means micro-benchmarks with rocprofv3 kernel stats
set -u
R=$PWD; out=$R/gpurun_out/r03; mkdir -p $out
timeout -k 10 500 python bench.py --config c4 --no-cpu-baseline > $out/r03_bench_c4_1gpu.json 2> $out/bench_c4.err; echo "[bench c4] rc=$?"
python - <<PY
import json
d=json.load(open("$out/r03_bench_c4_1gpu.json"))
print(d["value"], d["stage_ms_per_step"])
for s in d["secondary_rooflines"]: print("  ", {k:v for k,v in s.items() if k!="kernel"}, s["kernel"][:40])
PY
for k in 3 5 1; do timeout -k 10 120 python tools/sim_bench.py 126976 $k hipblaslt; done > $out/r03_sim_bench.txt 2>&1
grep -E "sim_topk|torch.mm" $out/r03_sim_bench.txt
timeout -k 10 200 python tools/kmeans_bench.py 768 0.8 95000 100 > $out/r03_kmeans_bench.txt 2>&1; tail -n 8 $out/r03_kmeans_bench.txt
timeout -k 10 200 python tools/kmeans_bench.py 512 0.8 160146 1000 > $out/r03_kmeans_bench_c4.txt 2>&1; grep -E "^estep|lloyd" $out/r03_kmeans_bench_c4.txt
timeout -k 10 200 python tools/sskm_phases.py 95000 768 100 > $out/r03_sskm_phases.txt 2>&1; tail -n 2 $out/r03_sskm_phases.txt | cut -c1-400
cd /tmp && export TMPDIR=/tmp
prof() { local tag=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/p_$tag --output-format csv -- python3 "$@" > $out/p_$tag.log 2>&1; echo "[rocprof $tag] rc=$?"
  f=$(find $out/p_$tag -name "*kernel_stats.csv" | head -n 1); [ -n "$f" ] && cp $f $out/r03_${tag}_kernel_stats.csv; rm -rf $out/p_$tag $out/p_$tag.log; }
prof sim_bench $R/tools/sim_bench.py 126976 3
prof kmeans_bench $R/tools/kmeans_bench.py 768 0.8 95000 100
prof kmeans_bench_c4 $R/tools/kmeans_bench.py 512 0.8 160146 1000
prof sskm_phases $R/tools/sskm_phases.py 95000 768 100
# the Lloyd loop's kernel timeline (last 70 kernels of a fit) and the seeding A/B
cd $R
bash tools/gpu_km_trace.sh 95000 768 100 70 > /dev/null 2>&1; cp gpurun_out/km_trace/timeline.txt $out/r03_lloyd_timeline.txt
{ echo "# tools/trace_gaps.py on a rocprofv3 --kernel-trace of tools/lloyd_trace.py 95000 768 100: start (us), duration, gap to the previous kernel's end"; cat gpurun_out/km_trace/run.log | grep "^fit"; } > $out/hdr.tmp; cat $out/hdr.tmp $out/r03_lloyd_timeline.txt > $out/t.tmp; mv $out/t.tmp $out/r03_lloyd_timeline.txt; rm -f $out/hdr.tmp
{ echo "# tools/gpu_seed_ab.sh: SSKM fit (tools/sskm_phases.py n d k), seeding rounds from Python (SEED_RUN=0), in C on the float32 tile kernel (FILTER=0), in C through the fp16 filter"; bash tools/gpu_seed_ab.sh; } > $out/r03_seed_ab.txt 2>&1
cat $out/r03_seed_ab.txt | cut -c1-150
