#!/bin/bash
# Board power / clocks / temperature sampled by rocm-smi (read-only) while bench.py runs -> gpurun_out/r05/r05_power_trace.txt
set -u
R=$PWD; out=$R/gpurun_out/r05; mkdir -p $out
rm -f $out/power_samples.txt
timeout -k 10 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $out/pw_bench.json 2> $out/pw.err &
BP=$!
for i in $(seq 1 110); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  { echo "== t=$(date +%s.%N)"; timeout 5 rocm-smi --showpower --showmaxpower --showclocks --showtemp --showuse 2>&1 | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|memory)|GPU use" ; } >> $out/power_samples.txt
  sleep 0.4
done
wait $BP; echo "[bench] rc=$?"
python3 - <<PY > $out/r05_power_trace.txt
import re, json
txt = open("$out/power_samples.txt").read().split("== t=")[1:]
rows = []
for blk in txt:
    t = float(blk.split()[0])
    def g(pat):
        m = re.search(pat, blk)
        return m.group(1) if m else "-"
    rows.append((t, g(r"Current Socket[^:]*:\s*([0-9.]+)"), g(r"sclk[^(]*\(([0-9]+)Mhz\)"), g(r"mclk[^(]*\(([0-9]+)Mhz\)"), g(r"junction\)[^:]*:\s*([0-9.]+)"), g(r"memory\)[^:]*:\s*([0-9.]+)"), g(r"GPU use[^:]*:\s*([0-9]+)")))
d = json.load(open("$out/pw_bench.json"))
print("# rocm-smi --showpower --showclocks --showtemp --showuse every ~0.5 s while 'python bench.py --steps 4 --warmup 1 --no-cpu-baseline' runs")
print("# bench line: %.1f images/s, encode %.1f ms/step, fc1 frac %.4f" % (d["value"], d["stage_ms_per_step"]["encode"], d["roofline"]["frac"]))
m = re.search(r"Max Graphics Package Power[^:]*:\s*([0-9.]+)", open("$out/power_samples.txt").read())
print("# power cap (rocm-smi --showmaxpower): %s W" % (m.group(1) if m else "n/a"))
print("# t (s)   power (W)  sclk (MHz)  mclk (MHz)  junction (C)  memory (C)  GPU use (%)")
t0 = rows[0][0]
for r in rows: print("%7.1f  %9s  %9s  %9s  %11s  %10s  %10s" % ((r[0] - t0,) + r[1:]))
PY
head -n 5 $out/r05_power_trace.txt; awk 'NR>3' $out/r05_power_trace.txt | sort -k2 -n -r | head -n 12
