set -u
mkdir -p gpurun_out/r04
for b in 3990 1995 1330 665 332 7980; do
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $b > gpurun_out/r04/bs.json 2> gpurun_out/r04/bs.err || { tail -n 3 gpurun_out/r04/bs.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r04/bs.json"))
print("batch %5d  %9.1f images/s  encode %.1f ms  fc1 frac %.4f  %s W  sclk %s" % ($b, d["value"], d["stage_ms_per_step"]["encode"], d["roofline"]["frac"], d["board_power"]["median_w"], d["board_power"]["sclk_mhz_median"]))
PY
done
