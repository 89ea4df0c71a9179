"""Summarise FETCH_SIZE / WRITE_SIZE passes of rocprofv3 for the dominant GEMM (ViT fc1) and the LayerNorm calibration kernel.

usage: python tools/pmc_traffic.py FETCH_DIR WRITE_DIR M N K OUT.json
Full-size launches are those within 10% of the largest counter value of the kernel (the last, partial batch is dropped).
FETCH_SIZE is doubled (gfx950: 128-byte requests tallied at 64 B, MI355X guide); the doubling is checked in the same run on
im2col_kernel, a pure copy: it reads batch*3*224*224*2 bytes of fp16 pixels and writes M*768*2 bytes of patch rows (round 2 / 3
calibrated on row_stats_kernel, which round 4 folded into the kernel that writes x).
"""
import csv, glob, json, sys

def load(d, counter):
    out = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                out.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return out

def full(vals):
    m = max(vals)
    sel = [v for v in vals if v > 0.9 * m]
    return len(sel), sum(sel) / len(sel)

fd, wd, M, N, K, outp = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
fetch, write = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
fc1 = [k for k in fetch if "gemm_w4_kernel" in k and "Li1ELb1ELb0ELi1" in k]   # QuickGELU, bias, no residual, LN-folded
ln = [k for k in fetch if "im2col_kernel" in k]      # a copy of known size, but its reads are 32-byte pieces of image rows (<= 64-B requests)
av = [k for k in fetch if "assemble_visual" in k]   # reads the patch rows as whole 512-B wave loads (128-B requests), writes x (round 6: assemble_visual_rows_kernel)
assert fc1 and (ln or av), (list(fetch)[:5])
nf, f_kb = full(fetch[fc1[0]]); nw, w_kb = full(write[fc1[0]])
# round 6: the patch GEMM gathers from the image and im2col_kernel no longer runs on fp16 pixels; its same-run check is then absent
nlf, lf_kb = full(fetch[ln[0]]) if ln else (0, float("nan"))
nlw, lw_kb = full(write[ln[0]]) if ln else (0, float("nan"))
BATCH = int(sys.argv[7]) if len(sys.argv) > 7 else M // 197          # images per launch
ln_bytes = BATCH * 3 * 224 * 224 * 2
ln_wbytes = (M // 197 * 196 + 255) // 256 * 256 * 768 * 2
cal128 = None
if av:
    naf, af_kb = full(fetch[av[0]]); naw, aw_kb = full(write[av[0]])
    rows_p = BATCH * 196
    cal128 = {"kernel": av[0].split("(")[0][:60], "raw_fetch_mb": af_kb * 1024 / 1e6, "expected_read_mb": rows_p * 768 * 2 / 1e6,
              "ratio_raw_over_expected": af_kb * 1024 / (rows_p * 768 * 2), "write_mb": aw_kb * 1024 / 1e6, "expected_write_mb": M * 768 * 2 / 1e6}
res = {
    "kernel": "gemm_w4_kernel<QuickGELU,bias,no-residual,LN-folded> (ViT fc1) m=%d n=%d k=%d" % (M, N, K),
    "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} --output-format csv -- python3 bench.py --steps 1 --warmup 0 --images 7980 --no-cpu-baseline (two separate passes)",
    "raw": {"FETCH_SIZE": {"full_launches": nf, "avg_kb": f_kb}, "WRITE_SIZE": {"full_launches": nw, "avg_kb": w_kb},
            "im2col_FETCH_SIZE": {"full_launches": nlf, "avg_kb": lf_kb}, "im2col_WRITE_SIZE": {"full_launches": nlw, "avg_kb": lw_kb}},
    "correction": "FETCH_SIZE x2: gfx950 tallies every request pattern this library uses at half the bytes read (tools/micro/fetch_calib.hip, profiles/r04_fetch_calib.txt: 1,024-byte wave loads, global_load_lds_dwordx4 fills with 128- and 64-byte row segments, 32-byte pieces: 0.500-0.504); same-run checks below: assemble_visual_kernel raw ratio ~0.5 (= 1.0 x its input after the correction); im2col_kernel raw fetch %.1f MB for a %.1f MB read (raw ratio %.3f, i.e. it fetches %.2f x its input: neighbouring patches re-request the same 128-byte lines); WRITE_SIZE needs no correction: %.1f MB for im2col's %.1f MB of patch rows.  The counter is on the L2's fabric side: Infinity-Cache hits are included"
                  % (lf_kb * 1024 / 1e6, ln_bytes / 1e6, lf_kb * 1024 / ln_bytes, 2 * lf_kb * 1024 / ln_bytes, lw_kb * 1024 / 1e6, ln_wbytes / 1e6),
    "calibration_128B_requests": cal128,
    "fetch_bytes_per_launch": 2 * f_kb * 1024,
    "write_bytes_per_launch": w_kb * 1024,
    "traffic_bytes_per_launch": 2 * f_kb * 1024 + w_kb * 1024,
    "algorithmic_bytes_per_launch": 2 * (M * K + N * K + M * N),
    "rows": M,
}
json.dump(res, open(outp, "w"), indent=1)
print(json.dumps(res, indent=1))
