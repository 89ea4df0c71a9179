"""Time x board power of one GEMM shape under the timing ablations of the -DSCD_ABLATE build (SCD_GEMM_X bits: 2 no stores, 4 every
tile loads the same L2-resident panels, 16 no epilogue).  The chip sits at its power cap during these launches, so time per launch x
watts = energy per launch, and the differences between the ablations attribute it.  One process per flag value (the switch is read once):
  SCD_HIP_LIB=scd_amd/lib/libscd_hip_ablate.so SCD_GEMM_X=4 python tools/gemm_energy.py M N K act res [seconds]"""
import glob, os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import ops


def hwmon_dir():
    pr = torch.cuda.get_device_properties(0)
    bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
    c = glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % bdf)
    return c[0] if c else None


def main():
    m, n, k, act, res = [int(v) for v in sys.argv[1:6]]
    secs = float(sys.argv[6]) if len(sys.argv) > 6 else 2.5
    a = (torch.randn(m, k, device="cuda") * 0.5).half()
    w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
    b = torch.randn(n, device="cuda")
    r = torch.randn(m, n, device="cuda").half() if res else None
    for _ in range(3):
        ops.gemm_f16(a, w, b, r, act)
    torch.cuda.synchronize()
    hd = hwmon_dir()
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            try:
                samples.append((float(open(hd + "/power1_input").read()) / 1e6, float(open(hd + "/freq1_input").read()) / 1e6))
            except Exception:
                pass
            stop.wait(0.1)
    th = threading.Thread(target=sampler, daemon=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # warm the clocks for a second, then measure
    t_end = time.time() + 1.0
    while time.time() < t_end:
        for _ in range(20):
            ops.gemm_f16(a, w, b, r, act)
        torch.cuda.synchronize()
    if hd:
        th.start()
    iters = 0
    e0.record()
    t_end = time.time() + secs
    while time.time() < t_end:
        for _ in range(20):
            ops.gemm_f16(a, w, b, r, act)
        iters += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    us = e0.elapsed_time(e1) * 1e3 / iters
    pw = sorted(s[0] for s in samples) or [float("nan")]
    fq = sorted(s[1] for s in samples) or [float("nan")]
    print("X=%-4s m=%d n=%d k=%d act=%d res=%d : %8.1f us  %7.1f TFLOP/s  %6.0f W  sclk %4.0f MHz  -> %6.3f J per launch"
          % (os.environ.get("SCD_GEMM_X", "0"), m, n, k, act, res, us, 2.0 * m * n * k / us / 1e6, pw[len(pw) // 2], fq[len(fq) // 2],
             us * 1e-6 * pw[len(pw) // 2]), flush=True)


if __name__ == "__main__":
    main()
