"""Build libscd_hip.so (gfx950) in-tree with hipcc.  `python -m scd_amd.build [--force]`.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container;
the resulting .so travels to the GPU box with the repo snapshot (it is git-ignored).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libscd_hip.so")
OBJDIR = os.path.join(LIBDIR, "obj")

SOURCES = ["api.cpp", "munkres.cpp", "munkres_sparse.cpp", "transport.cpp", "comm.cpp", "kmeans.hip", "mstep.hip", "sim.hip", "vote.hip", "gemm.hip", "encoder.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result",
         "-fno-gpu-rdc"]
# per-file additions.  sim.hip: no SLP vectorisation - hipcc pairs scalar float ops of the hand-placed epilogue slots into
# v_pk_*_f32 (an anti-lever beside MFMAs, MI355X guide) and spills the packed operands it builds for them inside the ring loop
EXTRA = {"sim.hip": ["-fno-slp-vectorize"]}


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA.items())).encode())
    return h.hexdigest()


def _deps():
    hdrs = [os.path.join(dp, f) for dp, _, fs in os.walk(CSRC) for f in fs if f.endswith(".h")]      # csrc/*.h and csrc/ablate/*.h
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "scd_hip.h"))
    return hdrs


def build(force=False, verbose=True, ablate=False):
    """ablate=True: lib/libscd_hip_ablate.so with -DSCD_ABLATE (timing ablations and the A/B kernels of earlier rounds; load it
    through SCD_HIP_LIB).  The default library contains neither."""
    objdir = OBJDIR + ("_ablate" if ablate else "")
    lib = os.path.join(LIBDIR, "libscd_hip_ablate.so") if ablate else LIB
    flags = FLAGS + (["-DSCD_ABLATE"] if ablate else [])
    os.makedirs(objdir, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    stamp = os.path.join(LIBDIR, "build_ablate.sha256" if ablate else "build.sha256")
    dig = _digest(srcs + _deps())
    if not force and os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return lib
    hdr_dig = _digest(_deps())

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        tag = obj + ".sha256"
        d = _digest([src]) + hdr_dig
        if not force and os.path.exists(obj) and os.path.exists(tag) and open(tag).read() == d:
            return obj
        cmd = [HIPCC] + flags + EXTRA.get(os.path.basename(src), []) + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", src, "-o", obj]
        if verbose:
            print("[scd_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        with open(tag, "w") as f:
            f.write(d)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"]
    if verbose:
        print("[scd_amd.build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    with open(stamp, "w") as f:
        f.write(dig)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, ablate="--ablate" in sys.argv))
