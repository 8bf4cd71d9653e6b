"""Builds eemflow_amd/libeemflow_hip.so (HIP kernels + C ABI) for gfx950 with hipcc.

In-tree on purpose: the .so travels with the repo snapshot to the GPU box; nothing is JIT-compiled
at import time.  `python -m eemflow_amd.build` or __graft_entry__.build().
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
# EEM_BUILD_TAG=<tag> (with EEM_EXTRA_FLAGS, e.g. -DEEM_DIAG): a second library beside the release one - objects under csrc/build_<tag>/,
# libeemflow_hip_<tag>.so - which tools load through EEM_LIB_PATH
TAG = os.environ.get("EEM_BUILD_TAG", "")
LIB = os.path.join(PKG, f"libeemflow_hip_{TAG}.so" if TAG else "libeemflow_hip.so")
SOURCES = ["api.hip", "conv_enc.hip", "conv_enc1.hip", "conv_enc2.hip", "conv_s2.hip", "conv_s2r.hip", "conv_bx3.hip", "conv_wino.hip", "conv_wino32.hip", "conv_wino4.hip", "conv_enc12.hip", "conv_wnc.hip", "tail.hip", "tail_fused.hip", "voxel.hip", "metrics.hip", "gconv.hip", "gconv16.hip", "gconvb.hip", "conv_stem7.hip", "eraft_kernels.hip",
           "eraft_api.hip", "train.hip", "wgrad_enc.hip", "wgrad_ring.hip", "wgrad_tail.hip", "dgrad_s2.hip", "train_api.hip", "plus_kernels.hip", "plus_api.hip", "bwd_ops.hip", "ops.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-fvisibility=hidden", "-fvisibility-inlines-hidden"]   # hidden: only include/eemflow_hip.h's entry points are dynamic symbols
EXTRA = {"voxel.hip": ["-ffp-contract=off"],
         "conv_wino.hip": ["-fno-slp-vectorize"], "conv_wino32.hip": ["-fno-slp-vectorize"], "conv_wino4.hip": ["-fno-slp-vectorize"], "conv_enc12.hip": ["-fno-slp-vectorize"], "conv_wnc.hip": ["-fno-slp-vectorize"],     # packed f32 VALU beside MFMAs is slower than scalar (guide: anti-lever)      # bit-exact f64 time scaling
         "plus_kernels.hip": ["-ffp-contract=off"],
         "bwd_ops.hip": ["-ffp-contract=off"]}      # same coordinate arithmetic (and mask) as the forward warp   # the warp mask depends on the last bit of the weight sum


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X library cannot be built")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True):
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, f"build_{TAG}" if TAG else "build")
    os.makedirs(objdir, exist_ok=True)
    # every header of csrc/ (a stale object behind a changed struct layout links and misbehaves silently), the public header, this file
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(PKG, "..", "include", "eemflow_hip.h"),
               os.path.abspath(__file__)]
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, *FLAGS, *EXTRA.get(src, []), *os.environ.get("EEM_EXTRA_FLAGS", "").split(), "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
