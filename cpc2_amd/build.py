"""Builds libcpc2_hip.so (gfx950) in-tree with hipcc.  `python -m cpc2_amd.build`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcpc2_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = ["gemm_f32.hip", "gemm_planes.hip", "rowops.hip", "encoder.hip", "gru.hip", "lstm.hip", "infonce.hip", "transformer.hip", "negidx.cpp", "flac.cpp"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-pthread"]


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=True):
    objdir = os.path.join(HERE, "..", "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "rowcfg.h"), os.path.join(CSRC, "coop.h"), os.path.join(CSRC, "ldsdma.h"), os.path.join(HERE, "..", "include", "cpc2_hip.h")]
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    if not force and _newer(LIB, srcs + headers):
        return LIB

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if not force and _newer(obj, [src] + headers):
            return obj
        cmd = [HIPCC] + FLAGS + (["--offload-arch=gfx950"] if src.endswith(".hip") else []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=4) as pool:
        objs = list(pool.map(compile_one, srcs))
    cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
