#!/usr/bin/env python3
"""Build libuniter_hip.so (gfx950) in-tree with hipcc.

    python uniter_amd/csrc/build.py [--force] [--verbose]

One object per .hip file (parallel), linked into uniter_amd/csrc/build/libuniter_hip.so.  Rebuilds only
objects whose sources (or shared headers) are newer.  hipcc cross-compiles without a GPU.
"""
import argparse
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
# (variant builds for A/B runs: UNITER_BUILD_DIR=build_b UNITER_EXTRA_FLAGS="-DUNITER_STORE_POLICY=2" python build.py; a process picks
#  a variant up through UNITER_AMD_LIB=<path to its libuniter_hip.so> / LD_LIBRARY_PATH for the native harness)
BUILD = os.path.join(HERE, os.environ.get("UNITER_BUILD_DIR", "build"))
LIB = os.path.join(BUILD, "libuniter_hip.so")
SOURCES = ["capi.hip", "gemm.hip", "attention.hip", "layernorm.hip", "embed.hip", "adamw.hip", "encoder.hip", "comm.hip", "ot.hip", "pool.hip", "lmhead.hip", "head.hip"]
HEADERS = [os.path.join(HERE, h) for h in ("common.cuh", "kernels.h", "gemm_lds.cuh", "gemm_args.cuh", "gemm8.cuh", "attention_fwd.cuh", "layernorm_fwd.cuh")] + \
    [os.path.join(ROOT, "include", "uniter_hip.h")]
ARCH = "gfx950"


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = find_hipcc()
    os.makedirs(BUILD, exist_ok=True)
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
             "-DNDEBUG"] + os.environ.get("UNITER_EXTRA_FLAGS", "").split()
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(BUILD, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + flags + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        return s, r

    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for s, r in ex.map(compile_one, jobs):
                if r.returncode != 0:
                    sys.stderr.write(r.stdout + r.stderr)
                    raise RuntimeError("hipcc failed on %s" % s)
                if verbose and r.stderr:
                    sys.stderr.write(r.stderr)
    if jobs or force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link failed")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    print(build(force=a.force, verbose=a.verbose))
