#!/usr/bin/env python3
"""Build the native kernel test/benchmark binary: tests/native/build/test_kernels (links libuniter_hip.so)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "uniter_amd", "csrc"))
import build as libbuild  # noqa: E402


def build():
    lib = libbuild.build()
    out_dir = os.path.join(HERE, "build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "test_kernels")
    src = os.path.join(HERE, "test_kernels.cpp")
    if (not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src)
            or os.path.getmtime(exe) < os.path.getmtime(lib)):
        libdir = os.path.dirname(lib)
        cmd = [libbuild.find_hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-x", "hip", src, "-o", exe,
               "-L" + libdir, "-luniter_hip", "-Wl,-rpath,$ORIGIN/../../../uniter_amd/csrc/build"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("building test_kernels failed")
    return exe


if __name__ == "__main__":
    print(build())
