#!/usr/bin/env python3
"""Regenerate uniter_amd/tuned/gfx950.json: the tile choices shipped for the reference's standard shapes.

Runs the in-situ tuner in N independent processes (each `python bench.py` with its own UNITER_AMD_TUNE_CACHE file and the
shipped table ignored), times the benchmark step with each result, then re-times the two best candidates and keeps the
faster one.  The table is nothing but that tuner's output; it removes the run-to-run spread of the timing-based descent
(about +-1 % of a step) from every later process.  Needs one MI355X; ~15 s per run."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "uniter_amd", "tuned", "gfx950.json")


def bench(cache, steps):
    env = dict(os.environ, UNITER_AMD_TUNE_CACHE=cache, UNITER_AMD_FACTORY_TUNE="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-kernel-timing", "--steps", str(steps)],
                         env=env, stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
    line = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")][-1]
    return json.loads(line)["ms_per_step"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    tmp = tempfile.mkdtemp(prefix="tune_")
    runs = []
    for i in range(n):
        cache = os.path.join(tmp, "c%d.json" % i)
        ms = bench(cache, 30)                       # tunes (cache absent), then times 30 steps
        runs.append((ms, cache))
        print("candidate %d: %.3f ms/step" % (i, ms), flush=True)
    runs.sort()
    finals = []
    for ms, cache in runs[:2]:                      # re-time the two best with their choices pinned
        again = min(bench(cache, 40), bench(cache, 40))
        finals.append((again, cache))
        print("re-timed %s: %.3f ms/step" % (os.path.basename(cache), again), flush=True)
    finals.sort()
    table = json.load(open(finals[0][1]))
    table["source"] = "scripts/make_factory_tune.py: best of %d in-situ tuning runs, %.3f ms/step on the box that made it" % (n, finals[0][0])
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    json.dump(table, open(OUT, "w"), indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
