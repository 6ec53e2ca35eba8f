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


CONFIG = "c2"


def bench(cache, steps):
    env = dict(os.environ, UNITER_AMD_TUNE_CACHE=cache, UNITER_AMD_FACTORY_TUNE="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-kernel-timing", "--config", CONFIG,
                          "--steps", str(steps)],
                         env=env, stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    if not lines:
        sys.stderr.write(out.stderr.decode()[-3000:])
        raise RuntimeError("bench.py printed no result line (rc %d)" % out.returncode)
    return json.loads(lines[-1])["ms_per_step"]


def main():
    """usage: make_factory_tune.py [runs] [config]   — entries of other shapes already in the table are kept"""
    global CONFIG
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    CONFIG = sys.argv[2] if len(sys.argv) > 2 else "c2"
    steps = 30 if CONFIG == "c2" else 8
    tmp = tempfile.mkdtemp(prefix="tune_")
    runs = []
    for i in range(n):
        cache = os.path.join(tmp, "c%d.json" % i)
        ms = bench(cache, steps)                    # tunes (cache absent), then times the steps
        runs.append((ms, cache))
        print("candidate %d: %.3f ms/step" % (i, ms), flush=True)
    runs.sort()
    finals = []
    for ms, cache in runs[:2]:                      # re-time the two best with their choices pinned
        again = min(bench(cache, steps + 10), bench(cache, steps + 10))
        finals.append((again, cache))
        print("re-timed %s: %.3f ms/step" % (os.path.basename(cache), again), flush=True)
    finals.sort()
    table = json.load(open(finals[0][1]))
    note = "%s: best of %d in-situ tuning runs, %.3f ms/step on the box that made it" % (CONFIG, n, finals[0][0])
    if os.path.exists(OUT):                         # keep the entries of the other workloads' shapes
        old = json.load(open(OUT))
        if old.get("n_tiles") == table.get("n_tiles"):
            have = {(e["kind"], e["M"], e["N"], e["K"]) for e in table["gemm"]}
            table["gemm"] += [e for e in old.get("gemm", []) if (e["kind"], e["M"], e["N"], e["K"]) not in have]
            prev = old.get("source", "")
            prev = [x for x in prev.split(" | ") if x and not x.startswith(CONFIG + ":") and not x.startswith("scripts/")]
            note = " | ".join(prev + [note])
    table["source"] = note
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    json.dump(table, open(OUT, "w"), indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
