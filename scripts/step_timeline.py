#!/usr/bin/env python3
"""Timeline of ONE optimizer step from a `rocprofv3 --kernel-trace` run of bench.py: every launch of the last complete step
(steps are delimited by the adamw_kernel launches) with its start offset, duration, the idle gap in front of it (start minus the
latest end of anything launched before it, on any queue) and its queue — where the GPU waits for the host, where launches
overlap, where the chip idles between dependent kernels.

    python scripts/step_timeline.py <rocprofv3 output dir> [--step -2] [--out file]
"""
import argparse
import csv
import glob
import gzip
import os
import re


def short(name):
    n = name.replace('(anonymous namespace)::', '').replace('void ', '')
    n = re.sub(r'\(.*$', '', n)
    n = re.sub(r'at::native::', '', n)
    return n[:90]


def load(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv*"), recursive=True)
    if not files:
        raise SystemExit("no kernel trace under " + d)
    f = max(files, key=os.path.getmtime)
    op = gzip.open if f.endswith(".gz") else open
    with op(f, "rt") as fh:
        rows = list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--step", type=int, default=-2, help="which step (index into the list of adamw launches; default: second to last)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    rows = load(a.dir)
    adam = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
    if len(adam) < 3:
        raise SystemExit("fewer than 3 optimizer steps in the trace")
    hi = adam[a.step]
    lo = adam[a.step - 1] + 1
    step = rows[lo:hi + 1]
    t0 = int(step[0]["Start_Timestamp"])
    prev_end = int(rows[lo - 1]["End_Timestamp"])
    lines = []
    idle = 0.0
    busy_end = prev_end
    queues = {}
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - busy_end) * 1e-3
        if gap > 0:
            idle += gap
        q = queues.setdefault(r.get("Queue_Id", "?"), len(queues))
        lines.append("%9.1f  %8.2f  %s%7.2f  q%d  %s" % ((s - t0) * 1e-3, (e - s) * 1e-3, " " if gap >= 0 else "~", gap, q, short(r["Kernel_Name"])))
        busy_end = max(busy_end, e)
    span = (int(step[-1]["End_Timestamp"]) - prev_end) * 1e-3
    head = ["step of %d launches: span %.1f us (previous step's last kernel end -> this step's adamw end), idle %.1f us (sum of positive gaps)" % (len(step), span, idle),
            "   start us    dur us   gap us  queue  kernel        (~gap: starts before everything earlier has ended = overlap)"]
    txt = "\n".join(head + lines)
    print(txt)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
