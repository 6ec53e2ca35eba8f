#!/usr/bin/env python3
"""Turn the rocprofv3 output of scripts/profile_round.sh (gpurun_out/<tag>/) into the committed summaries:

  profiles/<tag>_kernel_stats.csv      rocprofv3 --stats per-kernel table (names shortened, all rows)
  profiles/<tag>_step_breakdown.txt    kernels of the last two optimizer steps grouped by family: time, share, launches
  profiles/<tag>_pmc_traffic.json      FETCH_SIZE / WRITE_SIZE per dispatch of the GEMM kernels (separate PMC passes),
                                       gfx950 correction applied (FETCH_SIZE x2 for wide coalesced reads,
                                       MI355X_MICROARCH.md "HBM"); bench.py reads it for roofline.traffic

usage: python scripts/summarize_profile.py <tag>
"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEMM_RE = re.compile(r'gemm_kernel<(\d+), (\d+), (\w+), (\w+), (\d), (\d), (\w+)>')
EPI_KIND = {0: "gemm fwd +bias", 1: "gemm fwd +bias+gelu", 2: "gemm fwd +bias+dropout+residual", 3: "gemm dgrad",
            4: "gemm dgrad x gelu'", 5: "gemm wgrad"}


def short(name):
    name = name.replace('(anonymous namespace)::', '')
    name = re.sub(r'\(.*$', '', name)
    return name.replace('void ', '')[:110]


def one(pattern):
    files = glob.glob(pattern)
    return files[0] if files else None


def kernel_stats(src, dst):
    rows = list(csv.DictReader(open(src)))
    with open(dst, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], "%.1f" % float(r["AverageNs"]),
                        r["Percentage"], r["MinNs"], r["MaxNs"]])
    return len(rows)


def family(name):
    n = name.replace('(anonymous namespace)::', '')
    m = GEMM_RE.search(n)
    if m:
        return "%s  [tile %sx%s st%s %s]" % (EPI_KIND[int(m.group(5))], m.group(1), m.group(2), m.group(6),
                                               {"0": "plain", "false": "plain", "1": "ws4+4", "true": "ws4+4", "2": "ws8+4"}.get(m.group(7), m.group(7)))
    if n.startswith("Cijk_"):
        return "rocBLAS/hipBLASLt GEMM (task heads, torch)"
    if "at::native" in n:
        return "ATen elementwise/reduce (task heads, loss, torch)"
    return short(n)[:70]


def step_breakdown(trace, dst):
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'adamw_kernel' in r['Kernel_Name']]
    if len(idx) < 3:
        return
    a, b = idx[-3] + 1, idx[-1] + 1
    step = rows[a:b]
    t0, t1 = int(step[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in step)
    iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in step)
    busy, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    tot, cnt = collections.Counter(), collections.Counter()
    for r in step:
        k = family(r['Kernel_Name'])
        tot[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 2e3
        cnt[k] += 0.5
    total = sum(tot.values())
    with open(dst, 'w') as f:
        f.write("last two optimizer steps of `python bench.py` under rocprofv3 --kernel-trace (per step):\n")
        f.write("  wall span %.0f us | GPU busy (union of kernel intervals) %.0f us | sum of kernel durations %.0f us "
                "(> busy: side-stream kernels overlap) | %d launches\n\n" % ((t1 - t0) / 2e3, busy / 2e3, total, len(step) // 2))
        f.write("%10s %7s %9s %9s  %s\n" % ("us/step", "share", "launches", "avg us", "kernel family"))
        for k, v in tot.most_common():
            f.write("%10.1f %6.1f%% %9.1f %9.2f  %s\n" % (v, 100 * v / total, cnt[k], v / cnt[k], k))


def pmc(path, counter):
    """{(kernel short name, grid): [values]} of one counter."""
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            out[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return out


def pmc_traffic(fetch_csv, write_csv, dst):
    """Per GEMM flavour (the EPI template argument): mean FETCH_SIZE / WRITE_SIZE per dispatch.  The passes run with the
    tile choices pinned (UNITER_AMD_TUNE_CACHE), so every dispatch of a flavour is a steady-state kernel of the step."""
    fetch, write = pmc(fetch_csv, "FETCH_SIZE"), pmc(write_csv, "WRITE_SIZE")

    def by_kind(table):
        acc = collections.defaultdict(lambda: [0.0, 0, set()])
        for (name, grid), vals in table.items():
            m = GEMM_RE.search(name.replace('(anonymous namespace)::', ''))
            if not m:
                continue
            a = acc[int(m.group(5))]
            a[0] += sum(vals)
            a[1] += len(vals)
            a[2].add((short(name), grid))
        return acc

    fk, wk = by_kind(fetch), by_kind(write)
    out = {}
    for kind, (fsum, fn, fkeys) in sorted(fk.items()):
        wsum, wn, _ = wk.get(kind, [0.0, 0, set()])
        f_kib = fsum / fn
        w_kib = wsum / wn if wn else None
        # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; gfx950 tallies the 128-B requests of wide coalesced reads
        # at 64 B, so the read side is doubled (MI355X_MICROARCH.md "HBM").  WRITE_SIZE is used as reported.
        rd = 2.0 * f_kib * 1024.0
        wr = w_kib * 1024.0 if w_kib is not None else None
        out[str(kind)] = {"kind": EPI_KIND[kind], "dispatches": fn, "kernels": sorted("%s grid %d" % k for k in fkeys),
                          "single_shape": len(fkeys) == 1,
                          "fetch_size_kib_raw": round(f_kib, 1), "write_size_kib_raw": None if w_kib is None else round(w_kib, 1),
                          "hbm_read_bytes": round(rd), "hbm_write_bytes": None if wr is None else round(wr),
                          "hbm_bytes": round(rd + (wr or 0.0))}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over "
                         "`python bench.py --steps 3 --warmup 2` with pinned tile choices",
               "correction": "KiB -> bytes; FETCH_SIZE doubled on gfx950 (128-B requests of wide coalesced reads are tallied "
                             "at 64 B, MI355X_MICROARCH.md 'HBM'); WRITE_SIZE as reported; counters sit on the L2's fabric "
                             "side, so Infinity-Cache hits are included",
               "by_kind": out}, open(dst, 'w'), indent=1)
    return len(out)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    st = one(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
    tr = one(os.path.join(src, "trace", "*", "*kernel_trace.csv"))
    if st:
        print("kernel stats rows:", kernel_stats(st, os.path.join(dst, tag + "_kernel_stats.csv")))
    if tr:
        step_breakdown(tr, os.path.join(dst, tag + "_step_breakdown.txt"))
    fc = one(os.path.join(src, "pmc_fetch", "*", "*counter_collection.csv"))
    wc = one(os.path.join(src, "pmc_write", "*", "*counter_collection.csv"))
    if fc and wc:
        print("pmc kernels:", pmc_traffic(fc, wc, os.path.join(dst, tag + "_pmc_traffic.json")))


if __name__ == "__main__":
    main()
