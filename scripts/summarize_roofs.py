#!/usr/bin/env python3
"""Fold the PMC passes of scripts/gpu_r5_call1.sh (part `roofs`) into one table per chain GEMM shape.

    python scripts/summarize_roofs.py gpurun_out/r05c1 [profiles/r05_chain_gemm_roofs.json]

Alone passes (`test_kernels --roofs 5`: eight shapes in a fixed order, 3 warm-up + 5 launches each) are attributed to shapes by
dispatch order; in-situ passes (`test_kernels --enc`) by (kernel name, grid) — shapes that share a kernel and a grid are averaged.
Every counter is reported per launch.  Derived columns (definitions in the JSON):
  mfma_busy      SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE-equivalent cycles x 1024 SIMDs), clock from duration x 2.4 GHz when
                 GRBM_GUI_ACTIVE is not in the same pass
  l2_hit         TCC_HIT / (TCC_HIT + TCC_MISS)
  lds_busy       SQ_LDS_IDX_ACTIVE (quad... LDS-array cycles summed over CUs) / (duration x clock x 256 CUs)
"""
import collections
import csv
import glob
import gzip
import json
import os
import sys

SHAPES = ["qkv_fwd", "out_fwd", "ffn1_fwd_gelu", "ffn2_fwd", "ffn2_dgrad_gelu", "ffn1_dgrad", "out_dgrad", "qkv_dgrad"]
DIMS = {"qkv_fwd": (3072, 2304, 768), "out_fwd": (3072, 768, 768), "ffn1_fwd_gelu": (3072, 3072, 768), "ffn2_fwd": (3072, 768, 3072),
        "ffn2_dgrad_gelu": (3072, 3072, 768), "ffn1_dgrad": (3072, 768, 3072), "out_dgrad": (3072, 768, 768), "qkv_dgrad": (3072, 768, 2304)}
# algorithmic bytes: both operands once + every output once (+ the epilogue's residual / pre-activation operand), bf16
ALG = {"qkv_fwd": 2 * (3072 * 768 + 2304 * 768 + 3072 * 2304), "out_fwd": 2 * (3072 * 768 * 3 + 768 * 768),
       "ffn1_fwd_gelu": 2 * (3072 * 768 + 3072 * 768 + 2 * 3072 * 3072), "ffn2_fwd": 2 * (3072 * 3072 + 768 * 3072 + 2 * 3072 * 768),
       "ffn2_dgrad_gelu": 2 * (3072 * 768 + 768 * 3072 + 2 * 3072 * 3072), "ffn1_dgrad": 2 * (3072 * 3072 + 3072 * 768 + 2 * 3072 * 768),
       "out_dgrad": 2 * (3072 * 768 * 2 + 768 * 768), "qkv_dgrad": 2 * (3072 * 2304 + 2304 * 768 + 2 * 3072 * 768)}
PER_SHAPE = 8          # 3 warm-up + 5 timed launches of --roofs 5


def read_csv(path):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as f:
        return list(csv.DictReader(f))


def find(d, pat):
    g = glob.glob(os.path.join(d, "*", pat)) + glob.glob(os.path.join(d, "*", pat + ".gz")) + glob.glob(os.path.join(d, pat)) + glob.glob(os.path.join(d, pat + ".gz"))
    return g[0] if g else None


def load_pass(d):
    cc = find(d, "*counter_collection.csv")
    if not cc:
        return None
    per = collections.OrderedDict()
    for r in read_csv(cc):
        k = int(r["Dispatch_Id"])
        e = per.setdefault(k, {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"]), "wg": int(r.get("Workgroup_Size", 0) or 0), "c": collections.defaultdict(float)})
        e["c"][r["Counter_Name"]] += float(r["Counter_Value"])
        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
            e["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tr = find(d, "*kernel_trace.csv")
    if tr:
        for r in read_csv(tr):
            k = int(r["Dispatch_Id"])
            if k in per and "ns" not in per[k]:
                per[k]["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return [per[k] for k in sorted(per)]


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("_GLOBAL__N_1", "")
    return name[:110]


def is_gemm(name):
    return "gemm" in name and "splitk_reduce" not in name


def avg(entries):
    n = len(entries)
    out = {"launches": n, "kernel": short(entries[0]["name"]), "grid": entries[0]["grid"], "workgroup": entries[0]["wg"]}
    ns = [e.get("ns") for e in entries if e.get("ns")]
    if ns:
        out["avg_us_under_counters"] = round(sum(ns) / len(ns) / 1e3, 2)
    keys = set()
    for e in entries:
        keys |= set(e["c"])
    for k in sorted(keys):
        out[k] = round(sum(e["c"].get(k, 0.0) for e in entries) / n, 1)
    return out


def alone(dispatches):
    g = [e for e in dispatches if is_gemm(e["name"])]
    segs, cur = [], []
    for e in g:
        if cur and ((e["name"], e["grid"]) != (cur[0]["name"], cur[0]["grid"]) or len(cur) == PER_SHAPE):
            segs.append(cur)
            cur = []
        cur.append(e)
    if cur:
        segs.append(cur)
    out = {}
    if len(segs) != len(SHAPES):
        out["_warning"] = "expected %d segments, found %d (%s)" % (len(SHAPES), len(segs), [len(s) for s in segs])
    for s, seg in zip(SHAPES, segs):
        out[s] = avg(seg[3:] if len(seg) > 3 else seg)
    return out


def insitu(dispatches):
    groups = collections.OrderedDict()
    for e in dispatches:
        if is_gemm(e["name"]) and "multi" not in e["name"] and "group" not in e["name"]:
            groups.setdefault((e["name"], e["grid"]), []).append(e)
    return {"%s grid %d" % (short(k[0]), k[1]): avg(v) for k, v in groups.items() if len(v) >= 12}


def derive(row, clock_ghz=None):
    us = row.get("avg_us_under_counters")
    d = {}
    if "GRBM_GUI_ACTIVE" in row and us:
        d["clock_ghz_under_counters"] = round(row["GRBM_GUI_ACTIVE"] / (us * 1e3), 3)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in row and us:
        d["mfma_busy_at_2p4ghz"] = round(row["SQ_VALU_MFMA_BUSY_CYCLES"] / (us * 1e3 * 2.4 * 1024), 4)
    if "SQ_BUSY_CYCLES" in row and "SQ_VALU_MFMA_BUSY_CYCLES" in row and row["SQ_BUSY_CYCLES"]:
        d["mfma_busy_over_sq_busy"] = round(row["SQ_VALU_MFMA_BUSY_CYCLES"] / row["SQ_BUSY_CYCLES"], 4)
    if "SQ_WAVE_CYCLES" in row and row["SQ_WAVE_CYCLES"]:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if k in row:
                d[k.lower() + "_frac_of_wave_cycles"] = round(row[k] / row["SQ_WAVE_CYCLES"], 4)
    if "TCC_HIT_sum" in row and (row["TCC_HIT_sum"] + row.get("TCC_MISS_sum", 0)):
        d["l2_hit"] = round(row["TCC_HIT_sum"] / (row["TCC_HIT_sum"] + row["TCC_MISS_sum"]), 4)
    if "SQ_LDS_IDX_ACTIVE" in row and us:
        d["lds_idx_active_per_cu_cycle_at_2p4ghz"] = round(row["SQ_LDS_IDX_ACTIVE"] / (us * 1e3 * 2.4 * 256), 4)
    if "FETCH_SIZE" in row:
        d["fetch_MB_x2_gfx950"] = round(row["FETCH_SIZE"] * 2 * 1024 / 1e6, 2)     # FETCH_SIZE is in KiB; gfx950 reports half (MI355X_MICROARCH.md)
    if "WRITE_SIZE" in row:
        d["write_MB"] = round(row["WRITE_SIZE"] * 1024 / 1e6, 2)
    return d


def main():
    src = sys.argv[1]
    dst = sys.argv[2] if len(sys.argv) > 2 else None
    res = {"alone": {}, "in_situ": {}, "notes": {
        "alone": "test_kernels --roofs 5: each shape launched back to back on hot operands with the shipped tile; counters per launch, averaged over 5",
        "in_situ": "test_kernels --enc (12-layer forward + backward loops): grouped by (kernel, grid)",
        "algorithmic_MB": {k: round(v / 1e6, 2) for k, v in ALG.items()},
        "units": "SQ_* wave counters are quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES are cycles summed over SIMDs / SEs; FETCH_SIZE/WRITE_SIZE KiB"}}
    for d in sorted(glob.glob(os.path.join(src, "pmc_alone_*"))):
        if not os.path.isdir(d):
            continue
        disp = load_pass(d)
        if disp is None:
            res["alone"][os.path.basename(d)] = "no counter CSV (invalid counter name?)"
            continue
        tab = alone(disp)
        for s, row in tab.items():
            if isinstance(row, dict):
                row.update(derive(row))
        res["alone"][os.path.basename(d)[len("pmc_alone_"):]] = tab
    for d in sorted(glob.glob(os.path.join(src, "pmc_insitu_*"))):
        if not os.path.isdir(d):
            continue
        disp = load_pass(d)
        if disp is None:
            res["in_situ"][os.path.basename(d)] = "no counter CSV"
            continue
        tab = insitu(disp)
        for s, row in tab.items():
            row.update(derive(row))
        res["in_situ"][os.path.basename(d)[len("pmc_insitu_"):]] = tab
    tm = os.path.join(src, "roofs_timing.txt")
    if os.path.exists(tm):
        res["timing_without_counters"] = [l.strip() for l in open(tm) if "ROOF" in l]
    txt = json.dumps(res, indent=1)
    if dst:
        open(dst, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
