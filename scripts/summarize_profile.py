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
DEEP_RE = re.compile(r'gemm([86])_kernel<(\w+), (\w+), (\d)>')          # the deep-pipelined 256x256 / 192x192 tiles (gemm8.cuh)


class _M(object):
    """One parsed GEMM kernel name in the group() layout of GEMM_RE (bm, bn, tra, trb, epi, stages, ws)."""
    def __init__(self, g):
        self.g = g

    def group(self, i):
        return self.g[i - 1]


def gemm_match(name):
    n = name.replace('(anonymous namespace)::', '')
    m = GEMM_RE.search(n)
    if m:
        return m
    d = DEEP_RE.search(n)
    if d:
        edge = "256" if d.group(1) == "8" else "192"
        return _M((edge, edge, d.group(2), d.group(3), d.group(4), "2" if d.group(1) == "8" else "3", "3" if d.group(1) == "8" else "4"))
    return None
EPI_KIND = {0: "gemm fwd +bias", 1: "gemm fwd +bias+gelu", 2: "gemm fwd +bias+dropout+residual", 3: "gemm dgrad",
            4: "gemm dgrad x gelu'", 5: "gemm wgrad", 6: "gemm fwd QKV + self-attention (one launch)"}


def short(name):
    name = name.replace('(anonymous namespace)::', '')
    name = re.sub(r'\(.*$', '', name)
    return name.replace('void ', '')[:110]


def one(pattern):
    files = glob.glob(pattern)             # gpurun merges into existing directories: take the newest run
    return max(files, key=os.path.getmtime) if files else None


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
    m = gemm_match(n)
    if m:
        return "%s  [tile %sx%s st%s %s]" % (EPI_KIND[int(m.group(5))], m.group(1), m.group(2), m.group(6),
                                               {"0": "plain", "false": "plain", "1": "ws4+4", "true": "ws4+4", "2": "ws8+4",
                                                "3": "deep 8-phase", "4": "deep 3-phase"}.get(m.group(7), m.group(7)))
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
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))                 # program order (needed to split shared templates)
    for r in rows:
        out[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return out


def pmc_traffic(fetch_csv, write_csv, dst):
    """Per GEMM flavour (the EPI template argument): mean FETCH_SIZE / WRITE_SIZE per dispatch.  The passes run with the
    tile choices pinned (UNITER_AMD_TUNE_CACHE), so every dispatch of a flavour is a steady-state kernel of the step."""
    fetch, write = pmc(fetch_csv, "FETCH_SIZE"), pmc(write_csv, "WRITE_SIZE")

    def by_kind(table):
        acc = collections.defaultdict(lambda: [0.0, 0, set()])
        for (name, grid), vals in table.items():
            m = gemm_match(name)
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
    # per (kernel, grid) entries: a flavour that runs at several shapes is told apart by its dispatch grid
    per = []
    for (name, grid), fv in sorted(fetch.items(), key=lambda kv: (kv[0][0], kv[0][1])):
        m = gemm_match(name)
        if not m:
            continue
        wv = write.get((name, grid), [])
        rd = 2.0 * (sum(fv) / len(fv)) * 1024.0
        wr = (sum(wv) / len(wv)) * 1024.0 if wv else None
        per.append({"kind_id": int(m.group(5)), "kernel": short(name), "grid": grid, "dispatches": len(fv),
                    "hbm_read_bytes": round(rd), "hbm_write_bytes": None if wr is None else round(wr),
                    "hbm_bytes": round(rd + (wr or 0.0))})
    # exact attribution to (kind, M, N, K): the tile choices the profiled runs used are in tune_cache.json, the tile table
    # is parsed from gemm.hip, so the template arguments and the dispatch grid of every encoder GEMM are known.  Shapes
    # that share one template + grid are told apart by program order (they alternate in a fixed per-layer sequence).
    by_shape = {}
    cache_path = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(fetch_csv))), "tune_cache.json")
    if not os.path.exists(cache_path):
        # the passes ran with the shipped tile table (scripts/profile_round.sh): that IS the set of choices they used
        cache_path = os.path.join(ROOT, "uniter_amd", "tuned", "gfx950.json")
    tiles = parse_tiles()
    if os.path.exists(cache_path) and tiles:
        # per-layer launch order of the encoder (encoder.hip): (kind, EPI, N, K) with H / I recovered from the cache
        entries = json.load(open(cache_path)).get("gemm", [])
        # the table may hold several model sizes: keep the one whose token count the profiled run used (the grouped
        # launch's grid tells; default: the headline shape, the smallest hidden size at M = 3072)
        m_run = int(os.environ.get("UNITER_PROFILE_TOKENS", "3072"))
        entries = [e for e in entries if e["M"] == m_run]
        h_run = min([e["K"] for e in entries if e["kind"] == 0] or [0])
        entries = [e for e in entries if e["kind"] == 3 or min(e["N"], e["K"]) == h_run]
        g3all = [e for e in entries if e["kind"] == 3]
        entries = [e for e in entries if e["kind"] != 3] + [e for e in g3all if e["N"] == 9 * h_run]   # N = 5H + I with I = 4H
        hs = sorted({e["K"] for e in entries if e["kind"] < 3} | {e["N"] for e in entries if e["kind"] < 3})
        H, I = (hs[0], hs[-1]) if hs else (0, 0)
        order = [(0, 0, 3 * H, H), (0, 2, H, H), (0, 1, I, H), (0, 2, H, I),                       # forward
                 (1, 4, H, I), (1, 3, I, H), (1, 3, H, H), (1, 3, 3 * H, H),                       # backward, main stream
                 (2, 5, H, I), (2, 5, I, H), (2, 5, H, H), (2, 5, 3 * H, H)]                       # backward, side stream
        lookup = {(e["kind"], e["N"], e["K"]): e for e in entries}
        groups = collections.defaultdict(list)            # (kernel name, grid) -> [(epi, M, N, K)] in program order
        names = {}
        for kind, epi, N, K in order:
            e = lookup.get((kind, N, K))
            if e is None:
                continue
            M, cfg, splits = e["M"], e["cfg"], e["splits"]
            bm, bn, st, ws = tiles[cfg]
            rows, cols = (M, N) if kind == 0 else ((M, K) if kind == 1 else (N, K))      # output of the launch
            threads = 256 if ws == 0 else (512 if ws in (1, 3, 4) else 768)
            grid = ((rows + bm - 1) // bm) * (cols // bn) * threads * splits
            layout = {0: "false, false", 1: "false, true", 2: "true, true"}[kind]
            if ws >= 3:
                want = "gemm%d_kernel<%s, %d>" % (8 if ws == 3 else 6, layout, epi)
            else:
                want = "gemm_kernel<%d, %d, %s, %d, %d, %d>" % (bm, bn, layout, epi, st, ws)
            for (name, g2) in fetch:
                if g2 == grid and want in name.replace('(anonymous namespace)::', ''):
                    groups[(name, grid)].append((epi, M, N, K))
                    names[(name, grid)] = short(name)
        for key, shapes in groups.items():
            fv, wv = fetch.get(key, []), write.get(key, [])
            n = len(shapes)
            for idx, (epi, M, N, K) in enumerate(shapes):
                f_sel, w_sel = fv[idx::n], wv[idx::n]
                if len(f_sel) < 8 or not w_sel:
                    continue
                rd = 2.0 * (sum(f_sel) / len(f_sel)) * 1024.0
                wr = (sum(w_sel) / len(w_sel)) * 1024.0
                by_shape["%d:%d:%d:%d" % (epi, M, N, K)] = {
                    "kernel": names[key], "grid": key[1], "dispatches": len(f_sel), "shares_template_with": n - 1,
                    "hbm_read_bytes": round(rd), "hbm_write_bytes": round(wr), "hbm_bytes": round(rd + wr)}
        # the grouped weight-gradient launch (timing kind 13: N = sum N_i*K_i weight elements, K = problems per group)
        g3 = [e for e in entries if e["kind"] == 3]
        if g3 and H and I:
            welems = H * I + I * H + H * H + 3 * H * H
            for (name, grid), fv in fetch.items():
                if "gemm_group_kernel<" not in name and "gemm8_group_kernel" not in name and "gemm6_group_kernel" not in name:
                    continue
                if len(fv) < 8 * 4:            # (the NLVR2 head's small grouped launches share the template: 1-2 per step)
                    continue
                wv = write.get((name, grid), [])
                if len(fv) < 8 or not wv:
                    continue
                rd = 2.0 * (sum(fv) / len(fv)) * 1024.0
                wr = (sum(wv) / len(wv)) * 1024.0
                by_shape["13:%d:%d:4" % (g3[0]["M"], welems)] = {
                    "kernel": short(name), "grid": grid, "dispatches": len(fv), "shares_template_with": 0,
                    "hbm_read_bytes": round(rd), "hbm_write_bytes": round(wr), "hbm_bytes": round(rd + wr)}
        # the deferred launch (gemm8_multi_kernel): every weight / bias / LayerNorm parameter gradient of one backward call.
        # Timing kind 13 reports it as M tokens, N = weight elements of all its problems, K = problems (4 per layer).
        if g3 and H and I:
            layers = int(os.environ.get("UNITER_PROFILE_LAYERS", "12"))
            for (name, grid), fv in fetch.items():
                if "gemm8_multi_kernel" not in name:
                    continue
                wv = write.get((name, grid), [])
                if len(fv) < 3 or not wv:
                    continue
                rd = 2.0 * (sum(fv) / len(fv)) * 1024.0
                wr = (sum(wv) / len(wv)) * 1024.0
                by_shape["13:%d:%d:%d" % (g3[0]["M"], layers * (H * I + I * H + H * H + 3 * H * H), 4 * layers)] = {
                    "kernel": short(name), "grid": grid, "dispatches": len(fv), "shares_template_with": 0,
                    "hbm_read_bytes": round(rd), "hbm_write_bytes": round(wr), "hbm_bytes": round(rd + wr)}
    # every kernel family of the step (not only the GEMMs): bytes per optimizer step = sum over the family's dispatches / steps,
    # steps = number of AdamW launches of the pass (bench.py's roofline.traffic sums the non-optimizer rows)
    n_steps = max([len(v) for (name, _), v in fetch.items() if "adamw_kernel" in name] or [1])
    fam = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for (name, grid), fv in fetch.items():
        a = fam[family(name)]
        a[0] += 2.0 * sum(fv) * 1024.0
        a[2] += len(fv)
    for (name, grid), wv in write.items():
        fam[family(name)][1] += sum(wv) * 1024.0
    by_kernel = [{"kernel": k, "launches_per_step": round(v[2] / n_steps, 2), "hbm_read_bytes_per_step": round(v[0] / n_steps),
                  "hbm_write_bytes_per_step": round(v[1] / n_steps), "hbm_bytes_per_step": round((v[0] + v[1]) / n_steps)}
                 for k, v in sorted(fam.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))]
    json.dump({"by_shape": by_shape, "per_kernel": per, "by_kernel": by_kernel, "steps_in_pass": n_steps,
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over "
                         "`python bench.py --steps 3 --warmup 2` with pinned tile choices",
               "correction": "KiB -> bytes; FETCH_SIZE doubled on gfx950 (128-B requests of wide coalesced reads are tallied "
                             "at 64 B, MI355X_MICROARCH.md 'HBM'); WRITE_SIZE as reported; counters sit on the L2's fabric "
                             "side, so Infinity-Cache hits are included",
               "by_kind": out}, open(dst, 'w'), indent=1)
    return len(out)


def mfma_util(counter_csv, cal_csv, dst):
    """MFMA utilisation of the most expensive kernels of a step: SQ_VALU_MFMA_BUSY_CYCLES (matrix-pipe busy cycles summed
    over the chip's 1024 SIMDs) per dispatch / (dispatch duration x 1024 SIMDs x clock).  The clock is not a counter we
    can read per dispatch, so the unit is calibrated on a GEMM of known size run under the same counters: a 4096^3 bf16
    GEMM issues 4096^3 / (16*16*32) MFMAs of 16 busy cycles each."""
    def load(path):
        rows = list(csv.DictReader(open(path)))
        per = collections.defaultdict(lambda: collections.defaultdict(float))   # dispatch -> counter -> value
        meta = {}
        for r in rows:
            d = int(r["Dispatch_Id"])
            per[d][r["Counter_Name"]] += float(r["Counter_Value"])
            meta[d] = r
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                per[d]["_ns"] = float(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        return per, meta

    def durations(path):
        tr = one(os.path.join(os.path.dirname(path), "*kernel_trace.csv"))
        out = {}
        if tr:
            for r in csv.DictReader(open(tr)):
                out[int(r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        return out

    per, meta = load(counter_csv)
    dur = durations(counter_csv)
    for d, v in per.items():
        if "_ns" in v:
            dur[d] = v["_ns"]
    cal = None
    if cal_csv:
        cper, cmeta = load(cal_csv)
        cdur = durations(cal_csv)
        for d, v in cper.items():
            if "_ns" in v:
                cdur[d] = v["_ns"]
        busy = [v["SQ_VALU_MFMA_BUSY_CYCLES"] for d, v in cper.items() if "gemm8_kernel" in cmeta[d]["Kernel_Name"]]
        ns = [cdur.get(d) for d in cper if "gemm8_kernel" in cmeta[d]["Kernel_Name"] and cdur.get(d)]
        if busy and ns:
            expect = 4096.0 ** 3 / (16 * 16 * 32) * 16.0          # busy SIMD-cycles one launch must account for
            mean_busy = sum(busy) / len(busy)
            cal = {"kernel": "gemm8_kernel<false, false, 0> 4096^3", "launches": len(busy), "counter_per_launch": mean_busy,
                   "expected_busy_simd_cycles": expect, "counter_units_per_simd_cycle": mean_busy / expect,
                   "avg_ns": sum(ns) / len(ns),
                   "implied_clock_ghz_at_full_utilisation_of_busy_cycles": None}
    unit = cal["counter_units_per_simd_cycle"] if cal else 1.0
    acc = collections.defaultdict(lambda: [0.0, 0.0, 0, 0.0])     # name -> [busy, ns, launches, gui]
    for d, v in per.items():
        name = family(meta[d]["Kernel_Name"])
        a = acc[name]
        a[0] += v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / unit
        a[1] += dur.get(d, 0)
        a[2] += 1
        a[3] += v.get("GRBM_GUI_ACTIVE", 0.0)
    rows = []
    for name, (busy, ns, n, gui) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:8]:
        if ns <= 0:
            continue
        clock = 2.4
        rows.append({"kernel": name, "launches": n, "avg_us": round(ns / n / 1e3, 2),
                     "mfma_busy_simd_cycles_per_launch": round(busy / n),
                     "mfma_util_at_2p4ghz": round(busy / (ns * clock * 1024.0), 4),
                     "grbm_gui_active_per_launch": round(gui / n)})
    json.dump({"kernels": rows, "calibration": cal,
               "definition": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES (calibrated to SIMD-cycles) / (dispatch duration x 2.4 GHz x 1024 "
                             "SIMDs): the fraction of the chip's matrix-pipe cycles at the nominal clock that were busy; the chip "
                             "clocks below 2.4 GHz under load, so this is a lower bound of the in-kernel busy fraction",
               "source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace over `python bench.py --steps 3 --warmup 2`"},
              open(dst, 'w'), indent=1)
    return len(rows)


def parse_tiles():
    """[(bm, bn, stages, ws)] from the kTiles initialiser of uniter_amd/csrc/gemm.hip."""
    src = open(os.path.join(ROOT, "uniter_amd", "csrc", "gemm.hip")).read()
    m = re.search(r"constexpr TileShape kTiles\[\] = \{(.*?)\};", src, re.S)
    if not m:
        return []
    return [tuple(int(v) for v in t) for t in re.findall(r"\{(\d+), (\d+), (\d+), (\d+)\}", m.group(1))]


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
    mc = one(os.path.join(src, "pmc_mfma", "*", "*counter_collection.csv"))
    cc = one(os.path.join(src, "pmc_mfma_cal", "*", "*counter_collection.csv"))
    if mc:
        print("mfma util kernels:", mfma_util(mc, cc, os.path.join(dst, tag + "_mfma_util.json")))


if __name__ == "__main__":
    main()
