#!/usr/bin/env python3
"""Vendor yardstick for the eight per-layer ("chain") GEMM shapes of UNITER-base at 32 x 96 tokens.

NOT on the product path: this script times what `torch.nn.functional.linear` / `torch.matmul` (hipBLASLt / rocBLAS Tensile
kernels) reach in bf16 on the very shapes the hand-written tiles of uniter_amd/csrc/gemm.hip serve, so that DESIGN.md's
statement about the ceiling of those shapes is a measurement and not a model.

    python scripts/vendor_gemm_yardstick.py [--iters 20] [--layers 12] [--out gpurun_out/r06/vendor_gemm_yardstick.txt]

Two regimes per shape:
  hot    the same operands launched `iters` times back to back (what `tests/native/build/test_kernels --roofs` measures for ours)
  chain  `layers` distinct weight / activation sets walked in the order of the encoder's forward (qkv, out, ffn1, ffn2 per layer) and
         of its data-gradient chain (ffn2, ffn1, out, qkv per layer, last layer first): operands as cold as they are in the step
Times are HIP events around the whole loop / launches (queue boundaries included, as in `--roofs`).  Run it under
`rocprofv3 --kernel-trace --stats` to get the per-kernel durations and the Tensile kernel names (macro tile `MT..`, depth-U `DU..`,
`DTL` = direct-to-LDS); `scripts/vendor_gemm_yardstick.py --fold <dir>` folds such a trace into per-shape rows: phases are fenced
by an int32 fill kernel that nothing else in the run launches.
"""
import argparse
import csv
import glob
import os
import sys

SHAPES = [  # name, flavour, Linear out features N, Linear in features K  (M = tokens)
    ("qkv_fwd", "fwd", 2304, 768), ("out_fwd", "fwd", 768, 768), ("ffn1_fwd", "fwd", 3072, 768), ("ffn2_fwd", "fwd", 768, 3072),
    ("ffn2_dgrad", "dgrad", 768, 3072), ("ffn1_dgrad", "dgrad", 3072, 768), ("out_dgrad", "dgrad", 768, 768),
    ("qkv_dgrad", "dgrad", 2304, 768)]


def fold(trace_dir, out, skip=0, iters=20):
    files = glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit("no *kernel_trace.csv under " + trace_dir)
    rows = list(csv.DictReader(open(max(files, key=os.path.getmtime))))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    phases, cur = [], None
    for r in rows:
        name = r["Kernel_Name"]
        if "FillFunctor<int>" in name:
            cur = []
            phases.append(cur)
            continue
        if cur is not None:
            cur.append((name, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3))
    lines = []
    phases = phases[skip:]
    labels = [s[0] + " hot" for s in SHAPES] + ["chain forward", "chain data-gradient"]
    flop = {s[0] + " hot": 2.0 * 3072 * s[2] * s[3] for s in SHAPES}
    for label, ph in zip(labels, phases):
        by = {}
        for name, us in ph:
            if name.startswith("Cijk_"):               # Tensile GEMM kernels only (the phase also holds the next shape's set-up)
                by.setdefault(name, []).append(us)
        for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
            if len(v) < iters and "hot" in label:      # (< iters launches: the NEXT shape's three warm-up launches)
                continue
            v2 = sorted(v)
            avg = sum(v) / len(v)
            tf = ("%7.1f TF" % (flop[label] / avg * 1e-6)) if label in flop else "          "
            lines.append("%-22s x%-4d avg %7.2f us  med %7.2f  min %7.2f  %s  %s" % (label, len(v), avg, v2[len(v2) // 2], v2[0], tf, name))
    txt = "\n".join(lines)
    print(txt)
    if out:
        with open(out, "a") as f:
            f.write("\n== rocprofv3 --kernel-trace: kernel durations per phase ==\n" + txt + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--tokens", type=int, default=3072)
    ap.add_argument("--out", default=None)
    ap.add_argument("--skip", type=int, default=0, help="--fold: marker phases to drop at the front")
    ap.add_argument("--fold", default=None, help="fold the kernel trace under this rocprofv3 output directory")
    a = ap.parse_args()
    if a.fold:
        fold(a.fold, a.out, a.skip, a.iters)
        return
    import torch
    import torch.nn.functional as F
    assert torch.cuda.is_available(), "needs the GPU"
    dev = torch.device("cuda", 0)
    M = a.tokens
    g = torch.Generator(device=dev).manual_seed(5)

    def rnd(*shape, scale=1.0):
        return (torch.rand(*shape, device=dev, generator=g, dtype=torch.float32) * 2 - 1).mul_(scale).to(torch.bfloat16)

    marker = torch.empty(64, dtype=torch.int32, device=dev)      # (empty, not zeros: its fill kernel is the phase fence)

    def timed(fn, n):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    out = []
    out.append("vendor bf16 GEMM yardstick, %s, torch %s, M = %d tokens, %d launches per figure (HIP events around the loop)" % (
        torch.cuda.get_device_name(0), torch.__version__, M, a.iters))
    ops = {}
    for name, flav, N, K in SHAPES:
        if flav == "fwd":
            x, w, b = rnd(M, K), rnd(N, K, scale=0.05), rnd(N, scale=0.1)
            fn = (lambda x, w, b: (lambda i: F.linear(x, w, b)))(x, w, b)
        else:
            dy, w = rnd(M, N), rnd(N, K, scale=0.05)
            fn = (lambda dy, w: (lambda i: torch.matmul(dy, w)))(dy, w)
        ops[name] = fn
        for i in range(3):
            fn(i)
        marker.fill_(1)
        us = timed(fn, a.iters)
        out.append("  HOT   %-12s M%d N%d K%d : %7.2f us  %7.1f TF" % (name, M, N, K, us, 2.0 * M * N * K / us * 1e-6))
    # chain regime: distinct operands per layer, forward order then data-gradient order
    L = a.layers
    xs = {K: [rnd(M, K) for _ in range(L)] for K in (768, 3072)}
    dys = {N: [rnd(M, N) for _ in range(L)] for N in (768, 2304, 3072)}
    ws = {(N, K): [rnd(N, K, scale=0.05) for _ in range(L)] for _, _, N, K in SHAPES}
    bs = {N: rnd(N, scale=0.1) for N in (768, 2304, 3072)}
    fwd_order = [s for s in SHAPES if s[1] == "fwd"]
    bwd_order = [s for s in SHAPES if s[1] == "dgrad"]

    def chain_fwd(i):
        for l in range(L):
            for _, _, N, K in fwd_order:
                F.linear(xs[K][l], ws[(N, K)][l], bs[N])

    def chain_bwd(i):
        for l in reversed(range(L)):
            for _, _, N, K in bwd_order:
                torch.matmul(dys[N][l], ws[(N, K)][l])

    chain_fwd(0)
    chain_bwd(0)
    marker.fill_(1)
    reps = max(2, a.iters // 4)
    us_f = timed(chain_fwd, reps)
    marker.fill_(1)
    us_b = timed(chain_bwd, reps)
    gf = sum(2.0 * M * N * K for _, _, N, K in fwd_order) * L
    out.append("  CHAIN forward        %d layers x (qkv, out, ffn1, ffn2): %8.1f us per pass = %6.2f us per launch, %7.1f TF" % (
        L, us_f, us_f / (4 * L), gf / us_f * 1e-6))
    out.append("  CHAIN data-gradient  %d layers x (ffn2, ffn1, out, qkv): %8.1f us per pass = %6.2f us per launch, %7.1f TF" % (
        L, us_b, us_b / (4 * L), gf / us_b * 1e-6))
    txt = "\n".join(out)
    print(txt)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    sys.exit(main())
