#!/bin/bash
# Round 6, call 5: (1) A/B of the weight-gradient stream's priority (lowest vs default) on the c2 line, alternating; (2) attention alone
# (baseline for the attention work); (3) the GPU suite with the round's test changes (flat 5e-2 hidden bound, joint one-element rule,
# c4 at B = 32, heads that raise), every failure listed.  Output: gpurun_out/r06c5/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06c5
mkdir -p "$OUT"
cd "$ROOT"
T=$ROOT/tests/native/build/test_kernels
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['ms_per_step'], d['timed_windows']['ms_per_step'], 'fwd/bwd', r['encoder_fwd_bwd']['fwd_ms'], r['encoder_fwd_bwd']['bwd_ms'])"; }
for rep in 1 2; do
  for v in 1 0; do
    UNITER_AMD_WGRAD_STREAM_PRIO=$v timeout 300 python bench.py --no-cpu-baseline --no-traffic --steps 30 --warmup 8 2>/dev/null | tee "$OUT/c2_prio${v}_$rep.json" | line "wgrad stream prio(low=1)=$v"
  done
done 2>&1 | tee "$OUT/wgrad_stream_priority_ab.txt"
timeout 120 $T --attn 32 96 12 0.1 2>&1 | tail -12 | tee "$OUT/attn_alone.txt"
timeout 1200 python -m pytest tests -q -m gpu -s > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -3; grep -E "^FAILED|^ERROR" "$OUT/pytest_gpu.log" | cut -c1-300 | head -30
grep -E "headline parity|c4 B=32" "$OUT/pytest_gpu.log" | head
