#!/bin/bash
# Round 6, call 9: 16-byte attention output stores (in the tree) and the 16-byte LayerNorm kernels (UNITER_AMD_LN_WIDE A/B): harness,
# encoder harness, c2 bench line, GPU suite.  Output: gpurun_out/r06c9/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06c9
mkdir -p "$OUT"
cd "$ROOT"
T=$ROOT/tests/native/build/test_kernels
export UNITER_TUNED_JSON=$ROOT/uniter_amd/tuned/gfx950.json
timeout 900 $T --quick > "$OUT/native_harness.log" 2>&1; echo "harness rc=$? FAIL lines: $(grep -c '^\[FAIL' "$OUT/native_harness.log")"; grep -E "^\[FAIL" "$OUT/native_harness.log" | head; tail -1 "$OUT/native_harness.log"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['ms_per_step'], d['timed_windows']['ms_per_step'], 'fwd/bwd', r['encoder_fwd_bwd']['fwd_ms'], r['encoder_fwd_bwd']['bwd_ms'], 'frac', r['frac'], 'loss', d['final_loss'])"; }
for rep in 1 2; do
  for v in 1 0; do
    UNITER_AMD_LN_WIDE=$v UNITER_BENCH_XCD_ONLY=1 UNITER_BENCH_SKIP_CHAIN_CHECK=1 timeout 200 $T --enc 2>&1 | grep -E "ENCODER|in-situ (layernorm|ln|attn)" | sed "s/^/ln_wide=$v /" | tee -a "$OUT/enc_ab.txt"
    UNITER_AMD_LN_WIDE=$v timeout 300 python bench.py --no-cpu-baseline --no-traffic --steps 30 --warmup 8 2>/dev/null | tee "$OUT/c2_lnwide${v}_$rep.json" | line "c2 ln_wide=$v"
  done
done 2>&1 | tee "$OUT/ab.txt"
timeout 1200 python -m pytest tests -q -m gpu -s -x > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -3; grep -E "^FAILED|^ERROR" "$OUT/pytest_gpu.log" | cut -c1-300 | head -30
