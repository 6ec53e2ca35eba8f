#!/bin/bash
# Round 6, call 18: folded gradient norm (AdamW.fold_norm: per-tile sums of squares out of the deferred weight-gradient launch,
# uniter_encoder_last_grad_sq -> uniter_adamw_grad_norm_ex).  Native harness (the 256 x 256 tile's epilogue changed), the optimizer /
# headline tests, same-box A/B of the c2 line (UNITER_AMD_FOLD_NORM=0/1), whole GPU suite.  Output: gpurun_out/r06c18/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06c18
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 tests/native/build/test_kernels > "$OUT/harness.log" 2>&1; echo "harness rc=$?"; grep -c "^\[ OK \]" "$OUT/harness.log"; grep "FAIL" "$OUT/harness.log" | head -5; tail -1 "$OUT/harness.log"
timeout 900 python -m pytest tests -x -q -m gpu -k "folded or lazy or adamw or overlapped or headline or accum" > "$OUT/pytest_first.log" 2>&1; echo "pytest first rc=$?"; tail -3 "$OUT/pytest_first.log"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['ms_per_step'], d['timed_windows']['ms_per_step'], 'fwd/bwd', r['encoder_fwd_bwd']['fwd_ms'], r['encoder_fwd_bwd']['bwd_ms'], 'frac', r['frac'], 'loss', d['final_loss'])"; }
B="timeout 300 python bench.py --no-cpu-baseline --no-traffic --steps 30 --warmup 8"
for rep in 1 2 3; do
  for v in 1 0; do
    UNITER_AMD_FOLD_NORM=$v $B 2>/dev/null | tee "$OUT/c2_fold${v}_$rep.json" | line "c2 fold_norm=$v"
  done
done 2>&1 | tee "$OUT/ab.txt"
python scripts/segment_times.py 2>&1 | tail -8 | tee "$OUT/segments.txt"
timeout 1500 python -m pytest tests -x -q -m gpu > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
