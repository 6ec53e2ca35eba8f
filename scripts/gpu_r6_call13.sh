#!/bin/bash
# Round 6, call 13: FFN1 saves gelu'(u) instead of u (UNITER_AMD_SAVE_ACT_GRAD, default on): native harness (all checks), the
# roofs with the two new forms, the encoder / headline parity tests, same-box A/B of the c2 line.  Output: gpurun_out/r06c13/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06c13
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 tests/native/build/test_kernels > "$OUT/harness.log" 2>&1; echo "harness rc=$?"; grep -c "^\[ OK \]" "$OUT/harness.log"; grep "FAIL" "$OUT/harness.log" | head -5; tail -1 "$OUT/harness.log"
export UNITER_TUNED_JSON=$ROOT/uniter_amd/tuned/gfx950.json
timeout 200 tests/native/build/test_kernels --roofs 20 > "$OUT/roofs.txt" 2>&1; tail -12 "$OUT/roofs.txt"
timeout 1500 python -m pytest tests -x -q -m gpu > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['ms_per_step'], d['timed_windows']['ms_per_step'], 'fwd/bwd', r['encoder_fwd_bwd']['fwd_ms'], r['encoder_fwd_bwd']['bwd_ms'], 'frac', r['frac'], 'loss', d['final_loss'])"; }
for rep in 1 2 3; do
  for v in 1 0; do
    UNITER_AMD_SAVE_ACT_GRAD=$v timeout 300 python bench.py --no-cpu-baseline --no-traffic --steps 30 --warmup 8 2>/dev/null | tee "$OUT/c2_savegrad${v}_$rep.json" | line "c2 save_act_grad=$v"
  done
done 2>&1 | tee "$OUT/ab.txt"
