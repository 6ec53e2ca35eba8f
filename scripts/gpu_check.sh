#!/bin/bash
# Harness + GPU test suite + two bench lines (c2), output under gpurun_out/check.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/check
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 tests/native/build/test_kernels > "$OUT/harness.log" 2>&1; echo "harness rc=$?"; grep -c "^\[ OK \]" "$OUT/harness.log"; grep "FAIL" "$OUT/harness.log" | head -5; tail -1 "$OUT/harness.log"
timeout 1500 python -m pytest tests -x -q -m gpu > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('c2', d['ms_per_step'], d['value'], r['achieved'], r['avg_launch_us'], r['encoder_fwd_bwd'])"
done
