#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/gelu
mkdir -p "$OUT"
cd "$ROOT"
T=tests/native/build/test_kernels
timeout 600 $T > "$OUT/harness.log" 2>&1; echo "harness rc=$?"; grep -c "^\[ OK \]" "$OUT/harness.log"; grep "FAIL" "$OUT/harness.log" | head -5; tail -1 "$OUT/harness.log"
timeout 300 $T --g8 short > "$OUT/g8_short.log" 2>&1; echo "g8 short rc=$?"; grep "TIME\|FAIL" "$OUT/g8_short.log" | cut -c1-200
export UNITER_BENCH_SKIP_XCD_CHECK=1
for rep in 1 2; do
  timeout 160 $T --enc > "$OUT/enc_$rep.log" 2>&1; echo "enc: $(grep 'ENCODER\|FAIL' $OUT/enc_$rep.log | tail -2 | cut -c1-140)"
done
grep "in-situ" "$OUT/enc_1.log" | tail -14
