#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/multi
mkdir -p "$OUT"
cd "$ROOT"
T=tests/native/build/test_kernels
export UNITER_BENCH_SKIP_XCD_CHECK=1
timeout 120 $T --enc > /dev/null 2>&1     # warm the box
for rep in 1 2; do
  timeout 160 $T --enc > "$OUT/enc_multi_$rep.log" 2>&1; echo "multi rc=$?"; grep "ENCODER\|FAIL\|deferred" "$OUT/enc_multi_$rep.log" | tail -4
  UNITER_BENCH_NO_STAGE=1 timeout 160 $T --enc > "$OUT/enc_layer_$rep.log" 2>&1; echo "per-layer rc=$?"; grep "ENCODER\|FAIL" "$OUT/enc_layer_$rep.log" | tail -2
done
grep "in-situ" "$OUT/enc_multi_1.log" | tail -16
for c in large large178; do
  timeout 240 $T --enc $c > "$OUT/enc_${c}_multi.log" 2>&1; echo "$c multi: $(grep 'ENCODER\|FAIL\|deferred' $OUT/enc_${c}_multi.log | tail -3)"
  UNITER_BENCH_NO_STAGE=1 timeout 240 $T --enc $c > "$OUT/enc_${c}_layer.log" 2>&1; echo "$c per-layer: $(grep 'ENCODER\|FAIL' $OUT/enc_${c}_layer.log | tail -2)"
done
