#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/g8d
mkdir -p "$OUT"
cd "$ROOT"
T=tests/native/build/test_kernels
timeout 300 $T --quick > "$OUT/quick.log" 2>&1; echo "quick rc=$?"; tail -1 "$OUT/quick.log"
timeout 300 $T --g8 short > "$OUT/short.log" 2>&1; echo "short rc=$?"
grep "FAIL\|TIME" "$OUT/short.log"
export UNITER_BENCH_SKIP_XCD_CHECK=1
for v in A G H I J A; do
  if [ $v = A ]; then J=uniter_amd/tuned/gfx950.json; else J=aux_bin/tune_$v.json; fi
  UNITER_TUNED_JSON=$J timeout 120 $T --enc > "$OUT/enc_$v.log" 2>&1; echo "enc $v rc=$?"
  grep "ENCODER\|FAIL" "$OUT/enc_$v.log" | tail -3
done
