#!/bin/bash
# Second GPU pass: the 192x192 three-phase tile next to the 256x256 one; encoder A/B by tile table.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/g8b
mkdir -p "$OUT"
cd "$ROOT"
T=tests/native/build/test_kernels
timeout 300 $T --quick > "$OUT/quick.log" 2>&1; echo "quick rc=$?"
grep -c "\[ OK \]" "$OUT/quick.log"; grep "FAIL" "$OUT/quick.log" | head -20; tail -2 "$OUT/quick.log"
timeout 600 $T --g8 quick > "$OUT/g8.log" 2>&1; echo "g8 rc=$?"
grep "FAIL\|TIME\|failed" "$OUT/g8.log" | head -100
export UNITER_BENCH_SKIP_XCD_CHECK=1
for v in A G H I J; do
  if [ $v = A ]; then J=uniter_amd/tuned/gfx950.json; else J=aux_bin/tune_$v.json; fi
  UNITER_TUNED_JSON=$J timeout 120 $T --enc > "$OUT/enc_$v.log" 2>&1; echo "enc $v rc=$?"
  grep "ENCODER\|FAIL" "$OUT/enc_$v.log" | tail -3
done
