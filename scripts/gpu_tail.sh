#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/tail
mkdir -p "$OUT"
cd "$ROOT"
T=tests/native/build/test_kernels
export UNITER_BENCH_SKIP_XCD_CHECK=1
timeout 160 $T --enc > /dev/null 2>&1
for rep in 1 2; do for sp in 1 0; do
  UNITER_AMD_MULTI_TAIL_SPLIT=$sp timeout 160 $T --enc > "$OUT/enc_${sp}_$rep.log" 2>&1; echo "split=$sp: $(grep 'ENCODER\|FAIL\|deferred' $OUT/enc_${sp}_$rep.log | tail -2 | cut -c1-150 | tr '\n' ' ') $(grep 'wgrad group' $OUT/enc_${sp}_$rep.log | tail -1)"
done; done
UNITER_AMD_MULTI_TAIL_SPLIT=1 timeout 300 $T --enc large > "$OUT/enc_large.log" 2>&1; grep 'ENCODER\|FAIL\|deferred' $OUT/enc_large.log | tail -2 | cut -c1-150
timeout 300 $T --quick 2>&1 | grep "FAIL\|failed" | head -5
