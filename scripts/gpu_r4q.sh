#!/bin/bash
# forward / data-gradient GEMMs as two K slices combined inside the launch: harness checks, then encoder alone with a fresh
# autotune, candidates with and without the two-slice form
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04splitk
mkdir -p "$OUT"
cd "$ROOT"
T=tests/native/build/test_kernels
timeout 900 $T > "$OUT/harness.log" 2>&1; tail -1 "$OUT/harness.log"; grep "FAIL" "$OUT/harness.log" | head -10
for sk in 1 0 1; do
  UNITER_AMD_SPLITK2=$sk UNITER_TUNED_JSON=/nonexistent UNITER_BENCH_SKIP_CHAIN_CHECK=1 timeout 600 $T --enc > "$OUT/enc_autotune_sk$sk.log" 2>&1
  echo "two-slice candidates: $sk"; grep -E "ENCODER|in-order" "$OUT/enc_autotune_sk$sk.log"
done
grep -E "gemm (fwd|dgrad)" "$OUT/enc_autotune_sk1.log" | head -20
