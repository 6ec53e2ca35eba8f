#!/bin/bash
# LayerNorm tails: bit-identity against separate launches and timing, encoder alone; row blocks on one XCD (local) or spread
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04tail
mkdir -p "$OUT"
cd "$ROOT"
T=tests/native/build/test_kernels
for loc in 1 0 1; do
  UNITER_AMD_LN_TAIL_LOCAL=$loc timeout 300 $T --enc > "$OUT/native_encoder_local$loc.log" 2>&1; echo "local=$loc"; grep -E "ENCODER|LayerNorm|FAIL" "$OUT/native_encoder_local$loc.log"
done
UNITER_AMD_LN_TAIL_LOCAL=1 timeout 300 $T --enc large > "$OUT/native_encoder_large96.log" 2>&1; grep -E "ENCODER|LayerNorm tails|FAIL" "$OUT/native_encoder_large96.log" | tail -4
UNITER_AMD_LN_TAIL_LOCAL=1 timeout 300 $T --enc large178 > "$OUT/native_encoder_large178.log" 2>&1; grep -E "ENCODER|LayerNorm tails|FAIL" "$OUT/native_encoder_large178.log" | tail -4
