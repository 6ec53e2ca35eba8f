#!/bin/bash
# attention backward (L <= 128): row term in one pass — harness checks, kernel alone, encoder alone, parity tests
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04attn
mkdir -p "$OUT"
cd "$ROOT"
T=tests/native/build/test_kernels
timeout 600 $T > "$OUT/harness.log" 2>&1; tail -3 "$OUT/harness.log"; grep -c "FAIL" "$OUT/harness.log"
timeout 120 $T --attn 32 96 12 0.1 > "$OUT/attn_alone.log" 2>&1; grep -iE "bwd|fwd" "$OUT/attn_alone.log" | head -8
timeout 300 $T --enc > "$OUT/native_encoder.log" 2>&1; grep -E "ENCODER|FAIL" "$OUT/native_encoder.log"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "headline or attention or c5_large or other_tasks or packed or long_sequences" 2>&1 | tail -12
