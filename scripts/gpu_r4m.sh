#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04final
mkdir -p "$OUT"
cd "$ROOT"
T=tests/native/build/test_kernels
timeout 300 $T --enc > "$OUT/native_encoder.log" 2>&1; grep -E "ENCODER|overlapped|in-order" "$OUT/native_encoder.log"
timeout 300 $T --enc large > "$OUT/native_encoder_large96.log" 2>&1; grep "ENCODER" "$OUT/native_encoder_large96.log" | tail -1
timeout 300 $T --enc large178 > "$OUT/native_encoder_large178.log" 2>&1; grep "ENCODER" "$OUT/native_encoder_large178.log" | tail -1
