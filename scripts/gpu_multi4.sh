#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/multi4
mkdir -p "$OUT"
cd "$ROOT"
T=tests/native/build/test_kernels
timeout 400 $T --quick > "$OUT/quick.log" 2>&1; echo "quick rc=$?"; grep "deferred\|FAIL" "$OUT/quick.log" | cut -c1-250; tail -1 "$OUT/quick.log"
bash scripts/gpu_tune.sh "c2 c4 c5"
cp uniter_amd/tuned/gfx950.json "$OUT/gfx950.json"
