#!/bin/bash
# Round 6, call 6: fragment-order weight probe (go / no-go on FFN1 forward) + the GPU suite after the test fixes.  Output: gpurun_out/r06c6/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06c6
mkdir -p "$OUT"
cd "$ROOT"
export UNITER_TUNED_JSON=$ROOT/uniter_amd/tuned/gfx950.json
for rep in 1 2; do timeout 120 aux_bin/wfrag_probe 20 2>&1 | tee -a "$OUT/wfrag_probe.txt"; echo "rc=$?"; done
timeout 1200 python -m pytest tests -q -m gpu -s > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -3; grep -E "^FAILED|^ERROR" "$OUT/pytest_gpu.log" | cut -c1-300 | head -30
grep -E "headline parity|c4 B=32" "$OUT/pytest_gpu.log" | head
