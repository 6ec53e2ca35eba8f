#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4b
mkdir -p "$OUT"
cd "$ROOT"
for w in 3 10 25; do timeout 120 aux_bin/anyorder_probe 100 $w >> "$OUT/anyorder_probe.log" 2>&1; echo "probe rc=$?"; done; cat "$OUT/anyorder_probe.log"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "headline or c5_large or conditioning" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"; grep -E "parity|conditioning|FAILED|out of tolerance|^E  " "$OUT/pytest.log" | head -60
