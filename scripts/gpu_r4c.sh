#!/bin/bash
# overlapped chains: native harness (default checks), --enc (bit identity + A/B timing), the chain / c5 GPU tests, bench A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4c
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 tests/native/build/test_kernels --enc > "$OUT/enc.log" 2>&1; echo "enc rc=$?"; grep -E "^\[|in-order|ENCODER|overlapped" "$OUT/enc.log" | head -20
timeout 600 tests/native/build/test_kernels > "$OUT/harness.log" 2>&1; echo "harness rc=$?"; grep -c "^\[ OK \]" "$OUT/harness.log"; grep "FAIL" "$OUT/harness.log" | head -5; tail -1 "$OUT/harness.log"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "overlapped or c5_large or determinism or deferred_parameter" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"; grep -E "parity|FAILED|out of tolerance|^E  " "$OUT/pytest.log" | head -40
for chain in 0 1; do
  UNITER_AMD_CHAIN=$chain timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('c2 chain=$chain', d['ms_per_step'], d['value'], r['achieved'], r['avg_launch_us'], r['encoder_fwd_bwd'])"
done
