#!/bin/bash
# Round 4, first call: any-order launch probe, native harness, the GPU test suite (-s: parity prints), one bench line.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4a
mkdir -p "$OUT"
cd "$ROOT"
timeout 120 aux_bin/anyorder_probe 120 6 > "$OUT/anyorder_probe.log" 2>&1; echo "probe rc=$?"; cat "$OUT/anyorder_probe.log"
timeout 600 tests/native/build/test_kernels > "$OUT/harness.log" 2>&1; echo "harness rc=$?"; grep -c "^\[ OK \]" "$OUT/harness.log"; grep "FAIL" "$OUT/harness.log" | head -5; tail -1 "$OUT/harness.log"
timeout 1500 python -m pytest tests -q -m gpu -s > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"; grep -E "parity|conditioning|FAILED|out of tolerance" "$OUT/pytest.log" | head -40
timeout 400 python bench.py --steps 20 --warmup 5 > "$OUT/bench_c2.json.log" 2> "$OUT/bench_c2.err"; echo "bench rc=$?"
tail -1 "$OUT/bench_c2.json.log" | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('c2', d['ms_per_step'], d['value'], r['achieved'], r['avg_launch_us'], r['encoder_fwd_bwd']); print(d.get('parity')); print(d.get('cpu_baseline'))"
