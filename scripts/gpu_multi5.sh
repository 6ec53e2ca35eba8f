#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/multi5
mkdir -p "$OUT"
cd "$ROOT"
T=tests/native/build/test_kernels
timeout 400 $T --quick > "$OUT/quick.log" 2>&1; echo "quick rc=$?"; grep "deferred\|FAIL" "$OUT/quick.log" | cut -c1-250; tail -1 "$OUT/quick.log"
export UNITER_BENCH_SKIP_XCD_CHECK=1
for rep in 1 2; do
  timeout 160 $T --enc > "$OUT/enc_$rep.log" 2>&1; echo "enc: $(grep 'ENCODER\|FAIL' $OUT/enc_$rep.log | tail -2 | cut -c1-120)"
done
grep "in-situ" "$OUT/enc_1.log" | tail -15
timeout 1200 python -m pytest tests -x -q -m gpu > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
for rep in 1 2; do for c in c2 c4; do
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --config $c --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', d['ms_per_step'], d['value'])"
done; done
