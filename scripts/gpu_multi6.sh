#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/multi6
mkdir -p "$OUT"
cd "$ROOT"
T=tests/native/build/test_kernels
timeout 400 $T --quick > "$OUT/quick.log" 2>&1; echo "quick rc=$?"; grep "deferred\|FAIL" "$OUT/quick.log" | cut -c1-200 | tail -12; tail -1 "$OUT/quick.log"
export UNITER_BENCH_SKIP_XCD_CHECK=1
for rep in 1 2; do
  timeout 160 $T --enc > "$OUT/enc_$rep.log" 2>&1; echo "enc: $(grep 'ENCODER\|FAIL' $OUT/enc_$rep.log | tail -2 | cut -c1-140)"
done
grep "in-situ\|wgrad group" "$OUT/enc_1.log" | tail -16
for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('c2', d['ms_per_step'], d['value'], r['kernel'][:60], r['achieved'], r['avg_launch_us'])"
done
