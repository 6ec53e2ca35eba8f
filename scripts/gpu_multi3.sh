#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/multi3
mkdir -p "$OUT"
cd "$ROOT"
T=tests/native/build/test_kernels
export UNITER_BENCH_SKIP_XCD_CHECK=1
timeout 120 $T --enc > /dev/null 2>&1     # warm the box
for rep in 1 2; do
  for st in 1 0; do
    UNITER_AMD_MULTI_SPLIT_TAIL=$st timeout 160 $T --enc > "$OUT/enc_tail$st.log" 2>&1; echo "split_tail=$st: $(grep 'ENCODER\|FAIL\|deferred' $OUT/enc_tail$st.log | tail -2 | tr '\n' ' ' | cut -c1-260)"
  done
done
grep "wgrad group" "$OUT/enc_tail1.log" "$OUT/enc_tail0.log"
python bench.py --no-cpu-baseline --no-kernel-timing --config c2 --steps 20 --warmup 5 > /dev/null 2>&1
for rep in 1 2; do
for c in c2 c4; do
for v in "1 1" "1 0" "0 0"; do
  set -- $v
  UNITER_AMD_WGRAD_STAGE=$1 UNITER_AMD_DEFER_WGRAD_JOIN=$2 timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --config $c --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stage=$1 deferjoin=$2', '$c', d['ms_per_step'], d['value'])"
done; done; done
