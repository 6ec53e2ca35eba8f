#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
for cfg in "0,128,0" "64,192,64" "0,0,0" "0,128,0" "64,192,64"; do
  UNITER_BENCH_SKIP_CHAIN_CHECK=1 UNITER_AMD_MULTI_STAGGER=$cfg timeout 200 tests/native/build/test_kernels --enc 2>&1 | grep -E "in-situ gemm wgrad group|ENCODER" | tr '\n' ' '; echo " <- stagger $cfg"
done
for rep in 1 2 3; do for cfg in "0,128,0" "64,192,64"; do
  UNITER_AMD_MULTI_STAGGER=$cfg timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stagger=$cfg', d['ms_per_step'], d['value'])"
done; done
