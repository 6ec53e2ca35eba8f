#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541
for rep in 1 2; do for cfg in "0,128,0" "64,192,64" "96,128,32" "128,128,0"; do
  UNITER_AMD_MULTI_STAGGER=$cfg timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stagger=$cfg', d['ms_per_step'], d['value'])"
done; done
for cfg in "0,128,0" "64,192,64"; do
  UNITER_DIST_FORCE=1 UNITER_AMD_MULTI_STAGGER=$cfg timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one-rank RCCL stagger=$cfg', d['ms_per_step'], d['value'])"
done
