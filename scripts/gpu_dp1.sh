#!/bin/bash
# One-rank RCCL group (UNITER_DIST_FORCE=1): the data-parallel path's bucket hooks, collectives and joins on one GPU.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541
for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', d['ms_per_step'], d['value'])"
  UNITER_DIST_FORCE=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 2>gpurun_out/dp1.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one-rank RCCL group', d['ms_per_step'], d['value'], d['config'].get('parallelism'))"
done
tail -3 gpurun_out/dp1.err
