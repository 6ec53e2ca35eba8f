#!/bin/bash
# One-rank RCCL group: bucket size x (compute stream joins the weight-gradient stream between ranges, or not)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541
timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', d['ms_per_step'], d['value'])"
for lpb in 3 4 6 12; do for dj in 1 0; do
  UNITER_DIST_FORCE=1 UNITER_BENCH_LAYERS_PER_BUCKET=$lpb UNITER_AMD_DEFER_JOIN=$dj timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('layers/bucket $lpb defer_join $dj:', d['ms_per_step'], d['value'])"
done; done
