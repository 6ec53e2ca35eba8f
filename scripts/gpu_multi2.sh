#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
python bench.py --no-cpu-baseline --no-kernel-timing --config c2 --steps 20 --warmup 5 > /dev/null 2>&1   # warm the box
for rep in 1 2; do
for c in c2 c4 c5 c3; do
for m in 1 0; do
  UNITER_AMD_WGRAD_STAGE=$m timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --config $c --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stage=$m', '$c', d['ms_per_step'], d['value'])"
done; done; done
