#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
for rep in 1 2 3; do for f in 0 1; do
  UNITER_AMD_EMB_FUSED=$f timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('emb_fused=$f', d['ms_per_step'], d['value'])"
done; done
