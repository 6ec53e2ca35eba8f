#!/bin/bash
# In-situ tuning with the deep-pipelined tiles among the candidates; before/after bench lines per configuration.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/tune
mkdir -p "$OUT"
cd "$ROOT"
for c in ${1:-c5 c4 c2}; do
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --config $c --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('before', '$c', d['ms_per_step'], d['value'])"
  timeout 900 python scripts/make_factory_tune.py 3 $c 2>&1 | tail -8
  cp uniter_amd/tuned/gfx950.json "$OUT/gfx950_after_$c.json"
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --config $c --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('after', '$c', d['ms_per_step'], d['value'])"
done
cp uniter_amd/tuned/gfx950.json "$OUT/gfx950.json"
