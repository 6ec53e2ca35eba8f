#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4k
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "nlvr2 or paired or headline or conditioning" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"; grep -E "FAILED|^E  " "$OUT/pytest.log" | head
for rep in 1 2 3; do for f in 0 1; do
  UNITER_AMD_HEAD_GROUP=$f timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('head_group=$f', d['ms_per_step'], d['value'])"
done; done
