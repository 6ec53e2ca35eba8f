#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4f
mkdir -p "$OUT"
cd "$ROOT"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541
timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', d['ms_per_step'], d['value'])"
for lpb in 4 12; do
  UNITER_DIST_FORCE=1 UNITER_BENCH_LAYERS_PER_BUCKET=$lpb timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 2> "$OUT/bench_dp_${lpb}.err" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one-rank RCCL layers/bucket $lpb:', d['ms_per_step'], d['value'])"
done
timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', d['ms_per_step'], d['value'])"
timeout 1500 python -m pytest tests -q -m gpu -s > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"; grep -E "parity|conditioning|FAILED|out of tolerance" "$OUT/pytest.log" | head -30
