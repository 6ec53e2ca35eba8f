#!/bin/bash
# DP single-launch path: rccl tests, one-rank RCCL step A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4d
mkdir -p "$OUT"
cd "$ROOT"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541
timeout 600 python -m pytest tests/test_rccl_gpu.py -q -m gpu -x > "$OUT/pytest_rccl.log" 2>&1; echo "pytest rccl rc=$?"; tail -3 "$OUT/pytest_rccl.log"
timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', d['ms_per_step'], d['value'])"
for sl in 1; do for lpb in 4 12; do
  UNITER_DIST_FORCE=1 UNITER_AMD_DP_SINGLE_LAUNCH=$sl UNITER_BENCH_LAYERS_PER_BUCKET=$lpb timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 2> "$OUT/bench_dp_${sl}_${lpb}.err" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one-rank RCCL single_launch=$sl layers/bucket $lpb:', d['ms_per_step'], d['value'])"
done; done
UNITER_DIST_FORCE=1 UNITER_AMD_DP_SPARSE_WORD=0 UNITER_BENCH_LAYERS_PER_BUCKET=4 timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one-rank RCCL single launch, dense word table:', d['ms_per_step'], d['value'])"
