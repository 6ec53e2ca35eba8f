#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4g
mkdir -p "$OUT"
cd "$ROOT"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "golden or base_model_mlm" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"; grep -E "FAILED|out of tolerance|^E  " "$OUT/pytest.log" | head -30
for f in 0 1 0 1; do
  UNITER_AMD_EMB_FUSED=$f timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('emb_fused=$f', d['ms_per_step'], d['value'])"
done
UNITER_DIST_FORCE=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one-rank RCCL:', d['ms_per_step'], d['value'])"
