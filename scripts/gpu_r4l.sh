#!/bin/bash
# last-arriver finalisation in the embedding backward: parity tests with the new library, then A/B against the base library
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4l
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "golden or base_model_mlm" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"; grep -E "FAILED|^E  " "$OUT/pytest.log" | head
LIB=uniter_amd/csrc/build/libuniter_hip.so
for rep in 1 2 3; do for v in base new; do
  if [ $v = base ]; then cp aux_bin/base/libuniter_hip.so $LIB; else cp aux_bin/base/libuniter_hip_new.so $LIB; fi
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'])"
done; done
cp aux_bin/base/libuniter_hip_new.so $LIB
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541
UNITER_DIST_FORCE=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new, one-rank RCCL', d['ms_per_step'], d['value'])"
