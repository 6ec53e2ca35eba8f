#!/bin/bash
# Round 6, call 17: lazy zero_grad (AdamW.lazy_zero: the fused step leaves the encoder's parameter gradients alone, the next backward
# replaces them — uniter_encoder_set_grad_overwrite / uniter_adamw_plan_keep_grads).  Optimizer / headline / accumulation tests first,
# then the whole GPU suite, then same-box A/B of the c2 line (UNITER_AMD_LAZY_ZERO=0/1) and of c4 (gradient accumulation 4).
# Output: gpurun_out/r06c17/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06c17
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests -x -q -m gpu -k "lazy or adamw or overlapped or headline or accum or determinism or merge" > "$OUT/pytest_first.log" 2>&1; echo "pytest first rc=$?"; tail -3 "$OUT/pytest_first.log"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['ms_per_step'], d['timed_windows']['ms_per_step'], 'fwd/bwd', r['encoder_fwd_bwd']['fwd_ms'], r['encoder_fwd_bwd']['bwd_ms'], 'frac', r['frac'], 'loss', d['final_loss'])"; }
B="timeout 300 python bench.py --no-cpu-baseline --no-traffic --steps 30 --warmup 8"
for rep in 1 2 3; do
  for v in 1 0; do
    UNITER_AMD_LAZY_ZERO=$v $B 2>/dev/null | tee "$OUT/c2_lazy${v}_$rep.json" | line "c2 lazy_zero=$v"
  done
done 2>&1 | tee "$OUT/ab.txt"
for v in 1 0; do
  UNITER_AMD_LAZY_ZERO=$v timeout 400 python bench.py --config c4 --no-cpu-baseline --no-traffic --steps 10 --warmup 3 2>/dev/null | line "c4 lazy_zero=$v"
done 2>&1 | tee -a "$OUT/ab.txt"
timeout 1500 python -m pytest tests -x -q -m gpu > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
