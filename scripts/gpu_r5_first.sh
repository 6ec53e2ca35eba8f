#!/bin/bash
# Round 5, first GPU call: validate and measure what round 4's second session built without GPU minutes.
#   a) tests/test_experiments_gpu.py: XCD affinity leaves two optimizer steps bit-identical (c2, c3); merged micro-batches give the
#      accumulation loop's gradients (c3, c4)
#   b) c2 A/B of UNITER_AMD_XCD_AFFINITY, alternating, one box; AdamW alone and in the step with non-temporal streams
#   c) c3 / c4 / c5 with and without --merge-accum
# Output under gpurun_out/r05first.  ~12 GPU minutes.  `bash scripts/gpu_r5_first.sh a` runs one part only.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05first
mkdir -p "$OUT"
cd "$ROOT"
PART=${1:-abc}
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$1:', d['ms_per_step'], 'ms/step', d['value'], 'ex/s', 'fwd+bwd frac', (r.get('encoder_fwd_bwd') or {}).get('frac'), 'step frac', (r.get('step') or {}).get('frac'))"; }
if [[ $PART == *a* ]]; then
  UNITER_AMD_RUN_EXPERIMENTS=1 timeout 1500 python -m pytest tests/test_experiments_gpu.py -m gpu -q -s > "$OUT/pytest_experiments.log" 2>&1
  echo "experiments rc=$?"; grep -E "identical|merged vs|passed|failed|Error" "$OUT/pytest_experiments.log" | tail -20
fi
if [[ $PART == *b* ]]; then
  {
    for rep in 1 2; do
      timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tee "$OUT/c2_default_$rep.json" | line "c2 default maps (run $rep)"
      UNITER_AMD_XCD_AFFINITY=1 timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tee "$OUT/c2_affinity_$rep.json" | line "c2 UNITER_AMD_XCD_AFFINITY=1 (run $rep)"
    done
    echo "--- per-kernel, default (A) vs affinity (B), second pair"; python scripts/compare_bench.py "$OUT/c2_default_2.json" "$OUT/c2_affinity_2.json" | head -30
    timeout 200 python scripts/time_adamw.py 2>/dev/null | tail -1 | sed 's/^/default policy: /'
    UNITER_AMD_ADAMW_NT=1 timeout 200 python scripts/time_adamw.py 2>/dev/null | tail -1 | sed 's/^/UNITER_AMD_ADAMW_NT=1: /'
    UNITER_AMD_ADAMW_NT=1 timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | line "c2 UNITER_AMD_ADAMW_NT=1"
    T=tests/native/build/test_kernels
    timeout 300 $T --enc 2>&1 | grep -E "ENCODER" | tail -1
    UNITER_AMD_XCD_AFFINITY=1 timeout 300 $T --enc 2>&1 | grep -E "ENCODER" | tail -1
  } > "$OUT/xcd_affinity_ab.txt" 2>&1
  cat "$OUT/xcd_affinity_ab.txt"
fi
if [[ $PART == *c* ]]; then
  {
    for c in c3 c4 c5; do
      timeout 400 python bench.py --config $c --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | line "$c accumulation loop"
      timeout 400 python bench.py --config $c --no-cpu-baseline --steps 10 --warmup 3 --merge-accum 2>"$OUT/merge_$c.err" | line "$c --merge-accum"
    done
  } > "$OUT/merge_accum_ab.txt" 2>&1
  cat "$OUT/merge_accum_ab.txt"
fi
