#!/bin/bash
# Round 6, call 19: folded gradient norm with the training loop's DeferWgradJoin hook admitted (call 18 measured a fold that never ran
# in bench.py).  Same-box A/B of the c2 line, segment times, the affected tests.  Output: gpurun_out/r06c19/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06c19
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests -x -q -m gpu -k "folded or lazy or adamw or overlapped or headline or accum or merge" > "$OUT/pytest_first.log" 2>&1; echo "pytest first rc=$?"; tail -3 "$OUT/pytest_first.log"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['ms_per_step'], d['timed_windows']['ms_per_step'], 'fwd/bwd', r['encoder_fwd_bwd']['fwd_ms'], r['encoder_fwd_bwd']['bwd_ms'], 'frac', r['frac'], 'loss', d['final_loss'])"; }
B="timeout 300 python bench.py --no-cpu-baseline --no-traffic --steps 30 --warmup 8"
for rep in 1 2 3; do
  for v in 1 0; do
    UNITER_AMD_FOLD_NORM=$v $B 2>/dev/null | tee "$OUT/c2_fold${v}_$rep.json" | line "c2 fold_norm=$v"
  done
done 2>&1 | tee "$OUT/ab.txt"
for v in 1 0; do echo "fold_norm=$v"; UNITER_AMD_FOLD_NORM=$v python scripts/segment_times.py 2>&1 | tail -8; done | tee "$OUT/segments.txt"
