#!/bin/bash
# Same-box A/B of tile tables inside the 12-layer encoder harness and the bench step: A = shipped, B = sweep-best alone (all eight
# chain shapes), C = sweep-best forward only.  Alternating, three rounds.  Output: gpurun_out/r05tab/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05tab; mkdir -p "$OUT"; cd "$ROOT"
T=$ROOT/tests/native/build/test_kernels
export UNITER_BENCH_XCD_ONLY=1 UNITER_BENCH_SKIP_CHAIN_CHECK=1
for rep in 1 2 3; do
  for v in A B C; do
    tj=$ROOT/uniter_amd/tuned/gfx950.json; [[ $v == B ]] && tj=$ROOT/scripts/tables/table_sweep_best.json; [[ $v == C ]] && tj=$ROOT/scripts/tables/table_fwd_best.json
    echo "--- $v run $rep"; UNITER_TUNED_JSON=$tj timeout 200 $T --enc 2>&1 | grep -E "ENCODER"
  done
done | tee "$OUT/enc_tables.txt"
unset UNITER_BENCH_XCD_ONLY UNITER_BENCH_SKIP_CHAIN_CHECK
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$1:', d['ms_per_step'], 'ms/step', 'windows', (d.get('timed_windows') or {}).get('ms_per_step'), 'fwd+bwd', (r.get('encoder_fwd_bwd') or {}).get('fwd_ms'), (r.get('encoder_fwd_bwd') or {}).get('bwd_ms'))"; }
for rep in 1 2; do
  for v in A B C; do
    tj=$ROOT/uniter_amd/tuned/gfx950.json; [[ $v == B ]] && tj=$ROOT/scripts/tables/table_sweep_best.json; [[ $v == C ]] && tj=$ROOT/scripts/tables/table_fwd_best.json
    UNITER_AMD_FACTORY_TUNE=0 UNITER_AMD_TUNE_CACHE=$tj timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | line "c2 table $v run $rep"
  done
done | tee "$OUT/bench_tables.txt"
