#!/bin/bash
# Round 5, fourth GPU call: the round's GEMM family (scalar-origin DMA, loader waves in the epilogue, pipelined fragment reads,
# three deeper 8 + 4 tiles) — harness, alone sweep, in-situ tuning of c2 / c4 / c5, merged accumulation c3, bench lines.
# Output: gpurun_out/r05c4/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05c4
mkdir -p "$OUT"
cd "$ROOT"
PARTS="${*:-check sweep tune tests lines}"
T=$ROOT/tests/native/build/test_kernels
export UNITER_TUNED_JSON=$ROOT/uniter_amd/tuned/gfx950.json
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$1:', d['ms_per_step'], 'ms/step', d['value'], 'ex/s', 'windows', (d.get('timed_windows') or {}).get('ms_per_step'), 'fwd+bwd', (r.get('encoder_fwd_bwd') or {}).get('fwd_ms'), (r.get('encoder_fwd_bwd') or {}).get('bwd_ms'), 'frac', r.get('frac'))"; }
stamp() { echo "[$(date +%H:%M:%S)] $*"; }
if [[ $PARTS == *check* ]]; then
  stamp "native harness"
  timeout 900 $T --quick > "$OUT/native_harness.log" 2>&1
  echo "harness rc=$? ; FAIL lines: $(grep -c '^\[FAIL' "$OUT/native_harness.log") ; OK lines: $(grep -c '^\[ OK' "$OUT/native_harness.log")"
  grep '^\[FAIL' "$OUT/native_harness.log" | head -20; tail -2 "$OUT/native_harness.log"
fi
if [[ $PARTS == *sweep* ]]; then
  stamp "every tile on every chain shape"
  timeout 300 $T --sweep 12 > "$OUT/sweep.txt" 2>&1; grep BEST "$OUT/sweep.txt"
  timeout 120 $T --roofs 20 | tee "$OUT/roofs_shipped_table.txt"
fi
if [[ $PARTS == *tune* ]]; then
  stamp "c2: bench (shipped table), in-situ tuning, bench"
  timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 8 2>"$OUT/c2_before.err" | tee "$OUT/c2_before_tune.json" | line "c2 shipped table"
  cp uniter_amd/tuned/gfx950.json "$OUT/gfx950_before.json"
  for c in c2 c4 c5; do
    timeout 1200 python scripts/make_factory_tune.py 3 $c > "$OUT/tune_$c.log" 2>&1; tail -6 "$OUT/tune_$c.log"
    cp uniter_amd/tuned/gfx950.json "$OUT/gfx950_after_$c.json"
  done
  timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 8 2>"$OUT/c2_after.err" | tee "$OUT/c2_after_tune.json" | line "c2 re-tuned table"
  UNITER_BENCH_XCD_ONLY=1 UNITER_BENCH_SKIP_CHAIN_CHECK=1 timeout 200 $T --enc 2>&1 | grep -E "in-situ|ENCODER" > "$OUT/enc_after_tune.txt"; tail -1 "$OUT/enc_after_tune.txt"
fi
if [[ $PARTS == *tests* ]]; then
  stamp "merged accumulation c3 + c4 on the GPU"
  timeout 900 python -m pytest tests/test_merge_accumulation_gpu.py -m gpu -q -s > "$OUT/pytest_merge.log" 2>&1
  echo "merge tests rc=$?"; grep -E "merged vs|passed|failed|Error" "$OUT/pytest_merge.log" | tail -12
fi
if [[ $PARTS == *lines* ]]; then
  stamp "c3 / c4 / c5 lines, loop and merged"
  for c in c3 c4 c5; do
    timeout 400 python bench.py --config $c --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tee "$OUT/${c}_loop.json" | line "$c accumulation loop"
    timeout 400 python bench.py --config $c --no-cpu-baseline --steps 8 --warmup 2 --merge-accum 2>/dev/null | tee "$OUT/${c}_merged.json" | line "$c --merge-accum"
  done
fi
stamp done
