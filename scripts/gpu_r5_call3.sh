#!/bin/bash
# Round 5, third GPU call: scalar-origin LDS-DMA (plan_kc / plan_ks), loader waves in the epilogue, pipelined fragment reads
# where registers allow, the ws = 5 tiles (one workgroup per CU).  Libraries: A = all on (uniter_amd/csrc/build), B = PIPE off
# (build_b), D = EPI_ALL off (build_d).  Output: gpurun_out/r05c3/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05c3
mkdir -p "$OUT"
cd "$ROOT"
PARTS="${*:-check ab sweep tune}"
T=$ROOT/tests/native/build/test_kernels
export UNITER_TUNED_JSON=$ROOT/uniter_amd/tuned/gfx950.json
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$1:', d['ms_per_step'], 'ms/step', d['value'], 'ex/s', 'windows', (d.get('timed_windows') or {}).get('ms_per_step'), 'fwd+bwd', (r.get('encoder_fwd_bwd') or {}).get('fwd_ms'), (r.get('encoder_fwd_bwd') or {}).get('bwd_ms'), 'frac', r.get('frac'))"; }
stamp() { echo "[$(date +%H:%M:%S)] $*"; }
if [[ $PARTS == *check* ]]; then
  stamp "native harness, library A"
  timeout 900 $T --quick > "$OUT/native_harness.log" 2>&1
  echo "harness rc=$? ; FAIL lines: $(grep -c '^\[FAIL' "$OUT/native_harness.log") ; OK lines: $(grep -c '^\[ OK' "$OUT/native_harness.log")"
  grep '^\[FAIL' "$OUT/native_harness.log" | head -20; tail -2 "$OUT/native_harness.log"
fi
if [[ $PARTS == *ab* ]]; then
  stamp "A / B / D on the chain shapes (shipped tiles) and in the encoder harness"
  {
    for rep in 1 2; do
      for v in A B D; do
        lp=""; [[ $v == B ]] && lp=$ROOT/uniter_amd/csrc/build_b; [[ $v == D ]] && lp=$ROOT/uniter_amd/csrc/build_d
        echo "--- $v roofs, run $rep"; LD_LIBRARY_PATH=$lp timeout 120 $T --roofs 20
      done
    done
    export UNITER_BENCH_XCD_ONLY=1 UNITER_BENCH_SKIP_CHAIN_CHECK=1
    for rep in 1 2; do
      for v in A B D; do
        lp=""; [[ $v == B ]] && lp=$ROOT/uniter_amd/csrc/build_b; [[ $v == D ]] && lp=$ROOT/uniter_amd/csrc/build_d
        echo "--- $v encoder, run $rep"; LD_LIBRARY_PATH=$lp timeout 200 $T --enc 2>&1 | grep -E "in-situ gemm|ENCODER"
      done
    done
    unset UNITER_BENCH_XCD_ONLY UNITER_BENCH_SKIP_CHAIN_CHECK
  } > "$OUT/ab.txt" 2>&1
  grep -E "^---|ROOF|ENCODER" "$OUT/ab.txt"
fi
if [[ $PARTS == *sweep* ]]; then
  stamp "every tile on every chain shape, library A and B"
  timeout 300 $T --sweep 12 > "$OUT/sweep_A.txt" 2>&1; grep BEST "$OUT/sweep_A.txt"
  LD_LIBRARY_PATH=$ROOT/uniter_amd/csrc/build_b timeout 300 $T --sweep 12 > "$OUT/sweep_B.txt" 2>&1; grep BEST "$OUT/sweep_B.txt"
fi
if [[ $PARTS == *tune* ]]; then
  stamp "c2: bench (shipped table), in-situ tuning, bench"
  timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | tee "$OUT/c2_before_tune.json" | line "c2 library A, shipped table"
  cp uniter_amd/tuned/gfx950.json "$OUT/gfx950_before.json"
  timeout 900 python scripts/make_factory_tune.py 3 c2 2>&1 | tail -8
  cp uniter_amd/tuned/gfx950.json "$OUT/gfx950_after_c2.json"
  timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | tee "$OUT/c2_after_tune.json" | line "c2 library A, re-tuned table"
  UNITER_AMD_ADAMW_NT=1 timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | tee "$OUT/c2_after_tune_nt.json" | line "c2 library A, re-tuned, UNITER_AMD_ADAMW_NT=1"
  UNITER_BENCH_XCD_ONLY=1 UNITER_BENCH_SKIP_CHAIN_CHECK=1 timeout 200 $T --enc 2>&1 | grep -E "in-situ|ENCODER" > "$OUT/enc_after_tune.txt"; tail -1 "$OUT/enc_after_tune.txt"
fi
stamp done
