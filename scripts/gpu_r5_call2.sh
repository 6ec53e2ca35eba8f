#!/bin/bash
# Round 5, second GPU call: the software-pipelined fragment reads of the LDS-ring GEMM family (gemm.hip: pipe_step).
#   check   native harness (every tile x layout x epilogue against the host reference; bit-identity screens)
#   ab      the eight chain shapes alone and the 12-layer encoder harness: library A (pipelined, uniter_amd/csrc/build) against
#           library B (UNITER_GEMM_PIPE=0, uniter_amd/csrc/build_b), alternating
#   roofs   PMC passes over the chain shapes with the shipped tiles (the record for profiles/r05_chain_gemm_roofs.json)
#   tests   digests (non-temporal AdamW), merged accumulation c4
#   tune    in-situ tile tuning of c2 with the pipelined kernels, bench before / after
# Output: gpurun_out/r05c2/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05c2
mkdir -p "$OUT"
cd "$ROOT"
PARTS="${*:-check ab roofs tests tune}"
T=$ROOT/tests/native/build/test_kernels
B=$ROOT/uniter_amd/csrc/build_b
export UNITER_TUNED_JSON=$ROOT/uniter_amd/tuned/gfx950.json
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$1:', d['ms_per_step'], 'ms/step', d['value'], 'ex/s', 'windows', (d.get('timed_windows') or {}).get('ms_per_step'), 'fwd+bwd', (r.get('encoder_fwd_bwd') or {}).get('fwd_ms'), (r.get('encoder_fwd_bwd') or {}).get('bwd_ms'), 'frac', r.get('frac'))"; }
stamp() { echo "[$(date +%H:%M:%S)] $*"; }

if [[ $PARTS == *check* ]]; then
  stamp "native harness (pipelined library)"
  timeout 600 $T --quick > "$OUT/native_harness.log" 2>&1
  echo "harness rc=$? ; FAIL lines: $(grep -c '^\[FAIL' "$OUT/native_harness.log") ; OK lines: $(grep -c '^\[ OK' "$OUT/native_harness.log")"
  grep '^\[FAIL' "$OUT/native_harness.log" | head -20
  tail -2 "$OUT/native_harness.log"
fi

if [[ $PARTS == *ab* ]]; then
  stamp "A/B: pipelined (A) vs unpipelined (B) fragment reads"
  {
    for rep in 1 2; do
      echo "--- A (pipelined), run $rep"; timeout 120 $T --roofs 20
      echo "--- B (unpipelined), run $rep"; LD_LIBRARY_PATH=$B timeout 120 $T --roofs 20
    done
    export UNITER_BENCH_XCD_ONLY=1 UNITER_BENCH_SKIP_CHAIN_CHECK=1
    for rep in 1 2; do
      echo "--- A encoder, run $rep"; timeout 200 $T --enc 2>&1 | grep -E "in-situ gemm|ENCODER"
      echo "--- B encoder, run $rep"; LD_LIBRARY_PATH=$B timeout 200 $T --enc 2>&1 | grep -E "in-situ gemm|ENCODER"
    done
    unset UNITER_BENCH_XCD_ONLY UNITER_BENCH_SKIP_CHAIN_CHECK
    # every LDS-ring tile on the two long-K H-wide shapes and on FFN1 (which tile wins now?)
    for cfg in 10 16 25 28 38 31 40 54 55 56 57 50 52 53 20 33 43 44 45 46 47 48 49 51 4 21 5 22; do
      timeout 60 $T --one fwd 3072 768 3072 $cfg 1 20 2>&1 | grep " us "
    done
    for cfg in 0 20 33 43 44 45 46 4 21 47 48 49 50 51 6 7 59; do
      timeout 60 $T --one gelu 3072 3072 768 $cfg 1 20 2>&1 | grep " us "
    done
    for cfg in 0 20 33 43 44 45 46 4 21 47 48 49 50 51 6 7 59; do
      timeout 60 $T --one fwd 3072 2304 768 $cfg 1 20 2>&1 | grep " us "
    done
  } > "$OUT/ab.txt" 2>&1
  grep -E "^---|ROOF|ENCODER" "$OUT/ab.txt"
fi

if [[ $PARTS == *roofs* ]]; then
  stamp "roofs: PMC passes, shipped tiles, pipelined library"
  cd /tmp; export TMPDIR=/tmp
  $T --roofs 20 > "$OUT/roofs_timing.txt" 2>&1
  pass() { local name=$1; shift
    timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/pmc_alone_$name" -- $T --roofs 5 > "$OUT/pmc_alone_$name.log" 2>&1; echo "pass alone $name rc=$?"; }
  pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  pass tcc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
  pass tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum
  pass lds SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  insitu() { local name=$1; shift
    UNITER_BENCH_XCD_ONLY=1 UNITER_BENCH_SKIP_CHAIN_CHECK=1 timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/pmc_insitu_$name" -- $T --enc > "$OUT/pmc_insitu_$name.log" 2>&1; echo "pass in situ $name rc=$?"; }
  insitu tcc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
  insitu sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  cd "$ROOT"
  python scripts/summarize_roofs.py "$OUT" "$OUT/chain_gemm_roofs.json" > /dev/null 2>&1; ls -la "$OUT/chain_gemm_roofs.json"
  find "$OUT" -name "*_agent_info.csv" -delete 2>/dev/null
  find "$OUT" -name "*.csv" -size +200k -exec gzip -f {} \; 2>/dev/null
fi

if [[ $PARTS == *tests* ]]; then
  stamp "digests (non-temporal AdamW), merged accumulation (c4)"
  UNITER_AMD_RUN_EXPERIMENTS=1 UNITER_EXPERIMENTS_QUICK=1 timeout 900 python -m pytest tests/test_experiments_gpu.py -m gpu -q -s -k "non_temporal or (merged and c4)" > "$OUT/pytest_experiments.log" 2>&1
  echo "experiments rc=$?"; grep -E "merged vs|passed|failed|Error" "$OUT/pytest_experiments.log" | tail -8
fi

if [[ $PARTS == *tune* ]]; then
  stamp "c2: bench, in-situ tuning with the pipelined kernels, bench"
  timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | tee "$OUT/c2_before_tune.json" | line "c2 pipelined, shipped table"
  cp uniter_amd/tuned/gfx950.json "$OUT/gfx950_before.json"
  timeout 900 python scripts/make_factory_tune.py 3 c2 2>&1 | tail -8
  cp uniter_amd/tuned/gfx950.json "$OUT/gfx950_after_c2.json"
  timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | tee "$OUT/c2_after_tune.json" | line "c2 pipelined, re-tuned table"
  UNITER_AMD_ADAMW_NT=1 timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | tee "$OUT/c2_after_tune_nt.json" | line "c2 pipelined, re-tuned, UNITER_AMD_ADAMW_NT=1"
fi
stamp done
