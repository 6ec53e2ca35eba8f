#!/bin/bash
# Round 5, first GPU call (one box, ~15 min): (1) A/B of the three switches round 4 left unmeasured, (2) which roof the eight
# chain GEMM shapes sit on — PMC passes alone (hot operands) and in situ (the 12-layer encoder harness), (3) fill / depth sweeps
# of the deep 192 x 192 tile.  Output under gpurun_out/r05c1/.  Parts: `bash scripts/gpu_r5_call1.sh ab roofs sweep digests merge`.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05c1
mkdir -p "$OUT"
cd "$ROOT"
PARTS="${*:-ab roofs sweep digests merge}"
T=$ROOT/tests/native/build/test_kernels
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$1:', d['ms_per_step'], 'ms/step', d['value'], 'ex/s', 'fwd+bwd', (r.get('encoder_fwd_bwd') or {}).get('fwd_ms'), (r.get('encoder_fwd_bwd') or {}).get('bwd_ms'), 'frac', (r.get('encoder_fwd_bwd') or {}).get('frac'))"; }
stamp() { echo "[$(date +%H:%M:%S)] $*"; }

if [[ $PARTS == *ab* ]]; then
  stamp "A/B: XCD affinity, non-temporal AdamW"
  {
    export UNITER_BENCH_XCD_ONLY=1 UNITER_BENCH_SKIP_CHAIN_CHECK=1
    timeout 200 $T --enc 2>&1 | grep -E "in-situ|ENCODER" > "$OUT/enc_default.txt"; tail -1 "$OUT/enc_default.txt"
    UNITER_AMD_XCD_AFFINITY=1 timeout 200 $T --enc 2>&1 | grep -E "in-situ|ENCODER" > "$OUT/enc_affinity.txt"; tail -1 "$OUT/enc_affinity.txt"
    timeout 200 $T --enc 2>&1 | grep -E "ENCODER" | tail -1
    UNITER_AMD_XCD_AFFINITY=1 timeout 200 $T --enc 2>&1 | grep -E "ENCODER" | tail -1
    unset UNITER_BENCH_XCD_ONLY UNITER_BENCH_SKIP_CHAIN_CHECK
    timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tee "$OUT/c2_default_1.json" | line "c2 default maps"
    UNITER_AMD_XCD_AFFINITY=1 timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tee "$OUT/c2_affinity_1.json" | line "c2 UNITER_AMD_XCD_AFFINITY=1"
    UNITER_AMD_ADAMW_NT=1 timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tee "$OUT/c2_adamw_nt.json" | line "c2 UNITER_AMD_ADAMW_NT=1"
    timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tee "$OUT/c2_default_2.json" | line "c2 default maps (again)"
    echo "--- per-kernel, default (A) vs affinity (B)"; python scripts/compare_bench.py "$OUT/c2_default_1.json" "$OUT/c2_affinity_1.json" | head -30
    timeout 200 python scripts/time_adamw.py 2>/dev/null | tail -1 | sed 's/^/default policy: /'
    UNITER_AMD_ADAMW_NT=1 timeout 200 python scripts/time_adamw.py 2>/dev/null | tail -1 | sed 's/^/UNITER_AMD_ADAMW_NT=1: /'
  } > "$OUT/ab.txt" 2>&1
  cat "$OUT/ab.txt"
fi

if [[ $PARTS == *roofs* ]]; then
  stamp "roofs: PMC passes over the eight chain shapes"
  cd /tmp; export TMPDIR=/tmp
  rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
  $T --roofs 20 > "$OUT/roofs_timing.txt" 2>&1; cat "$OUT/roofs_timing.txt"
  pass() {   # name, counters...   (alone on hot operands)
    local name=$1; shift
    timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/pmc_alone_$name" -- $T --roofs 5 > "$OUT/pmc_alone_$name.log" 2>&1
    echo "pass alone $name rc=$?"
  }
  pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  pass tcc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
  pass tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum
  pass lds SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  pass ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
  insitu() {
    local name=$1; shift
    UNITER_BENCH_XCD_ONLY=1 UNITER_BENCH_SKIP_CHAIN_CHECK=1 timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/pmc_insitu_$name" -- $T --enc > "$OUT/pmc_insitu_$name.log" 2>&1
    echo "pass in situ $name rc=$?"
  }
  insitu tcc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
  insitu sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  insitu fetch FETCH_SIZE
  cd "$ROOT"
  python scripts/summarize_roofs.py "$OUT" > "$OUT/roofs_summary.txt" 2>&1; tail -60 "$OUT/roofs_summary.txt"
  # the raw per-dispatch CSVs are large: keep the counter CSVs only, gzip'd
  find "$OUT" -name "*_agent_info.csv" -delete 2>/dev/null
  find "$OUT" -name "*.csv" -size +200k -exec gzip -f {} \; 2>/dev/null
  du -sh "$OUT"
fi

if [[ $PARTS == *sweep* ]]; then
  stamp "sweeps: fill (tiles) and depth (K tiles) of the deep tiles and of the shipped choices"
  {
    for M in 768 1536 2304 3072; do $T --one gelu $M 3072 768 59 1 20; done          # 192x192: 64 / 128 / 192 / 256 tiles, 12 K tiles
    for K in 768 1536 3072; do $T --one fwd 3072 3072 $K 59 1 20; done                 # 256 tiles, 12 / 24 / 48 K tiles
    for K in 768 1536 3072; do $T --one fwd 768 3072 $K 59 1 20; done                  # 64 tiles
    for M in 768 1536 3072; do $T --one fwd $M 768 3072 38 1 20; done                  # 96x96 (FFN2's tile): 64 / 128 / 256 tiles, 48 K steps
    for K in 768 1536 3072; do $T --one fwd 3072 768 $K 38 1 20; done
    for K in 768 1536 3072; do $T --one fwd 3072 3072 $K 58 1 20; done                 # 256x256: 144 tiles
    $T --one fwd 4096 4096 4096 58 1 10
    $T --one fwd 4096 4096 4096 59 1 10 2>&1 | tail -1
  } 2>&1 | grep -E "us " > "$OUT/sweeps.txt"
  cat "$OUT/sweeps.txt"
fi

if [[ $PARTS == *digests* ]]; then
  stamp "digests: bit-identity of the switches (c2)"
  UNITER_AMD_RUN_EXPERIMENTS=1 UNITER_EXPERIMENTS_QUICK=1 timeout 900 python -m pytest tests/test_experiments_gpu.py -m gpu -q -s -k "c2 or non_temporal" > "$OUT/pytest_digests.log" 2>&1
  echo "digests rc=$?"; tail -5 "$OUT/pytest_digests.log"
fi

if [[ $PARTS == *merge* ]]; then
  stamp "merged accumulation: c4 gradient test + A/B lines"
  UNITER_AMD_RUN_EXPERIMENTS=1 timeout 900 python -m pytest tests/test_experiments_gpu.py -m gpu -q -s -k "merged and c4" > "$OUT/pytest_merge.log" 2>&1
  echo "merge test rc=$?"; tail -5 "$OUT/pytest_merge.log"
  {
    for c in c4 c5; do
      timeout 400 python bench.py --config $c --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tee "$OUT/${c}_loop.json" | line "$c accumulation loop"
      timeout 400 python bench.py --config $c --no-cpu-baseline --steps 8 --warmup 2 --merge-accum 2>"$OUT/merge_$c.err" | tee "$OUT/${c}_merged.json" | line "$c --merge-accum"
    done
  } > "$OUT/merge_accum_ab.txt" 2>&1
  cat "$OUT/merge_accum_ab.txt"
fi
stamp done
