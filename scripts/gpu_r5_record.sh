#!/bin/bash
# Round 5 record call: (1) scripts/profile_round.sh r05 (bench line, kernel trace + stats, FETCH / WRITE / MFMA PMC passes of the
# bench command), (2) PMC passes over the eight chain shapes alone and in the encoder harness (-> profiles/r05_chain_gemm_roofs.json),
# (3) the bench line with the cpu_baseline leg as the driver runs it.  Output: gpurun_out/r05/ and gpurun_out/r05roofs/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
bash scripts/profile_round.sh r05
OUT=$ROOT/gpurun_out/r05roofs; mkdir -p "$OUT"
T=$ROOT/tests/native/build/test_kernels
export UNITER_TUNED_JSON=$ROOT/uniter_amd/tuned/gfx950.json
cd /tmp; export TMPDIR=/tmp
$T --roofs 20 > "$OUT/roofs_timing.txt" 2>&1; cat "$OUT/roofs_timing.txt"
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
insitu fetch FETCH_SIZE
insitu write WRITE_SIZE
cd "$ROOT"
python scripts/summarize_roofs.py "$OUT" "$OUT/chain_gemm_roofs.json" > /dev/null 2>&1; ls -la "$OUT/chain_gemm_roofs.json"
find "$ROOT/gpurun_out/r05" "$OUT" -name "*_agent_info.csv" -delete 2>/dev/null
find "$ROOT/gpurun_out/r05" "$OUT" -name "*.csv" -size +300k -exec gzip -f {} \; 2>/dev/null
UNITER_BENCH_XCD_ONLY=1 UNITER_BENCH_SKIP_CHAIN_CHECK=1 timeout 200 $T --enc 2>&1 | grep -E "in-situ|ENCODER" > "$ROOT/gpurun_out/r05/native_encoder.txt"; tail -1 "$ROOT/gpurun_out/r05/native_encoder.txt"
timeout 600 python bench.py --steps 20 --warmup 5 > "$ROOT/gpurun_out/r05/bench_record.json.log" 2> "$ROOT/gpurun_out/r05/bench_record.err"; tail -c 600 "$ROOT/gpurun_out/r05/bench_record.json.log"
du -sh "$ROOT/gpurun_out/r05" "$OUT"
