#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace + stats of the bench command, then two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE — never combined with other trace domains) for the HBM-traffic figure of bench.py's
# roofline object.  Output under gpurun_out/<tag>/ ; scripts/summarize_profile.py turns it into profiles/<tag>_*.
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-kernel-timing"
# the GEMM tile choices are the shipped ones (uniter_amd/tuned/gfx950.json): every pass measures the same kernels and no
# pass contains tuning sweeps
timeout 300 python $ROOT/bench.py --no-cpu-baseline --steps 30 --warmup 5 > "$OUT/bench.json.log" 2> "$OUT/bench.err.log"
echo "bench rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python $ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 3 > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- $BENCH --steps 3 --warmup 2 > "$OUT/pmc_fetch.log" 2>&1
echo "pmc fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- $BENCH --steps 3 --warmup 2 > "$OUT/pmc_write.log" 2>&1
echo "pmc write rc=$?"
# MFMA utilisation: matrix-pipe busy cycles per kernel (own pass; PMC passes never share a run with trace domains other than
# --kernel-trace).  The harness GEMM of known FLOP count (4096^3 on the eight-phase tile) calibrates the counter's unit.
rocprofv3 -L 2>/dev/null | grep -i "MFMA_BUSY\|GRBM_GUI_ACTIVE\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES" | head -20 > "$OUT/counters_available.txt"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_mfma" -- $BENCH --steps 3 --warmup 2 > "$OUT/pmc_mfma.log" 2>&1
echo "pmc mfma rc=$?"
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_mfma_cal" -- $ROOT/tests/native/build/test_kernels --one fwd 4096 4096 4096 58 1 20 > "$OUT/pmc_mfma_cal.log" 2>&1
echo "pmc mfma calibration rc=$?"
ls -R "$OUT" | head -60
