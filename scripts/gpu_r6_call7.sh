#!/bin/bash
# Round 6, call 7: fused QKV projection + attention forward — bit-identity against the two launches, timing alone, encoder harness and
# the c2 bench line with / without it (UNITER_AMD_FUSED_QKV_ATTN).  Output: gpurun_out/r06c7/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06c7
mkdir -p "$OUT"
cd "$ROOT"
T=$ROOT/tests/native/build/test_kernels
export UNITER_TUNED_JSON=$ROOT/uniter_amd/tuned/gfx950.json
timeout 120 $T --qkvattn 32 12 0.1 2>&1 | tee "$OUT/qkvattn.txt"
timeout 120 $T --qkvattn 32 12 0.0 2>&1 | tee -a "$OUT/qkvattn.txt"
timeout 120 $T --qkvattn 32 16 0.1 2>&1 | tee -a "$OUT/qkvattn.txt"
timeout 120 $T --qkvattn 5 4 0.2 2>&1 | tee -a "$OUT/qkvattn.txt"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['ms_per_step'], d['timed_windows']['ms_per_step'], 'fwd/bwd', r['encoder_fwd_bwd']['fwd_ms'], r['encoder_fwd_bwd']['bwd_ms'], 'frac', r['frac'], 'loss', d['final_loss'])"; }
for rep in 1 2; do
  for v in 1 0; do
    UNITER_AMD_FUSED_QKV_ATTN=$v UNITER_BENCH_XCD_ONLY=1 UNITER_BENCH_SKIP_CHAIN_CHECK=1 timeout 200 $T --enc 2>&1 | grep -E "ENCODER" | sed "s/^/fused=$v /" | tee -a "$OUT/enc_ab.txt"
    UNITER_AMD_FUSED_QKV_ATTN=$v timeout 300 python bench.py --no-cpu-baseline --no-traffic --steps 30 --warmup 8 2>/dev/null | tee "$OUT/c2_fused${v}_$rep.json" | line "c2 fused=$v"
  done
done 2>&1 | tee "$OUT/ab.txt"
timeout 900 $T --quick > "$OUT/native_harness.log" 2>&1; echo "harness rc=$? FAIL lines: $(grep -c '^\[FAIL' "$OUT/native_harness.log")"; grep -E "^\[FAIL|fused qkv" "$OUT/native_harness.log" | head; tail -1 "$OUT/native_harness.log"
