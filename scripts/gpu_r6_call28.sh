#!/bin/bash
# Round 6, call 28: UNITER_GEMM_PIPE2 (gemm.hip: fragment reads a whole K tile ahead of their MFMAs, a second register set; variant
# build in uniter_amd/csrc/build_q2) against the shipped build: native harness on the variant (bit-identity / tolerance checks), the
# chain shapes alone, the encoder harness and the c2 line, alternating.  Output: gpurun_out/r06c28/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06c28
mkdir -p "$OUT"
cd "$ROOT"
T=$ROOT/tests/native/build/test_kernels
export UNITER_TUNED_JSON=$ROOT/uniter_amd/tuned/gfx950.json
LD_LIBRARY_PATH=$ROOT/uniter_amd/csrc/build_q2 timeout 600 $T > "$OUT/harness_q2.log" 2>&1; echo "harness (variant) rc=$?"; grep -c "^\[ OK \]" "$OUT/harness_q2.log"; grep FAIL "$OUT/harness_q2.log" | head -5; tail -1 "$OUT/harness_q2.log"
for rep in 1 2; do
  for v in build build_q2; do
    echo "== $v (rep $rep) =="
    LD_LIBRARY_PATH=$ROOT/uniter_amd/csrc/$v timeout 200 $T --roofs 20 2>&1 | grep ROOF
    LD_LIBRARY_PATH=$ROOT/uniter_amd/csrc/$v UNITER_BENCH_XCD_ONLY=1 UNITER_BENCH_SKIP_CHAIN_CHECK=1 timeout 200 $T --enc 2>&1 | grep -E "ENCODER"
  done
done > "$OUT/roofs_ab.txt" 2>&1; cat "$OUT/roofs_ab.txt"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]["encoder_fwd_bwd"]; print(sys.argv[1], d["ms_per_step"], d["timed_windows"]["ms_per_step"], "fwd/bwd", r["fwd_ms"], r["bwd_ms"], "loss", d["final_loss"])'
for rep in 1 2 3; do
  for v in build build_q2; do
    UNITER_AMD_LIB=$ROOT/uniter_amd/csrc/$v/libuniter_hip.so timeout 300 python bench.py --no-cpu-baseline --no-traffic --steps 30 --warmup 8 2>/dev/null | python -c "$P" "c2 $v"
  done
done 2>&1 | tee "$OUT/ab.txt"
