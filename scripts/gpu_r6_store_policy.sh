#!/bin/bash
# Round 6: output-store policy of the gemm_tile / LayerNorm / attention kernels (common.cuh UNITER_STORE_POLICY): nt (shipped) vs default
# vs sc1 write-through, three builds, alternating on the c2 line and the encoder harness.  Output: gpurun_out/r06sp/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06sp
mkdir -p "$OUT"
cd "$ROOT"
T=$ROOT/tests/native/build/test_kernels
export UNITER_TUNED_JSON=$ROOT/uniter_amd/tuned/gfx950.json
for rep in 1 2; do
  for v in build build_b build_d; do
    L=$ROOT/uniter_amd/csrc/$v
    LD_LIBRARY_PATH=$L UNITER_BENCH_XCD_ONLY=1 UNITER_BENCH_SKIP_CHAIN_CHECK=1 timeout 200 $T --enc 2>&1 | grep -E "ENCODER" | sed "s/^/$v /"
    UNITER_AMD_LIB=$L/libuniter_hip.so timeout 300 python bench.py --no-cpu-baseline --no-traffic --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('c2 $v', d['ms_per_step'], d['timed_windows']['ms_per_step'], 'fwd/bwd', r['encoder_fwd_bwd']['fwd_ms'], r['encoder_fwd_bwd']['bwd_ms'])"
  done
done 2>&1 | tee "$OUT/store_policy_ab.txt"
