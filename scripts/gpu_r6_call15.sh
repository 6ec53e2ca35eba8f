#!/bin/bash
# Round 6, call 15: feed probe of the gemm_tile family — three variant builds with WRONG results on purpose (gemm.hip,
# UNITER_GEMM_FEED_PROBE = 1: no N-side fragment reads, 2: no N-side LDS-DMA, 3: N-side LDS-DMA from one hot 1 KB piece) against the
# shipped build on the chain shapes alone; tile sweep of the NLVR2 head's two plain GEMMs.  Output: gpurun_out/r06c15/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06c15
mkdir -p "$OUT"
cd "$ROOT"
T=$ROOT/tests/native/build/test_kernels
export UNITER_TUNED_JSON=$ROOT/uniter_amd/tuned/gfx950.json
for rep in 1 2; do
  for v in build build_p1 build_p2 build_p3; do
    echo "== $v (rep $rep) =="
    LD_LIBRARY_PATH=$ROOT/uniter_amd/csrc/$v timeout 200 $T --roofs 20 2>&1 | grep ROOF
  done
done > "$OUT/feed_probe_roofs.txt" 2>&1
tail -52 "$OUT/feed_probe_roofs.txt"
timeout 300 $T --sweep 10 head > "$OUT/sweep_head.txt" 2>&1; grep "BEST\|SWEEP" "$OUT/sweep_head.txt" | sort -k7 -n | head -40
