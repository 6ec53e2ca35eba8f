#!/bin/bash
# rocprofv3 kernel trace of the one-rank RCCL bench (single deferred launch + bucket flags; the round-3 per-bucket calls) -> scripts/dp_timeline.py
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04dp
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp MASTER_ADDR=127.0.0.1 MASTER_PORT=29547
for mode in flags perbucket none; do
  E=""; [ $mode = flags ] && E="UNITER_DIST_FORCE=1"; [ $mode = perbucket ] && E="UNITER_DIST_FORCE=1 UNITER_AMD_DP_SINGLE_LAUNCH=0"
  env $E timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/$mode" -- python $ROOT/bench.py --no-cpu-baseline --no-kernel-timing --steps 6 --warmup 3 > "$OUT/$mode.log" 2>&1
  echo "$mode rc=$?"
  F=$(ls $OUT/$mode/*/*_kernel_trace.csv | head -1)
  python $ROOT/scripts/dp_timeline.py "$F" "$mode" > "$OUT/timeline_$mode.txt" 2>&1
  head -40 "$OUT/timeline_$mode.txt"
done
