#!/bin/bash
# kernel trace of the one-rank RCCL step (single launch, dense word table)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4e
mkdir -p "$OUT"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541
cd /tmp && export TMPDIR=/tmp
UNITER_DIST_FORCE=1 UNITER_AMD_DP_SPARSE_WORD=${SPARSE:-0} UNITER_BENCH_LAYERS_PER_BUCKET=4 timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python $ROOT/bench.py --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 3 > "$OUT/trace.log" 2>&1
echo "trace rc=$?"; tail -2 "$OUT/trace.log" | cut -c1-300
find "$OUT/trace" -name "*kernel_trace.csv" | head
