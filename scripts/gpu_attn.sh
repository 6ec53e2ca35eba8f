#!/bin/bash
# Attention kernels alone: forward / backward launch time, with the diagnostic switches of UNITER_AMD_ATTN_DBG.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/attn
mkdir -p "$OUT"
cd "$ROOT"
T=tests/native/build/test_kernels
timeout 60 $T --attn 32 96 12 0.1 > /dev/null 2>&1   # warm the box
for dbg in ${1:-0 1 2 4 6 7}; do
  echo "dbg=$dbg"; UNITER_AMD_ATTN_DBG=$dbg timeout 60 $T --attn 32 96 12 0.1 2>&1 | tail -18
done
echo "p=0"; timeout 60 $T --attn 32 96 12 0.0 2>&1 | tail -2
