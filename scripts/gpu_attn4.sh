#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
T=tests/native/build/test_kernels
timeout 60 $T --attn 32 96 12 0.1 > /dev/null 2>&1
for dbg in ${1:-8}; do echo "dbg=$dbg"; UNITER_AMD_ATTN_DBG=$dbg timeout 60 $T --attn 32 96 12 0.1 2>&1 | tail -16; done
