#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
T=tests/native/build/test_kernels
timeout 60 $T --attn 32 96 12 0.1 > /dev/null 2>&1
timeout 300 $T --quick 2>&1 | grep -i "attention_bwd\|FAIL\|failed" | cut -c1-160 | tail -14
for dbg in 0 16; do echo "dbg=$dbg"; UNITER_AMD_ATTN_DBG=$dbg timeout 60 $T --attn 32 96 12 0.1 2>&1 | tail -1; UNITER_AMD_ATTN_DBG=$dbg timeout 60 $T --attn 32 96 12 0.0 2>&1 | tail -1;  UNITER_AMD_ATTN_DBG=$dbg timeout 60 $T --attn 32 128 16 0.1 2>&1 | tail -1; UNITER_AMD_ATTN_DBG=$dbg timeout 60 $T --attn 32 64 12 0.1 2>&1 | tail -1; done
