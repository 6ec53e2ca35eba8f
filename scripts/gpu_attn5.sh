#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
T=tests/native/build/test_kernels
timeout 60 $T --attn 32 178 16 0.1 > /dev/null 2>&1
timeout 300 $T --quick 2>&1 | grep -i "attention_\|FAIL\|failed" | grep -v "^\[ OK \]" | head; timeout 300 $T --quick 2>&1 | grep -c "^\[ OK \] attention"
for L in 178 160 192 144; do timeout 60 $T --attn 32 $L 16 0.1 2>&1 | tail -2; done
