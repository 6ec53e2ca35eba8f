#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
export UNITER_BENCH_SKIP_XCD_CHECK=1
timeout 160 tests/native/build/test_kernels --enc > /dev/null 2>&1
UNITER_AMD_MULTI_STAMPS=1 timeout 200 tests/native/build/test_kernels --enc 2>&1 | grep -A22 "multi launch schedule" | tail -24
