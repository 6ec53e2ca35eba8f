#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
timeout 400 tests/native/build/test_kernels --quick 2>&1 | grep -A30 "== deferred" | cut -c1-200 | head -60
