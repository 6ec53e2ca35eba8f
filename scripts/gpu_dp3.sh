#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541
echo "== plain"; timeout 300 python scripts/segment_times.py 2>/dev/null | tail -7
for lpb in 12 4; do echo "== one-rank RCCL group, $lpb layers per bucket"; UNITER_DIST_FORCE=1 UNITER_AMD_LAYERS_PER_BUCKET=$lpb timeout 300 python scripts/segment_times.py 2>/dev/null | tail -7; done
