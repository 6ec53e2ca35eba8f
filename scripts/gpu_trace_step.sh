#!/bin/bash
# Kernel trace of the bench command only (step breakdown), output under gpurun_out/<tag>/trace.
set -u
TAG=${1:-r03b}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 python $ROOT/bench.py --no-cpu-baseline --steps 30 --warmup 5 > "$OUT/bench.json.log" 2> "$OUT/bench.err.log"
echo "bench rc=$?"; python -c "import json;d=json.loads(open('$OUT/bench.json.log').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'],d['roofline'].get('kernel'),d['roofline'].get('frac'))"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python $ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 3 > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
