#!/bin/bash
# Round 6, call 10: why does the asynchronous optimizer step (bench.py --overlap) not pay?  Kernel timeline of an overlapped step.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06c10
mkdir -p "$OUT"
cd "$ROOT"
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_overlap" -- python "$ROOT/bench.py" --overlap --no-cpu-baseline --no-kernel-timing --no-traffic --steps 8 --warmup 4 --windows 1 > "$OUT/trace_overlap.log" 2>&1; echo "rc=$?" )
python scripts/step_timeline.py "$OUT/trace_overlap" --out "$OUT/timeline_overlap.txt" | head -3
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_eager" -- python "$ROOT/bench.py" --no-cpu-baseline --no-kernel-timing --no-traffic --steps 8 --warmup 4 --windows 1 > "$OUT/trace_eager.log" 2>&1; echo "rc=$?" )
python scripts/step_timeline.py "$OUT/trace_eager" --out "$OUT/timeline_eager.txt" | head -3
for m in "" "--overlap"; do timeout 300 python bench.py $m --no-cpu-baseline --no-kernel-timing --no-traffic --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap', d['config']['optimizer_overlap'], d['ms_per_step'], (d.get('timed_windows') or {}).get('ms_per_step'))"; done
find "$OUT" -name "*_agent_info.csv" -delete 2>/dev/null
find "$OUT" -name "*.csv" -size +300k -exec gzip -f {} \; 2>/dev/null
