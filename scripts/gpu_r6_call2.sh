#!/bin/bash
# Round 6, second GPU call: where the step's time goes OUTSIDE the encoder kernels.  (1) host issue time vs GPU time per segment
# (scripts/segment_times.py), (2) kernel timeline of one eager step and of one hipGraph-replayed step,
# (3) the x gelu' shape without its epilogue beside the vendor figure.  Output: gpurun_out/r06c2/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06c2
mkdir -p "$OUT"
cd "$ROOT"
T=$ROOT/tests/native/build/test_kernels
export UNITER_TUNED_JSON=$ROOT/uniter_amd/tuned/gfx950.json
stamp() { echo "[$(date +%H:%M:%S)] $*"; }
stamp "segments: host issue vs GPU"
timeout 300 python scripts/segment_times.py 2>"$OUT/segments.err" | tee "$OUT/segment_times.txt"
stamp "roofs (+ plain ffn2_dgrad)"
timeout 120 $T --roofs 20 2>&1 | tee "$OUT/roofs_ours.txt"
stamp "eager timeline"
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_eager" -- python "$ROOT/bench.py" --no-cpu-baseline --no-kernel-timing --no-traffic --steps 8 --warmup 4 --windows 1 > "$OUT/trace_eager.log" 2>&1; echo "rc=$?" )
python scripts/step_timeline.py "$OUT/trace_eager" --out "$OUT/timeline_eager.txt" | head -3
stamp "hipGraph timeline"
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_graph" -- python "$ROOT/bench.py" --graph --no-cpu-baseline --no-kernel-timing --no-traffic --steps 8 --warmup 4 --windows 1 > "$OUT/trace_graph.log" 2>&1; echo "rc=$?" )
python scripts/step_timeline.py "$OUT/trace_graph" --out "$OUT/timeline_graph.txt" | head -3
stamp "eager vs graph, untraced"
for m in "" "--graph"; do timeout 300 python bench.py $m --no-cpu-baseline --no-kernel-timing --no-traffic --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mode', d['config']['launch'], d['ms_per_step'], (d.get('timed_windows') or {}).get('ms_per_step'))"; done
find "$OUT" -name "*_agent_info.csv" -delete 2>/dev/null
find "$OUT" -name "*.csv" -size +300k -exec gzip -f {} \; 2>/dev/null
du -sh "$OUT"
