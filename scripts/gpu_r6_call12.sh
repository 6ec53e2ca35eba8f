#!/bin/bash
# Round 6, call 12 (first call of the second session): the unedited GPU suite of HEAD, the round's profile record so far
# (scripts/profile_round.sh r06a: bench line, kernel trace + stats, FETCH / WRITE / MFMA PMC passes), the step timeline and the
# native harness's roofs.  Output: gpurun_out/r06a/, gpurun_out/r06c12/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06c12
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > "$OUT/pytest_gpu.log"; tail -3 "$OUT/pytest_gpu.log"
bash scripts/profile_round.sh r06a 2>&1 | grep "rc="
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python "$ROOT/bench.py" --no-cpu-baseline --no-kernel-timing --no-traffic --steps 8 --warmup 4 --windows 1 > "$OUT/trace.log" 2>&1; echo "trace rc=$?" )
python scripts/step_timeline.py "$OUT/trace" --out "$OUT/timeline.txt" | head -3
python scripts/segment_times.py > "$OUT/segments.txt" 2>&1; tail -12 "$OUT/segments.txt"
export UNITER_TUNED_JSON=$ROOT/uniter_amd/tuned/gfx950.json
timeout 200 tests/native/build/test_kernels --roofs 20 > "$OUT/roofs.txt" 2>&1; tail -12 "$OUT/roofs.txt"
find "$ROOT/gpurun_out/r06a" "$OUT" -name "*_agent_info.csv" -delete 2>/dev/null
find "$ROOT/gpurun_out/r06a" "$OUT" -name "*.csv" -size +300k -exec gzip -f {} \; 2>/dev/null
tail -c 1500 "$ROOT/gpurun_out/r06a/bench.json.log"
