#!/bin/bash
# Round 6, call 29: the deferred launch's problem table cached in four versions (gemm.hip MultiTable) instead of re-uploaded whenever
# the upstream-gradient / activation addresses alternate: does the per-step __amd_rocclr_copyBuffer in front of gemm8_multi_kernel go?
# Output: gpurun_out/r06c29/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06c29
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests -x -q -m gpu -k "headline or determinism or accum or merge or lazy or rccl or world1" > "$OUT/pytest_first.log" 2>&1; echo "pytest first rc=$?"; tail -3 "$OUT/pytest_first.log"
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/trace" -- python "$ROOT/bench.py" --no-cpu-baseline --no-kernel-timing --no-traffic --steps 8 --warmup 4 --windows 1 > "$OUT/trace.log" 2>&1; echo "trace rc=$?" )
python scripts/step_timeline.py "$OUT/trace" --out "$OUT/timeline.txt" | head -2
grep -c "copyBuffer" "$OUT/timeline.txt"
grep -B3 -A2 "gemm8_multi_kernel" "$OUT/timeline.txt" | cut -c1-110
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]["encoder_fwd_bwd"]; print(sys.argv[1], d["ms_per_step"], d["timed_windows"]["ms_per_step"], "fwd/bwd", r["fwd_ms"], r["bwd_ms"], "loss", d["final_loss"])'
for rep in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-traffic --steps 30 --warmup 8 2>/dev/null | python -c "$P" "c2"; done
find "$OUT" -name "*_agent_info.csv" -delete 2>/dev/null
