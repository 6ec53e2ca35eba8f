#!/bin/bash
# Round 6, call 20: deferred finalize steps of the embedding backward (uniter_finalize_defer / _flush: one launch instead of nine at
# the training loop's join): bit-identity tests, embedding / headline tests, same-box A/B of the c2 line (UNITER_AMD_DEFER_FINALIZE=0/1),
# the step timeline with it, merged c3 / c4 / c5 lines with enough warm-up for their one-off tile tuning.  Output: gpurun_out/r06c20/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06c20
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests -x -q -m gpu -k "deferred_finalize or embedding or golden or headline or mrfr or mrc or lazy or merge" > "$OUT/pytest_first.log" 2>&1; echo "pytest first rc=$?"; tail -3 "$OUT/pytest_first.log"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['ms_per_step'], d['timed_windows']['ms_per_step'], 'fwd/bwd', r['encoder_fwd_bwd']['fwd_ms'], r['encoder_fwd_bwd']['bwd_ms'], 'frac', r['frac'], 'loss', d['final_loss'])"; }
B="timeout 300 python bench.py --no-cpu-baseline --no-traffic --steps 30 --warmup 8"
for rep in 1 2 3; do
  for v in 1 0; do
    UNITER_AMD_DEFER_FINALIZE=$v $B 2>/dev/null | tee "$OUT/c2_deferfin${v}_$rep.json" | line "c2 defer_finalize=$v"
  done
done 2>&1 | tee "$OUT/ab.txt"
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python "$ROOT/bench.py" --no-cpu-baseline --no-kernel-timing --no-traffic --steps 8 --warmup 4 --windows 1 > "$OUT/trace.log" 2>&1; echo "trace rc=$?" )
python scripts/step_timeline.py "$OUT/trace" --out "$OUT/timeline.txt" | head -2
for c in c3 c4 c5; do
  timeout 500 python bench.py --config $c --merge-accum --no-cpu-baseline --no-traffic --steps 10 --warmup 12 2>/dev/null | tee "$OUT/bench_${c}_merged.json.log" | line "$c merged"
done 2>&1 | tee "$OUT/merged.txt"
find "$OUT" -name "*_agent_info.csv" -delete 2>/dev/null
timeout 1500 python -m pytest tests -q -m gpu > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
