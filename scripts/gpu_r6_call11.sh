#!/bin/bash
# Round 6, call 11: text-embedding backward on an auxiliary stream (UNITER_AMD_ASYNC_EMBED_BWD A/B on the c2 line), its bit-identity
# test, the step timeline with it.  Output: gpurun_out/r06c11/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06c11
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "async_text or headline or determinism" -x 2>&1 | tail -3
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['ms_per_step'], d['timed_windows']['ms_per_step'], 'fwd/bwd', r['encoder_fwd_bwd']['fwd_ms'], r['encoder_fwd_bwd']['bwd_ms'], 'frac', r['frac'], 'loss', d['final_loss'])"; }
for rep in 1 2 3; do
  for v in 1 0; do
    UNITER_AMD_ASYNC_EMBED_BWD=$v timeout 300 python bench.py --no-cpu-baseline --no-traffic --steps 30 --warmup 8 2>/dev/null | tee "$OUT/c2_async${v}_$rep.json" | line "c2 async_embed_bwd=$v"
  done
done 2>&1 | tee "$OUT/ab.txt"
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python "$ROOT/bench.py" --no-cpu-baseline --no-kernel-timing --no-traffic --steps 8 --warmup 4 --windows 1 > "$OUT/trace.log" 2>&1; echo "rc=$?" )
python scripts/step_timeline.py "$OUT/trace" --out "$OUT/timeline.txt" | head -3
find "$OUT" -name "*_agent_info.csv" -delete 2>/dev/null
find "$OUT" -name "*.csv" -size +300k -exec gzip -f {} \; 2>/dev/null
