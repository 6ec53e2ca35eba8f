#!/bin/bash
# Round 6, first GPU call: (1) the vendor bf16 GEMM yardstick on the eight chain shapes beside `test_kernels --roofs` (VERDICT r05
# item 1a), plain and under rocprofv3 --kernel-trace for the Tensile kernel names; (2) the c2 bench line of HEAD; (3) the full GPU
# suite of HEAD, unedited log.  Output: gpurun_out/r06c1/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06c1
mkdir -p "$OUT"
cd "$ROOT"
T=$ROOT/tests/native/build/test_kernels
export UNITER_TUNED_JSON=$ROOT/uniter_amd/tuned/gfx950.json
stamp() { echo "[$(date +%H:%M:%S)] $*"; }
stamp "vendor yardstick (plain)"
timeout 300 python scripts/vendor_gemm_yardstick.py --iters 20 --out "$OUT/vendor_gemm_yardstick.txt" 2>"$OUT/yardstick.err" | tail -14
stamp "ours, alone-hot"
timeout 120 $T --roofs 20 > "$OUT/roofs_ours.txt" 2>&1; cat "$OUT/roofs_ours.txt"
stamp "vendor yardstick under rocprofv3 --kernel-trace"
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/yard_trace" -- python "$ROOT/scripts/vendor_gemm_yardstick.py" --iters 20 > "$OUT/yard_trace.log" 2>&1; echo "rocprof rc=$?" )
python scripts/vendor_gemm_yardstick.py --fold "$OUT/yard_trace" --out "$OUT/vendor_gemm_yardstick.txt" | cut -c1-260 | head -40
stamp "ours under rocprofv3 --kernel-trace"
( cd /tmp; export TMPDIR=/tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/roofs_trace" -- $T --roofs 20 > "$OUT/roofs_trace.log" 2>&1; echo "rocprof rc=$?" )
f=$(find "$OUT/roofs_trace" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-200 "$f" | head -14 | tee "$OUT/roofs_ours_kernel_stats.csv"
find "$OUT" -name "*_agent_info.csv" -delete 2>/dev/null
find "$OUT" -name "*.csv" -size +300k -exec gzip -f {} \; 2>/dev/null
stamp "c2 bench line of HEAD"
timeout 400 python bench.py --no-cpu-baseline --steps 30 --warmup 8 2>"$OUT/c2_head.err" > "$OUT/c2_head.json"
python - "$OUT/c2_head.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get('roofline') or {}
print('c2 HEAD:', d['ms_per_step'], 'ms/step', d['value'], 'ex/s', (d.get('timed_windows') or {}).get('ms_per_step'), 'fwd/bwd', (r.get('encoder_fwd_bwd') or {}).get('fwd_ms'), (r.get('encoder_fwd_bwd') or {}).get('bwd_ms'), 'frac', r.get('frac'))
PY
stamp "full GPU suite of HEAD"
timeout 900 python -m pytest tests -q -m gpu -s > "$OUT/pytest_gpu_head.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu_head.log"
du -sh "$OUT"
