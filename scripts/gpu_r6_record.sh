#!/bin/bash
# Round 6 record call: scripts/profile_round.sh r06 (bench line, kernel trace + stats, FETCH / WRITE / MFMA PMC passes of the bench
# command), step timeline + segments, native roofs + encoder alone, the whole GPU suite, smoke(), the bench line as the driver runs it
# (with the cpu_baseline leg) and the c3 / c4 / c5 lines.  Output: gpurun_out/r06/, gpurun_out/r06rec/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06rec
mkdir -p "$OUT"
cd "$ROOT"
bash scripts/profile_round.sh r06 2>&1 | grep "rc="
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python "$ROOT/bench.py" --no-cpu-baseline --no-kernel-timing --no-traffic --steps 8 --warmup 4 --windows 1 > "$OUT/trace.log" 2>&1; echo "trace rc=$?" )
python scripts/step_timeline.py "$OUT/trace" --out "$OUT/timeline.txt" | head -2
python scripts/segment_times.py 2>&1 | tail -8 > "$OUT/segments.txt"; cat "$OUT/segments.txt"
export UNITER_TUNED_JSON=$ROOT/uniter_amd/tuned/gfx950.json
timeout 200 tests/native/build/test_kernels --roofs 20 2>&1 | grep ROOF > "$OUT/roofs.txt"; cat "$OUT/roofs.txt"
UNITER_BENCH_XCD_ONLY=1 UNITER_BENCH_SKIP_CHAIN_CHECK=1 timeout 200 tests/native/build/test_kernels --enc 2>&1 | grep -E "in-situ|ENCODER" > "$OUT/native_encoder.txt"; tail -1 "$OUT/native_encoder.txt"
timeout 600 tests/native/build/test_kernels > "$OUT/harness.log" 2>&1; echo "harness rc=$?"; grep -c "^\[ OK \]" "$OUT/harness.log"; tail -1 "$OUT/harness.log"
timeout 1500 python -m pytest tests -q -m gpu > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -2 "$OUT/smoke.log"
timeout 600 python bench.py > "$OUT/bench_c2.json.log" 2> "$OUT/bench_c2.err"; echo "bench c2 rc=$?"; tail -c 300 "$OUT/bench_c2.json.log"
for c in c3 c4 c5; do
  timeout 400 python bench.py --config $c --no-cpu-baseline --no-traffic --steps 10 --warmup 3 > "$OUT/bench_$c.json.log" 2> "$OUT/bench_$c.err"; echo "bench $c rc=$?"
  timeout 400 python bench.py --config $c --merge-accum --no-cpu-baseline --no-traffic --steps 10 --warmup 3 > "$OUT/bench_${c}_merged.json.log" 2> "$OUT/bench_${c}_merged.err"; echo "bench $c merged rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06rec/bench_*.json.log')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d.get('roofline') or {}
        print(f.split('/')[-1], d['ms_per_step'], d['value'], 'step frac', (r.get('step') or {}).get('frac'), 'fwd+bwd', (r.get('encoder_fwd_bwd') or {}).get('frac'), 'traffic', r.get('traffic'))
        if d.get('cpu_baseline'): print('  cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline'].get('kind'))
    except Exception as e:
        print(f, 'ERR', e)
PY
find "$ROOT/gpurun_out/r06" "$OUT" -name "*_agent_info.csv" -delete 2>/dev/null
du -sh "$ROOT/gpurun_out/r06" "$OUT"
