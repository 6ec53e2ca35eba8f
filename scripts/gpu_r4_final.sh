#!/bin/bash
# End-of-round record (round 4): GPU test suite, smoke, bench lines of all configurations, native harness, encoder alone,
# the any-order probe, the one-rank RCCL table.  Output under gpurun_out/r04final.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04final
mkdir -p "$OUT"
cd "$ROOT"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541
timeout 1500 python -m pytest tests -q -m gpu -s > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest_gpu.log"; grep -E "parity|conditioning" "$OUT/pytest_gpu.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -2 "$OUT/smoke.log"
timeout 500 python bench.py > "$OUT/bench_c2.json.log" 2> "$OUT/bench_c2.err"; echo "bench c2 rc=$?"
for c in c3 c4 c5; do
  timeout 400 python bench.py --config $c --steps 10 --warmup 3 > "$OUT/bench_$c.json.log" 2> "$OUT/bench_$c.err"; echo "bench $c rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04final/bench_*.json.log')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d.get('roofline') or {}
        print(f.split('/')[-1], d['ms_per_step'], d['value'], 'step frac', (r.get('step') or {}).get('frac'), 'fwd+bwd', (r.get('encoder_fwd_bwd') or {}).get('frac'), 'traffic', r.get('traffic'), d.get('traffic_source'), 'dominant', r.get('kernel'), r.get('frac'), r.get('avg_launch_us'))
        if 'parity' in d: print('  parity', {k:v for k,v in d['parity'].items() if k not in ('what','absolute_bounds')})
        if d.get('cpu_baseline'): print('  cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['kind'])
    except Exception as e:
        print(f, 'ERR', e)
PY
T=tests/native/build/test_kernels
timeout 600 $T > "$OUT/native_harness.log" 2>&1; echo "harness rc=$?"; grep -c "^\[ OK \]" "$OUT/native_harness.log"; tail -1 "$OUT/native_harness.log"
timeout 300 $T --enc > "$OUT/native_encoder.log" 2>&1; grep -E "ENCODER|overlapped|in-order" "$OUT/native_encoder.log"
timeout 300 $T --enc large > "$OUT/native_encoder_large96.log" 2>&1; grep "ENCODER" "$OUT/native_encoder_large96.log" | tail -1
timeout 300 $T --enc large178 > "$OUT/native_encoder_large178.log" 2>&1; grep "ENCODER" "$OUT/native_encoder_large178.log" | tail -1
{
  echo "# python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 5 ; UNITER_DIST_FORCE=1 = one-rank RCCL group (reducer, collectives, joins) on the one GPU of the box"
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no process group:', d['ms_per_step'], 'ms/step', d['value'], 'ex/s')"
  for lpb in 4 6; do
    UNITER_DIST_FORCE=1 UNITER_BENCH_LAYERS_PER_BUCKET=$lpb timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one-rank RCCL group, ONE backward call + bucket flags, $lpb layers per bucket:', d['ms_per_step'], 'ms/step', d['value'], 'ex/s')"
  done
  UNITER_DIST_FORCE=1 UNITER_AMD_DP_SINGLE_LAUNCH=0 UNITER_BENCH_LAYERS_PER_BUCKET=4 timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one-rank RCCL group, one backward call PER BUCKET (round 3), 4 layers per bucket:', d['ms_per_step'], 'ms/step', d['value'], 'ex/s')"
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no process group (again):', d['ms_per_step'], 'ms/step', d['value'], 'ex/s')"
} > "$OUT/dp_one_rank_rccl.txt" 2>&1
cat "$OUT/dp_one_rank_rccl.txt"
