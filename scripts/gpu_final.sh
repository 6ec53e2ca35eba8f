#!/bin/bash
# End-of-round record: GPU test suite, smoke, bench lines of all configurations.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/final
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests -x -q -m gpu -s > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest_gpu.log"; grep "headline parity" "$OUT/pytest_gpu.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -2 "$OUT/smoke.log"
timeout 500 python bench.py > "$OUT/bench_c2.json.log" 2> "$OUT/bench_c2.err"; echo "bench c2 rc=$?"
for c in c3 c4 c5; do
  timeout 400 python bench.py --config $c --steps 10 --warmup 3 > "$OUT/bench_$c.json.log" 2> "$OUT/bench_$c.err"; echo "bench $c rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/final/bench_*.json.log')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d.get('roofline') or {}
        print(f.split('/')[-1], d['ms_per_step'], d['value'], 'step frac', (r.get('step') or {}).get('frac'), 'fwd+bwd', (r.get('encoder_fwd_bwd') or {}).get('frac'), 'traffic', r.get('traffic'), 'dominant', r.get('kernel'), r.get('frac'))
        if 'parity' in d: print('  parity', {k:v for k,v in d['parity'].items() if k not in ('what','absolute_bounds')})
        if d.get('cpu_baseline'): print('  cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
    except Exception as e:
        print(f, 'ERR', e)
PY
