#!/bin/bash
# Round 6, third GPU call: A/B of HIP_FORCE_DEV_KERNARG (kernel arguments in device memory instead of host memory: the s_loads of a
# launch's first wave then do not cross PCIe) on the harness (--roofs, --enc) and on the c2 bench line.  Output: gpurun_out/r06c3/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06c3
mkdir -p "$OUT"
cd "$ROOT"
T=$ROOT/tests/native/build/test_kernels
export UNITER_TUNED_JSON=$ROOT/uniter_amd/tuned/gfx950.json
for rep in 1 2; do
for v in 0 1; do
  echo "=== HIP_FORCE_DEV_KERNARG=$v (rep $rep) ==="
  HIP_FORCE_DEV_KERNARG=$v timeout 120 $T --roofs 20 2>&1 | tee -a "$OUT/roofs_kernarg$v.txt"
  HIP_FORCE_DEV_KERNARG=$v UNITER_BENCH_XCD_ONLY=1 UNITER_BENCH_SKIP_CHAIN_CHECK=1 timeout 200 $T --enc 2>&1 | grep -E "ENCODER|in-situ" | tee -a "$OUT/enc_kernarg$v.txt" | tail -3
  HIP_FORCE_DEV_KERNARG=$v timeout 300 python bench.py --no-cpu-baseline --no-traffic --steps 30 --warmup 8 2>/dev/null | tee "$OUT/c2_kernarg${v}_$rep.json" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('c2 kernarg=$v', d['ms_per_step'], (d.get('timed_windows') or {}).get('ms_per_step'), 'fwd/bwd', r['encoder_fwd_bwd']['fwd_ms'], r['encoder_fwd_bwd']['bwd_ms'], 'frac', r['frac'])"
done
done
