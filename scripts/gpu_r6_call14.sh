#!/bin/bash
# Round 6, call 14: kernel arguments in device memory (HIP_FORCE_DEV_KERNARG=1) A/B on the native roofs and the c2 line; the host's
# lead over the GPU at the segment boundaries of a step (scripts/host_lead.py).  Output: gpurun_out/r06c14/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06c14
mkdir -p "$OUT"
cd "$ROOT"
export UNITER_TUNED_JSON=$ROOT/uniter_amd/tuned/gfx950.json
for v in 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$v =="
  HIP_FORCE_DEV_KERNARG=$v timeout 200 tests/native/build/test_kernels --roofs 20 2>&1 | grep ROOF
done > "$OUT/roofs_kernarg.txt" 2>&1; cat "$OUT/roofs_kernarg.txt"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['ms_per_step'], d['timed_windows']['ms_per_step'], 'fwd/bwd', r['encoder_fwd_bwd']['fwd_ms'], r['encoder_fwd_bwd']['bwd_ms'], 'frac', r['frac'], 'loss', d['final_loss'])"; }
for rep in 1 2 3; do
  for v in 0 1; do
    HIP_FORCE_DEV_KERNARG=$v timeout 300 python bench.py --no-cpu-baseline --no-traffic --steps 30 --warmup 8 2>/dev/null | tee "$OUT/c2_kernarg${v}_$rep.json" | line "c2 HIP_FORCE_DEV_KERNARG=$v"
  done
done 2>&1 | tee "$OUT/ab.txt"
for v in 0 1; do echo "== HIP_FORCE_DEV_KERNARG=$v =="; HIP_FORCE_DEV_KERNARG=$v timeout 200 python scripts/host_lead.py 2>&1 | tail -8; done | tee "$OUT/host_lead.txt"
for v in 0 1; do echo "== HIP_FORCE_DEV_KERNARG=$v =="; HIP_FORCE_DEV_KERNARG=$v timeout 200 python scripts/segment_times.py 2>&1 | tail -8; done | tee "$OUT/segments.txt"
