#!/bin/bash
# Small records for profiles/: attention kernels alone (+ cycle stamps of the backward), one-rank RCCL sweep, encoder alone.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/evidence
mkdir -p "$OUT"
cd "$ROOT"
T=tests/native/build/test_kernels
timeout 60 $T --attn 32 96 12 0.1 > /dev/null 2>&1
{
  echo "# tests/native/build/test_kernels --attn B L heads p   (UNITER_AMD_ATTN_DBG: 8 = cycle stamps of every wave, 16 = the recomputing kernel instead of the P~ / dS hand-off)"
  for dbg in 0 16 8; do echo "== UNITER_AMD_ATTN_DBG=$dbg"; UNITER_AMD_ATTN_DBG=$dbg timeout 60 $T --attn 32 96 12 0.1 2>&1 | tail -17; done
  echo "== p = 0"; timeout 60 $T --attn 32 96 12 0.0 2>&1 | tail -2
  echo "== large-96 (16 heads)"; timeout 60 $T --attn 32 96 16 0.1 2>&1 | tail -2
  echo "== large-178"; timeout 60 $T --attn 32 178 16 0.1 2>&1 | tail -2
} > "$OUT/attention_alone.log" 2>&1
export UNITER_BENCH_SKIP_XCD_CHECK=1
timeout 200 $T --enc > "$OUT/native_encoder.log" 2>&1; grep "ENCODER" "$OUT/native_encoder.log" | tail -1
timeout 300 $T --enc large > "$OUT/native_encoder_large96.log" 2>&1; grep "ENCODER" "$OUT/native_encoder_large96.log" | tail -1
timeout 300 $T --enc large178 > "$OUT/native_encoder_large178.log" 2>&1; grep "ENCODER" "$OUT/native_encoder_large178.log" | tail -1
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541
{
  echo "# python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 ; UNITER_DIST_FORCE=1 = one-rank RCCL group (bucket hooks, collectives, joins)"
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no process group:', d['ms_per_step'], 'ms/step', d['value'], 'ex/s')"
  for lpb in 3 4 6 12; do
    UNITER_DIST_FORCE=1 UNITER_BENCH_LAYERS_PER_BUCKET=$lpb timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one-rank RCCL group, $lpb layers per bucket:', d['ms_per_step'], 'ms/step', d['value'], 'ex/s')"
  done
} > "$OUT/dp_one_rank_rccl.txt" 2>&1
cat "$OUT/dp_one_rank_rccl.txt"
