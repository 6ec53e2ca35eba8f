#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/g8c
mkdir -p "$OUT"
cd "$ROOT"
T=tests/native/build/test_kernels
for P in nt pol2 pol1; do
  if [ $P = nt ]; then L=""; else L="$ROOT/aux_bin/$P"; fi
  LD_LIBRARY_PATH=$L timeout 300 $T --g8 short > "$OUT/short_$P.log" 2>&1; echo "short $P rc=$?"
  grep "FAIL\|TIME" "$OUT/short_$P.log"
done
export UNITER_BENCH_SKIP_XCD_CHECK=1
for v in A LB1; do
  if [ $v = A ]; then J=uniter_amd/tuned/gfx950.json; else J=aux_bin/tune_$v.json; fi
  UNITER_TUNED_JSON=$J timeout 180 $T --enc large > "$OUT/encL_$v.log" 2>&1; echo "enc large $v rc=$?"
  grep "ENCODER\|FAIL" "$OUT/encL_$v.log" | tail -3
done
for v in A L178B1; do
  if [ $v = A ]; then J=uniter_amd/tuned/gfx950.json; else J=aux_bin/tune_$v.json; fi
  UNITER_TUNED_JSON=$J timeout 240 $T --enc large178 > "$OUT/encL178_$v.log" 2>&1; echo "enc large178 $v rc=$?"
  grep "ENCODER\|FAIL" "$OUT/encL178_$v.log" | tail -3
done
