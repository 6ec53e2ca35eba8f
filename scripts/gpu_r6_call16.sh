#!/bin/bash
# Round 6, call 16: the NLVR2 head — one autograd node for regroup + cross attentions + cat (UNITER_AMD_NLVR2_CAT_TORCH=1 = the
# three-node form), 96 x 192 tile for the grouped input projections (UNITER_AMD_GROUP_FWD_192=0 = 96 x 96), the head's Linear(2H, H)
# tiles in the shipped table (UNITER_AMD_TUNE_CACHE=scripts/tables/gfx950_before_head_fc.json = without them): parity tests, then
# same-box A/B of the c2 line.  Output: gpurun_out/r06c16/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06c16
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests -x -q -m gpu -k "nlvr2 or paired or headline or golden or determinism or pool" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['ms_per_step'], d['timed_windows']['ms_per_step'], 'fwd/bwd', r['encoder_fwd_bwd']['fwd_ms'], r['encoder_fwd_bwd']['bwd_ms'], 'frac', r['frac'], 'loss', d['final_loss'])"; }
B="timeout 300 python bench.py --no-cpu-baseline --no-traffic --steps 30 --warmup 8"
cp scripts/tables/gfx950_before_head_fc.json /tmp/old_table.json
for rep in 1 2 3; do
  $B 2>/dev/null | tee "$OUT/c2_new_$rep.json" | line "c2 shipped"
  UNITER_AMD_NLVR2_CAT_TORCH=1 $B 2>/dev/null | line "c2 three-node cat"
  UNITER_AMD_GROUP_FWD_192=0 $B 2>/dev/null | line "c2 group fwd 96x96"
  UNITER_AMD_TUNE_CACHE=/tmp/old_table.json $B 2>/dev/null | line "c2 table without head fc"
  UNITER_AMD_NLVR2_CAT_TORCH=1 UNITER_AMD_GROUP_FWD_192=0 UNITER_AMD_TUNE_CACHE=/tmp/old_table.json $B 2>/dev/null | line "c2 all three off"
done 2>&1 | tee "$OUT/ab.txt"
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python "$ROOT/bench.py" --no-cpu-baseline --no-kernel-timing --no-traffic --steps 8 --warmup 4 --windows 1 > "$OUT/trace.log" 2>&1; echo "trace rc=$?" )
python scripts/step_timeline.py "$OUT/trace" --out "$OUT/timeline.txt" | head -3
find "$OUT" -name "*_agent_info.csv" -delete 2>/dev/null
find "$OUT" -name "*.csv" -size +300k -exec gzip -f {} \; 2>/dev/null
