#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/g8e
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
T=$ROOT/tests/native/build/test_kernels
export UNITER_BENCH_SKIP_XCD_CHECK=1
cd $ROOT
for v in ${VARIANTS:-A H G}; do
  if [ $v = A ]; then J=uniter_amd/tuned/gfx950.json; else J=aux_bin/tune_$v.json; fi
  UNITER_TUNED_JSON=$J timeout 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/tr_$v" -- $T --enc > "$OUT/tr_$v.log" 2>&1; echo "trace $v rc=$?"
  f=$(find "$OUT/tr_$v" -name "*kernel_trace.csv" | head -1)
  echo "$f"; wc -l "$f"
  # keep the file small: last 6000 rows, selected columns
  python3 - "$f" "$OUT/tr_$v.csv" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[-4000:]
t0=int(rows[0]['Start_Timestamp'])
with open(sys.argv[2],'w') as f:
    for r in rows:
        f.write("%d,%d,%s,%s\n"%(int(r['Start_Timestamp'])-t0,int(r['End_Timestamp'])-t0,r.get('Stream_Id',r.get('Queue_Id','')),r['Kernel_Name'][:90].replace(',',';')))
PY
  rm -rf "$OUT/tr_$v"
done
