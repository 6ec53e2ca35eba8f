#!/bin/bash
# LDS bank-conflict counters of the 256x256 tile: forward layout against the weight-gradient (both operands K-strided) layout.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/lds
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i "lds" | head -30 > "$OUT/lds_counters.txt"; cat "$OUT/lds_counters.txt" | cut -c1-160
T=$ROOT/tests/native/build/test_kernels
for kind in fwd wgrad dgrad; do
  timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d "$OUT/$kind" -- $T --one $kind 4096 4096 4096 58 1 10 > "$OUT/$kind.log" 2>&1; echo "$kind rc=$?"; tail -2 "$OUT/$kind.log" | cut -c1-200
done
python3 - <<'PY'
import csv,glob,collections
for kind in ('fwd','wgrad','dgrad'):
    fs=glob.glob('/root/repo/gpurun_out/lds/%s/*/*counter_collection.csv'%kind)
    if not fs: print(kind,'no csv'); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if 'gemm8' in r['Kernel_Name']: acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        print(kind, k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
