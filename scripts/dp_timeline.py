#!/usr/bin/env python
"""Fold a rocprofv3 kernel trace of a data-parallel bench run (UNITER_DIST_FORCE=1) into the timeline of its last optimizer step:
when the deferred weight-gradient launch, the flag waits of the communication stream, the bucket all-reduces, the embedding
backward and the optimizer kernels start and end, relative to the start of that step's deferred launch.
usage: python scripts/dp_timeline.py <kernel_trace.csv> [label]"""
import csv
import sys


def family(name):
    if 'gemm8_multi_kernel' in name:
        return 'deferred launch'
    if 'streamOpsWait' in name:
        return 'flag wait (hipStreamWaitValue32)'
    if 'nccl' in name.lower() or 'rccl' in name.lower():
        return 'all-reduce'
    if 'adamw_kernel' in name:
        return 'adamw'
    if 'gradsq' in name:
        return 'gradsq'
    return None


def main(path, label):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']))
    rows.sort()
    launches = [r for r in rows if 'gemm8_multi_kernel' in r[2]]
    adam = [r for r in rows if 'adamw_kernel' in r[2]]
    if len(launches) < 2 or len(adam) < 2:
        print("trace too short")
        return
    t0 = launches[-1][0]                                     # the last step's deferred launch
    end = [a for a in adam if a[0] > t0][0][1]
    prev_adam_end = [a for a in adam if a[1] < t0][-1][1]
    print("# %s" % label)
    print("# last optimizer step of the trace; times in us relative to the start of its deferred launch (queue = HIP stream's hardware queue)")
    print("# step span (end of the previous AdamW to the end of this one): %.1f us" % ((end - prev_adam_end) / 1e3))
    chain = [r for r in rows if prev_adam_end <= r[0] < t0 and family(r[2]) is None]
    if chain:
        print("%10.1f .. %10.1f  forward + backward chain on the compute stream: %d kernels" % ((chain[0][0] - t0) / 1e3, (chain[-1][1] - t0) / 1e3, len(chain)))
    beside = [r for r in rows if t0 <= r[0] < end and family(r[2]) is None]
    if beside:
        print("%10.1f .. %10.1f  embedding backward etc. beside / after the launch: %d kernels, %.1f us of kernel time" %
              ((beside[0][0] - t0) / 1e3, (beside[-1][1] - t0) / 1e3, len(beside), sum(r[1] - r[0] for r in beside) / 1e3))
    for r in rows:
        fam = family(r[2])
        if fam is None or r[1] < prev_adam_end or r[0] > end:
            continue
        print("%10.1f .. %10.1f  (%7.1f us)  queue %-3s %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], fam))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])
