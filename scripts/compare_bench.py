"""Per-kernel deltas between two bench.py lines (files holding the JSON line as their last line starting with '{', or driver
records with it under run.stdout_tail):   python scripts/compare_bench.py A.json B.json
Prints ms/step, the forward / backward segments and every timed kernel family's us per step in A and B, sorted by |delta|."""
import json
import sys


def load(path):
    txt = open(path).read()
    try:
        d = json.loads(txt)
        if isinstance(d, dict) and 'run' in d and 'stdout_tail' in d['run']:
            txt = d['run']['stdout_tail']
        elif isinstance(d, dict) and 'metric' in d:
            return d
    except ValueError:
        pass
    lines = [l for l in txt.splitlines() if l.startswith('{"metric"')]
    if not lines:
        raise SystemExit("%s: no bench line" % path)
    return json.loads(lines[-1])


def key(k):
    return (k['kernel'], tuple(k['shape']))


def main():
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    a, b = load(sys.argv[1]), load(sys.argv[2])
    print("ms/step           %9.3f %9.3f  %+7.1f %%" % (a['ms_per_step'], b['ms_per_step'], 100.0 * (b['ms_per_step'] / a['ms_per_step'] - 1)))
    for seg in ('fwd_ms', 'bwd_ms'):
        ea, eb = ((x.get('roofline') or {}).get('encoder_fwd_bwd') or {} for x in (a, b))
        if seg in ea and seg in eb:
            print("%-17s %9.3f %9.3f  %+7.1f %%" % (seg, ea[seg], eb[seg], 100.0 * (eb[seg] / ea[seg] - 1)))
    ka = {key(k): k for k in a.get('kernels') or []}
    kb = {key(k): k for k in b.get('kernels') or []}
    rows = []
    for k in sorted(set(ka) | set(kb)):
        ua = ka[k]['us_per_step'] if k in ka else 0.0
        ub = kb[k]['us_per_step'] if k in kb else 0.0
        rows.append((ub - ua, k, ua, ub, (ka.get(k) or kb.get(k))['launches_per_step']))
    rows.sort(key=lambda r: -abs(r[0]))
    print("%-44s %-22s %6s %9s %9s %8s" % ("kernel", "shape", "n/step", "A us", "B us", "delta"))
    for d, k, ua, ub, n in rows:
        print("%-44s %-22s %6.1f %9.1f %9.1f %+8.1f" % (k[0][:44], "x".join(str(v) for v in k[1]), n, ua, ub, d))
    print("sum of timed kernels %37.1f %9.1f %+8.1f" % (sum(r[2] for r in rows), sum(r[3] for r in rows), sum(r[0] for r in rows)))


if __name__ == "__main__":
    main()
