"""`python bench.py --gpus N` as the driver may type it (no torch.distributed.run in front): bench.py launches its own ranks,
meets on 127.0.0.1, runs the start-up self-check of the data-parallel exchange mode and prints ONE JSON line from rank 0.
`--dry-launch` runs exactly that start-up on the CPU over gloo with a stand-in for the model (reference: one process per GPU,
pretrain.py:169-173; allreduce semantics utils/distributed.py:16-43)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, timeout=150):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 only, one line
    return json.loads(lines[0]), r.stderr


def test_bench_self_launch_two_ranks_reach_the_group():
    d, _ = _run({})
    assert d["dry_launch"] is True and d["n_gpus"] == 2 and d["incarnation"] == 0
    assert d["dp_mode"] == "single_launch_flags", d


def test_bench_self_check_digest_mismatch_selects_per_bucket_mode():
    d, _ = _run({"UNITER_BENCH_SELFTEST_DIVERGE": "1"})
    assert d["dp_mode"] == "per_bucket" and "differ" in d["dp_mode_note"], d
    assert d["incarnation"] == 0                           # a wrong digest does not need a new process


def test_bench_self_check_hang_trips_the_watchdog_and_reruns_in_per_bucket_mode():
    d, err = _run({"UNITER_BENCH_SELFTEST_HANG": "1", "UNITER_BENCH_WATCHDOG_S": "4"})
    assert d["dp_mode"] == "per_bucket" and d["incarnation"] == 1, d
    assert "did not return within 4 s" in d["dp_mode_note"]
    assert "re-exec with UNITER_AMD_DP_SINGLE_LAUNCH=0" in err
