"""Opt-in switches of the library that are built but NOT yet validated on hardware: each must leave
the training step bit-identical, because it only re-orders order-independent work.  These tests are skipped unless
UNITER_AMD_RUN_EXPERIMENTS=1 — the switches are off by default, and a switch only becomes a default after this file has
passed on an MI355X and the A/B of scripts/gpu_r5_first.sh shows a gain.

    UNITER_AMD_RUN_EXPERIMENTS=1 python -m pytest tests/test_experiments_gpu.py -m gpu -x -q -s
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("UNITER_AMD_RUN_EXPERIMENTS") != "1",
                                 reason="unvalidated opt-in switches: set UNITER_AMD_RUN_EXPERIMENTS=1 to run")]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _digests(tmp_path, workload, steps, **switches):
    # one tile table for all runs of a comparison: the first run saves what it used (factory table or a fresh sweep), the
    # later ones load it — a switch must not be able to hide behind a different tile choice
    env = dict(os.environ, UNITER_AMD_TUNE_CACHE=str(tmp_path / ("tiles_%s.json" % workload)), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("UNITER_AMD_XCD_AFFINITY",):
        env.pop(k, None)
    env.update({k: str(v) for k, v in switches.items()})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "step_digest_script.py"), workload, str(steps)],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, (r.stdout[-2000:], r.stderr[-3000:])
    return json.loads(lines[-1])


def _same(a, b):
    diff = [n for n in a["per_param"] if a["per_param"][n] != b["per_param"][n]]
    return a["losses"] == b["losses"] and not diff, diff[:8]


@pytest.mark.parametrize("workload", ["c2", "c3"])
def test_xcd_affinity_leaves_the_step_bit_identical(tmp_path, workload):
    """UNITER_AMD_XCD_AFFINITY=1 (csrc/common.cuh: affine_block): GEMM tiles on an 8-row XCD grid,
    attention units and LayerNorm rows in XCD-contiguous order.  Index permutations only: two optimizer steps with dropout on
    must produce the same losses and parameters, bit for bit, as the default maps (and the default must repeat itself)."""
    base = _digests(tmp_path, workload, 2)
    again = _digests(tmp_path, workload, 2)
    ok, diff = _same(base, again)
    assert ok, ("the default step does not repeat itself", diff)
    aff = _digests(tmp_path, workload, 2, UNITER_AMD_XCD_AFFINITY=1)
    ok, diff = _same(base, aff)
    assert ok, ("XCD affinity changed the result", diff)
    print("xcd affinity, %s: %d parameters identical after 2 steps" % (workload, base["n_params"]))

