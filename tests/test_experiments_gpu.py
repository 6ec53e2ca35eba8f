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
    for k in ("UNITER_AMD_ADAMW_NT",):
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


def test_non_temporal_adamw_streams_leave_the_step_bit_identical(tmp_path):
    """UNITER_AMD_ADAMW_NT=1 (csrc/adamw.hip: adamw_kernel<true>): the fp32 master / moment streams of the update and the
    gradient read carry the non-temporal hint.  A cache policy: three optimizer steps must end on the same parameters."""
    base = _digests(tmp_path, "c2", 3)
    nt = _digests(tmp_path, "c2", 3, UNITER_AMD_ADAMW_NT=1)
    ok, diff = _same(base, nt)
    assert ok, ("non-temporal AdamW streams changed the result", diff)


@pytest.mark.parametrize("workload", ["c3", "c4"])
def test_merged_micro_batches_give_the_accumulation_loops_gradients(workload):
    """StepRunner(merge_accum=True) (uniter_amd/data/merge.py; bench.py --merge-accum): the micro-batches of an optimizer step as
    ONE batch.  Dropout off, every task of the workload's mix: the loss and the gradient of every parameter against the
    accumulation loop on the same weights.  Not bit-identical — the loop rounds the summed gradient to bf16 once per micro-step,
    the merged launch once; weight-gradient contractions run over twice / four times the tokens — so the bound is the one the
    parity tests use between two bf16 evaluations of the same function (relative L2 <= 2e-2 per tensor, cosine >= 0.999).
    The CPU half (exact equality of the merged batch with the collate's, fp32 equality of the loss and gradients through the
    oracle) is tests/test_merge_accumulation.py."""
    import torch
    from uniter_amd import _lib
    from uniter_amd.train import StepRunner
    from uniter_amd.utils.misc import set_dropout
    dev = torch.device("cuda", 0)
    out = {}
    for merged in (False, True):
        r = StepRunner(workload, dev, seed=77, merge_accum=merged)
        assert r.merge_accum == merged
        set_dropout(r.model, 0.0)
        res = {}
        for task, batch in r.batches.items():
            for p in r.model.parameters():
                p.grad = None
            total = 0.0
            for _ in range(1 if merged else r.w['accum']):
                loss = r._loss(task, batch)
                loss.backward()
                total += float(loss.detach())
            _lib.join_wgrads()
            torch.cuda.synchronize()
            res[task] = (total, {n: p.grad.detach().float().clone() for n, p in r.model.named_parameters() if p.grad is not None})
        out[merged] = res
        del r
        torch.cuda.empty_cache()
    for task in out[False]:
        l0, g0 = out[False][task]
        l1, g1 = out[True][task]
        assert abs(l1 - l0) <= 5e-3 * max(1.0, abs(l0)), (task, l0, l1)
        assert sorted(g0) == sorted(g1), task
        worst = (0.0, None)
        for n in g0:
            a, b = g1[n].double().flatten(), g0[n].double().flatten()
            if float(b.norm()) == 0.0:
                assert float(a.abs().max()) < 1e-6, (task, n)
                continue
            if n.endswith('attention.self.key.bias'):
                # softmax is invariant to a key bias: the true gradient is identically zero and both evaluations hold only
                # their own rounding residue (tests/test_gpu_parity.py treats it the same way): bounded, not compared
                assert float(a.abs().max()) <= 1e-2 * max(1.0, float(g0[n.replace('key.bias', 'query.bias')].abs().max())), (task, n)
                continue
            rel = float((a - b).norm() / b.norm())
            cos = float((a * b).sum() / (a.norm() * b.norm()))
            worst = max(worst, (rel, n))
            assert rel <= 2e-2 and cos >= 0.999, (task, n, rel, cos)
        print("merged vs accumulated, %s / %s: loss %.5f vs %.5f, %d gradients, worst rel-L2 %.2e (%s)"
              % (workload, task, l1, l0, len(g0), worst[0], worst[1]))
