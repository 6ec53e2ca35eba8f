"""Pins the CPU oracle (oracle/uniter_oracle.py) against vectors produced by the REAL reference
(tests/golden/make_golden.py -> tests/golden/uniter_tiny.npz).  fp32, no GPU needed."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import uniter_oracle as O
from tests.common import rel_l2


def _close(g, g_ref, tol=2e-4):
    """Relative L2 match, or both numerically zero (e.g. the key-bias gradient, which softmax makes vanish)."""
    return rel_l2(g, g_ref) < tol or float((g - g_ref).abs().max()) < 1e-7

TASK_FN = {
    'mlm': O.mlm_loss, 'mrfr': O.mrfr_loss, 'itm': O.itm_loss,
    'mrckl': lambda sd, cfg, b: O.mrc_loss(sd, cfg, b, kl=True),
    'mrc': lambda sd, cfg, b: O.mrc_loss(sd, cfg, b, kl=False),
}


def _leafs(sd):
    out = {}
    for k, v in sd.items():
        out[k] = v.clone().requires_grad_(True)
    out['cls.predictions.decoder.weight'] = out['uniter.embeddings.word_embeddings.weight']
    return out


@pytest.mark.parametrize("task", ['mlm', 'mrfr', 'mrckl', 'mrc', 'itm'])
def test_pretrain_tasks_match_reference(golden, task):
    sd = _leafs(golden.weights('pre'))
    batch = golden.batch(task)
    ref = golden.out(task)
    loss, seq = TASK_FN[task](sd, golden.cfg, batch)
    assert loss.shape == ref['loss'].shape
    torch.testing.assert_close(loss.detach(), ref['loss'], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(seq.detach(), ref['seq'], rtol=1e-4, atol=2e-5)
    loss.mean().backward()
    for name, g_ref in golden.grads(task).items():
        g = sd[name].grad
        assert g is not None, name
        assert _close(g, g_ref), (name, rel_l2(g, g_ref))


def test_mlm_all_gradients(golden):
    sd = _leafs(golden.weights('pre'))
    loss, _ = O.mlm_loss(sd, golden.cfg, golden.batch('mlm'))
    loss.mean().backward()
    grads = golden.grads('mlm')
    assert len(grads) > 50
    for name, g_ref in grads.items():
        g = sd[name].grad
        if g is None:
            assert float(g_ref.abs().max()) == 0.0, name
            continue
        if float(g_ref.abs().max()) == 0:
            assert float(g.abs().max()) < 1e-8, name
        else:
            assert _close(g, g_ref), (name, rel_l2(g, g_ref))


def test_vqa_matches_reference(golden):
    sd = _leafs({**golden.weights('pre'), **golden.weights('vqa')})
    loss, _ = O.vqa_loss(sd, golden.cfg, golden.batch('vqa'))
    torch.testing.assert_close(loss.detach(), golden.out('vqa')['loss'], rtol=1e-4, atol=1e-5)
    (loss.mean() * loss.shape[1]).backward()          # train_vqa.py:188
    for name, g_ref in golden.grads('vqa').items():
        assert _close(sd[name].grad, g_ref), name


def test_nlvr2_paired_attn_matches_reference(golden):
    w = golden.weights('pre')
    w.update(golden.weights('nlvr2'))                 # includes the grown 3-row token-type table
    sd = _leafs(w)
    loss, _ = O.nlvr2_paired_attn_loss(sd, golden.cfg, golden.batch('nlvr2'))
    torch.testing.assert_close(loss.detach(), golden.out('nlvr2')['loss'], rtol=1e-4, atol=1e-5)
    loss.mean().backward()
    for name, g_ref in golden.grads('nlvr2').items():
        assert _close(sd[name].grad, g_ref), name


def test_adamw_two_clipped_steps(golden):
    """optim/adamw.py + optim/misc.py grouping + clip_grad_norm_(0.5): weights after two steps on the MLM gradients."""
    sd = golden.weights('pre')
    grads = golden.grads('mlm')
    norm_ref, p_ref, v_ref = golden.adamw()
    total, coef = O.clip_coef(list(grads.values()), 0.5)
    assert math.isclose(total, norm_ref, rel_tol=1e-5)
    assert coef < 1.0
    for name, want in p_ref.items():
        p, g = sd[name], grads[name] * coef
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        wd = 0.0 if O.no_decay(name) else 0.01
        for step in (1, 2):
            p, m, v = O.adamw_step(p, g, m, v, step, lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=wd)
        torch.testing.assert_close(p, want, rtol=1e-5, atol=1e-7)
        if name in v_ref:
            torch.testing.assert_close(v, v_ref[name], rtol=1e-5, atol=1e-12)


def test_schedule_and_helpers():
    assert O.warmup_linear(0, 10, 100) == 0
    assert O.warmup_linear(5, 10, 100) == 0.5
    assert O.warmup_linear(10, 10, 100) == 1.0
    assert O.warmup_linear(100, 10, 100) == 0
    assert O.get_lr_sched(100, 3e-5, 10, 100) == 1e-8
    assert O.no_decay('uniter.encoder.layer.0.output.LayerNorm.weight')
    assert not O.no_decay('uniter.img_embeddings.img_layer_norm.weight')      # decayed: the reference quirk
    assert not O.no_decay('vqa_output.2.weight')
    a, b = torch.ones(3), torch.full((3,), 3.0)
    torch.testing.assert_close(O.allreduce_average([a, b], 2.0), torch.ones(3))
    gi = O.get_gather_index([2, 3], [2, 1], 2, 3, 5)
    assert gi.tolist() == [[0, 1, 3, 4, 4], [0, 1, 2, 3, 4]]


def test_inplace_adamw_equals_functional():
    g = torch.Generator().manual_seed(0)
    p, grad = torch.randn(50, generator=g), torch.randn(50, generator=g)
    m, v = torch.zeros(50), torch.zeros(50)
    p1, m1, v1 = p.clone(), m.clone(), v.clone()
    for step in (1, 2, 3):
        p, m, v = O.adamw_step(p, grad, m, v, step, 1e-2, (0.9, 0.98), 1e-6, 0.01)
        O.adamw_step_(p1, grad, m1, v1, step, 1e-2, (0.9, 0.98), 1e-6, 0.01)
    torch.testing.assert_close(p, p1, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(v, v1, rtol=1e-6, atol=1e-12)


# --------------------------------------------------------------------------------------------------------------
# optimal-transport distance: oracle vs the real reference's cost_matrix_cosine + ipot
# --------------------------------------------------------------------------------------------------------------
def _ot_case(name):
    import os
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ot_golden.npz"))

    def bf(a):
        return torch.from_numpy(a.astype(np.int16)).view(torch.bfloat16).float()

    return {k.split("/", 1)[1]: (bf(z[k]) if k.endswith("_bf16") else torch.from_numpy(z[k])) for k in z.files
            if k.startswith(name + "/")}


@pytest.mark.parametrize("name", ["small", "base", "long"])
def test_optimal_transport_matches_reference(name):
    c = _ot_case(name)
    x = c["x_bf16"].clone().requires_grad_(True)
    y = c["y_bf16"].clone().requires_grad_(True)
    dist, T = O.optimal_transport_dist(x, y, c["txt_pad"], c["img_pad"])
    torch.testing.assert_close(T, c["T"], rtol=1e-5, atol=1e-8)
    torch.testing.assert_close(dist.detach(), c["dist"], rtol=1e-5, atol=1e-7)
    dist.sum().backward()
    torch.testing.assert_close(x.grad, c["dx"], rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(y.grad, c["dy"], rtol=1e-4, atol=1e-7)
    # transport plan: rows of padded slots are exactly zero, the rest is a non-negative coupling
    assert float(T.min()) >= 0.0
    assert float(T[c["img_pad"]].abs().max() if c["img_pad"].any() else 0.0) == 0.0


# ---- the committed fixtures regenerate bit-exactly from the committed recipes (needs /root/reference: this container) ----
@pytest.mark.skipif(not os.path.isdir("/root/reference/model"), reason="the reference checkout only exists in the build container")
@pytest.mark.parametrize("script,fixture", [("make_golden.py", "uniter_tiny.npz"), ("make_golden_ot.py", "ot_golden.npz")])
def test_golden_recipe_regenerates_committed_fixture(tmp_path, script, fixture):
    import subprocess
    import sys
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    out = str(tmp_path / fixture)
    r = subprocess.run([sys.executable, os.path.join(here, script), out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    new, old = np.load(out), np.load(os.path.join(here, fixture))
    assert sorted(new.files) == sorted(old.files)
    for k in old.files:
        a, b = new[k], old[k]
        assert a.dtype == b.dtype and a.shape == b.shape, k
        assert a.tobytes() == b.tobytes(), "array %s differs from the committed fixture" % k


def test_staged_reference_runs_one_nlvr2_step(tmp_path):
    """oracle/make_ref.py + oracle/ref_runner.py (bench.py's cpu_baseline.kind == "reference"): the staged copy of the reference
    imports under private package names and runs the train_nlvr2.py step on a tiny configuration; without dropout its loss is the
    oracle's."""
    import json
    import os
    import pytest
    from oracle import make_ref, ref_runner, uniter_oracle as O
    if make_ref.stage() is None and not ref_runner.available():
        pytest.skip("no reference checkout and nothing staged")
    from uniter_amd.utils.synthetic import make_batch
    cfg = dict(vocab_size=96, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128,
               hidden_act="gelu", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, max_position_embeddings=32,
               type_vocab_size=2, initializer_range=0.02)
    path = tmp_path / "tiny.json"
    path.write_text(json.dumps(cfg))
    train = dict(learning_rate=1e-3, betas=(0.9, 0.98), weight_decay=0.01, warmup_steps=1, num_train_steps=10, grad_norm=2.0)
    import torch
    torch.manual_seed(0)
    nlvr2, _, _ = ref_runner._import_reference()
    seed_model = nlvr2.UniterForNlvr2PairedAttn.from_pretrained(str(path), {}, img_dim=64)
    seed_model.init_type_embedding()
    sd = {k: v.detach().clone() for k, v in seed_model.state_dict().items()}
    runner = ref_runner.ReferenceNlvr2Step(str(path), sd, train, img_dim=64)
    for m in runner.model.modules():
        if hasattr(m, 'dropout') and isinstance(m.dropout, float):
            m.dropout = 0.0
    batch = make_batch('nlvr2', 4, max_txt_len=9, num_bb=6, img_dim=64, vocab_size=96, seed=3, ragged=True, min_txt_len=4, min_bb=2)
    ref_loss, _ = O.nlvr2_paired_attn_loss(sd, cfg, batch)
    before = {k: v.detach().clone() for k, v in runner.model.state_dict().items()}
    loss = runner.step(batch, 1)
    assert abs(loss - float(ref_loss.mean())) <= 1e-4 * max(1.0, abs(loss))
    moved = sum(1 for k, v in runner.model.state_dict().items() if not torch.equal(v, before[k]))
    assert moved > 20                                   # the reference's AdamW really stepped
