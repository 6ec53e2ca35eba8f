"""Host-side logic of the drop-in boundary (CPU only): config, checkpoint loading, optimizer grouping, schedules,
arena layout, synthetic batch contract, and the no-fallback rule."""
import json

import pytest
import torch

from tests.common import IMG_DIM, LABEL_DIM, TINY_CONFIG


def _tiny():
    from uniter_amd.model.pretrain import UniterForPretraining
    return UniterForPretraining.from_pretrained(TINY_CONFIG, {}, img_dim=IMG_DIM, img_label_dim=LABEL_DIM)


def test_config_roundtrip(tmp_path):
    from uniter_amd.model.model import UniterConfig
    c = UniterConfig(100, hidden_size=128, num_attention_heads=2)
    d = json.loads(c.to_json_string())
    assert d["vocab_size"] == 100 and d["hidden_size"] == 128
    p = tmp_path / "c.json"
    p.write_text(c.to_json_string())
    c2 = UniterConfig.from_json_file(str(p))
    assert c2.to_dict() == c.to_dict()
    c3 = UniterConfig(str(p))
    assert c3.num_attention_heads == 2
    with pytest.raises(ValueError):
        UniterConfig(3.5)


def test_state_dict_keys_and_from_pretrained_renames(golden):
    model = _tiny()
    keys = set(model.state_dict().keys())
    assert keys == set(golden.pretrain_sd().keys())              # identical to the reference's key set
    sd = golden.pretrain_sd()
    old = {}
    for k, v in sd.items():                                       # TF-style names + "bert." prefix are accepted
        k2 = k.replace("LayerNorm.weight", "LayerNorm.gamma").replace("LayerNorm.bias", "LayerNorm.beta")
        old[k2] = v
    m2 = type(model).from_pretrained(TINY_CONFIG, old, img_dim=IMG_DIM, img_label_dim=LABEL_DIM)
    for k, v in m2.state_dict().items():
        torch.testing.assert_close(v, sd[k])
    with pytest.raises(RuntimeError):                             # shape errors are fatal, missing keys are not
        bad = dict(sd)
        bad["uniter.pooler.dense.weight"] = torch.zeros(3, 3)
        type(model).from_pretrained(TINY_CONFIG, bad, img_dim=IMG_DIM, img_label_dim=LABEL_DIM)
    # tied weights
    assert model.cls.predictions.decoder.weight is model.uniter.embeddings.word_embeddings.weight
    assert model.feat_regress.weight is model.uniter.img_embeddings.img_linear.weight


def test_init_weights_statistics():
    model = _tiny()
    lay = model.uniter.encoder.layer[0]
    assert float(lay.attention.self.query.bias.abs().max()) == 0
    assert float((lay.output.LayerNorm.weight - 1).abs().max()) == 0
    std = float(lay.intermediate.dense.weight.std())
    assert 0.015 < std < 0.025


def test_no_cpu_fallback():
    """The encoder hot path must fail loudly without the GPU kernels (no eager / CPU route)."""
    from uniter_amd._lib import UniterHipError
    from uniter_amd.utils.synthetic import make_batch
    model = _tiny()
    batch = make_batch('mlm', 2, max_txt_len=6, num_bb=4, img_dim=IMG_DIM, vocab_size=96)
    with pytest.raises(UniterHipError):
        model(batch, task='mlm')
    with pytest.raises(UniterHipError):
        model.uniter.encoder.layer[0](torch.zeros(1, 4, 128), torch.zeros(1, 1, 1, 4))


def test_optimizer_grouping_matches_reference_quirks():
    from uniter_amd.optim.misc import split_decay
    from uniter_amd.model.vqa import UniterForVisualQuestionAnswering
    model = UniterForVisualQuestionAnswering.from_pretrained(TINY_CONFIG, {}, img_dim=IMG_DIM, num_answer=13)
    named = dict(model.named_parameters())
    groups = split_decay(named.items(), 0.01)
    decayed = {id(p) for p in groups[0]['params']}
    assert id(named['uniter.encoder.layer.0.output.dense.weight']) in decayed
    assert id(named['uniter.encoder.layer.0.output.LayerNorm.weight']) not in decayed
    assert id(named['uniter.encoder.layer.0.output.dense.bias']) not in decayed
    # case-sensitive substring quirk: these LayerNorm weights ARE decayed in the reference
    assert id(named['uniter.img_embeddings.img_layer_norm.weight']) in decayed
    assert id(named['vqa_output.2.weight']) in decayed
    assert groups[0]['weight_decay'] == 0.01 and groups[1]['weight_decay'] == 0.0
    assert len(groups[0]['params']) + len(groups[1]['params']) == len(named)


def test_schedules():
    from uniter_amd.optim import get_lr_sched, noam_schedule, vqa_schedule, warmup_linear
    from uniter_amd.utils.misc import Struct
    from oracle import uniter_oracle as O
    opts = Struct(dict(learning_rate=3e-5, warmup_steps=800, num_train_steps=8000))
    for step in (0, 1, 400, 800, 801, 4000, 7999, 8000, 9000):
        assert get_lr_sched(step, opts) == O.get_lr_sched(step, 3e-5, 800, 8000)
        assert warmup_linear(step, 800, 8000) == O.warmup_linear(step, 800, 8000)
    assert get_lr_sched(8000, opts) == 1e-8
    assert noam_schedule(2000, 4000) == 0.5 and abs(noam_schedule(16000, 4000) - 0.5) < 1e-12
    assert vqa_schedule(0, 10, 5, 100, 0.5) == 0.25 and vqa_schedule(25, 10, 5, 100, 0.5) == 0.75
    assert vqa_schedule(50, 10, 5, 100, 0.5) == 1 and vqa_schedule(106, 10, 5, 100, 0.5) == 0.25


def test_arena_layout_and_fused_qkv():
    from uniter_amd.utils.arena import ParamArena
    model = _tiny()
    before = {k: v.clone() for k, v in model.state_dict().items()}
    n_params = sum(p.numel() for p in model.parameters())
    arena = ParamArena(model)
    assert arena.check()
    assert n_params <= arena.numel < n_params + 128 * len(arena.params)
    for k, v in model.state_dict().items():
        torch.testing.assert_close(v, before[k])
    att = model.uniter.encoder.layer[1].attention.self
    w, b = att.fused_qkv()                              # views, no copy: q/k/v are adjacent in the arena
    assert w.shape == (3 * 128, 128) and b.shape == (3 * 128,)
    assert w.data_ptr() == att.query.weight.data_ptr()
    torch.testing.assert_close(w[128:256], att.key.weight.data)
    gw, gb = att.fused_qkv_grad()
    assert gw.data_ptr() == att.query.weight.grad.data_ptr()
    lo, hi = arena.span(list(model.uniter.encoder.layer[0].parameters()))
    assert 0 < lo < hi <= arena.numel
    # zero_grad keeps the views
    att.query.weight.grad.fill_(1.0)
    arena.zero_grad()
    assert float(att.query.weight.grad.abs().max()) == 0 and arena.check()


def test_arena_grad_slot_is_cleared_when_reattached_after_zero_grad():
    """Module.zero_grad() / Optimizer.zero_grad(set_to_none=True) only drop the reference to an arena gradient slot; the old
    gradient stays in the slot.  Handing the slot out again must not carry it over (the kernels ACCUMULATE into `.grad`)."""
    from uniter_amd import _lib, ops
    from uniter_amd.utils.arena import ParamArena
    model = _tiny()
    ParamArena(model)
    att = model.uniter.encoder.layer[0].attention.self
    dense = model.uniter.encoder.layer[0].output.dense
    e0 = _lib.grad_attach_epoch()
    g = ops.ensure_grad(dense.weight)                       # first use: the slot is attached as it is (zeros)
    gw, gb = att.fused_qkv_grad()
    assert _lib.grad_attach_epoch() > e0                    # the optimizer's plan shortcut sees that gradients appeared
    assert float(g.abs().sum()) == 0 and float(gw.abs().sum()) == 0
    g += 1.0                                                # what a backward kernel does
    gw += 1.0
    gb += 1.0
    model.zero_grad()                                       # torch default: set_to_none=True
    assert dense.weight.grad is None and att.query.weight.grad is None
    g2 = ops.ensure_grad(dense.weight)
    gw2, gb2 = att.fused_qkv_grad()
    assert g2.data_ptr() == g.data_ptr() and gw2.data_ptr() == gw.data_ptr()      # the same storage ...
    assert float(g2.abs().sum()) == 0 and float(gw2.abs().sum()) == 0 and float(gb2.abs().sum()) == 0   # ... cleared
    g2 += 1.0
    assert float(g2.sum()) == g2.numel()                    # (was 2 * numel before the fix)
    # a gradient that is merely kept (zero_grad(set_to_none=False) semantics) is not touched by ensure_grad
    assert float(ops.ensure_grad(dense.weight).sum()) == g2.numel()


def test_fused_qkv_without_arena_refuses_after_cast():
    model = _tiny()
    att = model.uniter.encoder.layer[0].attention.self
    ref = torch.cat([att.query.weight.data, att.key.weight.data, att.value.weight.data], 0).clone()
    w, _ = att.fused_qkv()
    torch.testing.assert_close(w, ref)
    assert att.key.weight.data_ptr() == w.data_ptr() + 128 * 128 * 4
    model.bfloat16()                                     # .to()/.bfloat16() gives every parameter its own storage again
    w2, _ = att.fused_qkv()
    assert w2.dtype == torch.bfloat16
    torch.testing.assert_close(w2.float(), ref.bfloat16().float())
    assert att.value.weight.data_ptr() == w2.data_ptr() + 2 * 128 * 128 * 2


def test_synthetic_batch_contract():
    from uniter_amd.utils.synthetic import make_batch
    from oracle import uniter_oracle as O
    b = make_batch('mlm', 8, seed=1, ragged=True)
    assert b['input_ids'].dtype == torch.long and b['position_ids'].shape == (1, b['input_ids'].shape[1])
    assert b['img_feat'].shape[2] == 2048 and b['img_pos_feat'].shape[2] == 7
    lens = b['attn_masks'].sum(1)
    assert b['attn_masks'].shape[1] == int(lens.max())
    tl = (b['input_ids'] != 0).sum(1)
    nbb = lens - tl
    gi = O.get_gather_index(tl.tolist(), nbb.tolist(), 8, b['input_ids'].shape[1], b['attn_masks'].shape[1])
    assert torch.equal(gi, b['gather_index'])
    assert ((b['txt_labels'] != -1).sum(1) >= 1).all()
    m = make_batch('mrfr', 4, seed=2, ragged=True)
    assert m['feat_targets'].shape[0] == int(m['img_masks'].sum()) == int(m['img_mask_tgt'].sum())
    assert float(m['img_feat'][m['img_masks']].abs().max()) == 0.0
    n = make_batch('nlvr2', 6, seed=3)
    assert n['targets'].shape == (3,) and torch.equal(n['input_ids'][0], n['input_ids'][1])
    assert set(n['img_type_ids'].unique().tolist()) == {1, 2}
    v = make_batch('vqa', 3, seed=4)
    assert v['targets'].shape == (3, 3129) and float(v['targets'].max()) <= 1.0


def test_vqa_optimizer_has_four_groups_in_reference_order():
    """train_vqa.py:51-86: vqa_output (decayed, not decayed) first — their lr is multiplied by lr_mul every step —
    then the rest (decayed, not decayed).  `vqa_output.2.weight` is a LayerNorm weight inside nn.Sequential: decayed."""
    from uniter_amd.optim.misc import NO_DECAY
    from uniter_amd.utils.misc import Struct

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.body = torch.nn.Linear(4, 4)
            self.LayerNorm = torch.nn.LayerNorm(4)
            self.vqa_output = torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.GELU(), torch.nn.LayerNorm(8),
                                                  torch.nn.Linear(8, 3))

    model = Toy()
    opts = Struct(dict(learning_rate=1e-3, betas=(0.9, 0.98), weight_decay=0.01, optim='adam'))
    from uniter_amd.optim import build_vqa_optimizer
    opt = build_vqa_optimizer(model, opts)
    named = dict(model.named_parameters())
    ids = [[id(p) for p in g['params']] for g in opt.param_groups]
    assert len(ids) == 4
    want = [[], [], [], []]
    for n, p in named.items():
        top = 'vqa_output' in n
        nd = any(t in n for t in NO_DECAY)
        want[(0 if top else 2) + (1 if nd else 0)].append(id(p))
    assert ids == want
    assert id(named['vqa_output.2.weight']) in ids[0]                   # the quirk: Sequential LayerNorm weight is decayed
    assert [g['weight_decay'] for g in opt.param_groups] == [0.01, 0.0, 0.01, 0.0]


def test_pack_indices_layout():
    """Host tables of the padding-free layout (UniterEncoder.forward_packed): real-token row indices in example order,
    cumulative lengths, and all-zero dummy examples that round the row count up to a multiple of 64."""
    from uniter_amd.model.model import pack_indices
    idx, cu, total, extra = pack_indices([3, 0, 5], max_len=8, multiple=4)
    assert total == 8 and extra == []
    assert idx.tolist() == [0, 1, 2, 16, 17, 18, 19, 20]
    assert cu.tolist() == [0, 3, 3, 8] and cu.dtype == torch.int32
    idx, cu, total, extra = pack_indices(torch.tensor([60, 96, 41]), max_len=96)
    assert total == 197 and extra == [59] and cu.tolist() == [0, 60, 156, 197, 256]
    assert int(cu[-1]) % 64 == 0 and idx.numel() == total
    # a remainder longer than one sequence becomes several dummies
    _, cu, total, extra = pack_indices([5], max_len=16)
    assert total == 5 and extra == [16, 16, 16, 11] and int(cu[-1]) == 64
    with pytest.raises(ValueError):
        pack_indices([9], max_len=8)


def test_noop_and_parse_with_config(tmp_path):
    """utils/misc.py:17-36 of the reference: rank > 0 stand-ins, and --config JSON below the command line in precedence."""
    import argparse
    import json
    from uniter_amd.utils.misc import NoOp, parse_with_config
    n = NoOp()
    assert n.update(3) is None and n.add_scalar("x", 1.0, step=2) is None and n.close() is None
    cfg = tmp_path / "c.json"
    cfg.write_text(json.dumps({"learning_rate": 5e-5, "train_batch_size": 4096, "extra_key": [1, 2]}))
    ap = argparse.ArgumentParser()
    ap.add_argument("--config")
    ap.add_argument("--learning_rate", type=float, default=3e-5)
    ap.add_argument("--train_batch_size", type=int, default=1024)
    a = parse_with_config(ap, ["--config", str(cfg), "--learning_rate=1e-4"])
    assert a.learning_rate == 1e-4                    # spelled on the command line: wins over the file
    assert a.train_batch_size == 4096                 # only in the file (the parser's default loses)
    assert a.extra_key == [1, 2]                      # keys the parser does not know are attached too
    assert not hasattr(a, "config")
    b = parse_with_config(ap, ["--train_batch_size", "8"])
    assert b.train_batch_size == 8 and b.learning_rate == 3e-5 and not hasattr(b, "config")


def test_lazy_zero_grad_host_state_machine():
    """Host side of AdamW.lazy_zero / fold_norm (round 6), no GPU: which plan tensors lie inside the gradient storages an encoder
    backward overwrites (`_covered`: whole-tensor containment, the three thirds of a fused q|k|v storage each), and that a reader
    who comes before a backward gets zeros (`_lib.lazy_resolve`)."""
    from uniter_amd import _lib
    from uniter_amd.optim import AdamW
    opt = AdamW([torch.nn.Parameter(torch.zeros(4))], lr=1e-3)
    # plan tensors as (gradient address, bytes): three thirds of one fused storage, a neighbour, one straddling a range end, one apart
    opt._plan_grads = [(1000, 100), (1100, 100), (1200, 100), (1300, 64), (1990, 20), (5000, 8)]
    ranges = frozenset({(1000, 300), (1300, 64), (1900, 100)})
    assert opt._covered(ranges) == [True, True, True, True, False, False]
    assert opt._covered(frozenset()) == [False] * 6
    # undefined gradients are resolved to zeros on demand, once
    a, b = torch.ones(5), torch.full((3,), 2.0)
    saved = (_lib.lazy_tensors, _lib.lazy_undefined)
    try:
        _lib.lazy_tensors, _lib.lazy_undefined = [a, b], True
        _lib.lazy_resolve()
        assert not _lib.lazy_undefined and float(a.abs().sum()) == 0.0 and float(b.abs().sum()) == 0.0
        a.fill_(7.0)
        _lib.lazy_resolve()                      # nothing is undefined any more: untouched
        assert float(a[0]) == 7.0
    finally:
        _lib.lazy_tensors, _lib.lazy_undefined = saved
