"""GPU parity tests: the HIP path (through the drop-in modules and the C ABI) against
  (1) the golden vectors of the real reference (tests/golden/uniter_tiny.npz),
  (2) the CPU oracle on seeded inputs at the UNITER-base model size,
  (3) size-independent properties at the full benchmark shape (B=32, L=96).

Tolerances (ours = bf16 storage with fp32 accumulation / statistics, oracle = fp32):
  hidden states   |diff| <= 4e-2 + 1.6e-2*|ref| (5e-2 at |ref| ~ 0.6; the relative part is two bf16 ulps, for the
                  few LayerNorm outputs of magnitude > 2 whose single rounding step already is 1.6e-2) and
                  mean|diff| <= 5e-3
  per-example loss  rtol 2e-2 (+ small atol)
  gradients       cosine >= 0.99 and relative L2 error <= 5e-2 per tensor
  AdamW from identical fp32 gradients   rtol 1e-5 on the fp32 state
"""
import copy

import pytest
import torch

from oracle import uniter_oracle as O
from tests.common import IMG_DIM, LABEL_DIM, N_ANS, TINY_CONFIG, cosine, rel_l2

pytestmark = pytest.mark.gpu

# final hidden max-abs-diff <= 5e-2 (flat), mean-abs-diff <= 5e-3, per-example loss rtol 2e-2, per-tensor
# gradient cosine >= 0.99 and relative L2 <= 5e-2; secondary yardstick 2x the error of the oracle's torch ops run in bf16.
HID_ATOL, HID_RTOL, HID_MEAN, LOSS_RTOL, GRAD_COS, GRAD_L2 = 5e-2, 0.0, 5e-3, 2e-2, 0.99, 5e-2
# parameters of the task heads (NLVR2 cross attention / pooling, MLM / MRC / VQA heads: a short stack of small bf16 GEMMs behind
# the 12-24 encoder layers, whose accumulated bf16 error they inherit un-normalised):
GRAD_L2_HEAD = 1e-1


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda", 0)


def _to_dev(batch):
    from uniter_amd.utils.synthetic import to_device
    return to_device(batch, _dev())


def _prep(model):
    from uniter_amd.utils.misc import set_dropout
    model.to(_dev()).bfloat16()
    set_dropout(model, 0.0)
    for m in model.modules():                     # MultiheadAttention keeps its dropout as a float attribute
        if hasattr(m, 'dropout') and isinstance(m.dropout, float):
            m.dropout = 0.0
    model.train()
    return model


def _yardstick(fn, sd_fp32, cfg, batch):
    """Secondary yardstick: the oracle's plain torch ops run in bf16 on the GPU — i.e. what an
    unfused PyTorch-bf16 implementation of the same model achieves against the fp32 oracle.  Where the absolute
    tolerances are too tight for 12 stacked bf16 layers, the HIP path must stay within 2x of this error."""
    dev = _dev()
    sd = {k: v.detach().to(dev, torch.bfloat16).requires_grad_(True) for k, v in sd_fp32.items()
          if k != 'cls.predictions.decoder.weight'}
    if 'uniter.embeddings.word_embeddings.weight' in sd:
        sd['cls.predictions.decoder.weight'] = sd['uniter.embeddings.word_embeddings.weight']
    b = {k: ((v.to(dev, torch.bfloat16) if v.is_floating_point() else v.to(dev)) if torch.is_tensor(v) else v)
         for k, v in batch.items()}
    loss, seq = fn(sd, cfg, b)
    loss.float().mean().backward()
    grads = {k: v.grad.float().cpu() for k, v in sd.items() if v.grad is not None}
    return loss.detach().float().cpu(), seq.detach().float().cpu(), grads


def _yard_hidden(sd_fp32, cfg, batch, all_layers=False, text_only=False, txt_mask=None):
    """The survey's secondary yardstick for hidden states: the oracle's encoder run with torch ops in bf16 on the GPU."""
    dev = _dev()
    sd = {k: v.detach().to(dev, torch.bfloat16) for k, v in sd_fp32.items()}
    b = {k: ((v.to(dev, torch.bfloat16) if v.is_floating_point() else v.to(dev)) if torch.is_tensor(v) else v) for k, v in batch.items()}
    with torch.no_grad():
        if text_only:
            out = O.uniter_model(sd, cfg, b['input_ids'], b['position_ids'], None, None, txt_mask.to(dev))
        else:
            out = O.uniter_model(sd, cfg, b['input_ids'], b['position_ids'], b['img_feat'], b['img_pos_feat'], b['attn_masks'],
                                 b['gather_index'], all_layers=all_layers)
    return [o.float().cpu() for o in out] if all_layers else out.float().cpu()


def _check_hidden(got, ref, what, yard=None):
    d = (got.float().cpu() - ref).abs()
    excess = d - (HID_ATOL + HID_RTOL * ref.abs())
    ok_abs = float(excess.max()) <= 0 and float(d.mean()) <= HID_MEAN
    if ok_abs:
        return
    assert yard is not None, (what, float(d.max()), float(d.mean()))
    dy = (yard - ref).abs()
    assert float(d.max()) <= 2 * float(dy.max()) and float(d.mean()) <= 2 * float(dy.mean()), \
        (what, "ours max/mean", float(d.max()), float(d.mean()), "torch-bf16 max/mean", float(dy.max()), float(dy.mean()))


def _check_loss(loss, ref, atol=2e-2):
    """Per-example loss: heads that return element-wise losses (MRFR / MRC-KL / VQA: [n, dim]) are compared after the
    mean over their last dimension — a single squared-error element is not a meaningful unit at bf16."""
    got = loss.detach().float().cpu()
    if got.dim() > 1:
        got, ref = got.mean(-1), ref.mean(-1)
    torch.testing.assert_close(got, ref, rtol=LOSS_RTOL, atol=atol)


def _check_grad(name, g, g_ref, yard=None):
    """cosine >= 0.99 and relative L2 <= 5e-2 (1e-1 for the task heads' own parameters); where stacked bf16
    layers exceed the absolute bound, at most 2x the error of the oracle's own torch ops run in bf16 on the GPU (the survey's
    secondary yardstick).  No tensor-specific bounds: round 3's 3.5x for query / key went with the exact softmax row term of
    the attention backward, rounds 4-5's measured-amplification bound for AttentionPool's Linear(H, 1) went in round 6 (see
    _joint_one_element)."""
    g = g.float().cpu()
    scale = float(g_ref.abs().max())
    if scale < 1e-6:                               # mathematically zero gradients (e.g. key bias): absolute check
        assert float(g.abs().max()) < 1e-3, name
        return
    assert cosine(g, g_ref) >= GRAD_COS, (name, cosine(g, g_ref))
    limit = GRAD_L2 if name.startswith(('uniter.', 'encoder.', 'embeddings.', 'img_embeddings.')) else GRAD_L2_HEAD
    if yard is not None:
        limit = max(limit, 2.0 * rel_l2(yard, g_ref))
    assert rel_l2(g, g_ref) <= limit, (name, rel_l2(g, g_ref), None if yard is None else rel_l2(yard, g_ref))


def _joint_one_element(named_grads, ref_grads, yard_grads):
    """Cosine and relative L2 are per-TENSOR measures; on a one-element tensor they degenerate (the cosine of two scalars is +-1,
    the relative L2 is the relative error of ONE sum).  The only such parameter of the in-scope models is the bias of AttentionPool's
    Linear(H, 1) (model/nlvr2.py:113): sum over 32 x 96 tokens of softmax-backward terms that cancel to ~1/10 of their mass, so its
    relative error is a coin toss around 0.1 for ANY bf16 encoder in front of it (scripts/diag_pool_parity.py,
    profiles/r06_attention_pool_gradient_diag.txt: the pool kernels reproduce the fp32 formula on their own inputs to 2e-3, our
    pool input is closer to the oracle's than the torch-bf16 yardstick's, and ONE bf16 rounding of the oracle's own pool input
    already moves the scalar by 0.03).  Such a bias is therefore checked as what it is — the homogeneous coordinate of its module's
    weight: {weight, bias} of a module with a one-element bias are concatenated into ONE tensor and held to the ordinary bounds.
    Returns name -> (ours, ref, yard) for the joint tensors and the set of names they replace."""
    joint, replaced = {}, set()
    for name, g in named_grads.items():
        if not name.endswith('.bias') or g.numel() != 1:
            continue
        wname = name[:-len('bias')] + 'weight'
        if wname not in named_grads or wname not in ref_grads or name not in ref_grads:
            continue
        cat = lambda d: torch.cat([d[wname].float().cpu().reshape(-1), d[name].float().cpu().reshape(-1)])
        yard = cat(yard_grads) if (yard_grads and wname in yard_grads and name in yard_grads) else None
        joint[name[:-len('.bias')] + '.{weight,bias}'] = (cat(named_grads), cat(ref_grads), yard)
        replaced.update((name, wname))
    return joint, replaced


# --------------------------------------------------------------------------------------------------------------
# (1) golden vectors of the real reference
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("task", ['mlm', 'mrfr', 'mrckl', 'mrc', 'itm'])
def test_pretraining_tasks_vs_reference_golden(golden, task):
    from uniter_amd.model.pretrain import UniterForPretraining
    sd = golden.pretrain_sd()
    model = _prep(UniterForPretraining.from_pretrained(TINY_CONFIG, sd, img_dim=IMG_DIM, img_label_dim=LABEL_DIM))
    batch = golden.batch(task)
    ref = golden.out(task)
    loss = model(_to_dev(batch), task=task, compute_loss=True)
    if task == 'itm':
        loss = loss[0]
    seq = model.uniter(*[_to_dev(batch)[k] for k in ('input_ids', 'position_ids', 'img_feat', 'img_pos_feat',
                                                     'attn_masks', 'gather_index')],
                       output_all_encoded_layers=False, img_masks=_to_dev(batch).get('img_masks'))
    valid = batch['attn_masks'].bool()
    _check_hidden(seq.detach()[valid.to(seq.device)], ref['seq'][valid], task)
    _check_loss(loss, ref['loss'])
    model.zero_grad()
    for p in model.parameters():
        p.grad = None
    loss = model(_to_dev(batch), task=task, compute_loss=True)
    if task == 'itm':
        loss = loss[0]
    loss.mean().backward()
    named = dict(model.named_parameters())
    for name, g_ref in golden.grads(task).items():
        assert named[name].grad is not None, name
        _check_grad(name, named[name].grad, g_ref)


def test_vqa_vs_reference_golden(golden):
    from uniter_amd.model.vqa import UniterForVisualQuestionAnswering
    sd = {**golden.weights('pre'), **golden.weights('vqa')}
    model = _prep(UniterForVisualQuestionAnswering.from_pretrained(TINY_CONFIG, sd, img_dim=IMG_DIM, num_answer=N_ANS))
    batch = golden.batch('vqa')
    loss = model(_to_dev(batch), compute_loss=True)
    _check_loss(loss, golden.out('vqa')['loss'])
    (loss.mean() * loss.shape[1]).backward()
    named = dict(model.named_parameters())
    for name, g_ref in golden.grads('vqa').items():
        _check_grad(name, named[name].grad, g_ref)


def test_nlvr2_paired_attn_vs_reference_golden(golden):
    from uniter_amd.model.nlvr2 import UniterForNlvr2PairedAttn
    w = golden.weights('pre')
    nl = golden.weights('nlvr2')
    table3 = nl.pop('uniter.embeddings.token_type_embeddings.weight')
    model = UniterForNlvr2PairedAttn.from_pretrained(TINY_CONFIG, {**w, **nl}, img_dim=IMG_DIM)
    model.init_type_embedding()
    model.uniter.embeddings.token_type_embeddings.weight.data.copy_(table3)
    _prep(model)
    batch = golden.batch('nlvr2')
    loss = model(_to_dev(batch), compute_loss=True)
    _check_loss(loss, golden.out('nlvr2')['loss'])
    loss.mean().backward()
    named = dict(model.named_parameters())
    w_all = golden.weights('pre')
    w_all.update(golden.weights('nlvr2'))
    _, _, yard = _yardstick(O.nlvr2_paired_attn_loss, w_all, golden.cfg, batch)
    for name, g_ref in golden.grads('nlvr2').items():
        _check_grad(name, named[name].grad, g_ref, yard.get(name))


def test_state_dict_keys_match_reference(golden):
    from uniter_amd.model.pretrain import UniterForPretraining
    model = UniterForPretraining.from_pretrained(TINY_CONFIG, {}, img_dim=IMG_DIM, img_label_dim=LABEL_DIM)
    assert set(model.state_dict().keys()) == set(golden.pretrain_sd().keys())


# --------------------------------------------------------------------------------------------------------------
# (2) CPU oracle at the UNITER-base model size (config/uniter-base.json shapes), ragged batch
# --------------------------------------------------------------------------------------------------------------
BASE_CFG = dict(vocab_size=28996, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
                max_position_embeddings=512, type_vocab_size=2, initializer_range=0.02)


def _base_model(tmp_path, n_layers=12):
    import json
    from uniter_amd.model.pretrain import UniterForPretraining
    cfg = dict(BASE_CFG, num_hidden_layers=n_layers)
    path = tmp_path / "base.json"
    path.write_text(json.dumps(cfg))
    torch.manual_seed(7)
    model = UniterForPretraining.from_pretrained(str(path), {}, img_dim=2048, img_label_dim=1601)
    with torch.no_grad():                                   # non-trivial biases / LayerNorm parameters
        g = torch.Generator().manual_seed(8)
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.02)
            p.copy_(p.to(torch.bfloat16).float())            # bf16-representable: both sides start from identical weights
    return model, cfg


def test_base_model_mlm_step_vs_oracle(tmp_path):
    from uniter_amd.utils.synthetic import make_batch
    model, cfg = _base_model(tmp_path)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    batch = make_batch('mlm', 4, seed=3, ragged=True)
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != 'cls.predictions.decoder.weight'}
    leaf['cls.predictions.decoder.weight'] = leaf['uniter.embeddings.word_embeddings.weight']
    ref_loss, ref_seq = O.mlm_loss(leaf, cfg, batch)
    ref_loss.mean().backward()

    _prep(model)
    dbatch = _to_dev(batch)
    loss = model(dbatch, task='mlm', compute_loss=True)
    seq = model.uniter(dbatch['input_ids'], dbatch['position_ids'], dbatch['img_feat'], dbatch['img_pos_feat'],
                       dbatch['attn_masks'], dbatch['gather_index'], output_all_encoded_layers=False)
    valid = batch['attn_masks'].bool()
    _, yseq, ygrads = _yardstick(O.mlm_loss, sd, cfg, batch)
    _check_hidden(seq.detach()[valid.to(seq.device)], ref_seq.detach()[valid], "base mlm hidden", yseq[valid])
    _check_loss(loss, ref_loss.detach(), atol=3e-2)
    for p in model.parameters():
        p.grad = None
    model(dbatch, task='mlm', compute_loss=True).mean().backward()
    named = dict(model.named_parameters())
    checked = 0
    for name, p in named.items():
        ref_g = leaf[name].grad if name in leaf else None
        if ref_g is None or p.grad is None:
            continue
        _check_grad(name, p.grad, ref_g, ygrads.get(name))
        checked += 1
    assert checked > 200


def test_base_encoder_all_layers_and_text_only(tmp_path):
    """output_all_encoded_layers=True (+ a loss that uses an intermediate layer) and the text-only branch."""
    from uniter_amd.utils.synthetic import make_batch
    model, cfg = _base_model(tmp_path, n_layers=3)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    batch = make_batch('itm', 3, seed=5, ragged=True)
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    outs = O.uniter_model(leaf, cfg, batch['input_ids'], batch['position_ids'], batch['img_feat'], batch['img_pos_feat'],
                          batch['attn_masks'], batch['gather_index'], all_layers=True)
    valid = batch['attn_masks'].bool()
    w = torch.randn(outs[0].shape, generator=torch.Generator().manual_seed(1))
    ref_obj = ((outs[0] * w)[valid].sum() + (outs[2] * w)[valid].sum() * 0.5) / 100.0
    ref_obj.backward()

    _prep(model)
    d = _to_dev(batch)
    got = model.uniter(d['input_ids'], d['position_ids'], d['img_feat'], d['img_pos_feat'], d['attn_masks'],
                       d['gather_index'], output_all_encoded_layers=True)
    assert len(got) == 3
    youts = _yard_hidden(sd, cfg, batch, all_layers=True)
    for l in range(3):
        _check_hidden(got[l].detach()[valid.to(got[l].device)], outs[l].detach()[valid], "layer %d" % l, youts[l][valid])
    wd = w.to(got[0].device)
    obj = ((got[0].float() * wd)[valid.to(wd.device)].sum() + (got[2].float() * wd)[valid.to(wd.device)].sum() * 0.5) / 100.0
    obj.backward()
    named = dict(model.named_parameters())
    for name in ('uniter.encoder.layer.0.output.dense.weight', 'uniter.encoder.layer.1.attention.self.value.weight',
                 'uniter.encoder.layer.2.intermediate.dense.bias', 'uniter.img_embeddings.img_linear.weight'):
        _check_grad(name, named[name].grad, leaf[name].grad)
    # text-only branch (model/model.py:351-355)
    txt_mask = (batch['input_ids'] != 0).long()
    ref_txt = O.uniter_model(leaf, cfg, batch['input_ids'], batch['position_ids'], None, None, txt_mask)
    got_txt = model.uniter(d['input_ids'], d['position_ids'], None, None, txt_mask.to(_dev()), output_all_encoded_layers=False)
    _check_hidden(got_txt.detach()[txt_mask.bool().to(_dev())], ref_txt.detach()[txt_mask.bool()], "text only",
                  _yard_hidden(sd, cfg, batch, text_only=True, txt_mask=txt_mask)[txt_mask.bool()])


# --------------------------------------------------------------------------------------------------------------
# (3) properties at the benchmark shape: UNITER-base, B=32, L=60+36
# --------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def full_model(tmp_path_factory):
    model, cfg = _base_model(tmp_path_factory.mktemp("full"))
    _prep(model)
    return model


def _full_batch(seed=0, ragged=False):
    from uniter_amd.utils.synthetic import make_batch
    return _to_dev(make_batch('itm', 32, seed=seed, ragged=ragged))


def _run(model, b):
    return model.uniter(b['input_ids'], b['position_ids'], b['img_feat'], b['img_pos_feat'], b['attn_masks'],
                        b['gather_index'], output_all_encoded_layers=False)


def test_full_size_determinism_and_eval_equals_p0(full_model):
    b = _full_batch()
    y1 = _run(full_model, b)
    y2 = _run(full_model, b)
    assert torch.equal(y1, y2)                                   # bit-exact repeatability (no atomics in the forward)
    assert y1.shape == (32, 96, 768)
    full_model.eval()
    with torch.no_grad():
        y3 = _run(full_model, b)
    full_model.train()
    assert torch.equal(y1, y3)                                   # dropout p=0 training == inference
    # rows are LayerNorm outputs: per-row mean/var follow the LN affine parameters, finite everywhere
    assert torch.isfinite(y1.float()).all()


@pytest.mark.parametrize("mode", ["dense96", "ragged", "packed"])
def test_overlapped_kernel_chain_is_bit_identical_to_in_order_launches(full_model, mode):
    """The encoder's forward and data-gradient chains dispatched WITHOUT the queue barrier between dependent kernels and ordered
    by row-block flags (include/uniter_hip.h "Overlapped kernel chains", ABI v7) against the same kernels launched in order, at
    the benchmark shape with dropout on: every layer's output and every gradient, bit for bit.  dense96 = 32 x 96 tokens, every
    kernel of the call in the chain; ragged = a padded batch whose length is not a multiple of 32 (attention drops out of the
    chain, the GEMMs / LayerNorms around it stay in); packed = padding-free rows (cu_seqlens)."""
    from uniter_amd import _lib, ops
    from uniter_amd.utils.misc import set_dropout
    lib = _lib.load()
    b = _full_batch(seed=11, ragged=(mode != "dense96"))
    params = [p for p in full_model.uniter.parameters()]
    set_dropout(full_model, 0.1)
    full_model.uniter.pack_padding = (mode == "packed")

    def run(enable):
        lib.uniter_encoder_debug_chain(enable)
        ops.manual_seed(1234)
        for p in params:
            p.grad = None
        ys = full_model.uniter(b['input_ids'], b['position_ids'], b['img_feat'], b['img_pos_feat'], b['attn_masks'],
                               b['gather_index'], output_all_encoded_layers=True)
        (ys[-1].float() * b['attn_masks'].unsqueeze(-1)).square().mean().backward()
        torch.cuda.synchronize()
        return [y.detach().clone() for y in ys], [None if p.grad is None else p.grad.detach().clone() for p in params]

    try:
        y0, g0 = run(0)
        runs = [run(1) for _ in range(3)]            # (a race would not show on every launch)
    finally:
        lib.uniter_encoder_debug_chain(0)            # (the library's default: chains are opt-in, EXPERIMENTS.md section 10)
        set_dropout(full_model, 0.0)
        full_model.uniter.pack_padding = False
    assert len(y0) == 12
    for y1, g1 in runs:
        for a, c in zip(y0, y1):
            assert torch.equal(a, c)
        for a, c in zip(g0, g1):
            assert (a is None) == (c is None)
            if a is not None:
                assert torch.equal(a, c)


def test_full_size_padding_invariance(full_model):
    """Valid positions must not depend on what sits in padded slots (key mask -10000, model/model.py:342-345)."""
    b = _full_batch(seed=4, ragged=True)
    y1 = _run(full_model, b)
    b2 = dict(b)
    feat = b['img_feat'].clone()
    nbb_pad = (b['img_feat'].abs().sum(-1) == 0)                 # padded region rows are all-zero features
    feat[nbb_pad] = 3.0
    b2['img_feat'] = feat
    ids = b['input_ids'].clone()
    ids[ids == 0] = 777
    b2['input_ids'] = ids
    y2 = _run(full_model, b2)
    valid = b['attn_masks'].bool()
    assert torch.equal(y1[valid], y2[valid])


def test_full_size_backward_linearity_and_accumulation(full_model):
    b = _full_batch(seed=9)
    params = [p for p in full_model.uniter.parameters()]

    def grads(scale, repeat=1):
        for p in params:
            p.grad = None
        for _ in range(repeat):
            y = _run(full_model, b)
            (y.float().pow(2).mean() * scale).backward()
        return {n: p.grad.detach().float().clone() for n, p in full_model.uniter.named_parameters() if p.grad is not None}

    g1 = grads(1.0)
    g2 = grads(2.0)
    g11 = grads(1.0, repeat=2)                                   # accumulation over two micro-steps sums (pretrain.py:298-312)
    for name in ('encoder.layer.0.attention.self.query.weight', 'encoder.layer.11.output.dense.weight',
                 'encoder.layer.5.intermediate.dense.bias', 'encoder.layer.3.attention.output.LayerNorm.weight',
                 'img_embeddings.img_linear.weight', 'embeddings.position_embeddings.weight'):
        assert rel_l2(g2[name], 2 * g1[name]) < 2e-2, name
        assert rel_l2(g11[name], 2 * g1[name]) < 2e-2, name


def test_deferred_parameter_gradients_match_per_layer_launches(full_model, monkeypatch):
    """The backward call's one deferred launch (every weight / bias / LayerNorm-parameter gradient of its layers on the 256 x 256
    tile, after the data-gradient chain) against the per-layer grouped launches beside the chain (no stage registered) — the
    same numbers up to the summation order of the bias / LayerNorm column sums — and with the join of that launch left to the
    consumer (`ops.DeferWgradJoin` + `_lib.join_wgrads()`, what the training loop does) exactly the same bits again."""
    from uniter_amd import _lib, ops
    full_model.train()
    b = _full_batch(seed=21)
    params = dict(full_model.uniter.named_parameters())

    def grads():
        for p in params.values():
            p.grad = None
        y = _run(full_model, b)
        y.float().pow(2).mean().backward()
        _lib.join_wgrads()
        torch.cuda.synchronize()
        return {n: p.grad.detach().float().clone() for n, p in params.items() if p.grad is not None}

    assert ops._WGRAD_STAGE, "the stage is on by default"
    g_def = grads()
    enc = full_model.uniter.encoder
    old_hook = enc.grad_ready_hook
    try:
        enc.grad_ready_hook = ops.DeferWgradJoin()
        g_join = grads()
    finally:
        enc.grad_ready_hook = old_hook
    # data-parallel form of the launch: tiles walked bucket by bucket (4 layers each), a flag per bucket — same gradients (the plain
    # walk runs its few leftover tiles as two K slices, i.e. adds their fp32 partial sums in another order), flags raised
    import ctypes
    from uniter_amd.utils import distributed as D
    lib = _lib.load()
    seen = []

    def on_layer(layer):                                           # (runs on autograd's thread, whose launch this is)
        if layer != 0:
            return
        nb = ctypes.c_int32(0)
        lib.uniter_encoder_grad_bucket_count(ctypes.byref(nb))
        seen.append(nb.value)
        sp = ctypes.c_void_p()
        lib.uniter_encoder_side_stream(ctypes.byref(sp))
        done = torch.cuda.Event(enable_timing=True)
        done.record(torch.cuda.ExternalStream(sp.value))           # the end of the deferred launch (it is already enqueued there)
        raised.append(done)
        with torch.cuda.stream(watch):                             # a stream of its own waits for every bucket's flag in turn
            for k in range(nb.value):
                lib.uniter_encoder_bucket_wait(ctypes.c_int32(k), ctypes.c_void_p(watch.cuda_stream))
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(watch)
                raised.append(ev)
    watch, raised = torch.cuda.Stream(), []
    with torch.cuda.stream(watch):                                 # (a stream's hardware queue is set up at its first submission: not inside the measurement)
        torch.zeros(8, device=_dev()).add_(1)
    torch.cuda.synchronize()
    try:
        enc.grad_ready_hook = D._LayerHook(on_layer, set(), joins_side_stream=False, defer_wgrad_join=True, grad_buckets=lambda: 4)
        g_bucket = grads()
        assert seen == [3], seen
        # the buckets complete IN ORDER, a good part of the launch apart (a bucket's LayerNorm strips run ahead of its tiles; left to
        # the end of the launch they would hold every bucket but the first until then) — what lets an allreduce start early
        done, flags = raised[0], raised[1:]
        before_end = [flags[k].elapsed_time(done) * 1e3 for k in range(3)]
        print("bucket flags of the deferred launch raised %.0f / %.0f / %.0f us before its end" % tuple(before_end))
        assert before_end[0] > before_end[1] + 40.0 and before_end[1] > before_end[2] + 40.0, before_end
    finally:
        enc.grad_ready_hook = old_hook
    for n in g_def:
        r = rel_l2(g_bucket[n], g_def[n])
        assert r < (1e-3 if n.endswith('weight') and 'LayerNorm' not in n else 4e-3), ("bucketed", n, r)
    monkeypatch.setattr(ops, "_WGRAD_STAGE", False)
    g_layer = grads()
    assert set(g_def) == set(g_layer) == set(g_join)
    worst = ("", 0.0)
    for n in g_def:
        assert torch.equal(g_def[n], g_join[n]), n                 # who waits for the launch does not change what it computes
        if not n.startswith('encoder.'):
            assert torch.equal(g_def[n], g_layer[n]), n            # the data-gradient chain is untouched: embeddings bit-identical
            continue
        r = rel_l2(g_def[n], g_layer[n])
        worst = max(worst, (n, r), key=lambda t: t[1])
        # weights: the same products summed tile by tile in fp32 either way; biases / LayerNorm: fp32 column sums in another order
        assert r < (1e-3 if n.endswith('weight') and 'LayerNorm' not in n else 4e-3), (n, r)
    print("deferred vs per-layer parameter gradients: %d tensors, worst rel-L2 %.2e (%s)" % (len(g_def), worst[1], worst[0]))


def test_full_size_dropout_statistics():
    """Philox dropout: kept fraction ~ 1-p, scaled by 1/(1-p), same (seed, offset) -> same mask; different offset -> different."""
    import ctypes
    from uniter_amd._lib import C, ptr, stream_ptr
    M, N, K = 3072, 768, 64
    x = torch.zeros(M, K, dtype=torch.bfloat16, device=_dev())
    w = torch.zeros(N, K, dtype=torch.bfloat16, device=_dev())
    bias = torch.ones(N, dtype=torch.bfloat16, device=_dev())
    out = [torch.empty(M, N, dtype=torch.bfloat16, device=_dev()) for _ in range(3)]
    for o, (seed, off) in zip(out, [(5, 0), (5, 0), (5, 1)]):
        C.uniter_gemm_bias_dropout_residual_fwd(ptr(x), ptr(w), ptr(bias), None, ptr(o), M, N, K, ctypes.c_float(0.1), seed, off,
                                                stream_ptr())
    torch.cuda.synchronize()
    kept = (out[0] != 0).float().mean().item()
    assert abs(kept - 0.9) < 2e-3, kept
    vals = out[0][out[0] != 0].float()
    assert torch.allclose(vals, torch.full_like(vals, 1 / 0.9), rtol=1e-2)
    assert torch.equal(out[0], out[1])
    assert not torch.equal(out[0], out[2])


# --------------------------------------------------------------------------------------------------------------
# optimizer
# --------------------------------------------------------------------------------------------------------------
def test_adamw_vs_reference_golden(golden):
    """Two clipped AdamW steps on the golden MLM gradients == weights produced by the reference optimizer."""
    from uniter_amd.optim import AdamW, clip_grad_norm_
    from uniter_amd.optim.misc import split_decay
    sd = golden.weights('pre')
    grads = golden.grads('mlm')
    norm_ref, p_ref, v_ref = golden.adamw()
    names = [n for n in sd if n in grads]
    for dtype, rtol in ((torch.float32, 1e-5), (torch.bfloat16, 1e-5)):
        params = {n: torch.nn.Parameter(sd[n].to(_dev(), dtype)) for n in names}
        for n, p in params.items():
            p.grad = grads[n].to(_dev(), dtype)
        opt = AdamW(split_decay(params.items(), 0.01), lr=1e-3, betas=(0.9, 0.98))
        for _ in range(2):
            norm = clip_grad_norm_(opt, 0.5)
            opt.step()
        if dtype == torch.float32:
            assert abs(float(norm) - norm_ref) / norm_ref < 1e-5
            for n, want in p_ref.items():
                torch.testing.assert_close(params[n].detach().cpu(), want, rtol=rtol, atol=1e-7)
            for n, want in v_ref.items():
                torch.testing.assert_close(opt.state[params[n]]['exp_avg_sq'].cpu(), want, rtol=1e-5, atol=1e-12)
        else:
            # bf16 parameters: the fp32 master copy follows the reference up to the bf16 rounding of the gradients
            for n, want in p_ref.items():
                master = opt.state[params[n]]['master'].cpu()
                assert rel_l2(master, want) < 2e-3, n
                torch.testing.assert_close(params[n].detach().float().cpu(), master.to(torch.bfloat16).float(), rtol=0, atol=0)


def test_adamw_skips_params_without_grad_and_zero_grad_keeps_storage():
    from uniter_amd.optim import AdamW
    a = torch.nn.Parameter(torch.ones(1000, device=_dev()))
    b = torch.nn.Parameter(torch.ones(10, device=_dev()))
    opt = AdamW([a, b], lr=0.1)
    opt.zero_grad()
    opt.step()                                   # nothing has a gradient: no-op (pretrain.py:261-263 dummy step)
    assert len(opt.state[a]) == 0
    a.grad = torch.ones_like(a)
    ptr_before = a.grad.data_ptr()
    opt.step()
    assert opt.state[a]['step'] == 1 and len(opt.state[b]) == 0
    assert float(a.detach().max()) < 1.0 and float(b.detach().min()) == 1.0
    opt.zero_grad()
    assert a.grad is not None and a.grad.data_ptr() == ptr_before and float(a.grad.abs().max()) == 0.0


# --------------------------------------------------------------------------------------------------------------
# arena + data-parallel helpers at world size 1 on the GPU
# --------------------------------------------------------------------------------------------------------------
def test_arena_training_step_matches_unflattened(golden):
    from uniter_amd.model.pretrain import UniterForPretraining
    from uniter_amd.optim import AdamW, build_optimizer, clip_grad_norm_
    from uniter_amd.utils import distributed as D
    from uniter_amd.utils.arena import flatten_model
    from uniter_amd.utils.misc import Struct
    batch = _to_dev(golden.batch('mlm'))
    opts = Struct(dict(optim='adamw', learning_rate=1e-3, betas=(0.9, 0.98), weight_decay=0.01))
    results = []
    for use_arena in (False, True):
        model = _prep(UniterForPretraining.from_pretrained(TINY_CONFIG, golden.pretrain_sd(), img_dim=IMG_DIM, img_label_dim=LABEL_DIM))
        arena = flatten_model(model) if use_arena else None
        opt = build_optimizer(model, opts)
        assert isinstance(opt, AdamW)
        reducer = D.GradientReducer(arena, model.uniter.encoder) if use_arena else None
        for step in range(2):
            if reducer:
                reducer.begin()
            model(batch, task='mlm', compute_loss=True).mean().backward()
            if reducer:
                scale = reducer.finish()
            else:
                D.all_reduce_and_rescale_tensors([p.grad.data for p in model.parameters() if p.grad is not None], 1.0)
                scale = 1.0
            clip_grad_norm_(opt, 1.0, grad_scale=scale)
            opt.step()
            opt.zero_grad()
        if use_arena:
            assert arena.check()
        results.append({n: p.detach().float().cpu().clone() for n, p in model.named_parameters()})
    for n in results[0]:
        assert rel_l2(results[1][n], results[0][n]) < 1e-3, n


def test_overlapped_adamw_step_is_bit_identical_and_skips_untouched_params(golden):
    """AdamW.enable_overlap (segmented update on the optimizer stream, forward of the next step overlapping it, zero_grad
    folded into the kernel) must produce exactly the parameters of the synchronous step; parameters the loss never
    reaches keep grad None and are neither decayed nor stepped (optim/adamw.py:52-53)."""
    from uniter_amd.model.nlvr2 import UniterForNlvr2PairedAttn
    from uniter_amd.optim import build_optimizer, clip_grad_norm_, overlap_boundaries
    from uniter_amd.utils.arena import flatten_model
    from uniter_amd.utils.misc import Struct
    batch = _to_dev(golden.batch('nlvr2'))
    opts = Struct(dict(optim='adamw', learning_rate=1e-3, betas=(0.9, 0.98), weight_decay=0.01))
    results = []
    for mode in ('plain', 'fused_zero', 'overlap'):
        overlap = mode == 'overlap'
        w = golden.weights('pre')
        nl = golden.weights('nlvr2')
        table3 = nl.pop('uniter.embeddings.token_type_embeddings.weight')
        model = UniterForNlvr2PairedAttn.from_pretrained(TINY_CONFIG, {**w, **nl}, img_dim=IMG_DIM)
        model.init_type_embedding()
        model.uniter.embeddings.token_type_embeddings.weight.data.copy_(table3)
        _prep(model)
        arena = flatten_model(model)
        opt = build_optimizer(model, opts)
        pool0 = model.uniter.pooler.dense.weight.detach().clone()
        if overlap:
            b = overlap_boundaries(model)
            assert len(b) == len(model.uniter.encoder.layer) + 1 and b == sorted(b)
            opt.enable_overlap(b)
        opt.fuse_zero_grad = mode == 'fused_zero'       # zero_grad folded into the synchronous update kernel
        for step in range(3):
            model(batch, compute_loss=True).mean().backward()
            clip_grad_norm_(opt, 1.0)
            opt.step()
            opt.zero_grad()
        opt.synchronize()
        torch.cuda.synchronize()
        assert arena.check()
        # the pooler is unused by the paired-attention head: no gradient, no update, no optimizer state
        assert model.uniter.pooler.dense.weight.grad is None
        assert torch.equal(model.uniter.pooler.dense.weight, pool0)
        assert len(opt.state[model.uniter.pooler.dense.weight]) == 0
        from uniter_amd import _lib as L
        L.lazy_resolve()     # (lazy_zero: the encoder's gradients are left for the next backward to replace; a reader gets zeros this way)
        assert float(arena.grad.float().abs().max()) == 0.0            # zero_grad happened (fused or memset)
        results.append({n: p.detach().clone() for n, p in model.named_parameters()})
    for n in results[0]:
        assert torch.equal(results[0][n], results[1][n]), n
        assert torch.equal(results[0][n], results[2][n]), n


def test_lazy_zero_grad_is_bit_identical_over_steps(tmp_path):
    """AdamW.lazy_zero (round 6): the fused step leaves the encoder's parameter gradients un-zeroed and the next backward REPLACES
    them through the deferred launch (base width: the flow uniter_encoder_set_grad_overwrite is for).  Four optimizer steps, the
    second with two accumulated micro-batches and one without any backward in between (the kept gradients must read as zeros):
    parameters and the optimizer's moments bit for bit those of the eager zero_grad."""
    import json
    from uniter_amd import _lib as L, ops
    from uniter_amd.model.nlvr2 import UniterForNlvr2PairedAttn
    from uniter_amd.optim import build_optimizer, clip_grad_norm_
    from uniter_amd.utils.arena import flatten_model
    from uniter_amd.utils.misc import Struct, set_dropout
    from uniter_amd.utils.synthetic import make_batch
    cfg = dict(BASE_CFG, num_hidden_layers=2)
    path = tmp_path / "lz.json"
    path.write_text(json.dumps(cfg))
    batches = [_to_dev(make_batch('nlvr2', 8, seed=20 + k)) for k in range(2)]
    opts = Struct(dict(optim='adamw', learning_rate=1e-3, betas=(0.9, 0.98), weight_decay=0.01))
    out = []
    for lazy in (False, True):
        torch.manual_seed(11)
        model = UniterForNlvr2PairedAttn.from_pretrained(str(path), {}, img_dim=2048)
        model.init_type_embedding()
        _prep(model)
        set_dropout(model, 0.1)
        model.train()
        flatten_model(model)
        opt = build_optimizer(model, opts)
        opt.fuse_zero_grad = True
        opt.lazy_zero = lazy
        ops.manual_seed(99)
        seen_overwrite = []
        for step in range(4):
            if step != 2:                                          # step 2: an optimizer step with no backward before it
                for b in (batches if step == 1 else batches[:1]):  # step 1: two accumulated micro-batches
                    seen_overwrite.append(L.lazy_undefined)
                    model(b, compute_loss=True).mean().backward()
            clip_grad_norm_(opt, 1.0)
            opt.step()
            opt.zero_grad()
        torch.cuda.synchronize()
        assert seen_overwrite == ([False, True, False, True] if lazy else [False] * 4), seen_overwrite
        assert L.lazy_undefined == lazy
        L.lazy_resolve()
        state = {n: p.detach().clone() for n, p in model.named_parameters()}
        for n, p in model.named_parameters():
            if 'exp_avg' in opt.state[p]:
                state[n + '/m'] = opt.state[p]['exp_avg'].clone()
                state[n + '/v'] = opt.state[p]['exp_avg_sq'].clone()
        out.append(state)
    assert set(out[0]) == set(out[1]) and len(out[0]) > 100
    for n in out[0]:
        assert torch.equal(out[0][n], out[1][n]), n


def test_image_embeddings_dense_type_embeddings_form(golden):
    """UniterImageEmbeddings.forward(img_feat, img_pos_feat, type_embeddings [B, Li, H]) — the reference's own call convention
    (model/model.py:261-272) — against the (table, ids) form UniterModel uses: same output bit for bit; the gradient w.r.t. the
    dense tensor, summed per type id, is the table gradient of the other form; both against oracle.image_embeddings."""
    from uniter_amd.model.model import UniterImageEmbeddings, UniterConfig
    torch.manual_seed(41)
    conf = UniterConfig.from_json_file(TINY_CONFIG)
    emb = UniterImageEmbeddings(conf, IMG_DIM)
    for p in emb.parameters():
        torch.nn.init.normal_(p, std=0.1)
    for m in (emb.img_layer_norm, emb.pos_layer_norm, emb.LayerNorm):
        m.weight.data.add_(1.0)
    _prep(emb)
    H = conf.hidden_size
    B, Li = 3, 20
    dev = _dev()
    feat = torch.randn(B, Li, IMG_DIM, device=dev).bfloat16()
    pos = torch.rand(B, Li, 7, device=dev).bfloat16()
    table = (torch.randn(2, H, device=dev) * 0.3).bfloat16().requires_grad_(True)
    ids = torch.ones(B, Li, dtype=torch.int64, device=dev)
    ids[:, ::3] = 0
    go = torch.randn(B, Li, H, device=dev).bfloat16()
    out_t = emb(feat, pos, (table, ids))
    out_t.backward(go)
    g_table = table.grad.detach().float().cpu().clone()
    g_lin = emb.img_linear.weight.grad.detach().clone()
    for p in emb.parameters():
        p.grad = None
    dense = table.detach()[ids].clone().requires_grad_(True)
    out_d = emb(feat, pos, dense)
    out_d.backward(go)
    assert torch.equal(out_t, out_d)
    assert torch.equal(g_lin, emb.img_linear.weight.grad)
    gd = dense.grad.float().cpu()
    per_type = torch.stack([gd[(ids == k).cpu()].sum(0) for k in range(2)])
    assert rel_l2(per_type, g_table) <= 2e-2, rel_l2(per_type, g_table)
    sd = {'e.' + n: p.detach().float().cpu() for n, p in emb.named_parameters()}
    ref = O.image_embeddings(sd, 'e.', feat.float().cpu(), pos.float().cpu(), dense.detach().float().cpu())
    _check_hidden(out_d.detach(), ref, "image embeddings, dense type form")


def test_bert_sub_modules_called_on_their_own(tmp_path):
    """BertSelfAttention / BertSelfOutput / BertAttention / BertIntermediate / BertOutput.forward (model/layer.py:75-156) as separate
    calls — what third-party code written against the reference's modules does — against (1) the oracle's pieces in fp32 at the
    survey's tolerances and (2) BertLayer.forward (the fused stack) on the same parameters: outputs bit for bit (same kernels, one
    at a time), input and parameter gradients to 1e-2 relative L2 (the fused backward sums weight gradients in another tile order)."""
    import json
    from uniter_amd.model.model import UniterConfig
    from uniter_amd.model.layer import BertLayer
    cfg = dict(BASE_CFG)
    conf = UniterConfig.from_dict(cfg) if hasattr(UniterConfig, 'from_dict') else None
    if conf is None:
        path = tmp_path / "sub.json"
        path.write_text(json.dumps(cfg))
        conf = UniterConfig.from_json_file(str(path))
    torch.manual_seed(31)
    layer = BertLayer(conf)
    for p in layer.parameters():
        torch.nn.init.normal_(p, std=0.05)
    for m in (layer.attention.output.LayerNorm, layer.output.LayerNorm):
        m.weight.data.add_(1.0)
    _prep(layer)
    B, L, H = 4, 96, cfg['hidden_size']
    x = (torch.randn(B, L, H) * 0.7).to(_dev()).bfloat16()
    valid = torch.ones(B, L)
    valid[1, 70:] = 0
    valid[3, 50:] = 0
    ext = ((1.0 - valid) * -10000.0).view(B, 1, 1, L).to(_dev())
    go = (torch.randn(B, L, H) * 0.5).to(_dev()).bfloat16()

    def run(fused):
        for p in layer.parameters():
            p.grad = None
        xin = x.clone().requires_grad_(True)
        if fused:
            out = layer(xin, ext)
        else:
            att = layer.attention(xin, ext)                     # = output(self(x, mask), x)
            out = layer.output(layer.intermediate(att), att)
        out.backward(go)
        torch.cuda.synchronize()
        return out.detach().clone(), xin.grad.detach().clone(), {n: p.grad.detach().float().cpu() for n, p in layer.named_parameters()}

    out_s, dx_s, g_s = run(False)
    out_f, dx_f, g_f = run(True)
    assert torch.equal(out_s, out_f)
    assert rel_l2(dx_s.float().cpu(), dx_f.float().cpu()) <= 1e-2
    assert set(g_s) == set(g_f) and len(g_s) == 16
    kb = 'attention.self.key.bias'       # mathematically zero (a shift of every key's score leaves the softmax alone): noise only
    for g in (g_s, g_f):
        assert float(g[kb].abs().max()) <= 0.1 * float(g['attention.self.query.bias'].abs().max())
    for n in g_s:
        if n != kb:
            assert rel_l2(g_s[n], g_f[n]) <= 1e-2, (n, rel_l2(g_s[n], g_f[n]))
    # the oracle, piece by piece (fp32 on the CPU; model/layer.py:75-156)
    sd = {'l.' + n: p.detach().float().cpu().clone().requires_grad_(True) for n, p in layer.named_parameters()}
    xr = x.float().cpu().clone().requires_grad_(True)
    ref = O.bert_layer(xr, ext.float().cpu(), sd, 'l.', cfg['num_attention_heads'])
    ref.backward(go.float().cpu())
    _check_hidden(out_s, ref.detach(), "sub-module composition")
    assert rel_l2(dx_s.float().cpu(), xr.grad) <= GRAD_L2
    for n in g_s:
        if n != kb:
            assert cosine(g_s[n], sd['l.' + n].grad) >= GRAD_COS and rel_l2(g_s[n], sd['l.' + n].grad) <= GRAD_L2, n
    # the pieces' own interfaces: context of the attention alone, the intermediate alone
    ctx_ref = O.self_attention(x.float().cpu(), ext.float().cpu(), {k: v.detach() for k, v in sd.items()}, 'l.attention.self.', cfg['num_attention_heads'])
    with torch.no_grad():
        ctx_got = layer.attention.self(x, ext)
        inter = layer.intermediate(x)
    _check_hidden(ctx_got, ctx_ref, "BertSelfAttention alone")
    inter_ref = O.gelu(O.linear(x.float().cpu(), sd['l.intermediate.dense.weight'].detach(), sd['l.intermediate.dense.bias'].detach()))
    d = (inter.float().cpu() - inter_ref).abs()
    assert float((d - (2e-2 + 1e-2 * inter_ref.abs())).max()) <= 0


def test_folded_gradient_norm_equals_the_full_reduction(tmp_path, monkeypatch):
    """AdamW.fold_norm (round 6): the encoder weights' share of sum g^2 comes out of the deferred weight-gradient launch
    (uniter_encoder_last_grad_sq -> uniter_adamw_grad_norm_ex).  Same gradients, folded vs full reduction: equal to fp32 summation
    order; after a second accumulated micro-batch still equal; after an in-place edit of one gradient the fold must notice
    (version counters) and give the full reduction's number; with the lazy zero_grad in between as well."""
    import json
    from uniter_amd import _lib as L, ops
    from uniter_amd.model.nlvr2 import UniterForNlvr2PairedAttn
    from uniter_amd.optim import build_optimizer, clip_grad_norm_
    from uniter_amd.utils.arena import flatten_model
    from uniter_amd.utils.misc import Struct
    from uniter_amd.utils.synthetic import make_batch
    monkeypatch.setattr(ops, "_FOLD_NORM", True)              # (opt-in: the backward is asked for the per-tile sums)
    cfg = dict(BASE_CFG, num_hidden_layers=2)
    path = tmp_path / "fn.json"
    path.write_text(json.dumps(cfg))
    torch.manual_seed(12)
    model = UniterForNlvr2PairedAttn.from_pretrained(str(path), {}, img_dim=2048)
    model.init_type_embedding()
    _prep(model)
    flatten_model(model)
    opt = build_optimizer(model, Struct(dict(optim='adamw', learning_rate=1e-3, betas=(0.9, 0.98), weight_decay=0.01)))
    opt.fuse_zero_grad = True
    batches = [_to_dev(make_batch('nlvr2', 8, seed=30 + k)) for k in range(2)]

    def both():
        opt.fold_norm = True
        folded_used = L.sq_state is not None
        a = float(clip_grad_norm_(opt, 1.0))
        opt.fold_norm = False
        b = float(clip_grad_norm_(opt, 1.0))
        opt.fold_norm = True
        return a, b, folded_used

    for step in range(3):
        model(batches[0], compute_loss=True).mean().backward()
        a, b, used = both()
        assert used and b > 0 and abs(a - b) <= 2e-5 * b, (step, a, b)
        model(batches[1], compute_loss=True).mean().backward()          # accumulation: the sums are those of the running total
        a, b, used = both()
        assert used and abs(a - b) <= 2e-5 * b, (step, a, b)
        if step == 1:
            w = model.uniter.encoder.layer[0].output.dense.weight
            w.grad.mul_(3.0)                                            # someone edits a gradient in place
            a, b, _ = both()
            assert a == b, (a, b)                                       # the fold stood down: the very same reduction
        opt.step()
        opt.zero_grad()
    torch.cuda.synchronize()


def test_adamw_state_dict_roundtrip_keeps_fp32_state(golden):
    """save -> load -> step for bf16 parameters: torch would cast the fp32 moments / master weights to bf16 on load."""
    from uniter_amd.model.pretrain import UniterForPretraining
    from uniter_amd.optim import build_optimizer, clip_grad_norm_
    from uniter_amd.utils.arena import flatten_model
    from uniter_amd.utils.misc import Struct
    batch = _to_dev(golden.batch('itm'))
    opts = Struct(dict(optim='adamw', learning_rate=1e-3, betas=(0.9, 0.98), weight_decay=0.01))

    def fresh():
        m = _prep(UniterForPretraining.from_pretrained(TINY_CONFIG, golden.pretrain_sd(), img_dim=IMG_DIM, img_label_dim=LABEL_DIM))
        flatten_model(m)
        return m, build_optimizer(m, opts)

    def one_step(m, o):
        m(batch, task='itm', compute_loss=True)[0].mean().backward()
        clip_grad_norm_(o, 1.0)
        o.step()
        o.zero_grad()

    m1, o1 = fresh()
    one_step(m1, o1)
    ckpt_model = {k: v.detach().clone() for k, v in m1.state_dict().items()}
    import copy
    ckpt_opt = copy.deepcopy(o1.state_dict())      # state_dict() returns the live tensors; a saver serialises them at once
    one_step(m1, o1)
    m2, o2 = fresh()
    m2.load_state_dict(ckpt_model)
    one_step(m2, o2)                                     # builds a plan that load_state_dict must invalidate
    m2.load_state_dict(ckpt_model)
    o2.load_state_dict(ckpt_opt)
    for p in m2.parameters():
        st = o2.state.get(p, {})
        for k in ('exp_avg', 'exp_avg_sq', 'master'):
            if k in st:
                assert st[k].dtype == torch.float32 and st[k].is_contiguous()
    one_step(m2, o2)
    torch.cuda.synchronize()
    for (n, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.equal(a, b), n


# --------------------------------------------------------------------------------------------------------------
# (5) the other configs of SURVEY.md §8d as parity cases: uniter-large shapes, L = 128 + 50, VQA with 4 parameter groups
# --------------------------------------------------------------------------------------------------------------
LARGE_CFG = dict(vocab_size=28996, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                 hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
                 max_position_embeddings=512, type_vocab_size=2, initializer_range=0.02)


def test_large_config_vqa_l178_vs_oracle(tmp_path):
    """config/uniter-large.json widths (H=1024, 16 heads, I=4096) with 2 layers, ragged text up to 128 tokens + up to 50
    regions (the large-178 shape, config/pretrain-alldata-large-16gpu.json), VQA head; then one clipped AdamW step with
    the 4-group lr_mul optimizer of train_vqa.py:51-86 against the oracle's AdamW on the oracle's gradients."""
    import json
    from uniter_amd.model.vqa import UniterForVisualQuestionAnswering
    from uniter_amd.optim import build_vqa_optimizer, clip_grad_norm_
    from uniter_amd.utils.misc import Struct
    from uniter_amd.utils.synthetic import make_batch
    cfg = dict(LARGE_CFG, num_hidden_layers=2)
    path = tmp_path / "large.json"
    path.write_text(json.dumps(cfg))
    torch.manual_seed(11)
    model = UniterForVisualQuestionAnswering.from_pretrained(str(path), {}, img_dim=2048, num_answer=N_ANS)
    with torch.no_grad():
        g = torch.Generator().manual_seed(12)
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.02)
            p.copy_(p.to(torch.bfloat16).float())
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    batch = make_batch('vqa', 3, max_txt_len=128, num_bb=50, seed=9, ragged=True, num_answer=N_ANS,
                       min_txt_len=128, min_bb=20)              # full-length text, 20..50 regions: L = 178 with padding
    assert batch['attn_masks'].shape[1] == 178
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref_loss, ref_seq = O.vqa_loss(leaf, cfg, batch)
    (ref_loss.mean() * N_ANS).backward()                                  # train_vqa.py:188

    _prep(model)
    d = _to_dev(batch)
    seq = model.uniter(d['input_ids'], d['position_ids'], d['img_feat'], d['img_pos_feat'], d['attn_masks'],
                       d['gather_index'], output_all_encoded_layers=False)
    valid = batch['attn_masks'].bool()
    _, yseq, ygrads = _yardstick(O.vqa_loss, sd, cfg, batch)
    _check_hidden(seq.detach()[valid.to(seq.device)], ref_seq.detach()[valid], "large-178 hidden", yseq[valid])
    loss = model(d, compute_loss=True)
    _check_loss(loss, ref_loss.detach(), atol=3e-2)
    for p in model.parameters():
        p.grad = None
    (model(d, compute_loss=True).float().mean() * N_ANS).backward()
    named = dict(model.named_parameters())
    checked = 0
    for name, p in named.items():
        if leaf[name].grad is None or p.grad is None:
            continue
        yard = ygrads.get(name)
        _check_grad(name, p.grad / N_ANS, leaf[name].grad / N_ANS, None if yard is None else yard)
        checked += 1
    assert checked > 40

    # one optimizer step: 4 groups (vqa_output x lr_mul 10, decay / no-decay), clip 2.0 — from the ORACLE's gradients
    opts = Struct(dict(learning_rate=8e-5, lr_mul=10.0, betas=(0.9, 0.98), weight_decay=0.01, optim='adamw'))
    optimizer = build_vqa_optimizer(model, opts)
    assert len(optimizer.param_groups) == 4
    for group in optimizer.param_groups[:2]:                             # train_vqa.py:208-214
        group['lr'] = opts.learning_rate * opts.lr_mul
    with torch.no_grad():
        for name, p in named.items():
            p.grad = leaf[name].grad.to(p.device, p.dtype) if leaf[name].grad is not None else None
    ref_grads = {n: named[n].grad.float().cpu() for n in named if named[n].grad is not None}      # bf16-rounded, as fed
    _, coef = O.clip_coef(list(ref_grads.values()), 2.0)
    norm = clip_grad_norm_(optimizer, 2.0)
    total = float(torch.sqrt(sum((g.double() ** 2).sum() for g in ref_grads.values())))
    assert abs(float(norm) - total) <= 2e-3 * total
    optimizer.step()
    for name, p in named.items():
        if name not in ref_grads:
            continue
        lr = opts.learning_rate * (opts.lr_mul if 'vqa_output' in name else 1.0)
        wd = 0.0 if O.no_decay(name) else opts.weight_decay
        want, m, v = O.adamw_step(sd[name], ref_grads[name] * coef, torch.zeros_like(sd[name]), torch.zeros_like(sd[name]),
                                  1, lr, betas=opts.betas, eps=1e-6, weight_decay=wd)
        st = optimizer.state[p]
        torch.testing.assert_close(st['exp_avg'].float().cpu(), m, rtol=2e-4, atol=1e-9)
        torch.testing.assert_close(st['exp_avg_sq'].float().cpu(), v, rtol=4e-4, atol=1e-12)
        master = st['master'].float().cpu() if 'master' in st else p.detach().float().cpu()
        torch.testing.assert_close(master, want, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("long_seq", [False, True], ids=["96", "300"])
def test_paired_cross_attention_fused_equals_module_path(tmp_path, long_seq, monkeypatch):
    """The fused NLVR2 cross-attention op (strided GEMMs + the encoder's attention kernel on the packed partner layout) and
    the fused pooling against the plain module path of the same model (torch bf16 ops), ragged pairs, base width; at 60 + 36
    tokens and at up to 200 + 100 (beyond 256 the attention backward is the two-launch form; model/nlvr2.py:65-107 formats reach it)."""
    import json
    from uniter_amd.model.nlvr2 import UniterForNlvr2PairedAttn
    from uniter_amd.utils.synthetic import make_batch
    cfg = dict(BASE_CFG, num_hidden_layers=1)
    path = tmp_path / "pa.json"
    path.write_text(json.dumps(cfg))
    torch.manual_seed(3)
    model = UniterForNlvr2PairedAttn.from_pretrained(str(path), {}, img_dim=2048)
    model.init_type_embedding()
    _prep(model)
    if long_seq:
        batch = _to_dev(make_batch('nlvr2', 4, max_txt_len=200, num_bb=100, seed=4, ragged=True, min_txt_len=170, min_bb=90))
        assert 256 < batch['attn_masks'].shape[1] <= 300
    else:
        batch = _to_dev(make_batch('nlvr2', 8, seed=4, ragged=True))

    monkeypatch.setenv("UNITER_AMD_HEAD_TORCH", "1")          # the module path is a comparison path: it has to be asked for

    def run(fused):
        model._fused_pair_attention = (lambda seq: fused)
        model.attn_pool.force_module_path = not fused
        for p in model.parameters():
            p.grad = None
        loss = model(batch, compute_loss=True)
        loss.float().mean().backward()
        return loss.detach().float().cpu(), {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters()
                                             if p.grad is not None and n.startswith(('attn1', 'attn2', 'fc', 'uniter.encoder'))}

    loss_f, g_f = run(True)
    loss_m, g_m = run(False)
    torch.testing.assert_close(loss_f, loss_m, rtol=2e-2, atol=2e-2)
    assert set(g_f) == set(g_m) and len(g_f) > 20
    for name in g_f:
        if float(g_m[name].abs().max()) < 1e-6:
            continue
        assert cosine(g_f[name], g_m[name]) >= 0.99, (name, cosine(g_f[name], g_m[name]))
        assert rel_l2(g_f[name], g_m[name]) <= 1e-1, (name, rel_l2(g_f[name], g_m[name]))


def test_paired_cross_attention_cat_node_is_bit_identical_to_three_nodes(tmp_path, monkeypatch):
    """ops._PairedCrossAttnCatFn (regroup + both cross attentions + cat([attended, own]) written in place, round 6) against the
    round-4 form it replaces (regroup copy, _PairedCrossAttnFn, torch.cat: model.three_node_cat): same kernels on the same
    values through different row strides, dropout active with the same Philox offsets — loss and every gradient bit for bit."""
    import json
    from uniter_amd import ops
    from uniter_amd.model.nlvr2 import UniterForNlvr2PairedAttn
    from uniter_amd.utils.misc import set_dropout
    from uniter_amd.utils.synthetic import make_batch
    cfg = dict(BASE_CFG, num_hidden_layers=1)
    path = tmp_path / "pa.json"
    path.write_text(json.dumps(cfg))
    torch.manual_seed(5)
    model = UniterForNlvr2PairedAttn.from_pretrained(str(path), {}, img_dim=2048)
    model.init_type_embedding()
    _prep(model)
    set_dropout(model, 0.1)
    model.attn1.dropout = model.attn2.dropout = 0.1
    model.train()
    batch = _to_dev(make_batch('nlvr2', 8, seed=6, ragged=True))

    def run(three_nodes):
        model.three_node_cat = three_nodes
        ops.manual_seed(1234)
        for p in model.parameters():
            p.grad = None
        loss = model(batch, compute_loss=True)
        loss.float().mean().backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    loss_a, g_a = run(False)
    loss_b, g_b = run(True)
    assert torch.equal(loss_a, loss_b)
    assert set(g_a) == set(g_b) and len(g_a) > 30
    for name in g_a:
        assert torch.equal(g_a[name], g_b[name]), name


def test_timing_records_tuned_choice_and_cache(tmp_path, monkeypatch):
    """uniter_hip_timing_{begin,end}, uniter_gemm_set_tuned legality and the UNITER_AMD_TUNE_CACHE round trip."""
    import ctypes
    import json
    from uniter_amd import _lib, ops
    from uniter_amd._lib import C
    dev = _dev()
    M, N, K = 256, 128, 192
    x = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    _lib.timing_begin()
    for _ in range(3):
        C.uniter_gemm_bias_fwd(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), M, N, K, _lib.stream_ptr())
    recs = _lib.timing_end()
    mine = [r for r in recs if r["kind_id"] == 0 and (r["M"], r["N"], r["K"]) == (M, N, K)]
    assert len(mine) == 1 and mine[0]["calls"] == 3 and 0.0 < mine[0]["total_us"] < 1e5
    torch.testing.assert_close(y.float(), x.float() @ w.float().t(), rtol=2e-2, atol=2e-1)
    assert _lib.timing_end() == []                                       # nothing recorded outside begin/end
    # a 96-wide tile cannot serve a K-strided operand (dgrad), tile 3 (64x64) can
    with pytest.raises(_lib.UniterHipError):
        C.uniter_gemm_set_tuned(1, M, N, K, 10, 1)
    C.uniter_gemm_set_tuned(1, M, N, K, 3, 1)
    out = (ctypes.c_int32 * 2)()
    C.uniter_gemm_tuned_choice(1, M, N, K, out)
    assert (out[0], out[1]) == (3, 1)
    with pytest.raises(_lib.UniterHipError):
        C.uniter_gemm_set_tuned(0, M, N, K, 3, 2)                        # split-K is a wgrad-only option
    # cache round trip for an encoder shape
    cache = tmp_path / "tune.json"
    monkeypatch.setenv("UNITER_AMD_TUNE_CACHE", str(cache))
    s = ops._shape(dict(H=128, heads=2, I=256, p_hidden=0.0, p_attn=0.0, ln_eps=1e-12), 2, 32, True)
    ops._autotuned.discard((2, 32, 128, 256))
    ops._maybe_autotune(s, True)
    doc = json.loads(cache.read_text())
    assert doc["n_tiles"] == C.uniter_gemm_tile_count()
    saved = [e for e in doc["gemm"] if e["M"] == 2 * 32]               # (task-head groups tuned earlier in this process are saved too)
    assert len(saved) == 9 and all(e["cfg"] >= 0 for e in saved)      # 4 forward + 4 dgrad + the grouped wgrad launch
    ops._autotuned.discard((2, 32, 128, 256))
    assert ops._load_tune_cache(str(cache), s)


def test_native_kernel_harness():
    """The C++ harness (tests/native/test_kernels.cpp) checks every kernel against a host fp32 reference through the C
    ABI without Python or torch in the process: GEMM epilogues x all tile shapes, attention, LayerNorm, AdamW, probes."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "build", "test_kernels")
    if not os.path.exists(exe):
        pytest.skip("tests/native/build/test_kernels not built (python tests/native/build.py)")
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:]
    assert "== 0 check(s) failed ==" in res.stdout, res.stdout[-2000:]


# --------------------------------------------------------------------------------------------------------------
# (6) SURVEY.md §8 f-1: fused IPOT optimal-transport distance
# --------------------------------------------------------------------------------------------------------------
def _ot_golden(name):
    import os
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ot_golden.npz"))

    def bf(a):
        return torch.from_numpy(a.astype(np.int16)).view(torch.bfloat16)

    return {k.split("/", 1)[1]: (bf(z[k]) if k.endswith("_bf16") else torch.from_numpy(z[k])) for k in z.files
            if k.startswith(name + "/")}


@pytest.mark.parametrize("name", ["small", "base", "long"])
def test_ot_kernel_vs_reference_golden(name):
    """uniter_ot_fwd / uniter_ot_bwd against the REAL reference's cost_matrix_cosine + ipot (tests/golden/ot_golden.npz):
    transport plan, distance, and d dist / d embeddings.  The joint sequence is [text slots ; image slots] (identity
    scatter); the kernel's inputs are the same bf16 values the reference saw, so only fp32 summation order differs."""
    from uniter_amd import ops
    c = _ot_golden(name)
    x, y = c["x_bf16"], c["y_bf16"]
    B, M, D = x.shape
    N = y.shape[1]
    seq = torch.cat([x, y], dim=1).to(_dev()).requires_grad_(True)
    scatter = torch.arange(M + N).unsqueeze(0).repeat(B, 1)
    dist = ops.optimal_transport_dist(seq, scatter.to(_dev()), c["txt_pad"].to(_dev()), c["img_pad"].to(_dev()))
    torch.testing.assert_close(dist.detach().cpu(), c["dist"], rtol=2e-4, atol=1e-6)
    # the saved plan
    plan = dist.grad_fn.saved_tensors[4].cpu()
    torch.testing.assert_close(plan, c["T"], rtol=2e-3, atol=1e-7)
    dist.sum().backward()
    g = seq.grad.float().cpu()
    for got, want, what in ((g[:, :M], c["dx"], "dx"), (g[:, M:], c["dy"], "dy")):
        assert rel_l2(got, want) <= 1e-2, (name, what, rel_l2(got, want))                 # bf16 output rounding
        assert cosine(got, want) >= 0.9999, (name, what)
        assert float(got[want == 0].abs().max()) == 0.0                                  # padded slots get exact zeros


def test_itm_ot_loss_vs_oracle(tmp_path):
    """forward_itm with ot_inputs through the model (compact ragged sequences, real ot_scatter) and the loss mix of
    pretrain.py:270-290 against the oracle: ITM losses, OT distances, gradients."""
    from uniter_amd.utils.synthetic import make_batch
    model, cfg = _base_model(tmp_path, n_layers=2)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    batch = make_batch('itm', 6, seed=21, ragged=True, with_ot=True)
    batch['targets'] = torch.tensor([1, 0, 1, 1, 0, 0])
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != 'cls.predictions.decoder.weight'}
    ref_loss, ref_itm, ref_dist, _ = O.itm_ot_loss(leaf, cfg, batch, ot_lambda=0.1)
    ref_loss.backward()

    _prep(model)
    d = _to_dev(batch)
    itm, (pos, neg) = model(d, task='itm', compute_loss=True)
    _check_loss(itm, ref_itm.detach(), atol=3e-2)
    tgt = batch['targets']
    torch.testing.assert_close(pos.float().cpu(), ref_dist.detach()[tgt == 1], rtol=2e-2, atol=2e-3)
    torch.testing.assert_close(neg.float().cpu(), ref_dist.detach()[tgt == 0], rtol=2e-2, atol=2e-3)
    for p in model.parameters():
        p.grad = None
    itm, (pos, neg) = model(d, task='itm', compute_loss=True)
    loss = itm.float().mean() + 0.1 * (pos.float().sum() - neg.float().sum()) / (pos.numel() + neg.numel())
    loss.backward()
    named = dict(model.named_parameters())
    checked = 0
    for name in ('uniter.encoder.layer.1.output.dense.weight', 'uniter.encoder.layer.1.attention.self.query.weight',
                 'uniter.encoder.layer.0.intermediate.dense.weight', 'uniter.embeddings.word_embeddings.weight',
                 'uniter.img_embeddings.img_linear.weight', 'itm_output.weight'):
        _check_grad(name, named[name].grad, leaf[name].grad)
        checked += 1
    assert checked == 6


# --------------------------------------------------------------------------------------------------------------
# (7) SURVEY.md §8 f-3: padding-free (packed) execution
# --------------------------------------------------------------------------------------------------------------
def test_packed_execution_matches_dense_and_oracle(tmp_path):
    """UniterEncoder.forward_packed (only real tokens go through the layers; attention per example via cu_seqlens)
    against the dense path and the oracle on a ragged batch: hidden states at real positions, zeros at padded ones,
    loss and parameter gradients."""
    from uniter_amd.utils.synthetic import make_batch
    model, cfg = _base_model(tmp_path, n_layers=3)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    batch = make_batch('mlm', 5, seed=13, ragged=True)
    valid = batch['attn_masks'].bool()
    assert not bool(valid.all())                                       # the batch really has padding
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != 'cls.predictions.decoder.weight'}
    leaf['cls.predictions.decoder.weight'] = leaf['uniter.embeddings.word_embeddings.weight']
    ref_loss, ref_seq = O.mlm_loss(leaf, cfg, batch)

    _prep(model)
    d = _to_dev(batch)

    def run(pack):
        model.uniter.pack_padding = pack
        for p in model.parameters():
            p.grad = None
        seq = model.uniter(d['input_ids'], d['position_ids'], d['img_feat'], d['img_pos_feat'], d['attn_masks'],
                           d['gather_index'], output_all_encoded_layers=False)
        loss = model(d, task='mlm', compute_loss=True)
        loss.float().mean().backward()
        grads = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
        return seq.detach().float().cpu(), loss.detach().float().cpu(), grads

    seq_d, loss_d, g_d = run(False)
    seq_p, loss_p, g_p = run(True)
    model.uniter.pack_padding = False
    assert float(seq_p[~valid].abs().max()) == 0.0                     # padded positions are zeros in packed mode
    # same kernels on the same rows: real positions agree to bf16 rounding of a different tile choice at most
    torch.testing.assert_close(seq_p[valid], seq_d[valid], rtol=2e-2, atol=2e-2)
    assert float((seq_p[valid] == seq_d[valid]).float().mean()) > 0.9
    _check_hidden(seq_p[valid], ref_seq.detach()[valid], "packed hidden", _yard_hidden(sd, cfg, batch)[valid])
    torch.testing.assert_close(loss_p, loss_d, rtol=1e-2, atol=1e-2)
    _check_loss(loss_p, ref_loss.detach(), atol=3e-2)
    assert set(g_p) == set(g_d)
    for name in g_d:
        if float(g_d[name].abs().max()) < 1e-6:
            continue
        # both runs are bf16 paths with different tile choices / summation lengths: agreement to bf16 noise
        assert cosine(g_p[name], g_d[name]) >= 0.99, (name, cosine(g_p[name], g_d[name]))
        assert rel_l2(g_p[name], g_d[name]) <= 1e-1, (name, rel_l2(g_p[name], g_d[name]))


def test_packed_attention_kernel_matches_dense_masked():
    """uniter_attention_{fwd,bwd}_packed on concatenated sequences == the dense kernels with a key mask, per example."""
    from uniter_amd import _lib
    from uniter_amd._lib import C
    dev = _dev()
    heads, H = 4, 256
    lens = [37, 96, 5, 64]
    B, L = len(lens), max(lens)
    g = torch.Generator().manual_seed(3)
    qkv_d = torch.randn(B, L, 3 * H, generator=g).to(dev, torch.bfloat16)
    dctx_d = torch.randn(B, L, H, generator=g).to(dev, torch.bfloat16)
    mask = torch.zeros(B, L, dtype=torch.float32, device=dev)
    for b, n in enumerate(lens):
        mask[b, n:] = -10000.0
    st = _lib.stream_ptr()
    ctx_d = torch.zeros(B, L, H, dtype=torch.bfloat16, device=dev)
    lse_d = torch.zeros(B * heads * L, dtype=torch.float32, device=dev)
    dqkv_d = torch.zeros(B, L, 3 * H, dtype=torch.bfloat16, device=dev)
    C.uniter_attention_fwd(qkv_d.data_ptr(), mask.data_ptr(), ctx_d.data_ptr(), lse_d.data_ptr(), B, L, heads, 0.0, 0, 0, st)
    C.uniter_attention_bwd(qkv_d.data_ptr(), mask.data_ptr(), ctx_d.data_ptr(), lse_d.data_ptr(), dctx_d.data_ptr(),
                           dqkv_d.data_ptr(), B, L, heads, 0.0, 0, 0, st)
    rows = torch.cat([torch.arange(n) + b * L for b, n in enumerate(lens)]).to(dev)
    qkv_p = qkv_d.view(B * L, -1).index_select(0, rows).contiguous()
    dctx_p = dctx_d.view(B * L, -1).index_select(0, rows).contiguous()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
    T = int(cu[-1])
    ctx_p = torch.zeros(T, H, dtype=torch.bfloat16, device=dev)
    lse_p = torch.zeros(B * heads * L, dtype=torch.float32, device=dev)
    dqkv_p = torch.zeros(T, 3 * H, dtype=torch.bfloat16, device=dev)
    C.uniter_attention_fwd_packed(qkv_p.data_ptr(), cu.data_ptr(), ctx_p.data_ptr(), lse_p.data_ptr(), B, L, heads, 0.0, 0, 0, st)
    C.uniter_attention_bwd_packed(qkv_p.data_ptr(), cu.data_ptr(), ctx_p.data_ptr(), lse_p.data_ptr(), dctx_p.data_ptr(),
                                  dqkv_p.data_ptr(), B, L, heads, 0.0, 0, 0, st)
    torch.testing.assert_close(ctx_p.float(), ctx_d.view(B * L, -1).index_select(0, rows).float(), rtol=0, atol=0)
    # dense backward also sends (zero-probability) contributions of real queries to nothing else: identical on real rows
    got, want = dqkv_p.float(), dqkv_d.view(B * L, -1).index_select(0, rows).float()
    # dK / dV of real keys receive contributions from PADDED query rows in the dense run (those rows attend to real keys);
    # compare the dQ third exactly and the dK / dV thirds after removing that effect by zeroing dctx of padded rows
    torch.testing.assert_close(got[:, :H], want[:, :H], rtol=0, atol=0)
    dctx_z = dctx_d.clone()
    for b, n in enumerate(lens):
        dctx_z[b, n:] = 0
    dqkv_z = torch.zeros_like(dqkv_d)
    C.uniter_attention_bwd(qkv_d.data_ptr(), mask.data_ptr(), ctx_d.data_ptr(), lse_d.data_ptr(), dctx_z.data_ptr(),
                           dqkv_z.data_ptr(), B, L, heads, 0.0, 0, 0, st)
    torch.testing.assert_close(got, dqkv_z.view(B * L, -1).index_select(0, rows).float(), rtol=0, atol=0)


@pytest.mark.parametrize("L", [96, 400])
def test_attention_pool_kernel_vs_torch_fp32(L):
    """uniter_attn_pool_{fwd,bwd} (model/nlvr2.py:110-125) against oracle.attention_pool in fp32 on the bf16 inputs."""
    from uniter_amd import ops
    dev = _dev()
    B, H = 6, 768
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, L, H, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
    lin = torch.nn.Linear(H, 1).to(dev).bfloat16()
    with torch.no_grad():
        lin.weight.copy_((torch.randn(1, H, generator=g) * 0.05).to(dev, torch.bfloat16))
        lin.bias.fill_(0.1)
    pad = torch.zeros(B, L, dtype=torch.bool)
    for b in range(1, B):
        pad[b, (30 + 10 * b) * L // 96:] = True
    pad = pad.to(dev)
    w_out = torch.randn(B, H, generator=g).to(dev)
    out = ops.attention_pool(x, pad, lin, 0.0, True)
    (out.float() * w_out).sum().backward()
    got = (out.detach().float().cpu(), x.grad.float().cpu(), lin.weight.grad.float().cpu(), lin.bias.grad.float().cpu())

    xr = x.detach().float().requires_grad_(True)
    wr = lin.weight.detach().float().requires_grad_(True)
    br = lin.bias.detach().float().requires_grad_(True)
    ref = O.attention_pool({'attn_pool.fc.0.weight': wr, 'attn_pool.fc.0.bias': br}, xr, pad)     # (the pool oracle.nlvr2_paired_attn_loss uses)
    (ref * w_out).sum().backward()
    torch.testing.assert_close(got[0], ref.detach().cpu(), rtol=1e-2, atol=1e-2)
    assert rel_l2(got[1], xr.grad.cpu()) <= 1e-2 and cosine(got[1], xr.grad.cpu()) >= 0.9999
    assert rel_l2(got[2], wr.grad.cpu()) <= 2e-2, rel_l2(got[2], wr.grad.cpu())
    assert abs(float(got[3]) - float(br.grad)) <= 2e-2 * max(1.0, abs(float(br.grad)))
    # dropout: weights are either 0 or sm / (1 - p), and the mean survives
    out_d = ops.attention_pool(x.detach(), pad, lin, 0.5, True)
    assert torch.isfinite(out_d.float()).all()


@pytest.mark.gpu
def test_mlm_head_fused_vs_torch_fp32():
    """transform + tied 28996-way decoder + cross entropy through the HIP path
    (ops.mlm_head_loss: library GEMMs over a padded bf16 logits buffer + uniter_ce_fwd/_bwd + uniter_gelu_bwd)
    against the module formula (model/layer.py:188-222, model/pretrain.py:129-133) in torch fp32 on the same bf16
    parameters: losses, the gradient into the encoder output, and every parameter gradient incl. the tied embedding."""
    from uniter_amd import ops
    from uniter_amd.model.layer import BertOnlyMLMHead
    from uniter_amd.model.model import UniterConfig
    dev = _dev()
    V, H, n = 28996, 768, 339
    g = torch.Generator().manual_seed(11)
    emb = torch.nn.Parameter((torch.randn(V, H, generator=g) * 0.05).to(dev, torch.bfloat16))
    head = BertOnlyMLMHead(UniterConfig(V, hidden_size=H, num_hidden_layers=1), emb).to(dev).bfloat16()
    pr = head.predictions
    with torch.no_grad():
        pr.transform.dense.weight.copy_((torch.randn(H, H, generator=g) * 0.05).to(dev, torch.bfloat16))
        pr.transform.dense.bias.copy_((torch.randn(H, generator=g) * 0.1).to(dev, torch.bfloat16))
        pr.transform.LayerNorm.weight.copy_((1 + 0.1 * torch.randn(H, generator=g)).to(dev, torch.bfloat16))
        pr.transform.LayerNorm.bias.copy_((0.1 * torch.randn(H, generator=g)).to(dev, torch.bfloat16))
        pr.bias.copy_((0.1 * torch.randn(V, generator=g)).to(dev, torch.bfloat16))
    assert pr.decoder.weight is emb
    x = torch.randn(n, H, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
    labels = torch.randint(0, V, (n,), generator=g)
    labels[:3] = torch.tensor([V - 1, V - 2, V - 4])          # classes in the sliver beyond the last full GEMM tile
    labels[5] = -1                                             # an ignored row
    labels = labels.to(dev)
    wrow = torch.rand(n, generator=g).to(dev)

    loss = ops.mlm_head_loss(x, labels, pr)
    assert loss.shape == (n,) and loss.dtype == torch.float32
    (loss * wrow).sum().backward()
    params = [pr.transform.dense.weight, pr.transform.dense.bias, pr.transform.LayerNorm.weight, pr.transform.LayerNorm.bias,
              emb, pr.bias]
    got = [loss.detach().cpu(), x.grad.float().cpu()] + [p.grad.float().cpu() for p in params]

    # the oracle's head (oracle.mlm_head_loss, the function oracle.mlm_loss is built on) in fp32 on the same bf16-valued parameters
    xr = x.detach().float().requires_grad_(True)
    pf = [p.detach().float().requires_grad_(True) for p in params]
    t = 'cls.predictions.transform.'
    sd = {t + 'dense.weight': pf[0], t + 'dense.bias': pf[1], t + 'LayerNorm.weight': pf[2], t + 'LayerNorm.bias': pf[3],
          'uniter.embeddings.word_embeddings.weight': pf[4], 'cls.predictions.bias': pf[5]}
    ref = O.mlm_head_loss(sd, {'hidden_act': 'gelu'}, xr, labels, ignore_index=-1)
    (ref * wrow).sum().backward()
    assert float(got[0][5]) == 0.0 and float(ref.detach()[5]) == 0.0
    torch.testing.assert_close(got[0], ref.detach().cpu(), rtol=2e-2, atol=5e-2)
    assert rel_l2(got[1], xr.grad.cpu()) <= 3e-2 and cosine(got[1], xr.grad.cpu()) >= 0.999, rel_l2(got[1], xr.grad.cpu())
    names = ["dense.weight", "dense.bias", "LayerNorm.weight", "LayerNorm.bias", "word_embeddings (tied decoder)", "decoder bias"]
    for name, a, b in zip(names, got[2:], [p.grad.cpu() for p in pf]):
        assert rel_l2(a, b) <= 4e-2 and cosine(a, b) >= 0.999, (name, rel_l2(a, b), cosine(a, b))
    # the last rows of the embedding gradient (classes beyond the last full 64-wide tile) are covered
    assert float(got[6][V - 4:].abs().sum()) > 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["kl", "ce"])
def test_region_classification_head_fused_vs_torch_fp32(kind):
    """SURVEY.md section 8 f-2, MRC / MRC-KL (model/pretrain.py:36-47,206-229): dense+GELU+LN+Linear(1601) + loss through
    ops.head_kl_div / ops.head_cross_entropy against the module formula in torch fp32 (1601 = 25 GEMM tiles + 1 class)."""
    from uniter_amd import ops
    from uniter_amd.model.pretrain import RegionClassification
    dev = _dev()
    V, H, n = 1601, 768, 173
    g = torch.Generator().manual_seed(13)
    head = RegionClassification(H, V).to(dev).bfloat16()
    with torch.no_grad():
        for p in head.parameters():
            p.copy_((torch.randn(p.shape, generator=g) * (0.05 if p.dim() == 2 else 0.1)).to(dev, torch.bfloat16))
        head.net[2].weight.add_(1.0)
    x = torch.randn(n, H, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
    soft = torch.softmax(torch.randn(n, V, generator=g) * 2.0, dim=-1).to(dev)
    wel = torch.rand(n, V, generator=g).to(dev)
    net = head.net
    if kind == "kl":
        loss = ops.head_kl_div(x, soft, net[0], net[2], net[3].weight, net[3].bias)
        assert loss.shape == (n, V)
        (loss * wel).sum().backward()
    else:
        hard = torch.max(soft[:, 1:], dim=-1)[1] + 1
        hard[0] = V - 1                                          # the class beyond the last full tile
        loss = ops.head_cross_entropy(x, hard, net[0], net[2], net[3].weight, net[3].bias)
        assert loss.shape == (n,)
        (loss * wel[:, 0]).sum().backward()
    params = list(head.parameters())
    got = [loss.detach().cpu(), x.grad.float().cpu()] + [p.grad.float().cpu() for p in params]

    # the oracle's head (oracle.region_classification_loss, the function oracle.mrc_loss is built on) in fp32 on the same parameters
    xr = x.detach().float().requires_grad_(True)
    leaf = {'region_classifier.' + k: v.detach().float().requires_grad_(True) for k, v in head.named_parameters()}
    if kind == "kl":
        ref = O.region_classification_loss(leaf, xr, soft, kl=True)
        (ref * wel).sum().backward()
    else:
        onehot = torch.zeros_like(soft)
        onehot[torch.arange(n, device=dev), hard] = 1.0              # (the oracle derives the hard label from the soft one: argmax over classes 1..)
        ref = O.region_classification_loss(leaf, xr, onehot, kl=False)
        (ref * wel[:, 0]).sum().backward()
    torch.testing.assert_close(got[0], ref.detach().cpu(), rtol=3e-2, atol=(2e-3 if kind == "kl" else 5e-2))
    assert rel_l2(got[1], xr.grad.cpu()) <= 3e-2 and cosine(got[1], xr.grad.cpu()) >= 0.999, rel_l2(got[1], xr.grad.cpu())
    for (name, _), a in zip(head.named_parameters(), got[2:]):
        b = leaf['region_classifier.' + name].grad.cpu()
        assert rel_l2(a, b) <= 4e-2 and cosine(a, b) >= 0.999, (name, rel_l2(a, b), cosine(a, b))


def _check_all_grads(named, leaf, ygrads, scale=1.0, min_checked=1):
    """_check_grad over every parameter, reporting ALL violations at once (name, ours, torch-bf16 yardstick)."""
    bad, checked = [], 0
    for name, p in named.items():
        ref_g = leaf[name].grad if name in leaf else None
        if ref_g is None or p.grad is None:
            continue
        try:
            _check_grad(name, p.grad * scale, ref_g * scale, ygrads.get(name))
        except AssertionError as e:
            bad.append(str(e.args[0])[:160] if e.args else name)
        checked += 1
    assert not bad, "%d of %d gradients out of tolerance:\n  %s" % (len(bad), checked, "\n  ".join(bad))
    assert checked >= min_checked, checked
    return checked


# --------------------------------------------------------------------------------------------------------------
# (6) the headline workload itself and the other pre-training tasks at UNITER-base size, UNITER-large at full depth
# --------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c2_oracle(tmp_path_factory):
    """The oracle's forward + backward of the benchmarked step, run once for the two tests that need it (~40 s of CPU)."""
    from uniter_amd.train import WORKLOADS, build_model
    from uniter_amd.utils.synthetic import make_batch
    w = WORKLOADS['c2']
    cfg = w['cfg']
    model = build_model('nlvr2', cfg, torch.device('cpu'), 77, str(tmp_path_factory.mktemp("c2") / "base.json")).float()
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.to(torch.bfloat16).float())           # (build_model already rounded the weights to bf16)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    batch = make_batch('nlvr2', w['batch'], w['max_txt_len'], w['num_bb'], seed=1000)
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    taps = {}
    ref_loss, _ = O.nlvr2_paired_attn_loss(leaf, cfg, batch, taps)
    taps['pooled'].retain_grad()
    ref_loss.mean().backward()
    # conditioning of AttentionPool's Linear(H, 1) gradient: how far ONE bf16 rounding of the pooling input moves it (fp32 formula)
    g_out = taps['pooled'].grad.detach()

    def pool_formula(xs):
        W = sd['attn_pool.fc.0.weight'].clone().requires_grad_(True)
        b = sd['attn_pool.fc.0.bias'].clone().requires_grad_(True)
        outs = []
        for x, m in zip(xs, taps['pool_pad']):
            score = torch.relu(torch.nn.functional.linear(x, W, b)).squeeze(-1) + m.float() * -1e4
            outs.append(torch.softmax(score, dim=1).unsqueeze(1).matmul(x).squeeze(1))
        (torch.cat(outs, -1) * g_out).sum().backward()
        return W.grad, b.grad

    x32 = [x.detach() for x in taps['pool_in']]
    xbf = [x.to(torch.bfloat16).float() for x in x32]
    in_err = max(rel_l2(a, b) for a, b in zip(xbf, x32))
    fw, fb = pool_formula(xbf)
    amp = {'attn_pool.fc.0.weight': rel_l2(fw, leaf['attn_pool.fc.0.weight'].grad) / in_err,
           'attn_pool.fc.0.bias': rel_l2(fb, leaf['attn_pool.fc.0.bias'].grad) / in_err}
    return dict(w=w, cfg=cfg, model=model, sd=sd, batch=batch, leaf=leaf, taps=taps, ref_loss=ref_loss.detach(),
                pool_formula=pool_formula, pool_in_bf16=xbf, pool_in_err=in_err, pool_formula_grads=(fw, fb), pool_amplification=amp)


def test_headline_nlvr2_base_step_vs_oracle(c2_oracle):
    """The benchmarked step (bench.py / BASELINE.json configs[1]): UNITER-base NLVR2 paired-attention, 12 layers, B=32
    sequences (16 pairs), L=60+36, dropout 0 — per-pair loss, every gradient and the parameters after one clipped AdamW
    step against the oracle (model/nlvr2.py:163-204, train_nlvr2.py:153-195)."""
    from uniter_amd.optim import build_optimizer, clip_grad_norm_
    from uniter_amd.utils.arena import flatten_model
    from uniter_amd.utils.misc import Struct
    w, cfg, sd, batch, leaf, ref_loss = (c2_oracle[k] for k in ('w', 'cfg', 'sd', 'batch', 'leaf', 'ref_loss'))
    model = copy.deepcopy(c2_oracle['model'])

    _prep(model)
    arena = flatten_model(model)
    d = _to_dev(batch)
    d['img_feat'] = d['img_feat'].to(torch.bfloat16)
    d['img_pos_feat'] = d['img_pos_feat'].to(torch.bfloat16)
    seen = {}
    hook = model.attn_pool.register_forward_pre_hook(lambda m, args: seen.__setitem__('pool_in', args[0].detach().float().cpu()))
    loss = model(d, compute_loss=True)
    hook.remove()
    _check_loss(loss, ref_loss, atol=3e-2)
    loss.mean().backward()
    _, _, ygrads = _yardstick(O.nlvr2_paired_attn_loss, sd, cfg, batch)
    named = dict(model.named_parameters())
    # the pooling input: our error against the oracle's, beside the torch-bf16 yardstick's (what the head inherits from the encoder)
    pool_in_err = rel_l2(seen['pool_in'], torch.cat([x.detach() for x in c2_oracle['taps']['pool_in']], dim=0))
    print("headline parity: pooling input rel-L2 %.2e" % pool_in_err)
    assert pool_in_err <= 2e-2, pool_in_err
    ours = {n: p.grad for n, p in named.items() if p.grad is not None and leaf[n].grad is not None}
    refs = {n: leaf[n].grad for n in ours}
    joint, replaced = _joint_one_element(ours, refs, ygrads)
    assert set(joint) == {'attn_pool.fc.0.{weight,bias}'}, sorted(joint)
    checked, fallback = 0, 0
    for name, g in list(ours.items()) + [(k, v[0]) for k, v in joint.items()]:
        if name in replaced:
            continue
        ref_g, yard = (joint[name][1], joint[name][2]) if name in joint else (refs[name], ygrads.get(name))
        strict_ok = cosine(g.float().cpu(), ref_g) >= GRAD_COS and rel_l2(g.float().cpu(), ref_g) <= (
            GRAD_L2 if name.startswith('uniter.') else GRAD_L2_HEAD)
        fallback += 0 if strict_ok or float(ref_g.abs().max()) < 1e-6 else 1
        _check_grad(name, g, ref_g, yard)
        checked += 1
    assert checked >= 200
    print("headline parity: %d gradients checked, %d needed the torch-bf16 yardstick instead of the absolute bounds" % (checked, fallback))
    assert model.uniter.pooler.dense.weight.grad is None       # unused by this head: stays None like in the reference

    # one clipped AdamW step (lr at warm-up step 1 of the schedule would be ~4e-8: use the base lr so that the update is
    # visible in bf16) from the oracle's own gradients, as in test_large_config_vqa_l178_vs_oracle
    opts = Struct(dict(optim='adamw', learning_rate=w['learning_rate'], betas=w['betas'], weight_decay=w['weight_decay']))
    opt = build_optimizer(model, opts)
    with torch.no_grad():
        for name, p in named.items():
            if p.grad is not None:
                p.grad.copy_(leaf[name].grad.to(p.grad))
    ref_grads = {n: named[n].grad.float().cpu() for n in named if named[n].grad is not None}   # bf16-rounded, as the kernel sees them
    clip_grad_norm_(opt, w['grad_norm'])
    opt.step()
    torch.cuda.synchronize()
    _, coef = O.clip_coef(list(ref_grads.values()), w['grad_norm'])
    worst = 0.0
    for name, g in ref_grads.items():
        p_ref = sd[name].clone()
        O.adamw_step_(p_ref, g * coef, torch.zeros_like(g), torch.zeros_like(g), 1, w['learning_rate'], w['betas'], 1e-6,
                      0.0 if O.no_decay(name) else w['weight_decay'])
        master = opt.state[named[name]]['master'].float().cpu()
        worst = max(worst, float((master - p_ref).abs().max() / (p_ref.abs().max() + 1e-12)))
    assert worst < 1e-5, worst                                 # fp32 master weights vs the reference rule
    assert arena.check()


@pytest.mark.parametrize("task,bsz", [('mrfr', 4), ('mrckl', 4), ('itm_ot', 4), ('mlm', 32), ('mrfr', 32), ('mrckl', 32), ('itm_ot', 32)])
def test_base_model_other_tasks_vs_oracle(tmp_path, task, bsz):
    """MLM, MRFR, MRC-KL and ITM + 0.1 * OT (pretrain.py:270-290) at UNITER-base size, 12 layers: a ragged batch of 4, and the
    c3 micro-batch itself — 32 full-length examples = 3 072 tokens, the shape at which every parameter gradient of the
    encoder comes out of the ONE deferred launch (gemm8_multi_kernel) — against the oracle."""
    from uniter_amd.utils.synthetic import make_batch
    model, cfg = _base_model(tmp_path)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    tname = 'itm' if task == 'itm_ot' else task
    batch = make_batch(tname, bsz, seed=21, ragged=(bsz == 4), with_ot=(task == 'itm_ot'))
    if bsz == 32:
        assert batch['attn_masks'].shape == (32, 96) and int(batch['attn_masks'].sum()) == 3072
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != 'cls.predictions.decoder.weight'}
    leaf['cls.predictions.decoder.weight'] = leaf['uniter.embeddings.word_embeddings.weight']

    def ref_objective(sd_, cfg_, b_):
        if task == 'mlm':
            loss, seq = O.mlm_loss(sd_, cfg_, b_)
            return loss, seq, loss.mean()
        if task == 'mrfr':
            loss, seq = O.mrfr_loss(sd_, cfg_, b_)
            return loss, seq, loss.mean()
        if task == 'mrckl':
            loss, seq = O.mrc_loss(sd_, cfg_, b_, kl=True)
            return loss, seq, loss.mean()
        out = O.itm_ot_loss(sd_, cfg_, b_, ot_lambda=0.1)           # (scalar objective, itm losses, ot distances, seq)
        return out[0], out[3], out[0]

    ref_loss, ref_seq, ref_obj = ref_objective(leaf, cfg, batch)
    ref_obj.backward()

    _prep(model)
    d = _to_dev(batch)
    for p in model.parameters():
        p.grad = None
    out = model(d, task=tname, compute_loss=True)
    if task == 'itm_ot':
        itm_loss, (ot_pos, ot_neg) = out
        obj = itm_loss.mean() + 0.1 * (ot_pos.sum() - ot_neg.sum()) / (ot_pos.size(0) + ot_neg.size(0))
        torch.testing.assert_close(obj.detach().float().cpu(), ref_obj.detach(), rtol=3e-2, atol=3e-2)
    else:
        _check_loss(out, ref_loss.detach(), atol=3e-2)
        obj = out.mean()
    obj.backward()

    def yard_fn(sd_, cfg_, b_):
        loss_, seq_, obj_ = ref_objective(sd_, cfg_, b_)
        return obj_.reshape(1), seq_
    if task == 'itm_ot':
        ygrads = {}        # no torch-bf16 yardstick: 50 IPOT iterations in bf16 are not a meaningful reference (absolute bounds)
    else:
        _, _, ygrads = _yardstick(yard_fn, sd, cfg, batch)
    _check_all_grads(dict(model.named_parameters()), leaf, ygrads, min_checked=200)


def test_large_full_depth_vqa_vs_oracle(tmp_path):
    """config/uniter-large.json at its full 24 layers (H=1024, 16 heads, I=4096), VQA head, B=2, L=60+36."""
    import json
    from uniter_amd.model.vqa import UniterForVisualQuestionAnswering
    from uniter_amd.utils.synthetic import make_batch
    cfg = dict(LARGE_CFG)
    assert cfg['num_hidden_layers'] == 24
    path = tmp_path / "large24.json"
    path.write_text(json.dumps(cfg))
    torch.manual_seed(13)
    model = UniterForVisualQuestionAnswering.from_pretrained(str(path), {}, img_dim=2048, num_answer=N_ANS)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.to(torch.bfloat16).float())
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    batch = make_batch('vqa', 2, seed=14, ragged=True, num_answer=N_ANS)
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref_loss, ref_seq = O.vqa_loss(leaf, cfg, batch)
    (ref_loss.mean() * N_ANS).backward()
    _prep(model)
    d = _to_dev(batch)
    seq = model.uniter(d['input_ids'], d['position_ids'], d['img_feat'], d['img_pos_feat'], d['attn_masks'],
                       d['gather_index'], output_all_encoded_layers=False)
    valid = batch['attn_masks'].bool()
    _, yseq, ygrads = _yardstick(O.vqa_loss, sd, cfg, batch)
    _check_hidden(seq.detach()[valid.to(seq.device)], ref_seq.detach()[valid], "large 24-layer hidden", yseq[valid])
    loss = model(d, compute_loss=True)
    _check_loss(loss, ref_loss.detach(), atol=3e-2)
    for p in model.parameters():
        p.grad = None
    (model(d, compute_loss=True).float().mean() * N_ANS).backward()
    _check_all_grads(dict(model.named_parameters()), leaf, ygrads, scale=1.0 / N_ANS, min_checked=390)


def test_c4_large_full_depth_vqa_b32_hidden_and_loss_vs_oracle(tmp_path):
    """BASELINE.json configs[3] at the config's own micro-batch: UNITER-large, 24 layers, VQA head, B = 32 full-length sequences
    (60 + 36 tokens = 3 072 rows: the tile shapes and the one deferred launch of the benchmarked c4 step) — final hidden states and
    per-example losses against the oracle's forward (gradients of this configuration: test_large_full_depth_vqa_vs_oracle at
    B = 2, where the oracle's backward finishes in seconds)."""
    import json
    from uniter_amd.model.vqa import UniterForVisualQuestionAnswering
    from uniter_amd.utils.synthetic import make_batch
    cfg = dict(LARGE_CFG)
    path = tmp_path / "large24b32.json"
    path.write_text(json.dumps(cfg))
    torch.manual_seed(23)
    model = UniterForVisualQuestionAnswering.from_pretrained(str(path), {}, img_dim=2048, num_answer=3129)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.to(torch.bfloat16).float())
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    batch = make_batch('vqa', 32, seed=24, num_answer=3129)
    assert batch['attn_masks'].shape == (32, 96) and int(batch['attn_masks'].sum()) == 3072
    with torch.no_grad():
        ref_loss, ref_seq = O.vqa_loss(sd, cfg, batch)
        _prep(model)
        d = _to_dev(batch)
        seq = model.uniter(d['input_ids'], d['position_ids'], d['img_feat'], d['img_pos_feat'], d['attn_masks'],
                           d['gather_index'], output_all_encoded_layers=False)
        dev = _dev()
        sdb = {k: v.to(dev, torch.bfloat16) for k, v in sd.items()}
        bb = {k: ((v.to(dev, torch.bfloat16) if v.is_floating_point() else v.to(dev)) if torch.is_tensor(v) else v) for k, v in batch.items()}
        _, yseq = O.vqa_loss(sdb, cfg, bb)                       # the survey's secondary yardstick: the oracle's ops in torch bf16
    valid = batch['attn_masks'].bool()
    _check_hidden(seq.detach()[valid.to(seq.device)], ref_seq[valid], "c4 B=32 hidden", yseq.float().cpu()[valid])
    model.train()
    loss = model(d, compute_loss=True)
    _check_loss(loss, ref_loss, atol=3e-2)


@pytest.mark.parametrize("task", ['mlm', 'itm_ot'])
def test_c5_large_24_layers_l178_pretrain_accum2_vs_oracle(tmp_path, task):
    """BASELINE.json configs[4] as itself: config/uniter-large.json at its full 24 layers with the pre-training heads, text up to
    128 tokens + 50 regions (L = 178, config/pretrain-alldata-large-16gpu.json), gradient accumulation 2 (pretrain.py:298-312:
    the two micro-batches' gradients are SUMMED) — MLM, and ITM + 0.1 * OT (pretrain.py:270-290).  Per-example losses of both
    micro-batches and every accumulated gradient against the oracle; no tolerance beyond SURVEY 8c's (5e-2, or 2x the
    torch-bf16 yardstick)."""
    import json
    from uniter_amd.model.pretrain import UniterForPretraining
    from uniter_amd.utils.synthetic import make_batch
    cfg = dict(LARGE_CFG)
    assert cfg['num_hidden_layers'] == 24
    path = tmp_path / "large24pre.json"
    path.write_text(json.dumps(cfg))
    torch.manual_seed(23)
    model = UniterForPretraining.from_pretrained(str(path), {}, img_dim=2048, img_label_dim=1601)
    with torch.no_grad():
        g = torch.Generator().manual_seed(24)
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.02)
            p.copy_(p.to(torch.bfloat16).float())
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    tname = 'itm' if task == 'itm_ot' else task
    # micro-batch 0: both examples at the full 128 + 50 tokens (L = 178); micro-batch 1: ragged (padding, gather_index)
    batches = [make_batch(tname, 2, max_txt_len=128, num_bb=50, seed=31 + k, ragged=(k == 1), min_txt_len=90, min_bb=30,
                          with_ot=(task == 'itm_ot')) for k in range(2)]
    assert batches[0]['attn_masks'].shape[1] == 178 and int(batches[0]['attn_masks'].sum()) == 2 * 178
    assert 128 < batches[1]['attn_masks'].shape[1] <= 178
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != 'cls.predictions.decoder.weight'}
    leaf['cls.predictions.decoder.weight'] = leaf['uniter.embeddings.word_embeddings.weight']

    def ref_objective(sd_, cfg_, b_):
        if task == 'mlm':
            loss, seq = O.mlm_loss(sd_, cfg_, b_)
            return loss, seq, loss.mean()
        out = O.itm_ot_loss(sd_, cfg_, b_, ot_lambda=0.1)           # (scalar objective, itm losses, ot distances, seq)
        return out[0], out[3], out[0]

    refs = []
    for b in batches:                                               # accumulation: backward twice into the same leaves
        ref_loss, ref_seq, ref_obj = ref_objective(leaf, cfg, b)
        ref_obj.backward()
        refs.append((ref_loss.detach(), ref_seq.detach(), ref_obj.detach()))

    _prep(model)
    for p in model.parameters():
        p.grad = None
    for b, (ref_loss, ref_seq, ref_obj) in zip(batches, refs):
        d = _to_dev(b)
        out = model(d, task=tname, compute_loss=True)
        if task == 'itm_ot':
            itm_loss, (ot_pos, ot_neg) = out
            obj = itm_loss.mean() + 0.1 * (ot_pos.sum() - ot_neg.sum()) / (ot_pos.size(0) + ot_neg.size(0))
            torch.testing.assert_close(obj.detach().float().cpu(), ref_obj, rtol=3e-2, atol=3e-2)
        else:
            _check_loss(out, ref_loss, atol=3e-2)
            obj = out.float().mean()
        obj.backward()                                              # accumulates into .grad (no zero_grad in between)

    # torch-bf16 yardstick of the same accumulated objective
    dev = _dev()
    ysd = {k: v.detach().to(dev, torch.bfloat16).requires_grad_(True) for k, v in sd.items() if k != 'cls.predictions.decoder.weight'}
    ysd['cls.predictions.decoder.weight'] = ysd['uniter.embeddings.word_embeddings.weight']
    def yard_itm_ot(sd_, cfg_, b_):
        """The yardstick of ITM + OT: the ENCODER in torch bf16 ops, the ITM / optimal-transport head on its fp32 cast (50 IPOT
        iterations in bf16 are not a meaningful reference; our OT kernel works in fp32 on the bf16 encoder output too)."""
        seq = O.uniter_model(sd_, cfg_, b_['input_ids'], b_['position_ids'], b_['img_feat'], b_['img_pos_feat'], b_['attn_masks'],
                             b_['gather_index']).float()
        head = {k: v.float() for k, v in sd_.items() if k.startswith(('uniter.pooler.', 'itm_output.'))}
        scores = O.linear(O.pooler(head, 'uniter.pooler.', seq), head['itm_output.weight'], head['itm_output.bias'])
        itm = torch.nn.functional.cross_entropy(scores, b_['targets'], reduction='none')
        ot = b_['ot_inputs']
        txt, img = O.ot_scatter_split(seq, ot['ot_scatter'], ot['scatter_max'], b_['input_ids'].size(1), b_['img_feat'].size(1))
        dist, _ = O.optimal_transport_dist(txt, img, ot['txt_pad'], ot['img_pad'])
        pos, neg = dist[b_['targets'] == 1], dist[b_['targets'] == 0]
        return itm.mean() + 0.1 * (pos.sum() - neg.sum()) / (pos.numel() + neg.numel())

    def to_dev_bf16(v):
        if torch.is_tensor(v):
            return v.to(dev, torch.bfloat16) if v.is_floating_point() else v.to(dev)
        if isinstance(v, dict):
            return {k: to_dev_bf16(x) for k, x in v.items()}
        return v

    for b in batches:
        yb = {k: to_dev_bf16(v) for k, v in b.items()}
        if task == 'itm_ot':
            yard_itm_ot(ysd, cfg, yb).backward()
        else:
            ref_objective(ysd, cfg, yb)[2].float().backward()
    ygrads = {k: v.grad.float().cpu() for k, v in ysd.items() if v.grad is not None}
    named = dict(model.named_parameters())
    n_yard = sum(1 for n, p in named.items() if p.grad is not None and n in leaf and leaf[n].grad is not None
                 and float(leaf[n].grad.abs().max()) >= 1e-6
                 and rel_l2(p.grad.float().cpu(), leaf[n].grad) > (GRAD_L2 if n.startswith('uniter.') else GRAD_L2_HEAD))
    checked = _check_all_grads(named, leaf, ygrads, min_checked=400)
    print("c5 parity (%s, 24 layers, L=178, accumulation 2): %d gradients checked, %d over the absolute bound "
          "(held to 2x the torch-bf16 yardstick)" % (task, checked, n_yard))


def test_attention_pool_gradient_conditioning(c2_oracle):
    """Why the two parameters of AttentionPool's Linear(H, 1) (model/nlvr2.py:110-125) sit at ~0.13 relative L2 on the headline
    workload while every other tensor is under 5e-2: the gradient is sum_t p_t (dp_t - sum_s p_s dp_s) [relu'] x_t over 96
    tokens — the softmax backward cancels to a small remainder.  Shown on the oracle's own fp32 activations of the c2 step:
      (a) the HIP pooling kernels fed the oracle's pool input (rounded to the bf16 they take) reproduce the fp32 formula ON THAT
          SAME INPUT to <= 2e-2 — the kernel is not the source;
      (b) that single bf16 rounding of the input alone moves the fp32 formula's gradient by ~3e-2 (amplification ~20 over the
          input's 1.7e-3; measured on the CPU: weight 0.033, bias 0.028), so 12 stacked bf16 layers upstream land where the
          headline test sees them — inside 2x the torch-bf16 yardstick, without any tensor-specific exception."""
    from uniter_amd import ops
    cfg, sd, leaf, taps = (c2_oracle[k] for k in ('cfg', 'sd', 'leaf', 'taps'))
    g_out = taps['pooled'].grad.detach()                        # [pairs, 2H] = d loss / d pooled, left | right
    H = cfg['hidden_size']
    xbf, in_err, (fw, fb) = c2_oracle['pool_in_bf16'], c2_oracle['pool_in_err'], c2_oracle['pool_formula_grads']
    amp_w, amp_b = (c2_oracle['pool_amplification'][k] * in_err for k in ('attn_pool.fc.0.weight', 'attn_pool.fc.0.bias'))

    dev = _dev()
    lin = torch.nn.Linear(H, 1).to(dev).bfloat16()
    with torch.no_grad():
        lin.weight.copy_(sd['attn_pool.fc.0.weight'].to(dev, torch.bfloat16))
        lin.bias.copy_(sd['attn_pool.fc.0.bias'].to(dev, torch.bfloat16))
    for k, (x, m) in enumerate(zip(xbf, taps['pool_pad'])):
        out = ops.attention_pool(x.to(dev, torch.bfloat16), m.to(dev), lin, 0.0, True)
        (out.float() * g_out[:, k * H:(k + 1) * H].to(dev)).sum().backward()
    kw, kb = lin.weight.grad.float().cpu(), lin.bias.grad.float().cpu()
    print("attention pool conditioning: input rounding %.2e -> fp32 formula moves by w %.3f b %.3f; HIP kernel vs formula on the "
          "same input: w %.4f b %.4f" % (in_err, amp_w, amp_b, rel_l2(kw, fw), rel_l2(kb, fb)))
    assert rel_l2(kw, fw) <= 2e-2 and rel_l2(kb, fb) <= 2e-2, (rel_l2(kw, fw), rel_l2(kb, fb))
    assert amp_w >= 5 * in_err and amp_b >= 5 * in_err, (in_err, amp_w, amp_b)      # ill-conditioned: >= 5x amplification


def test_long_sequences_up_to_512_dense_and_packed_vs_oracle(tmp_path):
    """L = 284 text + 100 regions = 384 (max_position_embeddings is 512, pretrain.py:637-640; the NLVR2 triplet format of
    60 + 2 x 100 tokens already exceeds 256): the forward kernel keeps a 512-key score row in registers and the backward pass
    is the two-launch form.  Dense and packed execution against the oracle and against each other."""
    import json
    from uniter_amd.model.pretrain import UniterForPretraining
    from uniter_amd.utils.synthetic import make_batch
    cfg = dict(BASE_CFG, num_hidden_layers=2, hidden_size=256, num_attention_heads=4, intermediate_size=512)
    path = tmp_path / "long.json"
    path.write_text(json.dumps(cfg))
    torch.manual_seed(5)
    model = UniterForPretraining.from_pretrained(str(path), {}, img_dim=2048, img_label_dim=1601)
    with torch.no_grad():
        g = torch.Generator().manual_seed(6)
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.02)
            p.copy_(p.to(torch.bfloat16).float())
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    batch = make_batch('mlm', 3, max_txt_len=284, num_bb=100, seed=4, ragged=True, min_txt_len=200, min_bb=60)
    assert 256 < batch['attn_masks'].shape[1] <= 384           # longest example of the ragged batch: beyond the one-launch limit
    valid = batch['attn_masks'].bool()
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != 'cls.predictions.decoder.weight'}
    leaf['cls.predictions.decoder.weight'] = leaf['uniter.embeddings.word_embeddings.weight']
    ref_loss, ref_seq = O.mlm_loss(leaf, cfg, batch)
    ref_loss.mean().backward()
    _prep(model)
    d = _to_dev(batch)
    _, yseq, ygrads = _yardstick(O.mlm_loss, sd, cfg, batch)
    outs = {}
    for pack in (False, True):
        model.uniter.pack_padding = pack
        for p in model.parameters():
            p.grad = None
        seq = model.uniter(d['input_ids'], d['position_ids'], d['img_feat'], d['img_pos_feat'], d['attn_masks'],
                           d['gather_index'], output_all_encoded_layers=False)
        _check_hidden(seq.detach()[valid.to(seq.device)], ref_seq.detach()[valid], "L=384 hidden (pack=%s)" % pack, yseq[valid])
        loss = model(d, task='mlm', compute_loss=True)
        _check_loss(loss, ref_loss.detach(), atol=3e-2)
        loss.float().mean().backward()
        _check_all_grads(dict(model.named_parameters()), leaf, ygrads, min_checked=40)
        outs[pack] = seq.detach().float().cpu()
    model.uniter.pack_padding = False
    torch.testing.assert_close(outs[True][valid], outs[False][valid], rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("act", ["relu", "swish"])
def test_hidden_act_relu_and_swish_vs_oracle(tmp_path, act, monkeypatch):
    """config.hidden_act other than gelu (model/layer.py:44 ACT2FN): the FFN epilogues switch activation; the MLM head's HIP path
    covers the erf-GELU transform only, so its PyTorch module path has to be asked for (UNITER_AMD_HEAD_TORCH=1) — without the
    switch the head raises instead of degrading silently."""
    import json
    monkeypatch.setenv("UNITER_AMD_HEAD_TORCH", "1")
    from uniter_amd.model.pretrain import UniterForPretraining
    from uniter_amd.utils.synthetic import make_batch
    cfg = dict(BASE_CFG, num_hidden_layers=2, hidden_size=256, num_attention_heads=4, intermediate_size=512, hidden_act=act)
    path = tmp_path / "act.json"
    path.write_text(json.dumps(cfg))
    torch.manual_seed(15)
    model = UniterForPretraining.from_pretrained(str(path), {}, img_dim=2048, img_label_dim=1601)
    with torch.no_grad():
        g = torch.Generator().manual_seed(16)
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.02)
            elif 'intermediate' in n:
                p.mul_(4.0)                                    # larger pre-activations so that the non-linearity matters
            p.copy_(p.to(torch.bfloat16).float())
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    batch = make_batch('mlm', 4, seed=17, ragged=True)
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != 'cls.predictions.decoder.weight'}
    leaf['cls.predictions.decoder.weight'] = leaf['uniter.embeddings.word_embeddings.weight']
    ref_loss, ref_seq = O.mlm_loss(leaf, cfg, batch)
    ref_loss.mean().backward()
    gelu_seq = O.mlm_loss({k: v.detach() for k, v in leaf.items()}, dict(cfg, hidden_act='gelu'), batch)[1]
    valid = batch['attn_masks'].bool()
    assert float((gelu_seq - ref_seq.detach())[valid].abs().max()) > 0.1      # the activation really changes the result
    _prep(model)
    d = _to_dev(batch)
    seq = model.uniter(d['input_ids'], d['position_ids'], d['img_feat'], d['img_pos_feat'], d['attn_masks'],
                       d['gather_index'], output_all_encoded_layers=False)
    _, yseq, ygrads = _yardstick(O.mlm_loss, sd, cfg, batch)
    _check_hidden(seq.detach()[valid.to(seq.device)], ref_seq.detach()[valid], "hidden_act=%s" % act, yseq[valid])
    loss = model(d, task='mlm', compute_loss=True)
    _check_loss(loss, ref_loss.detach(), atol=3e-2)
    for p in model.parameters():
        p.grad = None
    model(d, task='mlm', compute_loss=True).mean().backward()
    _check_all_grads(dict(model.named_parameters()), leaf, ygrads, min_checked=40)
