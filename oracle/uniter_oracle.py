"""CPU oracle of the UNITER encoder training path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain functional restatement (torch CPU ops, fp32 or fp64, autograd for the backward pass) of the
reference algorithm for the hot path of SURVEY.md §8.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this module; the product (`uniter_amd/`) never does, and fails
loudly when its HIP library is missing.

Every function cites the reference lines it restates (paths relative to the ChenRocks/UNITER checkout).
Parameters are passed as a flat dict {state_dict key: tensor} using the reference's key names, so a
reference checkpoint / golden fixture feeds it directly.

Pinning: `tests/golden/make_golden.py` runs the REAL reference modules (imported from /root/reference under
an apex -> torch.nn.LayerNorm shim) on seeded tiny models and stores inputs + outputs + gradients +
post-AdamW weights in `tests/golden/*.npz`; `tests/test_oracle_golden.py` checks this oracle against those
vectors (fp32, rtol 1e-4).  The reference itself ships no tests or golden vectors; third-party
arithmetic it relies on (apex FusedLayerNorm, Horovod averaging) is restated with its documented semantics.
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------------------------------
def gelu(x):
    """model/layer.py:31-37 — exact erf GELU."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def layer_norm(x, weight, bias, eps=1e-12):
    """apex FusedLayerNorm(H, eps=1e-12) as used at model/layer.py:108,149: biased variance over the last dim."""
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * weight + bias


def linear(x, weight, bias=None):
    """nn.Linear: y = x W^T + b, W is [out, in]."""
    y = x.matmul(weight.t())
    return y if bias is None else y + bias


# ----------------------------------------------------------------------------------------------------
# encoder
# ----------------------------------------------------------------------------------------------------
def self_attention(x, ext_mask, sd, prefix, num_heads):
    """model/layer.py:75-101 BertSelfAttention.forward (dropout off)."""
    B, L, H = x.shape
    dh = H // num_heads

    def split(t):                                    # transpose_for_scores :70-73
        return t.view(B, L, num_heads, dh).permute(0, 2, 1, 3)

    q = split(linear(x, sd[prefix + 'query.weight'], sd[prefix + 'query.bias']))
    k = split(linear(x, sd[prefix + 'key.weight'], sd[prefix + 'key.bias']))
    v = split(linear(x, sd[prefix + 'value.weight'], sd[prefix + 'value.bias']))
    scores = q.matmul(k.transpose(-1, -2)) / math.sqrt(dh)          # :85-86 (division after the matmul)
    scores = scores + ext_mask                                      # :88   additive mask [B,1,1,L]
    probs = torch.softmax(scores, dim=-1)                           # :91
    ctx = probs.matmul(v)                                           # :97
    return ctx.permute(0, 2, 1, 3).contiguous().view(B, L, H)       # :98-100


def swish(x):
    """model/layer.py:40-41."""
    return x * torch.sigmoid(x)


ACT2FN = {"gelu": gelu, "relu": F.relu, "swish": swish}           # model/layer.py:44


def bert_layer(x, ext_mask, sd, prefix, num_heads, act="gelu"):
    """model/layer.py:166-170 BertLayer.forward = attention (:124-127) -> intermediate (:139-142) -> output (:152-156)."""
    ctx = self_attention(x, ext_mask, sd, prefix + 'attention.self.', num_heads)
    a = linear(ctx, sd[prefix + 'attention.output.dense.weight'], sd[prefix + 'attention.output.dense.bias'])
    a = layer_norm(a + x, sd[prefix + 'attention.output.LayerNorm.weight'], sd[prefix + 'attention.output.LayerNorm.bias'])
    i = ACT2FN[act](linear(a, sd[prefix + 'intermediate.dense.weight'], sd[prefix + 'intermediate.dense.bias']))   # :134-142
    o = linear(i, sd[prefix + 'output.dense.weight'], sd[prefix + 'output.dense.bias'])
    return layer_norm(o + a, sd[prefix + 'output.LayerNorm.weight'], sd[prefix + 'output.LayerNorm.bias'])


def text_embeddings(sd, prefix, input_ids, position_ids, token_type_ids=None):
    """model/model.py:232-245 UniterTextEmbeddings.forward (dropout off)."""
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)
    e = (sd[prefix + 'word_embeddings.weight'][input_ids]
         + sd[prefix + 'position_embeddings.weight'][position_ids]
         + sd[prefix + 'token_type_embeddings.weight'][token_type_ids])
    return layer_norm(e, sd[prefix + 'LayerNorm.weight'], sd[prefix + 'LayerNorm.bias'])


def image_embeddings(sd, prefix, img_feat, img_pos_feat, type_embeddings, img_masks=None):
    """model/model.py:261-272 UniterImageEmbeddings.forward (dropout off)."""
    if img_masks is not None:
        mask_table = sd[prefix + 'mask_embedding.weight'].clone()
        mask_table[0] = 0                                            # :263 row 0 forced to zero
        img_feat = img_feat + mask_table[img_masks.long()]           # :264-265
    t_im = layer_norm(linear(img_feat, sd[prefix + 'img_linear.weight'], sd[prefix + 'img_linear.bias']),
                      sd[prefix + 'img_layer_norm.weight'], sd[prefix + 'img_layer_norm.bias'])
    t_pos = layer_norm(linear(img_pos_feat, sd[prefix + 'pos_linear.weight'], sd[prefix + 'pos_linear.bias']),
                       sd[prefix + 'pos_layer_norm.weight'], sd[prefix + 'pos_layer_norm.bias'])
    e = t_im + t_pos + type_embeddings
    return layer_norm(e, sd[prefix + 'LayerNorm.weight'], sd[prefix + 'LayerNorm.bias'])


def uniter_model(sd, cfg, input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index=None,
                 img_masks=None, txt_type_ids=None, img_type_ids=None, prefix='uniter.', all_layers=False):
    """model/model.py:336-367 UniterModel.forward.  cfg: dict with num_hidden_layers, num_attention_heads."""
    dtype = sd[prefix + 'embeddings.word_embeddings.weight'].dtype
    ext_mask = (1.0 - attention_mask[:, None, None, :].to(dtype)) * -10000.0       # :342-345
    txt = img = None
    if input_ids is not None:
        txt = text_embeddings(sd, prefix + 'embeddings.', input_ids, position_ids, txt_type_ids)
    if img_feat is not None:
        if img_type_ids is None:
            img_type_ids = torch.ones(img_feat.shape[:2], dtype=torch.long, device=img_feat.device)   # :313-314
        type_emb = sd[prefix + 'embeddings.token_type_embeddings.weight'][img_type_ids]
        img = image_embeddings(sd, prefix + 'img_embeddings.', img_feat.to(dtype), img_pos_feat.to(dtype), type_emb,
                               img_masks)
    if txt is None:
        h = img
    elif img is None:
        h = txt
    else:                                                                           # :321-334
        H = txt.shape[-1]
        idx = gather_index.unsqueeze(-1).expand(-1, -1, H)
        h = torch.gather(torch.cat([txt, img], dim=1), dim=1, index=idx)
    outs = []
    for l in range(cfg['num_hidden_layers']):                                       # :282-292
        h = bert_layer(h, ext_mask, sd, '%sencoder.layer.%d.' % (prefix, l), cfg['num_attention_heads'],
                       act=cfg.get('hidden_act', 'gelu'))
        outs.append(h)
    return outs if all_layers else h


def pooler(sd, prefix, hidden):
    """model/layer.py:179-185 BertPooler."""
    return torch.tanh(linear(hidden[:, 0], sd[prefix + 'dense.weight'], sd[prefix + 'dense.bias']))


# ----------------------------------------------------------------------------------------------------
# heads (un-reduced losses, as the reference returns them)
# ----------------------------------------------------------------------------------------------------
def _head_transform(sd, prefix, x):
    """Linear -> GELU -> LayerNorm (model/layer.py:198-202; nn.Sequential version model/pretrain.py:23-25)."""
    return layer_norm(gelu(linear(x, sd[prefix + '0.weight'], sd[prefix + '0.bias'])),
                      sd[prefix + '2.weight'], sd[prefix + '2.bias'])


def mlm_head_loss(sd, cfg, h, labels, ignore_index=-100):
    """model/layer.py:188-222 BertOnlyMLMHead (transform, decoder tied to word_embeddings, bias) + the cross entropy of
    model/pretrain.py:129-133, un-reduced, over rows `h` [n, H]."""
    t = 'cls.predictions.transform.'
    act = ACT2FN[cfg.get('hidden_act', 'gelu')]                                     # model/layer.py:192-195
    h = layer_norm(act(linear(h, sd[t + 'dense.weight'], sd[t + 'dense.bias'])),
                   sd[t + 'LayerNorm.weight'], sd[t + 'LayerNorm.bias'])
    scores = linear(h, sd['uniter.embeddings.word_embeddings.weight']) + sd['cls.predictions.bias']
    return F.cross_entropy(scores, labels, ignore_index=ignore_index, reduction='none')


def mlm_loss(sd, cfg, batch):
    """model/pretrain.py:107-127 forward_mlm + model/layer.py:188-222 (decoder tied to word_embeddings)."""
    seq = uniter_model(sd, cfg, batch['input_ids'], batch['position_ids'], batch['img_feat'], batch['img_pos_feat'],
                       batch['attn_masks'], batch['gather_index'])
    txt_part = seq[:, :batch['input_ids'].size(1), :]                              # :114 text positions only
    picked = batch['txt_labels'] != -1
    return mlm_head_loss(sd, cfg, txt_part[picked], batch['txt_labels'][picked]), seq


def mrfr_loss(sd, cfg, batch):
    """model/pretrain.py:135-154 forward_mrfr + :19-33 RegionFeatureRegression (tied to img_linear.weight^T)."""
    seq = uniter_model(sd, cfg, batch['input_ids'], batch['position_ids'], batch['img_feat'], batch['img_pos_feat'],
                       batch['attn_masks'], batch['gather_index'], img_masks=batch['img_masks'])
    h = _head_transform(sd, 'feat_regress.net.', seq[batch['img_mask_tgt'].bool()])
    pred = linear(h, sd['uniter.img_embeddings.img_linear.weight'].t(), sd['feat_regress.bias'])
    return F.mse_loss(pred, batch['feat_targets'].to(pred.dtype), reduction='none'), seq


def region_classification_loss(sd, h, targets, kl=True, prefix='region_classifier.net.'):
    """model/pretrain.py:36-47 RegionClassification over rows `h` [n, H] + the KL / hard-label cross entropy of :213-229, un-reduced."""
    logits = linear(_head_transform(sd, prefix, h), sd[prefix + '3.weight'], sd[prefix + '3.bias'])
    targets = targets.to(logits.dtype)
    if kl:
        return F.kl_div(F.log_softmax(logits, dim=-1), targets, reduction='none')
    hard = torch.max(targets[:, 1:], dim=-1)[1] + 1
    return F.cross_entropy(logits, hard, ignore_index=0, reduction='none')


def mrc_loss(sd, cfg, batch, kl=True):
    """model/pretrain.py:201-229 forward_mrc + :36-47 RegionClassification."""
    seq = uniter_model(sd, cfg, batch['input_ids'], batch['position_ids'], batch['img_feat'], batch['img_pos_feat'],
                       batch['attn_masks'], batch['gather_index'], img_masks=batch['img_masks'])
    return region_classification_loss(sd, seq[batch['img_mask_tgt'].bool()], batch['label_targets'], kl), seq


def itm_loss(sd, cfg, batch):
    """model/pretrain.py:156-199 forward_itm without the OT term."""
    seq = uniter_model(sd, cfg, batch['input_ids'], batch['position_ids'], batch['img_feat'], batch['img_pos_feat'],
                       batch['attn_masks'], batch['gather_index'])
    scores = linear(pooler(sd, 'uniter.pooler.', seq), sd['itm_output.weight'], sd['itm_output.bias'])
    return F.cross_entropy(scores, batch['targets'], reduction='none'), seq


# ----------------------------------------------------------------------------------------------------
# word-region alignment: IPOT optimal-transport distance (model/ot.py) and its wiring into ITM
# ----------------------------------------------------------------------------------------------------
def ot_cost_matrix_cosine(x, y, eps=1e-5):
    """model/ot.py:11-22: 1 - cos(x_m, y_n); F.normalize divides by max(||.||, eps)."""
    xn = x / x.norm(dim=-1, keepdim=True).clamp_min(eps)
    yn = y / y.norm(dim=-1, keepdim=True).clamp_min(eps)
    return 1 - xn.matmul(yn.transpose(1, 2))


def ot_ipot(C, x_len, x_pad, y_len, y_pad, joint_pad, beta=0.5, iteration=50, k=1):
    """model/ot.py:36-69 with bool masks (the reference's uint8 masks no longer index in torch >= 1.2).
    C [B,M,N]; returns the transport plan T [B,N,M]; no gradient (the reference decorates it with no_grad)."""
    with torch.no_grad():
        b, m, n = C.shape
        sigma = torch.ones(b, m, dtype=C.dtype, device=C.device) / x_len.unsqueeze(1)                    # :39-40
        T = torch.ones(b, n, m, dtype=C.dtype, device=C.device)                                     # :41
        A = torch.exp(-C.transpose(1, 2) / beta)                                         # :42
        sigma = sigma.masked_fill(x_pad, 0)                                              # :45
        jp = joint_pad.transpose(1, 2)                                                   # :46
        T = T.masked_fill(jp, 0)                                                         # :47
        A = A.masked_fill(jp, 0)                                                         # :48
        xl = x_len.view(b, 1, 1)                                                         # :51-52
        yl = y_len.view(b, 1, 1)
        x_mask = (x_pad.to(C.dtype) * 1e4).unsqueeze(1)                                  # :55-56
        y_mask = (y_pad.to(C.dtype) * 1e4).unsqueeze(1)
        delta = None
        for _ in range(iteration):                                                       # :58-65
            Q = A * T
            sigma = sigma.view(b, m, 1)
            for _ in range(k):
                delta = 1 / (yl * Q.matmul(sigma).view(b, 1, n) + y_mask)
                sigma = 1 / (xl * delta.matmul(Q) + x_mask)
            T = delta.view(b, n, 1) * Q * sigma
        return T.masked_fill(jp, 0)                                                      # :66


def optimal_transport_dist(txt_emb, img_emb, txt_pad, img_pad, beta=0.5, iteration=50, k=1):
    """model/ot.py:70-85: trace(cost @ T) with T detached; gradient flows through the cosine cost only."""
    txt_pad, img_pad = txt_pad.bool(), img_pad.bool()
    cost = ot_cost_matrix_cosine(txt_emb, img_emb)
    joint_pad = txt_pad.unsqueeze(-1) | img_pad.unsqueeze(-2)                            # :75
    cost = cost.masked_fill(joint_pad, 0)                                                # :76
    txt_len = (txt_pad.size(1) - txt_pad.sum(dim=1)).to(cost.dtype)                      # :78-81
    img_len = (img_pad.size(1) - img_pad.sum(dim=1)).to(cost.dtype)
    T = ot_ipot(cost.detach(), txt_len, txt_pad, img_len, img_pad, joint_pad, beta, iteration, k)
    return torch.diagonal(cost.matmul(T), dim1=1, dim2=2).sum(-1), T                     # :83-84 (trace)


def ot_scatter_split(seq, ot_scatter, scatter_max, tl, il):
    """model/pretrain.py:168-181: undo the [txt_i ; img_i ; pad] compaction -> ([B,tl,H] text slots, [B,il,H] image slots)."""
    b, _, h = seq.shape
    max_l = max(int(scatter_max) + 1, tl + il)
    index = ot_scatter.unsqueeze(-1).expand_as(seq)
    ctx = torch.zeros(b, max_l, h, dtype=seq.dtype, device=seq.device).scatter(1, index, seq)
    return ctx[:, :tl, :], ctx[:, tl:tl + il, :]


def itm_ot_loss(sd, cfg, batch, ot_lambda=0.1):
    """forward_itm with ot_inputs (model/pretrain.py:156-199) and the loss mix of pretrain.py:270-290:
    itm.mean() + lambda * (sum(pos) - sum(neg)) / (n_pos + n_neg).  Returns (scalar loss, itm losses, ot distances, seq)."""
    seq = uniter_model(sd, cfg, batch['input_ids'], batch['position_ids'], batch['img_feat'], batch['img_pos_feat'],
                       batch['attn_masks'], batch['gather_index'])
    scores = linear(pooler(sd, 'uniter.pooler.', seq), sd['itm_output.weight'], sd['itm_output.bias'])
    itm = F.cross_entropy(scores, batch['targets'], reduction='none')
    ot = batch['ot_inputs']
    tl, il = batch['input_ids'].size(1), batch['img_feat'].size(1)
    txt, img = ot_scatter_split(seq, ot['ot_scatter'], ot['scatter_max'], tl, il)
    dist, _ = optimal_transport_dist(txt, img, ot['txt_pad'], ot['img_pad'])
    pos = dist[batch['targets'] == 1]
    neg = dist[batch['targets'] == 0]
    ot_loss = (pos.sum() - neg.sum()) / (pos.numel() + neg.numel())
    return itm.mean() + ot_lambda * ot_loss, itm, dist, seq


def vqa_loss(sd, cfg, batch):
    """model/vqa.py:30-52."""
    seq = uniter_model(sd, cfg, batch['input_ids'], batch['position_ids'], batch['img_feat'], batch['img_pos_feat'],
                       batch['attn_masks'], batch['gather_index'])
    h = _head_transform(sd, 'vqa_output.', pooler(sd, 'uniter.pooler.', seq))
    scores = linear(h, sd['vqa_output.3.weight'], sd['vqa_output.3.bias'])
    return F.binary_cross_entropy_with_logits(scores, batch['targets'].to(scores.dtype), reduction='none'), seq


def _mha(sd, prefix, query, key, value, key_padding_mask, num_heads):
    """model/attention.py:13-265 in the configuration UNITER uses ((L,N,E) layout, q != k == v, -inf key padding)."""
    L, N, E = query.shape
    S = key.shape[0]
    dh = E // num_heads
    w, b = sd[prefix + 'in_proj_weight'], sd[prefix + 'in_proj_bias']
    q = linear(query, w[:E], b[:E]) * (float(dh) ** -0.5)                          # :177 pre-scaled q
    k, v = linear(key, w[E:], b[E:]).chunk(2, dim=-1)                               # :103-127 kv_same branch
    q = q.contiguous().view(L, N * num_heads, dh).transpose(0, 1)
    k = k.contiguous().view(S, N * num_heads, dh).transpose(0, 1)
    v = v.contiguous().view(S, N * num_heads, dh).transpose(0, 1)
    att = torch.bmm(q, k.transpose(1, 2)).view(N, num_heads, L, S)
    att = att.masked_fill(key_padding_mask.bool()[:, None, None, :], float('-inf'))  # :243-249
    att = torch.softmax(att.view(N * num_heads, L, S), dim=-1)
    out = torch.bmm(att, v).transpose(0, 1).contiguous().view(L, N, E)
    return linear(out, sd[prefix + 'out_proj.weight'], sd[prefix + 'out_proj.bias'])


def attention_pool(sd, x, mask, prefix='attn_pool.'):
    """model/nlvr2.py:110-125 AttentionPool (dropout off): Linear(H,1) + ReLU scores, -1e4 at padded positions, softmax over
    the sequence, weighted sum.  x [B, L, H], mask [B, L] (True = padded) -> [B, H]."""
    score = torch.relu(linear(x, sd[prefix + 'fc.0.weight'], sd[prefix + 'fc.0.bias'])).squeeze(-1)
    score = score + mask.to(x.dtype) * -1e4
    return torch.softmax(score, dim=1).unsqueeze(1).matmul(x).squeeze(1)


def nlvr2_paired_attn_loss(sd, cfg, batch, taps=None):
    """model/nlvr2.py:163-204 UniterForNlvr2PairedAttn.forward + :110-125 AttentionPool (dropout off).
    `taps` (a dict, tests only) receives the attention pool's inputs and outputs: 'pool_in' [left, right], 'pool_pad', 'pooled'."""
    seq = uniter_model(sd, cfg, batch['input_ids'], batch['position_ids'], batch['img_feat'], batch['img_pos_feat'],
                       batch['attn_masks'], batch['gather_index'], img_type_ids=batch['img_type_ids'])
    bs, tl, d = seq.shape
    left, right = seq.contiguous().view(bs // 2, tl * 2, d).chunk(2, dim=1)
    pad = batch['attn_masks'] == 0
    lpad, rpad = pad.contiguous().view(bs // 2, tl * 2).chunk(2, dim=1)
    left, right = left.transpose(0, 1), right.transpose(0, 1)
    heads = cfg['num_attention_heads']
    l2r = _mha(sd, 'attn1.', left, right, right, rpad, heads)
    r2l = _mha(sd, 'attn2.', right, left, left, lpad, heads)

    def fc(x):
        return torch.relu(linear(x, sd['fc.0.weight'], sd['fc.0.bias']))

    left = fc(torch.cat([l2r, left], dim=-1)).transpose(0, 1)
    right = fc(torch.cat([r2l, right], dim=-1)).transpose(0, 1)

    pooled = torch.cat([attention_pool(sd, left, lpad), attention_pool(sd, right, rpad)], dim=-1)
    if taps is not None:
        taps.update(pool_in=[left, right], pool_pad=[lpad, rpad], pooled=pooled)
    scores = linear(pooled, sd['nlvr2_output.weight'], sd['nlvr2_output.bias'])
    return F.cross_entropy(scores, batch['targets'], reduction='none'), seq


# ----------------------------------------------------------------------------------------------------
# optimizer, schedule, clipping, allreduce semantics
# ----------------------------------------------------------------------------------------------------
def adamw_step(p, g, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0,
               correct_bias=True):
    """optim/adamw.py:74-101 for one tensor; `step` is the value AFTER the increment (>= 1).  Returns new tensors."""
    b1, b2 = betas
    exp_avg = exp_avg * b1 + (1.0 - b1) * g                                          # :77
    exp_avg_sq = exp_avg_sq * b2 + (1.0 - b2) * g * g                                # :78
    denom = exp_avg_sq.sqrt() + eps                                                  # :79
    step_size = lr
    if correct_bias:                                                                 # :81-86
        step_size = step_size * math.sqrt(1.0 - b2 ** step) / (1.0 - b1 ** step)
    p = p - step_size * exp_avg / denom                                              # :88
    if weight_decay > 0.0:                                                           # :100-101 (after the update, raw lr)
        p = p - lr * weight_decay * p
    return p, exp_avg, exp_avg_sq


def adamw_step_(p, g, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
    """In-place form of adamw_step with exactly the reference's op sequence (optim/adamw.py:77-101); used where the
    oracle is TIMED as the CPU baseline so that it does not pay for extra allocations the reference does not make."""
    b1, b2 = betas
    exp_avg.mul_(b1).add_(g, alpha=1.0 - b1)
    exp_avg_sq.mul_(b2).addcmul_(g, g, value=1.0 - b2)
    denom = exp_avg_sq.sqrt().add_(eps)
    step_size = lr
    if correct_bias:
        step_size = step_size * math.sqrt(1.0 - b2 ** step) / (1.0 - b1 ** step)
    p.addcdiv_(exp_avg, denom, value=-step_size)
    if weight_decay > 0.0:
        p.add_(p, alpha=-lr * weight_decay)


def no_decay(name):
    """optim/misc.py:14-22: case-sensitive substring match on the parameter name."""
    return any(tag in name for tag in ('bias', 'LayerNorm.bias', 'LayerNorm.weight'))


def clip_coef(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ as called at pretrain.py:329-331: (total_norm, multiplier)."""
    total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads))
    coef = max_norm / (total + 1e-6)
    return total, (coef if coef < 1.0 else 1.0)


def warmup_linear(step, warmup_step, tot_step):
    """optim/sched.py:17-21."""
    if step < warmup_step:
        return step / warmup_step
    return max(0, (tot_step - step) / (tot_step - warmup_step))


def get_lr_sched(global_step, learning_rate, warmup_steps, num_train_steps):
    """optim/sched.py:40-46."""
    lr = learning_rate * warmup_linear(global_step, warmup_steps, num_train_steps)
    return lr if lr > 0 else 1e-8


def allreduce_average(per_rank_tensors, rescale_denom=1.0):
    """utils/distributed.py:16-43 with Horovod 0.16's default average=True: mean over ranks, then / rescale_denom."""
    return sum(per_rank_tensors) / float(len(per_rank_tensors)) / float(rescale_denom)


def get_gather_index(txt_lens, num_bbs, batch_size, max_len, out_size):
    """data/data.py:271-279."""
    gi = torch.arange(0, out_size, dtype=torch.long).unsqueeze(0).repeat(batch_size, 1)
    for i, (tl, nbb) in enumerate(zip(txt_lens, num_bbs)):
        gi[i, tl:tl + nbb] = torch.arange(max_len, max_len + nbb, dtype=torch.long)
    return gi
