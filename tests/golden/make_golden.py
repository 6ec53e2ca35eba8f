#!/usr/bin/env python3
"""Generate tests/golden/uniter_tiny.npz by running the REAL reference (ChenRocks/UNITER @ /root/reference).

Run in the build container only (the GPU box has no /root/reference):

    python tests/golden/make_golden.py

The reference's model/*.py and optim/adamw.py are imported unmodified; apex is absent, so
`apex.normalization.fused_layer_norm.FusedLayerNorm` is shimmed with torch.nn.LayerNorm (same math, SURVEY.md
§8c).  A tiny configuration (hidden 128, 2 heads of 64, 2 layers) keeps the fixture small; batches are ragged
(different text / region counts per example) so `gather_index`, padding and masking are exercised.  Stored:
the reference's initial weights, the input batches, per-task un-reduced losses, sequence outputs, gradients
(all of them for MLM, a sample for the other tasks) and the weights / Adam state after one clipped AdamW step.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("UNITER_REFERENCE", "/root/reference")

for name in ('apex', 'apex.normalization', 'apex.normalization.fused_layer_norm'):
    sys.modules[name] = types.ModuleType(name)
sys.modules['apex.normalization.fused_layer_norm'].FusedLayerNorm = torch.nn.LayerNorm
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)

from model.model import UniterConfig  # noqa: E402  (reference)
from model.nlvr2 import UniterForNlvr2PairedAttn  # noqa: E402
from model.pretrain import UniterForPretraining  # noqa: E402
from model.vqa import UniterForVisualQuestionAnswering  # noqa: E402
from optim.adamw import AdamW  # noqa: E402

from uniter_amd.utils.synthetic import make_batch  # noqa: E402  (input generator only)

CFG = dict(vocab_size=96, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128,
           hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
           max_position_embeddings=32, type_vocab_size=2, initializer_range=0.02)
IMG_DIM, LABEL_DIM, N_ANS = 64, 11, 13
BATCH = dict(batch_size=4, max_txt_len=9, num_bb=6, img_dim=IMG_DIM, vocab_size=CFG['vocab_size'], ragged=True,
             img_label_dim=LABEL_DIM, num_answer=N_ANS, min_txt_len=4, min_bb=2, mask_prob=0.3)


def no_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, 'dropout') and isinstance(m.dropout, float):
            m.dropout = 0.0


def randomise(model, gen):
    """init_weights leaves biases 0 and LayerNorm at (1,0): perturb everything so every term is exercised."""
    with torch.no_grad():
        for n, p in model.named_parameters():
            if 'LayerNorm.weight' in n or 'layer_norm.weight' in n or n.endswith('net.2.weight') or n.endswith('vqa_output.2.weight'):
                p.add_(torch.randn(p.shape, generator=gen) * 0.1)
            elif p.dim() == 1:
                p.add_(torch.randn(p.shape, generator=gen) * 0.05)
            else:
                p.mul_(4.0)          # std 0.08: large enough that attention / GELU are not in their linear regime
        snap_bf16(model)


def snap_bf16(model):
    """Round every parameter to a bf16-representable value: the fixture stores weights as 2-byte bf16 patterns and the
    bf16 GPU path starts from EXACTLY the weights the fp32 reference used."""
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.to(torch.bfloat16).float())


def bf16_bits(t):
    return t.detach().to(torch.bfloat16).view(torch.int16).numpy().copy()


def store_batch(out, prefix, batch):
    """Only the tensor entries of a batch are fixture data: `seq_lens` (a host-side python list, a hint for the packed
    encoder path) and `ot_inputs` (None / a dict, covered by make_golden_ot.py) are derived from them and skipped."""
    for k, v in batch.items():
        if isinstance(v, torch.Tensor):
            out["%s/%s" % (prefix, k)] = v.numpy()


def main(out_path=None):
    torch.manual_seed(1234)
    gen = torch.Generator().manual_seed(4321)
    cfg_path = os.path.join(os.path.dirname(os.path.abspath(out_path)) if out_path else HERE, "uniter_tiny_config.json")
    with open(cfg_path, "w") as f:
        json.dump(CFG, f, indent=2, sort_keys=True)
    out = {"config_json": np.array(json.dumps(CFG))}

    pre = UniterForPretraining.from_pretrained(cfg_path, {}, img_dim=IMG_DIM, img_label_dim=LABEL_DIM)
    randomise(pre, gen)
    no_dropout(pre)
    pre.train()
    sd0 = {k: v.detach().clone() for k, v in pre.state_dict().items()}
    for k, v in sd0.items():
        if k != 'cls.predictions.decoder.weight':          # tied to word_embeddings.weight
            out["pre/" + k] = bf16_bits(v)

    tasks = [('mlm', 11), ('mrfr', 12), ('mrckl', 13), ('mrc', 14), ('itm', 15)]
    for task, seed in tasks:
        batch = make_batch(task, seed=seed, **BATCH)
        store_batch(out, "batch_%s" % task, batch)
        pre.zero_grad()
        rb = dict(batch)
        if 'img_masks' in rb:
            rb['img_mask_tgt'] = rb['img_mask_tgt']
        loss = pre(rb, task=task, compute_loss=True)
        if task == 'itm':
            loss = loss[0]
        seq = pre.uniter(rb['input_ids'], rb['position_ids'], rb['img_feat'], rb['img_pos_feat'], rb['attn_masks'],
                         rb['gather_index'], output_all_encoded_layers=False, img_masks=rb.get('img_masks'))
        out["out_%s/loss" % task] = loss.detach().numpy()
        out["out_%s/seq" % task] = seq.detach().numpy()
        loss.mean().backward()
        grads = {n: p.grad.detach().clone() for n, p in pre.named_parameters() if p.grad is not None}
        if task == 'mlm':
            for n, g in grads.items():
                out["grad_mlm/" + n] = g.numpy()
            # one clipped AdamW step with the reference optimizer (optim/adamw.py, optim/misc.py grouping)
            no_decay = ['bias', 'LayerNorm.bias', 'LayerNorm.weight']
            named = list(pre.named_parameters())
            groups = [{'params': [p for n, p in named if not any(nd in n for nd in no_decay)], 'weight_decay': 0.01},
                      {'params': [p for n, p in named if any(nd in n for nd in no_decay)], 'weight_decay': 0.0}]
            opt = AdamW(groups, lr=1e-3, betas=(0.9, 0.98))
            total = torch.nn.utils.clip_grad_norm_([p for _, p in named], 0.5)
            out["adamw/grad_norm"] = np.array(float(total))
            for _ in range(2):          # two steps with the same gradients: exercises step-dependent bias correction
                opt.step()
            keep = ('query.weight', 'key.bias', 'output.LayerNorm.weight', 'word_embeddings.weight', 'img_linear.weight',
                    'pos_linear.weight', 'intermediate.dense.bias', 'cls.predictions.bias', 'img_layer_norm.weight')
            for n, p in named:
                if n.endswith(keep):
                    out["adamw/p/" + n] = p.detach().numpy().copy()
            for n, p in named:
                if p in opt.state and 'exp_avg_sq' in opt.state[p]:
                    if n.endswith('query.weight') or n.endswith('LayerNorm.weight') or 'word_embeddings' in n:
                        out["adamw/v/" + n] = opt.state[p]['exp_avg_sq'].numpy().copy()
            pre.load_state_dict(sd0)
        else:
            for n in ('uniter.encoder.layer.0.attention.self.query.weight',
                      'uniter.img_embeddings.img_linear.weight',
                      'uniter.img_embeddings.pos_linear.weight', 'uniter.img_embeddings.mask_embedding.weight',
                      'uniter.embeddings.token_type_embeddings.weight', 'uniter.embeddings.position_embeddings.weight',
                      'uniter.img_embeddings.LayerNorm.weight', 'uniter.encoder.layer.0.attention.self.key.bias'):
                if n in grads:
                    out["grad_%s/%s" % (task, n)] = grads[n].numpy()

    # ---- VQA ----
    vqa = UniterForVisualQuestionAnswering.from_pretrained(cfg_path, {k: v.clone() for k, v in sd0.items()},
                                                           img_dim=IMG_DIM, num_answer=N_ANS)
    randomise(vqa.vqa_output, gen)
    no_dropout(vqa)
    vqa.train()
    for k, v in vqa.state_dict().items():
        if k.startswith('vqa_output.'):
            out["vqa/" + k] = bf16_bits(v)
    batch = make_batch('vqa', seed=21, **BATCH)
    store_batch(out, "batch_vqa", batch)
    loss = vqa(batch, compute_loss=True)
    out["out_vqa/loss"] = loss.detach().numpy()
    (loss.mean() * N_ANS).backward()
    out["grad_vqa/vqa_output.3.weight"] = vqa.vqa_output[3].weight.grad.numpy().copy()
    out["grad_vqa/uniter.encoder.layer.0.attention.self.query.weight"] = \
        vqa.uniter.encoder.layer[0].attention.self.query.weight.grad.numpy().copy()

    # ---- NLVR2 paired-attention ----
    nl = UniterForNlvr2PairedAttn.from_pretrained(cfg_path, {k: v.clone() for k, v in sd0.items()}, img_dim=IMG_DIM)
    nl.init_type_embedding()
    with torch.no_grad():
        nl.uniter.embeddings.token_type_embeddings.weight[2].add_(torch.randn(CFG['hidden_size'], generator=gen) * 0.05)
        for n, p in nl.named_parameters():
            if not n.startswith('uniter.'):
                if p.dim() == 1:
                    p.add_(torch.randn(p.shape, generator=gen) * 0.05)
                elif 'in_proj' not in n:
                    p.mul_(4.0)
    snap_bf16(nl)
    no_dropout(nl)
    nl.train()
    for k, v in nl.state_dict().items():
        if not k.startswith('uniter.') or 'token_type_embeddings' in k:
            out["nlvr2/" + k] = bf16_bits(v)
    batch = make_batch('nlvr2', seed=31, **BATCH)
    store_batch(out, "batch_nlvr2", batch)
    loss = nl(batch, compute_loss=True)
    out["out_nlvr2/loss"] = loss.detach().numpy()
    loss.mean().backward()
    out["grad_nlvr2/attn1.in_proj_bias"] = nl.attn1.in_proj_bias.grad.numpy().copy()
    out["grad_nlvr2/attn2.out_proj.weight"] = nl.attn2.out_proj.weight.grad.numpy().copy()
    out["grad_nlvr2/uniter.embeddings.token_type_embeddings.weight"] = \
        nl.uniter.embeddings.token_type_embeddings.weight.grad.numpy().copy()
    out["grad_nlvr2/uniter.encoder.layer.1.intermediate.dense.weight"] = \
        nl.uniter.encoder.layer[1].intermediate.dense.weight.grad.numpy().copy()

    path = out_path or os.path.join(HERE, "uniter_tiny.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.2f MB, %d arrays)" % (path, os.path.getsize(path) / 1e6, len(out)))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
