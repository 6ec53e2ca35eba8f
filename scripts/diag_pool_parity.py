#!/usr/bin/env python3
"""Where the error of AttentionPool's Linear(H, 1) gradients (attn_pool.fc.0.{weight,bias}) of the headline step comes from.

The two tensors sit at 0.10-0.13 relative L2 against the fp32 oracle (head bound 0.10; torch-bf16 yardstick 0.05-0.09).  The pool
kernels themselves work in fp32 on their bf16 inputs (pool.hip), so the error has to arrive with the inputs: x (the fc output the
pool reads) or dout (the gradient of the pooled vector).  This script separates the two with the fp32 formula of the pool's
backward: oracle inputs with one of them replaced by ours / by the torch-bf16 yardstick's.

    python scripts/diag_pool_parity.py        (GPU; ~1 min of CPU for the oracle)
"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import uniter_oracle as O
from tests.common import rel_l2
from uniter_amd.train import WORKLOADS, build_model
from uniter_amd.utils.misc import set_dropout
from uniter_amd.utils.synthetic import make_batch, to_device

dev = torch.device("cuda", 0)
w = WORKLOADS['c2']
cfg = w['cfg']
model = build_model('nlvr2', cfg, torch.device('cpu'), 77, "/tmp/diag_pool_base.json").float()
with torch.no_grad():
    for p in model.parameters():
        p.copy_(p.to(torch.bfloat16).float())
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
batch = make_batch('nlvr2', w['batch'], w['max_txt_len'], w['num_bb'], seed=1000)

# ---- oracle ----
leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
taps = {}
ref_loss, _ = O.nlvr2_paired_attn_loss(leaf, cfg, batch, taps)
taps['pooled'].retain_grad()
for x in taps['pool_in']:
    x.retain_grad()
ref_loss.mean().backward()
n = w['batch'] // 2
H = cfg['hidden_size']
x_ref = torch.cat([x.detach() for x in taps['pool_in']], 0)                # [2n, L, H] left block, right block
g_ref = taps['pooled'].grad.detach().view(n, 2, H).transpose(0, 1).reshape(2 * n, H)   # same order
pad = torch.cat(taps['pool_pad'], 0)
gw_ref, gb_ref = leaf['attn_pool.fc.0.weight'].grad, leaf['attn_pool.fc.0.bias'].grad


def formula(x, g):
    W = sd['attn_pool.fc.0.weight'].clone().requires_grad_(True)
    b = sd['attn_pool.fc.0.bias'].clone().requires_grad_(True)
    score = torch.relu(torch.nn.functional.linear(x, W, b)).squeeze(-1) + pad.float() * -1e4
    out = torch.softmax(score, dim=1).unsqueeze(1).matmul(x).squeeze(1)
    (out * g).sum().backward()
    return W.grad, b.grad


fw, fb = formula(x_ref, g_ref)
print("formula on oracle inputs vs oracle autograd: w %.2e b %.2e (self-check)" % (rel_l2(fw, gw_ref), rel_l2(fb, gb_ref)))

# ---- ours ----
m = copy.deepcopy(model).to(dev).bfloat16()
set_dropout(m, 0.0)
for mod in m.modules():
    if hasattr(mod, 'dropout') and isinstance(mod.dropout, float):
        mod.dropout = 0.0
m.train()
d = to_device(batch, dev)
d['img_feat'] = d['img_feat'].to(torch.bfloat16)
d['img_pos_feat'] = d['img_pos_feat'].to(torch.bfloat16)
seen = {}


def fwd_hook(mod, args, out):
    seen['x'] = args[0].detach().float().cpu()
    out.register_hook(lambda g: seen.__setitem__('g', g.detach().float().cpu()))


h = m.attn_pool.register_forward_hook(fwd_hook)
loss = m(d, compute_loss=True)
loss.mean().backward()
torch.cuda.synchronize()
h.remove()
named = dict(m.named_parameters())
gw_o, gb_o = named['attn_pool.fc.0.weight'].grad.float().cpu(), named['attn_pool.fc.0.bias'].grad.float().cpu()
x_o, g_o = seen['x'], seen['g']

# ---- torch-bf16 yardstick ----
sdb = {k: v.detach().to(dev, torch.bfloat16).requires_grad_(True) for k, v in sd.items()}
bb = {k: ((v.to(dev, torch.bfloat16) if v.is_floating_point() else v.to(dev)) if torch.is_tensor(v) else v) for k, v in batch.items()}
ytaps = {}
yl, _ = O.nlvr2_paired_attn_loss(sdb, cfg, bb, ytaps)
ytaps['pooled'].retain_grad()
yl.float().mean().backward()
x_y = torch.cat([x.detach().float().cpu() for x in ytaps['pool_in']], 0)
g_y = ytaps['pooled'].grad.detach().float().cpu().view(n, 2, H).transpose(0, 1).reshape(2 * n, H)
gw_y, gb_y = sdb['attn_pool.fc.0.weight'].grad.float().cpu(), sdb['attn_pool.fc.0.bias'].grad.float().cpu()

print("pool input x   rel-L2 vs oracle: ours %.3e   torch-bf16 %.3e" % (rel_l2(x_o, x_ref), rel_l2(x_y, x_ref)))
print("pooled grad g  rel-L2 vs oracle: ours %.3e   torch-bf16 %.3e" % (rel_l2(g_o, g_ref), rel_l2(g_y, g_ref)))
print("actual gradients vs oracle:      ours w %.4f b %.4f   torch-bf16 w %.4f b %.4f" % (
    rel_l2(gw_o, gw_ref), rel_l2(gb_o, gb_ref), rel_l2(gw_y, gw_ref), rel_l2(gb_y, gb_ref)))
for label, xs, gs in (("x ours, g oracle", x_o, g_ref), ("x oracle, g ours", x_ref, g_o), ("x ours, g ours", x_o, g_o),
                      ("x bf16(oracle), g oracle", x_ref.to(torch.bfloat16).float(), g_ref),
                      ("x yard, g oracle", x_y, g_ref), ("x oracle, g yard", x_ref, g_y), ("x yard, g yard", x_y, g_y)):
    a, b_ = formula(xs, gs)
    print("fp32 formula [%-26s] vs oracle: w %.4f b %.4f" % (label, rel_l2(a, gw_ref), rel_l2(b_, gb_ref)))
a, b_ = formula(x_o, g_o)
print("our kernel vs the fp32 formula on OUR inputs: w %.4f b %.4f   (the kernel's own arithmetic)" % (rel_l2(gw_o, a), rel_l2(gb_o, b_)))
print("values: db oracle %.6e ours %.6e yard %.6e ; |dW| oracle %.4e" % (float(gb_ref), float(gb_o), float(gb_y), float(gw_ref.norm())))
