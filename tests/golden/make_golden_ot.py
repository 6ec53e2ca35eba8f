#!/usr/bin/env python3
"""Generate tests/golden/ot_golden.npz by running the REAL reference's optimal-transport code (model/ot.py @ /root/reference).

Run in the build container only (the GPU box has no /root/reference):   python tests/golden/make_golden_ot.py

`cost_matrix_cosine` (model/ot.py:11-22) and `ipot` (model/ot.py:36-69) are called as they are, with bool padding masks
(the uint8 masks of data/itm.py:138-142 no longer index in current PyTorch).  `optimal_transport_dist` itself cannot run
here: its `trace` helper (model/ot.py:25-33) selects with a uint8 eye mask, which torch 2.x rejects — the four lines
around it (joint pad, masked_fill, lengths, trace(cost @ T)) are therefore executed from this script with
torch.diagonal instead of masked_select; everything numerical comes from the reference functions.
Stored per case: x, y as bf16 bit patterns (the GPU path consumes bf16 encoder outputs), pads, cost, T, dist and the
gradients of dist.sum() w.r.t. x and y.
"""
import importlib
import os
import sys

import numpy as np
import torch

REF = os.environ.get("UNITER_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def bf16_bits(t):
    return t.to(torch.bfloat16).view(torch.int16).numpy().astype(np.uint16)


def main(out_path=None):
    sys.path.insert(0, REF)
    ot = importlib.import_module("model.ot")
    out = {}
    cases = {"small": (4, 12, 9, 64, 0), "base": (3, 60, 36, 768, 1), "long": (2, 128, 50, 128, 2)}
    for name, (B, M, N, D, seed) in cases.items():
        g = torch.Generator().manual_seed(100 + seed)
        x = torch.randn(B, M, D, generator=g).to(torch.bfloat16).float()
        y = (torch.randn(B, N, D, generator=g) * 0.7 + 0.1).to(torch.bfloat16).float()
        txt_pad = torch.zeros(B, M, dtype=torch.bool)
        img_pad = torch.zeros(B, N, dtype=torch.bool)
        for b in range(1, B):                                    # example 0 is full length, the others ragged
            txt_pad[b, int(torch.randint(M // 3, M, (1,), generator=g)):] = True
            img_pad[b, int(torch.randint(N // 3, N, (1,), generator=g)):] = True
        x[txt_pad] = 0                                           # padded slots of the scattered sequence are zeros
        y[img_pad] = 0
        x.requires_grad_(True)
        y.requires_grad_(True)
        cost = ot.cost_matrix_cosine(x, y)                       # reference code
        joint_pad = txt_pad.unsqueeze(-1) | img_pad.unsqueeze(-2)
        cost = cost.masked_fill(joint_pad, 0)
        txt_len = (txt_pad.size(1) - txt_pad.sum(dim=1)).to(cost.dtype)
        img_len = (img_pad.size(1) - img_pad.sum(dim=1)).to(cost.dtype)
        T = ot.ipot(cost.detach(), txt_len, txt_pad, img_len, img_pad, joint_pad, 0.5, 50, 1)   # reference code
        dist = torch.diagonal(cost.matmul(T.detach()), dim1=1, dim2=2).sum(-1)
        dist.sum().backward()
        out[name + "/x_bf16"] = bf16_bits(x.detach())
        out[name + "/y_bf16"] = bf16_bits(y.detach())
        out[name + "/txt_pad"] = txt_pad.numpy()
        out[name + "/img_pad"] = img_pad.numpy()
        out[name + "/cost"] = cost.detach().numpy()
        out[name + "/T"] = T.numpy()
        out[name + "/dist"] = dist.detach().numpy()
        out[name + "/dx"] = x.grad.numpy()
        out[name + "/dy"] = y.grad.numpy()
        print(name, "dist", dist.detach().numpy())
    path = out_path or os.path.join(HERE, "ot_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
