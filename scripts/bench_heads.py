#!/usr/bin/env python3
"""SURVEY.md §8 f-2 measurement: what the pre-training task heads cost on top of the encoder.
Times the MLM head (BertOnlyMLMHead: dense+GELU+LN, tied 28996-way decoder, cross entropy) and the MRC-KL head
(dense+GELU+LN, Linear(1601), KL) forward+backward on the masked rows of a [B, L, H] bf16 encoder output: as plain
PyTorch modules (the reference's op sequence on the same GPU) and through the HIP path (uniter_head_ce_* /
uniter_head_kl_*).  Wall time per forward+backward with the row indices precomputed.  Prints one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.nn import functional as F  # noqa: E402

from uniter_amd.model.model import UniterConfig  # noqa: E402
from uniter_amd.model.pretrain import UniterForPretraining  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def main():
    dev = torch.device("cuda", 0)
    cfg = UniterConfig(28996, num_hidden_layers=1)        # uniter-base widths (config/uniter-base.json); the encoder is not run here
    model = UniterForPretraining(cfg, img_dim=2048, img_label_dim=1601).to(dev).bfloat16()
    model.train()
    B, Lt, H = 32, 60, cfg.hidden_size
    g = torch.Generator().manual_seed(3)
    seq = torch.randn(B, Lt, H, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
    labels = torch.full((B, Lt), -1, dtype=torch.long)
    pick = torch.rand(B, Lt, generator=g) < 0.15
    pick[:, 1] = True
    labels[pick] = torch.randint(1000, 28996, (int(pick.sum()),), generator=g)
    labels = labels.to(dev)
    n = int(pick.sum())
    idx = pick.reshape(-1).nonzero().squeeze(1).to(dev)          # row gather without a host sync inside the timed region
    lab_rows = labels.reshape(-1)[idx]

    def torch_mlm():
        seq.grad = None
        scores = model.cls(seq.view(-1, H).index_select(0, idx))
        F.cross_entropy(scores.float(), lab_rows, reduction='none').mean().backward()

    out = {"what": "pre-training task heads fwd+bwd on %d masked rows (B=32 x 60 text slots, 15%%), UNITER-base, bf16" % n,
           "mlm_pytorch_us": round(timed(torch_mlm, 20), 1)}
    from uniter_amd import ops

    def fused_mlm():
        seq.grad = None
        ops.mlm_head_loss(seq.view(-1, H).index_select(0, idx), lab_rows, model.cls.predictions).mean().backward()
    out["mlm_fused_hip_us"] = round(timed(fused_mlm, 20), 1)
    # MRC-KL head: 15 % of 36 region slots
    ni = 173
    rows = torch.randn(ni, H, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
    soft = torch.softmax(torch.randn(ni, 1601, generator=g), dim=-1).to(dev)
    net = model.region_classifier.net

    def torch_mrc():
        rows.grad = None
        F.kl_div(F.log_softmax(model.region_classifier(rows).float(), dim=-1), soft, reduction='none').mean().backward()

    def fused_mrc():
        rows.grad = None
        ops.head_kl_div(rows, soft, net[0], net[2], net[3].weight, net[3].bias).mean().backward()
    out["mrckl_rows"] = ni
    out["mrckl_pytorch_us"] = round(timed(torch_mrc, 20), 1)
    out["mrckl_fused_hip_us"] = round(timed(fused_mrc, 20), 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
