#!/usr/bin/env python3
"""SURVEY.md §8 f-1 measurement: the fused IPOT kernel (uniter_ot_fwd / _bwd) against the PyTorch module path
(uniter_amd/model/ot.py — the reference's op sequence, fp32, on the same GPU) at the pre-training shape
B=32, 60 text + 36 region slots, H=768.  Prints one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from uniter_amd import ops  # noqa: E402
from uniter_amd.model.ot import optimal_transport_dist as torch_ot  # noqa: E402
from uniter_amd.utils.synthetic import make_batch  # noqa: E402


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
    B, tl, il, H = 32, 60, 36, 768
    batch = make_batch('itm', B, tl, il, seed=5, ragged=True, with_ot=True)
    ot = batch['ot_inputs']
    L = batch['attn_masks'].size(1)
    seq = torch.randn(B, L, H, generator=torch.Generator().manual_seed(1)).to(dev, torch.bfloat16).requires_grad_(True)
    sc, tp, ip = ot['ot_scatter'].to(dev), ot['txt_pad'].to(dev), ot['img_pad'].to(dev)

    def fused():
        seq.grad = None
        ops.optimal_transport_dist(seq, sc, tp, ip).sum().backward()

    def module_path():
        seq.grad = None
        max_l = max(ot['scatter_max'] + 1, tl + il)
        ctx = torch.zeros(B, max_l, H, dtype=seq.dtype, device=dev).scatter(1, sc.unsqueeze(-1).expand_as(seq), seq)
        torch_ot(ctx[:, :tl].float(), ctx[:, tl:tl + il].float(), tp, ip).sum().backward()

    d_f = ops.optimal_transport_dist(seq, sc, tp, ip).detach()
    max_l = max(ot['scatter_max'] + 1, tl + il)
    ctx = torch.zeros(B, max_l, H, dtype=seq.dtype, device=dev).scatter(1, sc.unsqueeze(-1).expand_as(seq), seq)
    d_t = torch_ot(ctx[:, :tl].float(), ctx[:, tl:tl + il].float(), tp, ip).detach()
    def fused_fwd():
        with torch.no_grad():
            ops.optimal_transport_dist(seq, sc, tp, ip)

    us_f, us_t, us_ff = timed(fused, 50), timed(module_path, 10), timed(fused_fwd, 50)
    print(json.dumps({"what": "IPOT optimal-transport distance fwd+bwd, B=32 x (60 text + 36 region slots) x H=768, 50 iterations",
                      "fused_hip_us": round(us_f, 1), "fused_hip_fwd_only_us": round(us_ff, 1), "pytorch_module_path_us": round(us_t, 1),
                      "speedup": round(us_t / us_f, 1),
                      "max_abs_dist_diff_vs_pytorch_fp32": float((d_f - d_t).abs().max())}))


if __name__ == "__main__":
    main()
