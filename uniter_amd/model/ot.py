"""Word-region alignment loss: IPOT approximation of the optimal-transport distance (reference model/ot.py:11-85).

Written against bool padding masks (the reference's uint8 masks no longer work in current PyTorch,
SURVEY.md §8c) and kept in PyTorch: it is a "next" row of the scope table (§8f-1), not part of the
encoder hot path.  The transport plan is computed without gradient; the distance trace(C · T)
back-propagates through the cosine cost only, as in the reference.
"""
import torch
from torch.nn import functional as F


def cost_matrix_cosine(x, y, eps=1e-5):
    """Pairwise cosine distance, [B, Lx, D] x [B, Ly, D] -> [B, Lx, Ly]."""
    if x.dim() != y.dim() or x.size(0) != y.size(0) or x.size(2) != y.size(2):
        raise ValueError("incompatible shapes %s / %s" % (tuple(x.shape), tuple(y.shape)))
    xn = F.normalize(x, p=2, dim=-1, eps=eps)
    yn = F.normalize(y, p=2, dim=-1, eps=eps)
    return 1 - xn.matmul(yn.transpose(1, 2))


def trace(x):
    """Batched trace of square matrices [B, n, n] -> [B]."""
    if x.size(1) != x.size(2):
        raise ValueError("trace needs square matrices")
    return torch.diagonal(x, dim1=1, dim2=2).sum(dim=-1)


@torch.no_grad()
def ipot(C, x_len, x_pad, y_len, y_pad, joint_pad, beta, iteration, k):
    """Inexact proximal point OT.  C [B, M, N]; x_pad [B, M], y_pad [B, N], joint_pad [B, M, N] are bool."""
    b, m, n = C.size()
    sigma = torch.ones(b, m, dtype=C.dtype, device=C.device) / x_len.unsqueeze(1)
    T = torch.ones(b, n, m, dtype=C.dtype, device=C.device)
    A = torch.exp(-C.transpose(1, 2) / beta)

    sigma = sigma.masked_fill(x_pad, 0)
    jp = joint_pad.transpose(1, 2)
    T = T.masked_fill(jp, 0)
    A = A.masked_fill(jp, 0)

    x_len = x_len.unsqueeze(1).unsqueeze(2)
    y_len = y_len.unsqueeze(1).unsqueeze(2)
    # large additive constants keep padded entries of delta / sigma at ~0
    x_mask = (x_pad.to(C.dtype) * 1e4).unsqueeze(1)
    y_mask = (y_pad.to(C.dtype) * 1e4).unsqueeze(1)

    delta = None
    for _ in range(iteration):
        Q = A * T                                   # [B, N, M]
        sigma = sigma.view(b, m, 1)
        for _ in range(k):
            delta = 1 / (y_len * Q.matmul(sigma).view(b, 1, n) + y_mask)
            sigma = 1 / (x_len * delta.matmul(Q) + x_mask)
        T = delta.view(b, n, 1) * Q * sigma
    return T.masked_fill(jp, 0)


def optimal_transport_dist(txt_emb, img_emb, txt_pad, img_pad, beta=0.5, iteration=50, k=1):
    """[B, M, D], [B, N, D], [B, M], [B, N] -> [B] transport distance."""
    txt_pad = txt_pad.bool()
    img_pad = img_pad.bool()
    cost = cost_matrix_cosine(txt_emb, img_emb)
    joint_pad = txt_pad.unsqueeze(-1) | img_pad.unsqueeze(-2)
    cost = cost.masked_fill(joint_pad, 0)

    txt_len = (txt_pad.size(1) - txt_pad.sum(dim=1)).to(dtype=cost.dtype)
    img_len = (img_pad.size(1) - img_pad.sum(dim=1)).to(dtype=cost.dtype)

    T = ipot(cost.detach(), txt_len, txt_pad, img_len, img_pad, joint_pad, beta, iteration, k)
    return trace(cost.matmul(T.detach()))
