"""Multi-head cross attention used by the NLVR2 paired-attention head.

Reference: model/attention.py:13-402 (a vendored copy of torch.nn.MultiheadAttention, (L, N, E) layout).
This is a head-side module, so it stays in PyTorch; it is re-implemented compactly with the
same parameter names (`in_proj_weight [3E,E]`, `in_proj_bias`, `out_proj.{weight,bias}`), the same
semantics (q pre-scaled by head_dim**-0.5, `key_padding_mask` -> -inf, dropout on the probabilities, returned
weights averaged over heads) and the same call signature.  Unlike the vendored copy it does not compare
`query`/`key`/`value` with torch.equal (a device sync, model/attention.py:86-87): object identity decides
whether projections can share a GEMM.
"""
import torch
from torch import nn
from torch.nn import functional as F
from torch.nn.init import constant_, xavier_uniform_


def multi_head_attention_forward(query, key, value, embed_dim_to_check, num_heads, in_proj_weight, in_proj_bias,
                                 dropout_p, out_proj_weight, out_proj_bias, training=True,
                                 key_padding_mask=None, need_weights=True, attn_mask=None):
    tgt_len, bsz, embed_dim = query.size()
    if embed_dim != embed_dim_to_check:
        raise ValueError("embedding dimension mismatch")
    if key.size() != value.size():
        raise ValueError("key and value must have the same shape")
    head_dim = embed_dim // num_heads
    if head_dim * num_heads != embed_dim:
        raise ValueError("embed_dim must be divisible by num_heads")
    scaling = float(head_dim) ** -0.5

    def proj(x, lo, hi):
        b = None if in_proj_bias is None else in_proj_bias[lo:hi]
        return F.linear(x, in_proj_weight[lo:hi, :], b)

    if query is key and key is value:
        q, k, v = proj(query, 0, 3 * embed_dim).chunk(3, dim=-1)
    elif key is value:
        q = proj(query, 0, embed_dim)
        k, v = proj(key, embed_dim, 3 * embed_dim).chunk(2, dim=-1)
    else:
        q = proj(query, 0, embed_dim)
        k = proj(key, embed_dim, 2 * embed_dim)
        v = proj(value, 2 * embed_dim, 3 * embed_dim)
    q = q * scaling

    src_len = k.size(0)
    # (L, N, E) -> (N*heads, L, head_dim)
    q = q.contiguous().view(tgt_len, bsz * num_heads, head_dim).transpose(0, 1)
    k = k.contiguous().view(src_len, bsz * num_heads, head_dim).transpose(0, 1)
    v = v.contiguous().view(src_len, bsz * num_heads, head_dim).transpose(0, 1)

    weights = torch.bmm(q, k.transpose(1, 2))                       # [N*heads, L, S]
    if attn_mask is not None:
        weights = weights + attn_mask.unsqueeze(0)
    if key_padding_mask is not None:
        if key_padding_mask.size(0) != bsz or key_padding_mask.size(1) != src_len:
            raise ValueError("key_padding_mask must be [N, S]")
        weights = weights.view(bsz, num_heads, tgt_len, src_len)
        weights = weights.masked_fill(key_padding_mask.bool().unsqueeze(1).unsqueeze(2), float('-inf'))
        weights = weights.view(bsz * num_heads, tgt_len, src_len)
    weights = F.softmax(weights, dim=-1)
    weights = F.dropout(weights, p=dropout_p, training=training)

    out = torch.bmm(weights, v)                                     # [N*heads, L, head_dim]
    out = out.transpose(0, 1).contiguous().view(tgt_len, bsz, embed_dim)
    out = F.linear(out, out_proj_weight, out_proj_bias)
    if need_weights:
        weights = weights.view(bsz, num_heads, tgt_len, src_len)
        return out, weights.sum(dim=1) / num_heads
    return out, None


class MultiheadAttention(nn.Module):
    """attn = MultiheadAttention(embed_dim, num_heads, dropout); out, w = attn(query, key, value, key_padding_mask=...)."""

    def __init__(self, embed_dim, num_heads, dropout=0., bias=True, add_bias_kv=False, add_zero_attn=False,
                 kdim=None, vdim=None):
        super(MultiheadAttention, self).__init__()
        if add_bias_kv or add_zero_attn or (kdim not in (None, embed_dim)) or (vdim not in (None, embed_dim)):
            raise NotImplementedError("only the configuration UNITER uses is implemented "
                                      "(kdim = vdim = embed_dim, no bias_kv / zero_attn)")
        self.embed_dim = embed_dim
        self.kdim = embed_dim
        self.vdim = embed_dim
        self._qkv_same_embed_dim = True
        self.num_heads = num_heads
        self.dropout = dropout
        self.head_dim = embed_dim // num_heads
        if self.head_dim * num_heads != self.embed_dim:
            raise ValueError("embed_dim must be divisible by num_heads")
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        if bias:
            self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim))
        else:
            self.register_parameter('in_proj_bias', None)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.bias_k = self.bias_v = None
        self.add_zero_attn = False
        self._reset_parameters()

    def _reset_parameters(self):
        xavier_uniform_(self.in_proj_weight)
        if self.in_proj_bias is not None:
            constant_(self.in_proj_bias, 0.)
            constant_(self.out_proj.bias, 0.)

    def forward(self, query, key, value, key_padding_mask=None, need_weights=True, attn_mask=None):
        return multi_head_attention_forward(
            query, key, value, self.embed_dim, self.num_heads, self.in_proj_weight, self.in_proj_bias,
            self.dropout, self.out_proj.weight, self.out_proj.bias, training=self.training,
            key_padding_mask=key_padding_mask, need_weights=need_weights, attn_mask=attn_mask)
