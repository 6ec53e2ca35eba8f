"""UNITER for NLVR2: paired, triplet and paired-attention heads.  Reference: model/nlvr2.py:17-204.

All three share the encoder call and the token-type table growth (2 -> 3 rows, `init_type_embedding`,
model/nlvr2.py:26-34), factored into a base class here; public class names, constructor signatures and
sub-module names (state_dict keys) are the reference's.
"""
from collections import defaultdict

import torch
from torch import nn
from torch.nn import functional as F

from .attention import MultiheadAttention
from .model import UniterModel, UniterPreTrainedModel
from .. import _lib


class _Nlvr2Base(UniterPreTrainedModel):
    def __init__(self, config, img_dim):
        super().__init__(config)
        self.uniter = UniterModel(config, img_dim)

    def init_type_embedding(self):
        """Grow token_type_embeddings to 3 rows: rows 0/1 copied, row 2 starts as a copy of row 1."""
        hidden = self.uniter.config.hidden_size
        old = self.uniter.embeddings.token_type_embeddings.weight.data
        new_emb = nn.Embedding(3, hidden)
        new_emb.apply(self.init_weights)
        new_emb = new_emb.to(device=old.device, dtype=old.dtype)
        new_emb.weight.data[0, :].copy_(old[0, :])
        new_emb.weight.data[1, :].copy_(old[1, :])
        new_emb.weight.data[2, :].copy_(old[1, :])
        self.uniter.embeddings.token_type_embeddings = new_emb

    def _encode(self, batch):
        batch = defaultdict(lambda: None, batch)
        self.uniter.seq_lens_hint = batch['seq_lens']          # optional host-side lengths (packed execution, no sync)
        seq = self.uniter(batch['input_ids'], batch['position_ids'], batch['img_feat'], batch['img_pos_feat'],
                          batch['attn_masks'], batch['gather_index'], output_all_encoded_layers=False,
                          img_type_ids=batch['img_type_ids'])
        return batch, seq

    @staticmethod
    def _finish(answer_scores, batch, compute_loss):
        if not compute_loss:
            return answer_scores
        return F.cross_entropy(answer_scores.float(), batch['targets'], reduction='none')


class UniterForNlvr2Paired(_Nlvr2Base):
    """Concatenate the [CLS] states of the two (text, image_i) rows of a pair -> Linear(2H, 2)."""

    def __init__(self, config, img_dim):
        super().__init__(config, img_dim)
        self.nlvr2_output = nn.Linear(config.hidden_size * 2, 2)
        self.apply(self.init_weights)

    def forward(self, batch, compute_loss=True):
        batch, seq = self._encode(batch)
        pooled = self.uniter.pooler(seq)
        n_pair = pooled.size(0) // 2
        scores = self.nlvr2_output(pooled.contiguous().view(n_pair, -1))
        return self._finish(scores, batch, compute_loss)


class UniterForNlvr2Triplet(_Nlvr2Base):
    """One row per (text, image_left, image_right) triplet -> Linear(H, 2)."""

    def __init__(self, config, img_dim):
        super().__init__(config, img_dim)
        self.nlvr2_output = nn.Linear(config.hidden_size, 2)
        self.apply(self.init_weights)

    def forward(self, batch, compute_loss=True):
        batch, seq = self._encode(batch)
        scores = self.nlvr2_output(self.uniter.pooler(seq))
        return self._finish(scores, batch, compute_loss)


class AttentionPool(nn.Module):
    """Softmax-weighted average over the sequence; scores from Linear(H,1)+ReLU; padded slots get -1e4."""

    def __init__(self, hidden_size, drop=0.0):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(hidden_size, 1), nn.ReLU())
        self.dropout = nn.Dropout(drop)
        self.force_module_path = False          # tests: the plain PyTorch ops below even where the fused kernel applies

    def forward(self, input_, mask=None):
        """input_ [B, T, D], mask [B, T] (True = padding)."""
        if (not self.force_module_path and input_.is_cuda and input_.dtype == torch.bfloat16 and input_.size(1) <= 512 and input_.size(2) <= 1024
                and input_.size(2) % 8 == 0 and self.fc[0].weight.dtype == torch.bfloat16):
            from .. import ops
            return ops.attention_pool(input_, mask, self.fc[0], self.dropout.p, self.training)   # one fused kernel
        if not self.force_module_path:              # (force_module_path is the explicit per-module form of the switch)
            _lib.head_torch_path("AttentionPool", "needs a bf16 CUDA input with T <= 512, D <= 1024, D % 8 == 0")
        score = self.fc(input_).squeeze(-1)
        if mask is not None:
            score = score + mask.to(dtype=input_.dtype) * -1e4
        norm_score = self.dropout(F.softmax(score, dim=1))
        return norm_score.unsqueeze(1).matmul(input_).squeeze(1)


class UniterForNlvr2PairedAttn(_Nlvr2Base):
    """Paired format + bidirectional cross attention between the two images' sequences
    (config/train-nlvr2-base-1gpu.json "model": "paired-attn")."""

    def __init__(self, config, img_dim):
        super().__init__(config, img_dim)
        self.attn1 = MultiheadAttention(config.hidden_size, config.num_attention_heads,
                                        config.attention_probs_dropout_prob)
        self.attn2 = MultiheadAttention(config.hidden_size, config.num_attention_heads,
                                        config.attention_probs_dropout_prob)
        self.fc = nn.Sequential(nn.Linear(2 * config.hidden_size, config.hidden_size), nn.ReLU(),
                                nn.Dropout(config.hidden_dropout_prob))
        self.attn_pool = AttentionPool(config.hidden_size, config.attention_probs_dropout_prob)
        self.nlvr2_output = nn.Linear(2 * config.hidden_size, 2)
        self.three_node_cat = False      # tests: the round-4 form (regroup copy, cross-attention node, torch.cat) instead of the one node
        self.apply(self.init_weights)

    def forward(self, batch, compute_loss=True):
        batch, seq = self._encode(batch)
        bs, tl, d = seq.size()
        n = bs // 2
        # rows 2i / 2i+1 are the left / right image of pair i (model/nlvr2.py:172-176): regroup as [side, pair, L, H]
        fused = self._fused_pair_attention(seq) and d % 64 == 0 and self.fc[0].weight.dtype == torch.bfloat16
        if fused and batch['attn_masks'].dtype == torch.int64 and not self.three_node_cat:
            from .. import ops
            # the shipped path: masks from one kernel; regrouping, both cross attentions and cat([attended, own]) as one autograd
            # node that writes the fc input in place (no torch.cat, no regrouped copy of its own) — ops._PairedCrossAttnCatFn
            pad, partner_bias = ops.nlvr2_pair_masks(batch['attn_masks'])
            cat = ops.paired_cross_attention_cat(seq, partner_bias, self.attn1, self.attn2, self.attn1.dropout, self.training)
            hidden = ops.linear_relu_dropout(cat.view(bs * tl, 2 * d), self.fc[0], self.fc[2].p, self.training).view(bs, tl, d)
            pooled = self.attn_pool(hidden, pad)                                        # [2n, H]
            pooled = pooled.view(2, n, d).transpose(0, 1).reshape(n, 2 * d)
            if compute_loss and self.nlvr2_output.weight.dtype == torch.bfloat16:
                return ops.linear_cross_entropy(pooled, self.nlvr2_output, batch['targets'])
            return self._finish(self.nlvr2_output(pooled), batch, compute_loss)
        xs = seq.contiguous().view(n, 2, tl, d).transpose(0, 1).contiguous()
        if self._fused_pair_attention(seq) and batch['attn_masks'].dtype == torch.int64:
            from .. import ops
            # padding mask (left block, then right block) and the partner's key mask — instance block 0 (left queries) attends
            # to the right sequences and vice versa — from one kernel instead of six small PyTorch launches
            pad, partner_bias = ops.nlvr2_pair_masks(batch['attn_masks'])
            att = ops.paired_cross_attention(xs, None, self.attn1, self.attn2, self.attn1.dropout, self.training,
                                             partner_bias=partner_bias)
        elif self._fused_pair_attention(seq):
            from .. import ops
            valid = batch['attn_masks'].contiguous().view(n, 2, tl).transpose(0, 1)     # [2, n, L]
            pad = (valid == 0).reshape(bs, tl)                                          # left block, then right block
            att = ops.paired_cross_attention(xs, valid.flip(0).reshape(bs, tl), self.attn1, self.attn2,
                                             self.attn1.dropout, self.training)
        else:
            _lib.head_torch_path("NLVR2 paired cross attention", "needs a bf16 CUDA sequence, 64-wide heads, L <= 512")
            valid = batch['attn_masks'].contiguous().view(n, 2, tl).transpose(0, 1)     # [2, n, L]
            pad = (valid == 0).reshape(bs, tl)
            left, right = xs[0].transpose(0, 1), xs[1].transpose(0, 1)                  # (L, N, E) module layout
            l2r, _ = self.attn1(left, right, right, key_padding_mask=pad[n:], need_weights=False)
            r2l, _ = self.attn2(right, left, left, key_padding_mask=pad[:n], need_weights=False)
            att = torch.stack([l2r.transpose(0, 1), r2l.transpose(0, 1)], dim=0)
        # both sides at once: fc(cat([attended, self])) -> masked attention pooling -> cat(left, right) -> classifier
        fused = self._fused_pair_attention(seq) and d % 64 == 0 and self.fc[0].weight.dtype == torch.bfloat16
        cat = torch.cat([att, xs], dim=-1)
        if fused:
            from .. import ops
            # Linear(2H, H) + ReLU + Dropout as one GEMM with a fused epilogue
            hidden = ops.linear_relu_dropout(cat.view(bs * tl, 2 * d), self.fc[0], self.fc[2].p, self.training).view(bs, tl, d)
        else:
            _lib.head_torch_path("NLVR2 fc (Linear + ReLU + Dropout)", "needs the fused pair attention path, hidden size % 64 == 0, bf16 weights")
            hidden = self.fc(cat).view(bs, tl, d)
        pooled = self.attn_pool(hidden, pad)                                            # [2n, H]
        pooled = pooled.view(2, n, d).transpose(0, 1).reshape(n, 2 * d)
        if fused and compute_loss and self.nlvr2_output.weight.dtype == torch.bfloat16:
            return ops.linear_cross_entropy(pooled, self.nlvr2_output, batch['targets'])  # classifier + loss, one kernel
        return self._finish(self.nlvr2_output(pooled), batch, compute_loss)

    def _fused_pair_attention(self, seq):
        """The HIP path covers the shipped configuration (bf16 on the GPU, 64-wide heads, L <= 512 = max_position_embeddings)."""
        return (seq.is_cuda and seq.dtype == torch.bfloat16 and self.attn1.head_dim == 64 and seq.size(1) <= 512
                and self.attn1.in_proj_weight.dtype == torch.bfloat16)
