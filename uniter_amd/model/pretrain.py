"""UNITER pre-training heads: MLM, MRFR, MRC / MRC-KL, ITM (+OT).  Reference: model/pretrain.py:19-229.

The heads stay ordinary PyTorch modules ("plug in unchanged"): they consume the
autograd-tracked [B, L, H] output of the HIP encoder.  Class names, constructor signatures, sub-module
names (state_dict keys) and the `forward(batch, task, compute_loss)` contract follow the reference;
losses come back un-reduced.  Tied weights: the MLM decoder IS `word_embeddings.weight`
(model/pretrain.py:55-56) and the MRFR output projection IS `img_linear.weight` transposed (:57-59).
"""
from collections import defaultdict

import torch
from torch import nn
from torch.nn import functional as F

from .layer import GELU, BertLayerNorm as LayerNorm, BertOnlyMLMHead, gelu
from .model import UniterModel, UniterPreTrainedModel
from .. import _lib


class RegionFeatureRegression(nn.Module):
    """MRFR head: hidden -> Linear+GELU+LN -> tied projection back to the region-feature space."""

    def __init__(self, hidden_size, feat_dim, img_linear_weight):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(hidden_size, hidden_size), GELU(), LayerNorm(hidden_size, eps=1e-12))
        self.weight = img_linear_weight            # [hidden, feat_dim], used transposed
        self.bias = nn.Parameter(torch.zeros(feat_dim))

    def forward(self, input_):
        return F.linear(self.net(input_), self.weight.t(), self.bias)


class RegionClassification(nn.Module):
    """MRC(-KL) head: hidden -> Linear+GELU+LN -> Linear(label_dim)."""

    def __init__(self, hidden_size, label_dim):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(hidden_size, hidden_size), GELU(), LayerNorm(hidden_size, eps=1e-12),
                                 nn.Linear(hidden_size, label_dim))

    def forward(self, input_):
        return self.net(input_)


def _rows_where(hidden, mask):
    """Rows of hidden [B, L, H] selected by the boolean mask [B, L] -> [n, H] (only masked positions go through
    the heads, model/pretrain.py:129-133)."""
    return hidden[mask.bool()]


class UniterForPretraining(UniterPreTrainedModel):
    """UNITER with all pre-training heads."""

    def __init__(self, config, img_dim, img_label_dim):
        super().__init__(config)
        self.uniter = UniterModel(config, img_dim)
        self.cls = BertOnlyMLMHead(config, self.uniter.embeddings.word_embeddings.weight)
        self.feat_regress = RegionFeatureRegression(config.hidden_size, img_dim,
                                                    self.uniter.img_embeddings.img_linear.weight)
        self.region_classifier = RegionClassification(config.hidden_size, img_label_dim)
        self.itm_output = nn.Linear(config.hidden_size, 2)
        self.apply(self.init_weights)

    def _encode(self, batch, img_masks=None):
        return self.uniter(batch['input_ids'], batch['position_ids'], batch['img_feat'], batch['img_pos_feat'],
                           batch['attn_masks'], batch['gather_index'], output_all_encoded_layers=False,
                           img_masks=img_masks)

    def forward(self, batch, task, compute_loss=True):
        batch = defaultdict(lambda: None, batch)
        self.uniter.seq_lens_hint = batch['seq_lens']          # optional host-side lengths (packed execution, no sync)
        if task == 'mlm':
            return self.forward_mlm(batch['input_ids'], batch['position_ids'], batch['img_feat'],
                                    batch['img_pos_feat'], batch['attn_masks'], batch['gather_index'],
                                    batch['txt_labels'], compute_loss)
        if task == 'mrfr':
            return self.forward_mrfr(batch['input_ids'], batch['position_ids'], batch['img_feat'],
                                     batch['img_pos_feat'], batch['attn_masks'], batch['gather_index'],
                                     batch['img_masks'], batch['img_mask_tgt'], batch['feat_targets'], compute_loss)
        if task == 'itm':
            return self.forward_itm(batch['input_ids'], batch['position_ids'], batch['img_feat'],
                                    batch['img_pos_feat'], batch['attn_masks'], batch['gather_index'],
                                    batch['targets'], batch['ot_inputs'], compute_loss)
        if task.startswith('mrc'):
            return self.forward_mrc(batch['input_ids'], batch['position_ids'], batch['img_feat'],
                                    batch['img_pos_feat'], batch['attn_masks'], batch['gather_index'],
                                    batch['img_masks'], batch['img_mask_tgt'], batch['label_targets'], task,
                                    compute_loss)
        raise ValueError('invalid task')

    def forward_mlm(self, input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index,
                    txt_labels, compute_loss=True):
        sequence_output = self.uniter(input_ids, position_ids, img_feat, img_pos_feat, attention_mask,
                                      gather_index, output_all_encoded_layers=False)
        txt_part = sequence_output[:, :input_ids.size(1), :]
        picked = txt_labels != -1
        rows = _rows_where(txt_part, picked)
        if compute_loss and self._fused_heads(rows):
            from .. import ops
            # transform + tied 28996-way decoder + cross entropy without an fp32 logits tensor (ops._HeadCeFn)
            return ops.mlm_head_loss(rows, txt_labels[picked], self.cls.predictions)
        if compute_loss:
            _lib.head_torch_path("MLM head loss", "needs bf16 CUDA rows, erf-GELU transform, hidden size % 64 == 0")
        prediction_scores = self.cls(rows)           # (scores for inference: the reference's modules, as they are)
        if not compute_loss:
            return prediction_scores
        return F.cross_entropy(prediction_scores.float(), txt_labels[picked], reduction='none')

    def _fused_heads(self, rows):
        """The HIP head path covers the shipped configuration: bf16 on the GPU, erf-GELU transform."""
        tr = self.cls.predictions.transform
        return (rows.is_cuda and rows.dtype == torch.bfloat16 and tr.transform_act_fn is gelu
                and tr.dense.weight.dtype == torch.bfloat16 and rows.size(1) % 64 == 0)

    def _compute_masked_hidden(self, hidden, mask):
        return _rows_where(hidden, mask)

    def forward_mrfr(self, input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index,
                     img_masks, img_mask_tgt, feat_targets, compute_loss=True):
        sequence_output = self.uniter(input_ids, position_ids, img_feat, img_pos_feat, attention_mask,
                                      gather_index, output_all_encoded_layers=False, img_masks=img_masks)
        prediction_feat = self.feat_regress(_rows_where(sequence_output, img_mask_tgt))
        if not compute_loss:
            return prediction_feat
        return F.mse_loss(prediction_feat.float(), feat_targets.float(), reduction='none')

    def forward_itm(self, input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index,
                    targets, ot_inputs, compute_loss=True):
        sequence_output = self.uniter(input_ids, position_ids, img_feat, img_pos_feat, attention_mask,
                                      gather_index, output_all_encoded_layers=False)
        itm_scores = self.itm_output(self.uniter.pooler(sequence_output))

        ot_loss = None
        if ot_inputs is not None:
            tl, il = input_ids.size(1), img_feat.size(1)
            if sequence_output.is_cuda and sequence_output.dtype == torch.bfloat16 and self.config.hidden_size <= 1024:
                # fused HIP path: cosine cost, 50 IPOT iterations and trace(C T) in one workgroup per example, reading
                # the compact sequence through ot_scatter (no un-compaction copy); distances stay fp32
                from .. import ops
                ot_dist = ops.optimal_transport_dist(sequence_output, ot_inputs['ot_scatter'], ot_inputs['txt_pad'],
                                                     ot_inputs['img_pad'])
            else:
                _lib.head_torch_path("optimal-transport (IPOT) loss", "needs a bf16 CUDA sequence and hidden size <= 1024")
                from .ot import optimal_transport_dist
                # undo the compaction: scatter the joint sequence back to [txt(max_tl) ; img] slots
                b = sequence_output.size(0)
                max_l = max(ot_inputs['scatter_max'] + 1, tl + il)
                index = ot_inputs['ot_scatter'].unsqueeze(-1).expand_as(sequence_output)
                ctx_emb = torch.zeros(b, max_l, self.config.hidden_size, dtype=sequence_output.dtype,
                                      device=sequence_output.device).scatter_(dim=1, index=index, src=sequence_output)
                txt_emb, img_emb = ctx_emb[:, :tl, :], ctx_emb[:, tl:tl + il, :]
                # fp32 for stability, as the reference does
                ot_dist = optimal_transport_dist(txt_emb.float(), img_emb.float(), ot_inputs['txt_pad'],
                                                 ot_inputs['img_pad']).to(txt_emb)
            ot_loss = (ot_dist.masked_select(targets == 1), ot_dist.masked_select(targets == 0))

        if not compute_loss:
            return itm_scores, ot_loss
        return F.cross_entropy(itm_scores.float(), targets, reduction='none'), ot_loss

    def forward_mrc(self, input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index,
                    img_masks, img_mask_tgt, label_targets, task, compute_loss=True):
        sequence_output = self.uniter(input_ids, position_ids, img_feat, img_pos_feat, attention_mask,
                                      gather_index, output_all_encoded_layers=False, img_masks=img_masks)
        rows = _rows_where(sequence_output, img_mask_tgt)
        net = self.region_classifier.net
        if compute_loss and self._fused_heads(rows) and net[3].weight.dtype == torch.bfloat16:
            from .. import ops
            # dense+GELU+LN+Linear(1601)+loss in one call each way (ops._HeadLossFn), no fp32 logits / log-probs
            if "kl" in task:
                return ops.head_kl_div(rows, label_targets, net[0], net[2], net[3].weight, net[3].bias)
            hard = torch.max(label_targets[:, 1:], dim=-1)[1] + 1           # never 0, so ignore_index=0 is moot
            return ops.head_cross_entropy(rows, hard, net[0], net[2], net[3].weight, net[3].bias)
        if compute_loss:
            _lib.head_torch_path("MRC head loss", "needs bf16 CUDA rows and bf16 classifier weights")
        prediction_soft_label = self.region_classifier(rows)
        if not compute_loss:
            return prediction_soft_label
        if "kl" in task:
            log_probs = F.log_softmax(prediction_soft_label.float(), dim=-1)
            return F.kl_div(log_probs, label_targets.float(), reduction='none')
        # hard labels: arg-max over the non-background classes; class 0 (background) is never a target
        hard = torch.max(label_targets[:, 1:], dim=-1)[1] + 1
        return F.cross_entropy(prediction_soft_label.float(), hard, ignore_index=0, reduction='none')
