"""UNITER model classes with the reference's public surface (model/model.py) on the HIP kernels.

Kept identical to the reference: `UniterConfig` fields and (de)serialisation,
`UniterPreTrainedModel.init_weights / from_pretrained` semantics (gamma/beta renaming, optional `bert.`
prefix, lenient missing / unexpected keys), module attribute paths, state_dict keys and the
`UniterModel.forward` signature.  The embeddings, the compaction gather, the additive mask and the whole
encoder stack run as HIP kernels (uniter_amd/ops.py -> include/uniter_hip.h).

Attribution: `UniterConfig` and `UniterPreTrainedModel.{init_weights, from_pretrained}` restate the interface of
ChenRocks/UNITER model/model.py (Copyright (c) Microsoft Corporation, MIT license; itself modified from HuggingFace
transformers, Apache-2.0) line for line, because checkpoints, configs and the task heads depend on exactly that surface.
"""
import copy
import os
import json
import logging
from io import open

import torch
from torch import nn

from .. import ops
from .layer import run_layers_packed, BertLayer, BertLayerNorm, BertPooler, run_layers

logger = logging.getLogger(__name__)


class UniterConfig(object):
    """Attribute bag describing a `UniterModel` (model/model.py:24-114): built from an int vocabulary size
    plus keyword hyper-parameters, or from the path of a JSON file (config/uniter-base.json)."""

    def __init__(self, vocab_size_or_config_json_file, hidden_size=768, num_hidden_layers=12,
                 num_attention_heads=12, intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1,
                 attention_probs_dropout_prob=0.1, max_position_embeddings=512, type_vocab_size=2,
                 initializer_range=0.02):
        if isinstance(vocab_size_or_config_json_file, str):
            with open(vocab_size_or_config_json_file, "r", encoding='utf-8') as reader:
                for key, value in json.loads(reader.read()).items():
                    self.__dict__[key] = value
        elif isinstance(vocab_size_or_config_json_file, int):
            self.vocab_size = vocab_size_or_config_json_file
            self.hidden_size = hidden_size
            self.num_hidden_layers = num_hidden_layers
            self.num_attention_heads = num_attention_heads
            self.hidden_act = hidden_act
            self.intermediate_size = intermediate_size
            self.hidden_dropout_prob = hidden_dropout_prob
            self.attention_probs_dropout_prob = attention_probs_dropout_prob
            self.max_position_embeddings = max_position_embeddings
            self.type_vocab_size = type_vocab_size
            self.initializer_range = initializer_range
        else:
            raise ValueError("First argument must be either a vocabulary size "
                             "(int) or the path to a pretrained model config "
                             "file (str)")

    @classmethod
    def from_dict(cls, json_object):
        config = UniterConfig(vocab_size_or_config_json_file=-1)
        config.__dict__.update(json_object)
        return config

    @classmethod
    def from_json_file(cls, json_file):
        with open(json_file, "r", encoding='utf-8') as reader:
            return cls.from_dict(json.loads(reader.read()))

    def __repr__(self):
        return str(self.to_json_string())

    def to_dict(self):
        return copy.deepcopy(self.__dict__)

    def to_json_string(self):
        return json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n"


class UniterPreTrainedModel(nn.Module):
    """Weight initialisation + checkpoint loading shared by every UNITER model (model/model.py:117-214)."""

    def __init__(self, config, *inputs, **kwargs):
        super().__init__()
        if not isinstance(config, UniterConfig):
            raise ValueError(
                "Parameter config in `{}(config)` should be an instance of "
                "class `UniterConfig`. To create a model from a Google "
                "pretrained model use "
                "`model = {}.from_pretrained(PRETRAINED_MODEL_NAME)`".format(
                    self.__class__.__name__, self.__class__.__name__))
        self.config = config

    def init_weights(self, module):
        """N(0, initializer_range) for Linear / Embedding weights, zeros for Linear biases, (1, 0) for LayerNorm
        (model/model.py:133-146)."""
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        elif isinstance(module, BertLayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    @classmethod
    def from_pretrained(cls, config_file, state_dict, *inputs, **kwargs):
        """Build `cls(config, *inputs, **kwargs)` from a config JSON and load `state_dict` into it
        (model/model.py:148-214).  `{}` is the normal no-checkpoint call (pretrain.py:215-221)."""
        config = UniterConfig.from_json_file(config_file)
        logger.info("Model config {}".format(config))
        model = cls(config, *inputs, **kwargs)

        # TF-era checkpoints call the LayerNorm parameters gamma / beta
        renames = []
        for key in list(state_dict.keys()):
            new_key = None
            if 'gamma' in key:
                new_key = key.replace('gamma', 'weight')
            if 'beta' in key:
                new_key = key.replace('beta', 'bias')
            if new_key:
                renames.append((key, new_key))
        for old_key, new_key in renames:
            state_dict[new_key] = state_dict.pop(old_key)

        missing_keys, unexpected_keys, error_msgs = [], [], []
        metadata = getattr(state_dict, '_metadata', None)
        state_dict = state_dict.copy()
        if metadata is not None:
            state_dict._metadata = metadata

        def load(module, prefix=''):
            local_metadata = {} if metadata is None else metadata.get(prefix[:-1], {})
            module._load_from_state_dict(state_dict, prefix, local_metadata, True, missing_keys, unexpected_keys,
                                         error_msgs)
            for name, child in module._modules.items():
                if child is not None:
                    load(child, prefix + name + '.')

        start_prefix = ''
        if not hasattr(model, 'bert') and any(s.startswith('bert.') for s in state_dict.keys()):
            start_prefix = 'bert.'
        load(model, prefix=start_prefix)
        if len(missing_keys) > 0:
            logger.info("Weights of {} not initialized from pretrained model: {}".format(
                model.__class__.__name__, missing_keys))
        if len(unexpected_keys) > 0:
            logger.info("Weights from pretrained model not used in {}: {}".format(
                model.__class__.__name__, unexpected_keys))
        if len(error_msgs) > 0:
            raise RuntimeError('Error(s) in loading state_dict for {}:\n\t{}'.format(
                model.__class__.__name__, "\n\t".join(error_msgs)))
        return model


class UniterTextEmbeddings(nn.Module):
    """word + position + token-type embeddings -> LayerNorm -> dropout (model/model.py:217-245)."""

    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        # TF checkpoint naming: LayerNorm, not layer_norm
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, input_ids, position_ids, token_type_ids=None):
        # token_type_ids=None means "all zeros" (model/model.py:233-234); handled inside the kernel
        return ops.txt_embeddings(self, input_ids, position_ids, token_type_ids)


class UniterImageEmbeddings(nn.Module):
    """LN(Linear(img_feat)) + LN(Linear_7(box)) + type embedding -> LayerNorm -> dropout (model/model.py:248-272)."""

    def __init__(self, config, img_dim):
        super().__init__()
        self.img_linear = nn.Linear(img_dim, config.hidden_size)
        self.img_layer_norm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.pos_layer_norm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.pos_linear = nn.Linear(7, config.hidden_size)
        self.mask_embedding = nn.Embedding(2, img_dim, padding_idx=0)

        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, img_feat, img_pos_feat, type_embeddings, img_masks=None):
        """type_embeddings: either the already looked-up rows [B, Li, H] (reference call convention) or a
        `(table, ids)` pair, which lets the lookup be fused into the kernel (UniterModel uses this form)."""
        if isinstance(type_embeddings, tuple):
            table, ids = type_embeddings
            return ops.img_embeddings(self, img_feat, img_pos_feat, table, ids, img_masks)
        # the reference's own call form (model/model.py:261-272, e.g. from third-party code): the rows are already looked up
        return ops.img_embeddings(self, img_feat, img_pos_feat, type_embeddings, None, img_masks)


def pack_indices(seq_lens, max_len, multiple=64):
    """Host-side tables of the padding-free layout for B sequences stored as [B, max_len] rows.

    Returns (idx, cu_seqlens, total, extra): `idx` int64 [total] = flat row indices (b * max_len + t) of the real
    tokens in example order; `cu_seqlens` int32 [B + len(extra) + 1]; `extra` = lengths of the all-zero dummy examples
    that round the packed row count up to a multiple of `multiple` (the wave-specialised wgrad tiles contract over whole
    64-row steps); each dummy is at most max_len long."""
    lens = [int(v) for v in (seq_lens.tolist() if torch.is_tensor(seq_lens) else seq_lens)]
    if any(v < 0 or v > max_len for v in lens):
        raise ValueError("sequence lengths must lie in [0, max_len]")
    total = sum(lens)
    parts = [torch.arange(v, dtype=torch.int64) + b * max_len for b, v in enumerate(lens) if v > 0]
    idx = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64)
    pad = (-total) % multiple if multiple > 1 else 0
    extra = []
    while pad > 0:
        extra.append(min(pad, max_len))
        pad -= extra[-1]
    cu = torch.zeros(len(lens) + len(extra) + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(torch.tensor(lens + extra, dtype=torch.int64), dim=0).to(torch.int32)
    return idx, cu, total, extra


class UniterEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        layer = BertLayer(config)
        self.layer = nn.ModuleList([copy.deepcopy(layer) for _ in range(config.num_hidden_layers)])
        # optional callable(layer_index) fired during backward when a layer's gradients have been enqueued
        # (utils.distributed.GradientReducer uses it to overlap the allreduce with the rest of backward)
        self.grad_ready_hook = None

    def forward(self, input_, attention_mask, output_all_encoded_layers=True):
        if output_all_encoded_layers:
            outs = run_layers(self.layer, input_, attention_mask, output_all=True, hook=self.grad_ready_hook)
            return list(outs)
        return [run_layers(self.layer, input_, attention_mask, output_all=False, hook=self.grad_ready_hook)]

    def forward_packed(self, input_, valid_mask, output_all_encoded_layers=True, seq_lens=None):
        """Padding-free execution.  input_ [B, L, H]; valid_mask [B, L] (1 = real token; the real
        tokens of every row must form a prefix, which is how gather_index lays sequences out, data/data.py:271-279).
        Only the real tokens go through the layers; the returned [B, L, H] tensors are ZERO at padded positions (the
        reference computes values there that no head ever reads).  With `seq_lens` (host-side list / CPU tensor of the
        B real lengths) nothing synchronises; without it the token count is read back from the device (one sync)."""
        B, L, H = input_.shape
        dev = input_.device
        if seq_lens is None:
            # lengths from the mask: one device -> host copy of B integers (a sync); callers that know the lengths on the
            # host (every collate does) pass them and skip it
            seq_lens = (valid_mask.reshape(B, L) != 0).sum(dim=1).tolist()
        idx, cu_host, total, extra = pack_indices(seq_lens, L)
        idx = idx.to(dev, non_blocking=True)
        cu = cu_host.to(dev, non_blocking=True)
        x = input_.reshape(B * L, H).index_select(0, idx)
        if extra:
            x = torch.cat([x, x.new_zeros(sum(extra), H)], dim=0)
        outs = run_layers_packed(self.layer, x, cu, B + len(extra), int(L), output_all=output_all_encoded_layers,
                                 hook=self.grad_ready_hook)
        outs = list(outs) if output_all_encoded_layers else [outs]

        def unpack(y):
            return y.new_zeros(B * L, H).index_copy(0, idx, y[:total]).view(B, L, H)

        return [unpack(y) for y in outs]


class UniterModel(UniterPreTrainedModel):
    """Joint vision-language encoder (model/model.py:295-367)."""

    def __init__(self, config, img_dim):
        super().__init__(config)
        self.embeddings = UniterTextEmbeddings(config)
        self.img_embeddings = UniterImageEmbeddings(config, img_dim)
        self.encoder = UniterEncoder(config)
        self.pooler = BertPooler(config)
        # True: run the encoder on real tokens only (UniterEncoder.forward_packed); off by default because it costs a
        # host sync per step and changes nothing when batches have no padding.  UNITER_AMD_PACK_PADDING=1 turns it on.
        self.pack_padding = os.environ.get("UNITER_AMD_PACK_PADDING", "0") == "1"
        # host-side sequence lengths of the NEXT forward (list / CPU tensor of B ints = tl + nbb, which the collate
        # knows: data/data.py:255-279): lets the packed path build its index tables without a device sync.  The task
        # wrappers copy batch['seq_lens'] here; consumed (reset to None) by forward.
        self.seq_lens_hint = None
        self.apply(self.init_weights)

    def _compute_txt_embeddings(self, input_ids, position_ids, txt_type_ids=None):
        return self.embeddings(input_ids, position_ids, txt_type_ids)

    def _compute_img_embeddings(self, img_feat, img_pos_feat, img_masks=None, img_type_ids=None):
        # img_type_ids=None: every region gets token-type row 1 (model/model.py:313-316)
        table = self.embeddings.token_type_embeddings.weight
        return self.img_embeddings(img_feat, img_pos_feat, (table, img_type_ids), img_masks)

    def _compute_img_txt_embeddings(self, input_ids, position_ids, img_feat, img_pos_feat, gather_index,
                                    img_masks=None, txt_type_ids=None, img_type_ids=None):
        txt_emb = self._compute_txt_embeddings(input_ids, position_ids, txt_type_ids)
        img_emb = self._compute_img_embeddings(img_feat, img_pos_feat, img_masks, img_type_ids)
        # align back to the most compact [txt_i ; img_i ; pad] sequence
        return ops.gather_embeddings(txt_emb, img_emb, gather_index)

    def forward(self, input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index=None,
                img_masks=None, output_all_encoded_layers=True, txt_type_ids=None, img_type_ids=None):
        # additive key mask (1 - m) * -10000, kept in fp32 and shaped like the reference's [B,1,1,L]
        extended_attention_mask = ops.mask_bias(attention_mask).unsqueeze(1).unsqueeze(2)

        if input_ids is None:
            embedding_output = self._compute_img_embeddings(img_feat, img_pos_feat, img_masks, img_type_ids)
        elif img_feat is None:
            embedding_output = self._compute_txt_embeddings(input_ids, position_ids, txt_type_ids)
        else:
            embedding_output = self._compute_img_txt_embeddings(
                input_ids, position_ids, img_feat, img_pos_feat, gather_index, img_masks, txt_type_ids,
                img_type_ids)

        lens_hint, self.seq_lens_hint = self.seq_lens_hint, None
        if self.pack_padding:
            encoded_layers = self.encoder.forward_packed(embedding_output, attention_mask,
                                                         output_all_encoded_layers=output_all_encoded_layers,
                                                         seq_lens=lens_hint)
        else:
            encoded_layers = self.encoder(embedding_output, extended_attention_mask,
                                          output_all_encoded_layers=output_all_encoded_layers)
        if not output_all_encoded_layers:
            encoded_layers = encoded_layers[-1]
        return encoded_layers
