"""BERT encoder blocks with the reference's module tree (model/layer.py) on top of the HIP kernels.

The classes keep the reference names, constructor arguments, sub-module attribute names and therefore
state_dict keys (`attention.self.query.weight`, `attention.output.LayerNorm.bias`, ...),
so checkpoints and the task heads plug in unchanged.  What differs is underneath: a BertLayer does not
run its sub-modules one torch op at a time; `BertLayer.forward` (and `UniterEncoder.forward` for the
whole stack) hands the parameter pointers to `uniter_encoder_forward` which enqueues the fused kernels.
The sub-modules still exist (they own the parameters) and the small head-side modules (pooler, MLM head)
stay ordinary PyTorch, as in the reference.
"""
import math

import torch
from torch import nn

from .. import ops
from .._lib import UniterHipError

# The reference aliases apex FusedLayerNorm as BertLayerNorm (model/layer.py:25); the parameter holder here
# is torch's LayerNorm (same `weight` / `bias` names, same math for the head-side uses).
BertLayerNorm = nn.LayerNorm


def gelu(x):
    """Exact erf GELU, x * 0.5 * (1 + erf(x / sqrt(2)))  (model/layer.py:31-37; not the tanh approximation)."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def swish(x):
    return x * torch.sigmoid(x)


ACT2FN = {"gelu": gelu, "relu": torch.nn.functional.relu, "swish": swish}


class GELU(nn.Module):
    def forward(self, input_):
        return gelu(input_)


def _encoder_cfg(config):
    act = config.hidden_act
    codes = {"gelu": 0, "relu": 1, "swish": 2}                    # model/layer.py:44 ACT2FN
    if act not in codes:
        raise UniterHipError("hidden_act must be one of %s (model/layer.py:44), got %r" % (sorted(codes), act))
    return {"H": int(config.hidden_size), "heads": int(config.num_attention_heads),
            "I": int(config.intermediate_size), "ln_eps": 1e-12, "act": codes[act]}


class BertSelfAttention(nn.Module):
    """Owner of the query / key / value projections (model/layer.py:53-101).

    The three nn.Linear(H, H) stay separate named Parameters, but their storage is one [3H, H] weight and
    one [3H] bias buffer (`fused_qkv`) so the projection is a single MFMA GEMM; `.grad` is fused the same way.
    """

    def __init__(self, config):
        super(BertSelfAttention, self).__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError(
                "The hidden size (%d) is not a multiple of the number of attention "
                "heads (%d)" % (config.hidden_size, config.num_attention_heads))
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = int(config.hidden_size / config.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size

        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)

        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)
        self._qkv_w = None
        self._qkv_b = None
        self._qkv_gw = None
        self._qkv_gb = None

    # -- fused storage ----------------------------------------------------------------------------------
    def _linears(self):
        return (self.query, self.key, self.value)

    @staticmethod
    def _is_stacked(tensors, flat):
        if flat is None or tensors[0].device != flat.device or tensors[0].dtype != flat.dtype:
            return False
        step = tensors[0].numel() * tensors[0].element_size()
        base = flat.data_ptr()
        return all(t.data_ptr() == base + i * step and t.is_contiguous() for i, t in enumerate(tensors))

    @staticmethod
    def _adjacent(tensors):
        """Back to back inside ONE storage (views of an arena / of a fused buffer) — being neighbours in the
        allocator's address space by accident does not count: as_strided cannot span storages."""
        step = tensors[0].numel() * tensors[0].element_size()
        base = tensors[0].untyped_storage().data_ptr()
        return all(t.is_contiguous() and t.untyped_storage().data_ptr() == base
                   and t.data_ptr() == tensors[0].data_ptr() + i * step for i, t in enumerate(tensors))

    def fused_qkv(self):
        """([3H,H] weight, [3H] bias) whose thirds ARE query/key/value .weight/.bias (re-fused lazily after
        .to() / .bfloat16() / load_state_dict replaced the individual storages)."""
        ws = [m.weight for m in self._linears()]
        bs = [m.bias for m in self._linears()]
        H = ws[0].shape[1]
        if self._adjacent([w.data for w in ws]) and self._adjacent([b.data for b in bs]):
            # already laid out back to back (e.g. by utils.arena.ParamArena): build views without copying
            if not self._is_stacked([w.data for w in ws], self._qkv_w):
                self._qkv_w = torch.as_strided(ws[0].data, (3 * ws[0].shape[0], H), (H, 1))
                self._qkv_b = torch.as_strided(bs[0].data, (3 * bs[0].shape[0],), (1,))
            return self._qkv_w, self._qkv_b
        with torch.no_grad():
            fw = torch.cat([w.data for w in ws], dim=0).contiguous()
            fb = torch.cat([b.data for b in bs], dim=0).contiguous()
            n = ws[0].shape[0]
            for i, m in enumerate(self._linears()):
                m.weight.data = fw[i * n:(i + 1) * n]
                m.bias.data = fb[i * n:(i + 1) * n]
        self._qkv_w, self._qkv_b = fw, fb
        return fw, fb

    def fused_qkv_grad(self):
        """Fused gradient buffers whose thirds are query/key/value .weight.grad / .bias.grad."""
        ws = [m.weight for m in self._linears()]
        bs = [m.bias for m in self._linears()]
        H = ws[0].shape[1]
        n = ws[0].shape[0]
        for p in ws + bs:                          # a flat gradient arena hands out its (adjacent) slots on first use
            if p.grad is None:                     # (cleared if the slot still holds a gradient from before a zero_grad
                ops.attach_grad_slot(p)            #  that only dropped the reference)
        have = all(p.grad is not None for p in ws + bs)
        if have and self._adjacent([w.grad for w in ws]) and self._adjacent([b.grad for b in bs]):
            if not self._is_stacked([w.grad for w in ws], self._qkv_gw):
                self._qkv_gw = torch.as_strided(ws[0].grad, (3 * n, H), (H, 1))
                self._qkv_gb = torch.as_strided(bs[0].grad, (3 * n,), (1,))
            return self._qkv_gw, self._qkv_gb
        gw = torch.zeros(3 * n, H, dtype=ws[0].dtype, device=ws[0].device)
        gb = torch.zeros(3 * n, dtype=bs[0].dtype, device=bs[0].device)
        for i, m in enumerate(self._linears()):
            if m.weight.grad is not None:
                gw[i * n:(i + 1) * n].copy_(m.weight.grad)
            if m.bias.grad is not None:
                gb[i * n:(i + 1) * n].copy_(m.bias.grad)
            m.weight.grad = gw[i * n:(i + 1) * n]
            m.bias.grad = gb[i * n:(i + 1) * n]
        self._qkv_gw, self._qkv_gb = gw, gb
        return gw, gb

    def forward(self, hidden_states, attention_mask):
        """model/layer.py:75-101 on its own (BertLayer / UniterEncoder run the fused stack instead): the fused [3H, H] projection
        and the attention kernel as one autograd node."""
        return ops.self_attention(self, hidden_states, attention_mask)


class BertSelfOutput(nn.Module):
    """Parameter holder of the attention output projection + LayerNorm (model/layer.py:104-115)."""

    def __init__(self, config):
        super(BertSelfOutput, self).__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        """model/layer.py:111-115 on its own: bias + dropout + residual in the GEMM epilogue, then the LayerNorm kernel."""
        return ops.dense_dropout_residual_layernorm(self, hidden_states, input_tensor)


class BertAttention(nn.Module):
    def __init__(self, config):
        super(BertAttention, self).__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)

    def forward(self, input_tensor, attention_mask):
        """model/layer.py:124-127."""
        return self.output(self.self(input_tensor, attention_mask), input_tensor)


class BertIntermediate(nn.Module):
    def __init__(self, config):
        super(BertIntermediate, self).__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)
        if isinstance(config.hidden_act, str):
            self.intermediate_act_fn = ACT2FN[config.hidden_act]
        else:
            self.intermediate_act_fn = config.hidden_act

    def forward(self, hidden_states):
        """model/layer.py:139-142 on its own (erf GELU only: the other activations exist inside the fused stack)."""
        if self.intermediate_act_fn is not gelu:
            raise UniterHipError("BertIntermediate.forward on its own covers hidden_act = 'gelu'; relu / swish run inside BertLayer.forward")
        return ops.dense_gelu(self, hidden_states)


class BertOutput(nn.Module):
    def __init__(self, config):
        super(BertOutput, self).__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        """model/layer.py:152-156 on its own."""
        return ops.dense_dropout_residual_layernorm(self, hidden_states, input_tensor)


def layer_dropouts(layer):
    """(hidden p, attention p) currently set on a layer's nn.Dropout modules (utils.misc.set_dropout edits them)."""
    return float(layer.output.dropout.p), float(layer.attention.self.dropout.p)


def run_layers(layers, hidden_states, attention_mask, output_all=False, hook=None):
    """Shared entry of BertLayer.forward / UniterEncoder.forward."""
    layers = list(layers)
    first = layers[0]
    cfg = dict(first._enc_cfg)
    p_h, p_a = layer_dropouts(first)
    for lay in layers[1:]:
        if layer_dropouts(lay) != (p_h, p_a):
            raise UniterHipError("all layers of one encoder call must share their dropout probabilities")
    cfg["p_hidden"], cfg["p_attn"] = p_h, p_a
    training = first.training
    B, L = hidden_states.shape[0], hidden_states.shape[1]
    mask = attention_mask
    if mask.dtype != torch.float32:
        mask = mask.float()
    mask = mask.reshape(B, L)
    return ops.encoder_forward(layers, hidden_states, mask, cfg, training, need_all=output_all, hook=hook)


def run_layers_packed(layers, hidden_states, cu_seqlens, n_examples, max_len, output_all=False, hook=None):
    """Padding-free variant: hidden_states [T, H] holds only real tokens (see ops.encoder_forward_packed)."""
    layers = list(layers)
    first = layers[0]
    cfg = dict(first._enc_cfg)
    p_h, p_a = layer_dropouts(first)
    for lay in layers[1:]:
        if layer_dropouts(lay) != (p_h, p_a):
            raise UniterHipError("all layers of one encoder call must share their dropout probabilities")
    cfg["p_hidden"], cfg["p_attn"] = p_h, p_a
    return ops.encoder_forward_packed(layers, hidden_states, cu_seqlens, n_examples, max_len, cfg, first.training,
                                      need_all=output_all, hook=hook)


class BertLayer(nn.Module):
    """One transformer block; forward(hidden_states, attention_mask) as model/layer.py:166-170.

    attention_mask is the additive mask produced by UniterModel.forward ([B,1,1,L] or [B,L], (1-m)*-10000)."""

    def __init__(self, config):
        super(BertLayer, self).__init__()
        self.attention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)
        self._enc_cfg = _encoder_cfg(config)
        self._ptr_cache = None         # device-pointer row cached by ops._layer_row

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .bfloat16() re-allocate every parameter: forget cached device pointers
        self._ptr_cache = None
        return super(BertLayer, self)._apply(fn, *args, **kwargs)

    def forward(self, hidden_states, attention_mask):
        return run_layers([self], hidden_states, attention_mask)


class BertPooler(nn.Module):
    def __init__(self, config):
        super(BertPooler, self).__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.activation = nn.Tanh()

    def forward(self, hidden_states):
        # "pool" = transform of the first ([CLS]) token (model/layer.py:179-185)
        cls_state = hidden_states[:, 0]
        return self.activation(self.dense(cls_state))


class BertPredictionHeadTransform(nn.Module):
    def __init__(self, config):
        super(BertPredictionHeadTransform, self).__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        if isinstance(config.hidden_act, str):
            self.transform_act_fn = ACT2FN[config.hidden_act]
        else:
            self.transform_act_fn = config.hidden_act
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=1e-12)

    def forward(self, hidden_states):
        return self.LayerNorm(self.transform_act_fn(self.dense(hidden_states)))


class BertLMPredictionHead(nn.Module):
    """MLM decoder tied to the word-embedding matrix + output-only bias (model/layer.py:205-222)."""

    def __init__(self, config, bert_model_embedding_weights):
        super(BertLMPredictionHead, self).__init__()
        self.transform = BertPredictionHeadTransform(config)
        vocab, hidden = bert_model_embedding_weights.size(0), bert_model_embedding_weights.size(1)
        self.decoder = nn.Linear(hidden, vocab, bias=False)
        self.decoder.weight = bert_model_embedding_weights
        self.bias = nn.Parameter(torch.zeros(vocab))

    def forward(self, hidden_states):
        return self.decoder(self.transform(hidden_states)) + self.bias


class BertOnlyMLMHead(nn.Module):
    def __init__(self, config, bert_model_embedding_weights):
        super(BertOnlyMLMHead, self).__init__()
        self.predictions = BertLMPredictionHead(config, bert_model_embedding_weights)

    def forward(self, sequence_output):
        return self.predictions(sequence_output)
