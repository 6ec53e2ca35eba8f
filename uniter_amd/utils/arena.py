"""Flat parameter / gradient arenas (MI355X memory layout of the model).

All parameters of a model live back to back in ONE bf16 (or fp32) buffer and all gradients in a second one
of the same layout; every nn.Parameter and its `.grad` become views.  This gives
  * fused QKV: query/key/value weights (and biases, and their grads) of a BertSelfAttention are adjacent,
    so the projection is one [3H,H] GEMM without copies (model/layer.py:64-66 keeps three named Linears);
  * the data-parallel allreduce operates in place on contiguous ranges of the gradient arena, no
    flatten/unflatten copies (the reference copies 228 tensors in and out, utils/distributed.py:24-43);
  * `optimizer.zero_grad()` is one memset; the fused AdamW walks aligned 16-byte streams.
Tensors are aligned to 128 elements (256 B in bf16).

Gradients are attached lazily: a parameter's `.grad` stays None until a backward pass first produces a gradient for
it, exactly like the reference (optim/adamw.py:52-53 skips `p.grad is None`; `zero_grad()` keeps existing gradients
as zeros).  Parameters the loss never reaches — the pooler under the NLVR2 paired-attention head, `mask_embedding`
in fine-tuning, a pre-training head before its task is first drawn — are therefore neither decayed nor stepped, and
a head's bias correction starts at its own first update.  The arena slot is handed out by `ops.ensure_grad` (kernels
that write gradients themselves) or by a post-accumulate hook that moves autograd's freshly allocated gradient into
the slot (PyTorch modules).
"""
import torch

from .. import _lib

from ..model.layer import BertSelfAttention

ALIGN = 128


def _ordered_parameters(model):
    """Unique parameters in module order, with each BertSelfAttention emitting q.w, k.w, v.w, q.b, k.b, v.b."""
    seen = set()
    out = []

    def add(name, p):
        if p is not None and id(p) not in seen:
            seen.add(id(p))
            out.append((name, p))

    def visit(mod, prefix):
        if isinstance(mod, BertSelfAttention):
            for attr in ("weight", "bias"):
                for lin in ("query", "key", "value"):
                    add("%s%s.%s" % (prefix, lin, attr), getattr(getattr(mod, lin), attr))
            return
        for n, p in mod._parameters.items():
            add(prefix + n, p)
        for n, child in mod._modules.items():
            if child is not None:
                visit(child, prefix + n + ".")

    visit(model, "")
    return out


def _adopt_grad(param):
    """post-accumulate hook: autograd allocated a gradient of its own for a parameter whose `.grad` was None — move it
    into the arena slot (which is all zeros until then) and make the slot the gradient."""
    slot = getattr(param, '_uniter_grad_slot', None)
    g = param.grad
    if slot is None or g is None or g.data_ptr() == slot.data_ptr():
        return
    with torch.no_grad():
        slot.copy_(g)
    param.grad = slot
    param._uniter_slot_used = True
    _lib.note_grad_attached()


class ParamArena(object):
    """Re-homes every parameter (and gradient) of `model` into flat buffers.

    Call after the model has its final device / dtype (`model.cuda().bfloat16()`), before building the
    optimizer plan.  Parameters keep their identity, so optimizers and state_dict are unaffected.
    """

    def __init__(self, model, with_grad=True):
        named = _ordered_parameters(model)
        if not named:
            raise ValueError("model has no parameters")
        dtype, device = named[0][1].dtype, named[0][1].device
        for n, p in named:
            if p.dtype != dtype or p.device != device:
                raise ValueError("all parameters must share dtype/device (%s is %s on %s)" % (n, p.dtype, p.device))
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        self.offsets = []
        o = 0
        fused_tail = set()
        # q/k/v triples must be exactly adjacent (no alignment padding between them)
        for i, n in enumerate(self.names):
            if n.endswith(("key.weight", "value.weight", "key.bias", "value.bias")) and i > 0:
                prev = self.names[i - 1]
                if prev.rsplit(".", 2)[0] == n.rsplit(".", 2)[0] and prev.rsplit(".", 1)[1] == n.rsplit(".", 1)[1]:
                    fused_tail.add(i)
        for i, p in enumerate(self.params):
            if i not in fused_tail:
                o = (o + ALIGN - 1) // ALIGN * ALIGN
            self.offsets.append(o)
            o += p.numel()
        self.numel = (o + ALIGN - 1) // ALIGN * ALIGN
        self.data = torch.zeros(self.numel, dtype=dtype, device=device)
        self.grad = torch.zeros(self.numel, dtype=dtype, device=device) if with_grad else None
        if self.grad is not None:
            self.grad._uniter_flat_grad = True
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                view = self.data[off:off + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                if self.grad is not None:
                    g = self.grad[off:off + p.numel()].view(p.shape)
                    p._uniter_grad_slot = g
                    if p.grad is not None:              # an existing gradient moves in; otherwise it stays None
                        g.copy_(p.grad)
                        p.grad = g
                    if not getattr(p, '_uniter_grad_hook', False):
                        p.register_post_accumulate_grad_hook(_adopt_grad)
                        p._uniter_grad_hook = True
        for m in model.modules():                      # parameter storages moved: drop cached device pointers
            if hasattr(m, "_ptr_cache"):
                m._ptr_cache = None

    def span(self, params):
        """(start, end) element range of the arena covering the given parameters (must be arena members)."""
        index = {id(p): i for i, p in enumerate(self.params)}
        idx = [index[id(p)] for p in params]
        lo = min(self.offsets[i] for i in idx)
        hi = max(self.offsets[i] + self.params[i].numel() for i in idx)
        return lo, hi

    def zero_grad(self):
        if self.grad is not None:
            self.grad.zero_()

    def check(self):
        """True if every parameter / gradient still is a view at its arena offset (e.g. nobody called .to())."""
        esz = self.data.element_size()
        for p, off in zip(self.params, self.offsets):
            if p.data.data_ptr() != self.data.data_ptr() + off * esz:
                return False
            if self.grad is not None and p.grad is not None and p.grad.data_ptr() != self.grad.data_ptr() + off * esz:
                return False
        return True


def flatten_model(model, with_grad=True):
    """Convenience: build the arena and remember it on the model (`model._uniter_arena`)."""
    arena = ParamArena(model, with_grad=with_grad)
    model._uniter_arena = arena
    return arena
