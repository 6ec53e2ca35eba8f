"""AdamW with decoupled weight decay, as one fused HIP kernel over every parameter.

Reference: optim/adamw.py:11-103 (HF AdamW: a Python loop of ~6 elementwise ops per tensor).  Same
constructor, `param_groups` keys (`lr, betas, eps, weight_decay, correct_bias`, mutated from outside every
step, pretrain.py:317-318) and `state[p]` keys (`step, exp_avg, exp_avg_sq`); the update rule is the
reference's, including weight decay applied AFTER the Adam update with the un-corrected lr
(optim/adamw.py:88-101).

MI355X specifics:
  * all tensors are updated by ONE kernel launch (`uniter_adamw_step`); the tensor table lives on the
    device and is rebuilt only when the set of (param, grad) storages changes;
  * bf16 parameters (the apex-O2 replacement: bf16 model, fp32 master weights) keep an fp32 master copy
    in `state[p]['master']`; fp32 parameters are updated directly;
  * gradient clipping is fused: `clip_grad_norm_(optimizer, max_norm)` computes the global norm on the
    device (no host sync) and leaves a device-side coefficient that the next `step()` applies while it
    reads the gradients — identical to scaling the gradients in place first;
  * `enable_overlap(boundaries)`: the update (HBM-bound, 28 B per parameter) runs on a side stream in
    address-ordered segments and the next forward pass (MFMA-bound) overlaps it; every consumer of a
    parameter waits for that parameter's segment only (`uniter_params_wait`, called by the encoder per layer
    and by the embedding ops), `zero_grad()` is folded into the kernel.  `synchronize()` joins the side
    stream for code that reads parameters / optimizer state some other way.
"""
import ctypes
import os
import math

import torch
from torch.optim import Optimizer

from .. import _lib
from .._lib import C, UniterAdamGroup, UniterAdamTensor, ptr


class AdamW(Optimizer):
    """Adam with the weight-decay fix.

    Parameters:
        lr (float): learning rate. Default 1e-3.
        betas (tuple of 2 floats): Adam's beta parameters (b1, b2). Default: (0.9, 0.999)
        eps (float): Adam's epsilon (added OUTSIDE the square root). Default: 1e-6
        weight_decay (float): decoupled weight decay. Default: 0.0
        correct_bias (bool): apply the bias correction (False reproduces the BERT TF repository). Default True.
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[1]))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(eps))
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias)
        super(AdamW, self).__init__(params, defaults)
        self._plan = None
        self._plan_key = None
        self._plan_groups = None      # internal group id -> (param_group index, [params])
        self._keep = None
        self._clip = None             # device tensor [norm, coef] left by clip_grad_norm_
        self._norm_buf = None
        self._overlap = None          # ctypes array of segment boundaries (parameter addresses) or None
        self._grads_zeroed = False    # the last step zeroed the gradients itself (fused), zero_grad() has nothing to do
        self.fuse_zero_grad = False   # True: step() zeroes every gradient once read and the next zero_grad() is a no-op —
                                      # for loops that call zero_grad() right after step() (pretrain.py:332-334)
        # lazy_zero (with a fused zero_grad only): the gradients the encoder's backward produces are NOT zeroed by the step — the next
        # backward replaces them (uniter_encoder_set_grad_overwrite), so the update writes 2 bytes per parameter less and the deferred
        # weight-gradient launch does not read the old values.  Like zero_grad(set_to_none=True): between step() and the next
        # backward those `.grad` tensors hold stale values, not zeros; every reader inside this package (grad_norm, step, the
        # allreduce — all after a backward) sees finished gradients, and an optimizer call without a backward in between zeroes
        # them first (_lib.lazy_resolve).  UNITER_AMD_LAZY_ZERO=0 switches it off.
        self.lazy_zero = os.environ.get("UNITER_AMD_LAZY_ZERO", "1") != "0"
        self._keep_ranges = None      # the _lib.lazy_ranges the plan's keep flags were computed from
        # fold_norm: grad_norm() takes the encoder weights' share of sum g^2 from the per-tile sums their deferred launch left
        # (ops._EncoderFn.backward -> _lib.sq_state) instead of reading those 170 MB again; any doubt (a tensor touched since, a
        # data-parallel reduction, parameters outside the plan) falls back to the full reduction.  Opt-in (UNITER_AMD_FOLD_NORM=1 +
        # ops.set_fold_norm): on the c2 step the reduction gets 20 us shorter and the update 29 us longer — the reduction is also what
        # brings the gradients into the Infinity Cache for the update (profiles/r06_fold_norm_ab.txt).
        self.fold_norm = os.environ.get("UNITER_AMD_FOLD_NORM", "0") == "1"
        self._skip_ranges = None

    # ---- plan management ------------------------------------------------------------------------------
    def _active(self):
        """[(group_index, param)] for parameters that currently have a gradient (optim/adamw.py:52-53)."""
        out = []
        for gi, group in enumerate(self.param_groups):
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError('Adam does not support sparse gradients, please consider SparseAdam instead')
                out.append((gi, p))
        return out

    def _init_state(self, active):
        fresh = [p for _, p in active if len(self.state[p]) == 0]
        if not fresh:
            return
        dev = fresh[0].device
        align = 64      # elements: keeps every tensor's fp32 streams 256-byte aligned
        total = sum((p.numel() + align - 1) // align * align for p in fresh)
        n_master = sum((p.numel() + align - 1) // align * align for p in fresh if p.dtype == torch.bfloat16)
        moments = torch.zeros(2, total, dtype=torch.float32, device=dev)
        master = torch.empty(max(n_master, 1), dtype=torch.float32, device=dev)
        o = om = 0
        for p in fresh:
            n = p.numel()
            st = self.state[p]
            st['step'] = 0
            st['exp_avg'] = moments[0, o:o + n].view_as(p)
            st['exp_avg_sq'] = moments[1, o:o + n].view_as(p)
            o += (n + align - 1) // align * align
            if p.dtype == torch.bfloat16:
                m = master[om:om + n].view_as(p)
                m.copy_(p.data)
                st['master'] = m
                om += (n + align - 1) // align * align

    def _build_plan(self, active):
        for _, p in active:
            if not p.is_cuda:
                raise _lib.UniterHipError("uniter_amd AdamW updates CUDA(HIP) parameters only (fused gfx950 kernel)")
            if p.dtype not in (torch.bfloat16, torch.float32):
                raise _lib.UniterHipError("AdamW supports bf16 (with fp32 master) and fp32 parameters, got %s" % p.dtype)
            if not p.data.is_contiguous() or not p.grad.is_contiguous() or p.grad.dtype != p.dtype:
                raise _lib.UniterHipError("AdamW needs contiguous parameters / gradients of matching dtype")
        self._init_state(active)
        # internal groups = (param group, step cohort): parameters of one cohort advance their step together
        cohorts = {}
        for gi, p in active:
            cohorts.setdefault((gi, self.state[p]['step']), []).append(p)
        if len(cohorts) > 16:
            raise _lib.UniterHipError("more than 16 (param group x step) cohorts are not supported by the fused AdamW")
        table = (UniterAdamTensor * len(active))()
        groups = []
        i = 0
        for ig, ((gi, _), plist) in enumerate(sorted(cohorts.items(), key=lambda kv: kv[0])):
            groups.append((gi, plist))
            for p in plist:
                st = self.state[p]
                for name in ('exp_avg', 'exp_avg_sq') + (('master',) if p.dtype == torch.bfloat16 else ()):
                    buf = st[name]
                    if buf.dtype != torch.float32 or not buf.is_contiguous() or buf.device != p.device:
                        raise _lib.UniterHipError("AdamW state '%s' must be a contiguous fp32 tensor on the parameter's device "
                                                  "(got %s); load checkpoints through AdamW.load_state_dict" % (name, buf.dtype))
                t = table[i]
                t.param, t.grad = p.data.data_ptr(), p.grad.data_ptr()
                t.master = st['master'].data_ptr() if p.dtype == torch.bfloat16 else None
                t.exp_avg, t.exp_avg_sq = st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr()
                t.numel, t.group, t.param_is_bf16 = p.numel(), ig, 1 if p.dtype == torch.bfloat16 else 0
                i += 1
        self._destroy_plan()
        handle = ctypes.c_void_p()
        C.uniter_adamw_plan_create(table, len(active), ctypes.byref(handle))
        self._plan, self._plan_groups = handle, groups
        self._plan_grads = [(int(table[k].grad), int(table[k].numel) * (2 if table[k].param_is_bf16 else 4)) for k in range(len(active))]
        self._keep_ranges = self._skip_ranges = None      # (a new device table: no flags yet)
        self._keep_any = self._skip_any = False

    def _covered(self, ranges):
        """Per plan tensor: does its gradient storage lie inside one of `ranges` (frozenset of (first byte, byte length))?"""
        import bisect
        spans = sorted(ranges)
        starts = [a for a, _ in spans]
        out = []
        for g, nbytes in self._plan_grads:
            j = bisect.bisect_right(starts, g) - 1
            out.append(j >= 0 and g + nbytes <= spans[j][0] + spans[j][1])
        return out

    def _sync_flags(self, skip_ranges=None):
        """Bring the plan's per-tensor flags in line with lazy_zero (tensors the encoder's backward overwrites: KEEP_GRAD) and
        fold_norm (tensors whose sum of squares comes with the backward: SKIP_NORM).  Returns (any kept, folding possible)."""
        keep_ranges = _lib.lazy_ranges if self.lazy_zero else frozenset()
        if skip_ranges is None:
            skip_ranges = self._skip_ranges if self._skip_ranges is not None else frozenset()
        if keep_ranges == self._keep_ranges and skip_ranges == self._skip_ranges:
            return self._keep_any, self._skip_any
        keep = self._covered(keep_ranges)
        skip = self._covered(skip_ranges)
        # the per-tile sums stand for WHOLE tensors of the encoder: they can replace the reduction only if the plan holds every byte
        covered = sum(nb for (_, nb), f in zip(self._plan_grads, skip) if f)
        if covered != sum(nb for _, nb in skip_ranges):
            skip = [False] * len(skip)
        flags = (ctypes.c_uint8 * len(self._plan_grads))(*[(1 if k else 0) | (2 if f else 0) for k, f in zip(keep, skip)])
        C.uniter_adamw_plan_set_flags(self._plan, flags, len(self._plan_grads))
        self._keep_ranges, self._skip_ranges = keep_ranges, skip_ranges
        self._keep_any, self._skip_any = any(keep), any(skip)
        return self._keep_any, self._skip_any

    def _destroy_plan(self):
        if self._plan is not None:
            try:
                C.uniter_adamw_plan_destroy(self._plan)
            finally:
                self._plan = None

    def __del__(self):
        try:
            self._destroy_plan()
        except Exception:
            pass

    def _ensure_plan(self, in_step=False):
        """(Re)build the device-side tensor table when the set of (parameter, gradient) storages changed.

        Full validation (two data_ptr() calls per tensor, ~0.5 ms for 230 tensors) runs every 32nd call; in between a
        cheap check catches the two things that happen in practice: a gradient tensor was replaced (set_to_none,
        first backward) -> identity test per parameter, or the parameter groups were edited."""
        self._calls = getattr(self, '_calls', 0) + 1
        if (in_step and self._plan is not None and getattr(self, '_checked_for_step', False)
                and getattr(self, '_checked_epoch', -1) == _lib.grad_attach_epoch()):
            # grad_norm() validated the plan a moment ago in this same optimizer step (clip_grad_norm_ -> step()) and no
            # parameter has been given a gradient tensor since (a backward that lazily attaches new ones bumps the epoch)
            self._checked_for_step = False
            return True
        self._checked_for_step = False
        if self._plan is not None and (self._calls & 31):
            refs = self._plan_refs
            n = 0
            ok = True
            for group in self.param_groups:
                for p in group['params']:
                    g = p.grad
                    if g is None:
                        continue
                    if n >= len(refs) or refs[n][0] is not p or refs[n][1] is not g:
                        ok = False
                        break
                    n += 1
                if not ok:
                    break
            if ok and n == len(refs):
                return True
        active = self._active()
        if not active:
            return False
        key = tuple((gi, p.data.data_ptr(), p.grad.data_ptr(), p.numel(),
                     self.state[p]['exp_avg'].data_ptr() if 'exp_avg' in self.state[p] else 0) for gi, p in active)
        if self._plan is None or key != self._plan_key:
            self._build_plan(active)
            self._plan_key = key
        self._plan_refs = [(p, p.grad) for _, p in active]
        self._flat_grads = None
        return True

    # ---- asynchronous step ------------------------------------------------------------------------------
    def enable_overlap(self, boundaries, fuse_zero_grad=True):
        """Run every later step() asynchronously on the library's optimizer stream, cut into segments at the given
        parameter addresses (any iterable of ints / tensors; see `overlap_boundaries(model)`), so that the next forward
        pass overlaps the update.  Contract: between step() and the next backward the parameters are only read through
        uniter_amd modules (they wait per segment) or after `synchronize()`; with `fuse_zero_grad` the following
        `zero_grad()` is a no-op because the step zeroes each gradient right after reading it."""
        addrs = sorted(set(int(b.data_ptr()) if isinstance(b, torch.Tensor) else int(b) for b in boundaries))
        self._overlap = ((ctypes.c_void_p * len(addrs))(*addrs), len(addrs), bool(fuse_zero_grad))

    def disable_overlap(self):
        self.synchronize()
        self._overlap = None

    def synchronize(self):
        """Make the current stream wait for a pending asynchronous step (no host synchronisation)."""
        if _lib.async_pending():
            C.uniter_params_wait_all(_lib.stream_ptr())
            _lib.set_async_pending(False)

    def state_dict(self):
        self.synchronize()
        return super(AdamW, self).state_dict()

    def load_state_dict(self, state_dict):
        """torch's Optimizer.load_state_dict casts floating-point state to the PARAMETER dtype: for bf16 parameters the
        fp32 moments and master weights of the checkpoint would come back as bf16 while the kernel reads them as fp32.
        Restore them in fp32 (contiguous, 16-byte aligned) and drop the device plan, which points at the old buffers."""
        self.synchronize()
        import copy
        sd = copy.deepcopy(state_dict) if isinstance(state_dict, dict) else state_dict
        # keep fp32 copies of the saved state before torch casts them
        saved = {k: {n: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for n, v in st.items()}
                 for k, st in sd['state'].items()}
        super(AdamW, self).load_state_dict(sd)
        id_map = {}
        for g_saved, g_live in zip(sd['param_groups'], self.param_groups):
            for pid, p in zip(g_saved['params'], g_live['params']):
                id_map[pid] = p
        for pid, st in saved.items():
            p = id_map.get(pid)
            if p is None:
                continue
            live = self.state[p]
            for name in ('exp_avg', 'exp_avg_sq', 'master'):
                if name in st and isinstance(st[name], torch.Tensor):
                    live[name] = st[name].to(device=p.device, dtype=torch.float32).contiguous().view_as(p).clone()
            if 'step' in st:
                live['step'] = int(st['step']) if not isinstance(st['step'], torch.Tensor) else int(st['step'].item())
            if p.dtype == torch.bfloat16:
                if 'master' not in live:
                    live['master'] = p.detach().float().clone()
                else:
                    p.data.copy_(live['master'])          # the model restarts from the fp32 weights of the checkpoint
        self._destroy_plan()
        self._plan_key = None
        self._flat_grads = None

    # ---- public API -----------------------------------------------------------------------------------
    def grad_norm(self, max_norm=0.0, grad_scale=1.0):
        """Global L2 norm of all gradients times `grad_scale`, as a 0-dim device tensor (no sync), and arm the fused
        clipping coefficient `grad_scale * min(1, max_norm / (norm + 1e-6))` for the next step()."""
        _lib.join_wgrads()                # weight gradients a training loop left in flight (ops.defer_wgrad_join)
        _lib.lazy_resolve()               # (no backward since a lazy step: the kept gradients become the zeros zero_grad() promised)
        if not self._ensure_plan():
            return torch.zeros((), device='cuda')
        dev = self._plan_groups[0][1][0].device
        if self._norm_buf is None or self._norm_buf.device != dev:
            self._norm_buf = torch.zeros(2, dtype=torch.float32, device=dev)
        st = _lib.sq_state if self.fold_norm else None
        if st is not None and all(t._version == v for t, v in zip(st['tensors'], st['versions'])) and self._sync_flags(st['ranges'])[1]:
            C.uniter_adamw_grad_norm_ex(self._plan, float(grad_scale), float(max_norm), ptr(self._norm_buf), st['ptr'], st['n'],
                                        _lib.stream_ptr())
        else:
            C.uniter_adamw_grad_norm(self._plan, float(grad_scale), float(max_norm), ptr(self._norm_buf), _lib.stream_ptr())
        self._clip = self._norm_buf
        self._checked_for_step = True
        self._checked_epoch = _lib.grad_attach_epoch()
        self._grads_zeroed = False        # gradients have been produced since a fused zeroing: a later zero_grad() is real work
        return self._norm_buf[0]

    def step(self, closure=None):
        """One optimisation step over every parameter that has a gradient."""
        loss = None
        if closure is not None:
            loss = closure()
        _lib.join_wgrads()
        _lib.lazy_resolve()
        _lib.sq_state = None              # (the gradients are consumed by this step: per-tile sums of an earlier backward are history)
        if not self._ensure_plan(in_step=True):
            return loss           # nothing has a gradient: no-op, like the reference's dummy first step (pretrain.py:261-263)
        self._grads_zeroed = False    # whatever an earlier fused step zeroed has been written again by the backward in between
        hyper = (UniterAdamGroup * len(self._plan_groups))()
        for ig, (gi, plist) in enumerate(self._plan_groups):
            group = self.param_groups[gi]
            for p in plist:
                self.state[p]['step'] += 1
            h = hyper[ig]
            h.lr, h.beta1, h.beta2 = float(group['lr']), float(group['betas'][0]), float(group['betas'][1])
            h.eps, h.weight_decay = float(group['eps']), float(group['weight_decay'])
            h.correct_bias = 1 if group['correct_bias'] else 0
            h.step = int(self.state[plist[0]]['step'])
        clip = self._clip
        self._clip = None
        if self._overlap is not None and not torch.cuda.is_current_stream_capturing():
            arr, n, fuse = self._overlap
            kept = self._sync_flags()[0] if fuse else False
            C.uniter_adamw_step_async(self._plan, hyper, len(self._plan_groups), ptr(clip[1:]) if clip is not None else None,
                                      arr, n, 1 if fuse else 0, _lib.stream_ptr())
            self._grads_zeroed = fuse
            _lib.lazy_undefined = bool(kept)
            _lib.set_async_pending(True)
            return loss
        if self.fuse_zero_grad:
            kept = self._sync_flags()[0]
            C.uniter_adamw_step_zero(self._plan, hyper, len(self._plan_groups), ptr(clip[1:]) if clip is not None else None,
                                     _lib.stream_ptr())
            self._grads_zeroed = True
            _lib.lazy_undefined = bool(kept)      # the kept gradients are stale until the next encoder backward replaces them
            return loss
        C.uniter_adamw_step(self._plan, hyper, len(self._plan_groups), ptr(clip[1:]) if clip is not None else None,
                            _lib.stream_ptr())
        return loss

    def zero_grad(self, set_to_none=False):
        """Zero the gradients IN PLACE by default: the kernels accumulate into `.grad` storages that may be views of
        one flat arena (utils.arena), which must survive the step."""
        _lib.join_wgrads()
        if set_to_none:
            _lib.lazy_resolve()
            self.synchronize()
            self._grads_zeroed = False
            self._flat_grads = None
            return super(AdamW, self).zero_grad(set_to_none=True)
        if self._grads_zeroed:                   # the asynchronous step zeroes every gradient it has consumed
            self._grads_zeroed = False
            return
        flats = getattr(self, '_flat_grads', None)
        if flats is not None and self._plan is not None:
            for f in flats:                      # fast path: everything lives in flat arenas found earlier
                f.zero_()
            return
        all_flat = True
        bases = []
        seen = set()
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is not None:
                    base = p.grad._base if p.grad._base is not None else p.grad
                    key = (base.data_ptr(), base.numel())
                    if getattr(base, '_uniter_flat_grad', False):
                        if key not in seen:
                            seen.add(key)
                            base.zero_()
                            bases.append(base)
                    else:
                        all_flat = False
                        p.grad.zero_()
        self._flat_grads = bases if (all_flat and bases) else None


def overlap_boundaries(model, layers_per_segment=None):
    """Segment boundaries for `AdamW.enable_overlap`: the first parameter of every `layers_per_segment`-th BertLayer (default 1,
    UNITER_AMD_OVERLAP_LAYERS) and of whatever follows the last layer in module order.  With the parameters in a flat arena (utils.arena.flatten_model) these addresses ascend in
    the order the forward pass consumes them.  The arena is REQUIRED: the encoder waits for a layer's segment through the
    address of its fused query/key/value weight only (csrc/encoder.hip), which covers the layer's other parameters — and the
    fused weight itself, a view of the arena rather than a re-stacked copy — only when all of them sit in one arena segment."""
    from ..model.layer import BertLayer
    arena = getattr(model, '_uniter_arena', None)
    if arena is None or not arena.check():
        raise _lib.UniterHipError("overlap_boundaries: the model's parameters must live in a flat arena "
                                  "(uniter_amd.utils.arena.flatten_model) that is still intact; the asynchronous optimizer step "
                                  "orders a layer behind its update through one address per layer")
    layers = [m for m in model.modules() if isinstance(m, BertLayer)]
    if layers_per_segment is None:
        import os
        layers_per_segment = int(os.environ.get("UNITER_AMD_OVERLAP_LAYERS", "1"))
    out = []
    for lay in layers[::max(1, int(layers_per_segment))]:
        ps = [p.data_ptr() for p in lay.parameters()]
        if ps:
            out.append(min(ps))
    if layers:
        last = list(layers[-1].parameters())
        end = max(p.data_ptr() + p.numel() * p.element_size() for p in last)
        after = [p.data_ptr() for p in model.parameters() if p.data_ptr() >= end]
        if after:
            out.append(min(after))
    return out


def clip_grad_norm_(parameters_or_optimizer, max_norm, grad_scale=1.0):
    """Counterpart of `torch.nn.utils.clip_grad_norm_(amp.master_params(optimizer), max_norm)` (pretrain.py:329-331).

    Pass the uniter_amd AdamW instance: the norm is computed by one HIP reduction over all gradients and the
    clipping is deferred into the next `optimizer.step()` (fused).  Returns the (un-clipped) total norm as a
    0-dim device tensor — call `.item()` only when you need the number on the host.  `grad_scale` folds a global
    factor (e.g. 1/world_size of the averaged allreduce) into both the norm and the update."""
    if isinstance(parameters_or_optimizer, AdamW):
        return parameters_or_optimizer.grad_norm(max_norm, grad_scale)
    raise TypeError("clip_grad_norm_ expects the uniter_amd.optim.AdamW instance whose step() will consume the "
                    "fused clipping coefficient")
