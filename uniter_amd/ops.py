"""torch.autograd glue between the drop-in nn.Modules (uniter_amd/model) and the C ABI (include/uniter_hip.h).

Every function here launches HIP kernels of libuniter_hip.so on the current torch stream through ctypes;
PyTorch only provides device memory, streams and the autograd graph at block granularity.  There is no
eager / CPU fallback: non-CUDA or non-bf16 inputs raise.

Parameter gradients are written by the kernels straight into ``param.grad`` (accumulating, like autograd's
own AccumulateGrad would, pretrain.py:298-312 sums micro-step gradients) instead of being returned to
autograd: the wgrad GEMMs fuse the accumulation, which saves one read+write of every gradient per step.
"""
import ctypes
import os
import threading

import torch

from . import _lib
from ._lib import C, UniterEncoderShape, UniterLayerParams, ptr

_BF16 = torch.bfloat16


# ----------------------------------------------------------------------------------------------------
# dropout random stream: (seed, offset) pairs for the Philox generator inside the kernels
# ----------------------------------------------------------------------------------------------------
class _Rng(threading.local):
    def __init__(self):
        self.seed = None
        self.offset = 0


_rng = _Rng()

def manual_seed(seed):
    """Reset the dropout stream (utils.misc.set_random_seed calls this)."""
    _rng.seed = int(seed) & 0x7FFFFFFFFFFFFFFF
    _rng.offset = 0


def _next_offsets(n):
    if _rng.seed is None:
        _rng.seed = int(torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF
    off = _rng.offset
    _rng.offset += int(n)
    return _rng.seed, off


# ----------------------------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------------------------
def _check_dev(t, name, dtype=_BF16):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.UniterHipError(
            "%s must be a CUDA(HIP) tensor: the UNITER encoder path only exists as gfx950 kernels "
            "(no CPU / PyTorch fallback)" % name)
    if dtype is not None and t.dtype != dtype:
        raise _lib.UniterHipError("%s must be %s, got %s (cast the model with .bfloat16())" % (name, dtype, t.dtype))
    return t


def params_ready(*tensors):
    """Order the current stream behind the segments of a pending asynchronous optimizer step that hold these parameters
    (AdamW.enable_overlap).  Free when no such step is pending."""
    if not _lib.async_pending():
        return
    st = _lib.stream_ptr()
    for t in tensors:
        if t is not None:
            C.uniter_params_wait(t.data_ptr(), st)


def attach_grad_slot(p):
    """Make the parameter's slot of the flat gradient arena (utils.arena) its `.grad` — zeroed.  The slot is all zeros when
    the arena is built, but `Module.zero_grad()` / `Optimizer.zero_grad(set_to_none=True)` only drop the reference and leave
    the last gradient in the slot: handing that back would double-count it, so a slot that has been out before is cleared
    when it is attached again.  Returns False if the parameter has no slot."""
    slot = getattr(p, '_uniter_grad_slot', None)
    if slot is None:
        return False
    if getattr(p, '_uniter_slot_used', False):
        with torch.no_grad():
            slot.zero_()
    p._uniter_slot_used = True
    p.grad = slot
    _lib.note_grad_attached()
    return True


def ensure_grad(p):
    """param.grad as a zero-initialised contiguous tensor the kernels can accumulate into."""
    if p.grad is None:
        if not attach_grad_slot(p):
            p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
            _lib.note_grad_attached()
    elif not p.grad.is_contiguous() or p.grad.dtype != p.dtype:
        raise _lib.UniterHipError("param.grad must be contiguous and of the parameter's dtype")
    return p.grad


_scratch_cache = {}
_WGRAD_STAGE = os.environ.get("UNITER_AMD_WGRAD_STAGE", "1") != "0"     # 0: never register a stage (per-layer weight-gradient launches)


class DeferWgradJoin(object):
    """Set as `model.uniter.encoder.grad_ready_hook` by a TRAINING LOOP (uniter_amd/train.py) of a single process: encoder
    backward calls then return WITHOUT making the compute stream wait for the library's weight-gradient stream, so that what
    autograd runs next (the embedding backward) overlaps the deferred weight-gradient launch.  Contract: nothing reads or
    writes an encoder weight gradient until `_lib.join_wgrads()` has run on the stream — uniter_amd.optim.AdamW does it at the
    start of grad_norm() / step() / zero_grad().  Without it `.grad` is complete on the compute stream when backward() returns,
    as in PyTorch.  (A data-parallel GradientReducer installs its own hook and joins per bucket.)"""
    ready_layers = frozenset()
    joins_side_stream = False
    defer_wgrad_join = True

    def __call__(self, layer_index):
        return None


def _scratch(key, nbytes, device):
    buf = _scratch_cache.get(key)
    if buf is None or buf.numel() < nbytes or buf.device != device:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _scratch_cache[key] = buf
    return buf


def _dummy_grad_like(t, pool):
    """Scratch gradient buffer for a frozen parameter (result is discarded)."""
    g = pool.get(t.numel())
    if g is None:
        g = torch.empty(t.numel(), dtype=t.dtype, device=t.device)
        pool[t.numel()] = g
    return g


# ----------------------------------------------------------------------------------------------------
# encoder stack
# ----------------------------------------------------------------------------------------------------
class LayerView(object):
    """The 12 (+12 gradient) device pointers of one BertLayer, in UniterLayerParams order."""

    NAMES = ("wqkv", "bqkv", "wo", "bo", "ln1_g", "ln1_b", "w1", "b1", "w2", "b2", "ln2_g", "ln2_b")

    def __init__(self, layer):
        self.layer = layer

    def params(self):
        """12 parameter tensors; wqkv / bqkv are the fused [3H,H] / [3H] storages."""
        lay = self.layer
        att = lay.attention.self
        wqkv, bqkv = att.fused_qkv()
        return [wqkv, bqkv,
                lay.attention.output.dense.weight, lay.attention.output.dense.bias,
                lay.attention.output.LayerNorm.weight, lay.attention.output.LayerNorm.bias,
                lay.intermediate.dense.weight, lay.intermediate.dense.bias,
                lay.output.dense.weight, lay.output.dense.bias,
                lay.output.LayerNorm.weight, lay.output.LayerNorm.bias]

    def grads(self, pool):
        lay = self.layer
        att = lay.attention.self
        gw, gb = att.fused_qkv_grad()
        out = [gw, gb]
        for p in (lay.attention.output.dense.weight, lay.attention.output.dense.bias,
                  lay.attention.output.LayerNorm.weight, lay.attention.output.LayerNorm.bias,
                  lay.intermediate.dense.weight, lay.intermediate.dense.bias,
                  lay.output.dense.weight, lay.output.dense.bias,
                  lay.output.LayerNorm.weight, lay.output.LayerNorm.bias):
            out.append(ensure_grad(p) if p.requires_grad else _dummy_grad_like(p, pool))
        return out


def _layer_row(lay, with_grads, pool):
    """Device pointers of one BertLayer (12 params [+ 12 grads]), cached on the module.

    Building a row costs ~40 tensor-metadata calls, so it is cached and re-validated cheaply: parameter storages
    can only move through nn.Module._apply (.to / .bfloat16 — BertLayer._apply drops the cache) or through
    ParamArena (which also drops it); gradients are re-created when someone sets them to None, which is caught by
    an identity check of every cached grad tensor."""
    cache = getattr(lay, "_ptr_cache", None)
    if cache is not None:
        ps, pptrs, gs, gptrs, owners = cache
        ok = ps[2].data_ptr() == pptrs[2] and lay.attention.self.query.weight.data_ptr() == pptrs[0]
        if ok and with_grads:
            if gs is None:
                ok = False
            else:
                for p, g in owners:
                    if p.grad is not g:
                        ok = False
                        break
        if ok:
            return pptrs, gptrs, (ps, gs)
    view = LayerView(lay)
    ps = view.params()
    for t in ps:
        _check_dev(t, "encoder parameter")
        if not t.is_contiguous():
            raise _lib.UniterHipError("encoder parameters must be contiguous")
    pptrs = [t.data_ptr() for t in ps]
    gs = gptrs = None
    owners = []
    if with_grads:
        gs = view.grads(pool)
        gptrs = [t.data_ptr() for t in gs]
        att = lay.attention.self
        for m in (att.query, att.key, att.value, lay.attention.output.dense, lay.attention.output.LayerNorm,
                  lay.intermediate.dense, lay.output.dense, lay.output.LayerNorm):
            for p in (m.weight, m.bias):
                if p.requires_grad:
                    owners.append((p, p.grad))
    lay._ptr_cache = (ps, pptrs, gs, gptrs, owners)
    return pptrs, gptrs, (ps, gs)


_table_cache = {}


def _layer_table(layers, with_grads):
    """ctypes array of UniterLayerParams for a list of BertLayers.  Filling it costs ~25 ctypes attribute stores per layer,
    which sat on the critical path at the start of every backward pass (the GPU idles until the first launch), so the filled
    array is kept and reused while every row's device pointers are unchanged."""
    n = len(layers)
    pool = {}
    rows = [_layer_row(lay, with_grads, pool) for lay in layers]
    sig = tuple(id(r[0]) for r in rows) + tuple(id(r[1]) for r in rows)          # the cached pointer lists themselves
    key = (id(layers[0]) if n else 0, n, with_grads)
    hit = _table_cache.get(key)
    if hit is not None and hit[0] == sig:
        return hit[1], hit[2]
    table = (UniterLayerParams * n)()
    keep = []
    names = LayerView.NAMES
    for i, (pptrs, gptrs, alive) in enumerate(rows):
        keep.append(alive)
        row = table[i]
        for name, v in zip(names, pptrs):
            setattr(row, name, v)
        if with_grads:
            for name, v in zip(names, gptrs):
                setattr(row, "g_" + name, v)
    _table_cache[key] = (sig, table, keep, [r[0] for r in rows], [r[1] for r in rows])   # (keeps the id()s alive)
    return table, keep


# Folded gradient norm (optim.AdamW.fold_norm): opt-in (UNITER_AMD_FOLD_NORM=1 or set_fold_norm(True)).  Measured a wash on the c2 step:
# the gradient reduction gets 20 us shorter and AdamW 29 us longer, because the reduction was also what brought the gradients into the
# Infinity Cache for the update that follows it (profiles/r06_fold_norm_ab.txt).
_FOLD_NORM = os.environ.get("UNITER_AMD_FOLD_NORM", "0") == "1"


def set_fold_norm(enable):
    """Ask encoder backward calls for the per-tile sums of squares of their weight gradients (see optim.AdamW.fold_norm)."""
    global _FOLD_NORM
    _FOLD_NORM = bool(enable)
_range_cache = {}


def _layer_grad_ranges(layers, keep):
    """(frozenset of (first byte, byte length), [tensors]) of the parameter-gradient storages an encoder backward over `layers`
    writes — what AdamW.lazy_zero may leave un-zeroed.  `keep` is _layer_table(layers, True)[1]; cached per table."""
    key = (id(layers[0]) if layers else 0, len(layers), id(keep))
    hit = _range_cache.get(key)
    if hit is not None and hit[0] is keep:
        return hit[1], hit[2]
    tensors = [g for _, gs in keep for g in gs]
    ranges = frozenset((g.data_ptr(), g.numel() * g.element_size()) for g in tensors)
    _range_cache.clear()
    _range_cache[key] = (keep, ranges, tensors)
    return ranges, tensors


def _shape(cfg_like, B, L, training):
    s = UniterEncoderShape()
    s.B, s.L, s.H, s.heads, s.I = B, L, cfg_like["H"], cfg_like["heads"], cfg_like["I"]
    s.p_hidden = float(cfg_like["p_hidden"]) if training else 0.0
    s.p_attn = float(cfg_like["p_attn"]) if training else 0.0
    s.ln_eps = float(cfg_like["ln_eps"])
    s.hidden_act = int(cfg_like.get("act", 0))
    s.training = 1 if training else 0
    return s


_autotuned = set()


def _layer_gemm_shapes(s):
    """(kind, M, N, K) of the tuned launches of one BertLayer as the C ABI sees them: kind 0 forward and 1 dgrad for the
    four linear layers, kind 3 = the grouped launch of the four weight gradients, keyed (T, sum N, sum K)."""
    T, H, I = int(s.B) * int(s.L), int(s.H), int(s.I)
    shapes = [(kind, T, n, k) for kind in range(2) for n, k in ((3 * H, H), (H, H), (I, H), (H, I))]
    shapes.append((3, T, 5 * H + I, 3 * H + I))
    return shapes


FACTORY_TUNE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "gfx950.json")


def _load_tune_cache(path, s):
    """Install tile choices saved by an earlier process.  True if every GEMM of this shape was covered."""
    import json
    import os
    if not path or not os.path.exists(path):
        return False
    try:
        saved = json.load(open(path))
    except (OSError, ValueError):
        return False
    if saved.get("n_tiles", C.uniter_gemm_tile_count()) != C.uniter_gemm_tile_count():
        return False                                   # tile indices of another build of the GEMM family
    table = {(e["kind"], e["M"], e["N"], e["K"]): (e["cfg"], e["splits"]) for e in saved.get("gemm", [])}
    shapes = _layer_gemm_shapes(s)
    if any(k not in table for k in shapes):
        return False
    try:
        for k in shapes:
            C.uniter_gemm_set_tuned(k[0], k[1], k[2], k[3], table[k][0], table[k][1])
    except _lib.UniterHipError:
        return False                                   # a choice this build rejects: tune afresh
    for k, v in table.items():                          # auxiliary launches saved with them (task-head groups)
        if k not in shapes:
            try:
                C.uniter_gemm_set_tuned(k[0], k[1], k[2], k[3], v[0], v[1])
            except _lib.UniterHipError:
                pass
    return True


_aux_keys = []          # (kind, M, N, K) of tuned launches outside the encoder layers (grouped head weight gradients)


def _save_tune_cache(path, s):
    import json
    entries = []
    if os.path.exists(path):
        try:
            entries = json.load(open(path)).get("gemm", [])
        except (OSError, ValueError):
            entries = []
    have = {(e["kind"], e["M"], e["N"], e["K"]) for e in entries}
    out = (ctypes.c_int32 * 2)()
    for k in (_layer_gemm_shapes(s) if s is not None else []) + list(_aux_keys):
        C.uniter_gemm_tuned_choice(k[0], k[1], k[2], k[3], out)
        if out[0] >= 0 and k not in have:
            entries.append({"kind": k[0], "M": k[1], "N": k[2], "K": k[3], "cfg": int(out[0]), "splits": int(out[1])})
    tmp = "%s.%d.tmp" % (path, os.getpid())
    with open(tmp, "w") as f:
        json.dump({"n_tiles": int(C.uniter_gemm_tile_count()), "gemm": entries}, f)
    os.replace(tmp, path)


def _maybe_autotune(s, training):
    """First training-mode call of a new (B, L, H, I): let the library time its GEMM tile shapes for exactly these sizes
    (one-off, synchronous, ~0.1 s).  UNITER_AMD_AUTOTUNE=0 keeps the built-in cost model (run-to-run identical tiles);
    UNITER_AMD_TUNE_CACHE=<file> saves the winners and reuses them in later processes (no sweep, same kernels)."""
    key = (int(s.B), int(s.L), int(s.H), int(s.I))
    if key in _autotuned:
        return
    _autotuned.add(key)
    import os
    if os.environ.get("UNITER_AMD_AUTOTUNE", "1") == "0" or not training:
        return
    if torch.cuda.is_current_stream_capturing():
        return
    cache = os.environ.get("UNITER_AMD_TUNE_CACHE", "")
    if cache and _load_tune_cache(cache, s):
        return
    # tile choices shipped with the package for the reference's standard shapes (the same tuner's output, picked as the
    # best of several independent runs by scripts/make_factory_tune.py); UNITER_AMD_FACTORY_TUNE=0 ignores them
    if not cache and os.environ.get("UNITER_AMD_FACTORY_TUNE", "1") != "0" and _load_tune_cache(FACTORY_TUNE, s):
        return
    C.uniter_encoder_autotune(ctypes.byref(s), _lib.stream_ptr())
    if cache:
        _save_tune_cache(cache, s)


class _EncoderFn(torch.autograd.Function):
    """All BertLayers of a UniterEncoder in one autograd node (model/model.py:282-292)."""

    @staticmethod
    def forward(ctx, x, mask_bias, layers, cfg_like, training, need_all, hook, packed, *params):
        # dense: x [B, L, H], mask_bias [B, L];  packed: x [T, H], packed = (cu_seqlens int32 [B+1] on device, B, Lmax)
        if packed is None:
            B, L, H = x.shape
            s = _shape(cfg_like, B, L, training)
            out_shape = (B, L, H)
        else:
            cu, B, L = packed
            H = x.shape[1]
            s = _shape(cfg_like, B, L, training)
            s.total_tokens = x.shape[0]
            s.cu_seqlens = cu.data_ptr()
            out_shape = (x.shape[0], H)
        ctx.packed = packed
        n = len(layers)
        act_bytes = C.uniter_encoder_layer_act_bytes(ctypes.byref(s))
        if act_bytes == 0:
            raise _lib.UniterHipError("bad encoder shape: " + _lib.load().uniter_hip_last_error().decode())
        out_off = C.uniter_encoder_layer_out_offset(ctypes.byref(s))
        if packed is None:
            _maybe_autotune(s, training)
        else:
            # the token count changes every step: sweep once per power-of-two bucket on a dense stand-in shape; the
            # library then reuses the nearest tuned M for the actual count (gemm.hip tuned_lookup)
            bucket = 1 << max(int(x.shape[0]) - 1, 1).bit_length()
            _maybe_autotune(_shape(cfg_like, max(bucket // 64, 1), 64, training), training)
        acts = torch.empty(n * act_bytes, dtype=torch.uint8, device=x.device)
        table, keep = _layer_table(layers, with_grads=False)
        seed, off = _next_offsets(n * 8) if training else (0, 0)
        xc = x.contiguous()
        # the shared scratch buffer also holds the row-block flags of the overlapped kernel chain (include/uniter_hip.h, ABI v7)
        scratch = _scratch(("enc", x.device.index), C.uniter_encoder_scratch_bytes(ctypes.byref(s)), x.device)
        C.uniter_encoder_forward(ctypes.byref(s), table, 0, n, ptr(xc), None if packed is not None else ptr(mask_bias),
                                 ptr(acts), ptr(scratch), seed, off, _lib.stream_ptr())
        _lib.set_async_pending(False)          # the call ends with uniter_params_wait_all on this stream

        n_rows = out_shape[0] if packed is not None else B * L

        def layer_out(l):
            o = l * act_bytes + out_off
            return acts[o:o + n_rows * H * 2].view(_BF16).view(*out_shape)

        ctx.layers, ctx.cfg_like, ctx.s = layers, cfg_like, s
        ctx.acts, ctx.act_bytes = acts, act_bytes
        ctx.seed, ctx.off = seed, off
        ctx.need_all, ctx.hook = need_all, hook
        ctx.save_for_backward(xc, mask_bias)
        del keep
        if need_all:
            return tuple(layer_out(l) for l in range(n))
        return layer_out(n - 1)

    @staticmethod
    def backward(ctx, *grads):
        xc, mask_bias = ctx.saved_tensors
        layers, s = ctx.layers, ctx.s
        n = len(layers)
        if not s.training:
            raise _lib.UniterHipError("backward through an encoder forward that ran in eval mode / under no_grad")
        out_shape = tuple(xc.shape)
        H = out_shape[-1]
        scr_bytes = C.uniter_encoder_scratch_bytes(ctypes.byref(s))
        scratch = _scratch(("enc", xc.device.index), scr_bytes, xc.device)
        table, keep = _layer_table(layers, with_grads=True)
        # lazy zero_grad (optim.AdamW.lazy_zero): after a fused step that left exactly these gradient storages alone, this backward
        # REPLACES their contents (uniter_encoder_set_grad_overwrite); anything else that is marked undefined is zeroed first
        g_ranges, g_tensors = _layer_grad_ranges(layers, keep)
        overwrite = False
        if _lib.lazy_undefined:
            if g_ranges == _lib.lazy_ranges:
                overwrite = True
            else:
                for t in _lib.lazy_tensors:
                    t.zero_()
            _lib.lazy_undefined = False
        _lib.lazy_ranges, _lib.lazy_tensors = g_ranges, g_tensors
        _lib.sq_state = None
        # split the stack where an intermediate layer output received a gradient of its own
        if ctx.need_all:
            extra = [g for g in grads]
        else:
            extra = [None] * (n - 1) + [grads[0]]
        if extra[n - 1] is None:
            extra[n - 1] = torch.zeros(out_shape, dtype=_BF16, device=xc.device)
        dy = extra[n - 1].contiguous()
        end = n
        st = _lib.stream_ptr()
        act_bytes = ctx.act_bytes
        out_off = C.uniter_encoder_layer_out_offset(ctypes.byref(s))
        dx = torch.empty(out_shape, dtype=_BF16, device=xc.device)
        cuts = set(l for l in range(n - 1) if extra[l] is not None)
        hook = ctx.hook
        # a hook that joins the library's weight-gradient stream itself (uniter_encoder_side_join on its own stream) lets
        # the ranges below follow each other without serialising the two streams at every range boundary
        defer = hook is not None and getattr(hook, "joins_side_stream", False)
        defer_only = False
        while end > 0:
            inner = [c for c in cuts if c + 1 < end]
            begin = (max(inner) + 1) if inner else 0
            if hook is not None:
                # hand control back where the hook has work (a gradient bucket completes), else after every layer
                ready = getattr(hook, "ready_layers", None)
                if ready is None:
                    begin = end - 1
                else:
                    stops = [l for l in ready if begin < l < end]
                    if stops:
                        begin = max(stops)
            if begin == 0:
                x_in = ptr(xc)
            else:
                x_in = ctx.acts.data_ptr() + (begin - 1) * act_bytes + out_off
            # the library's flag is per thread and sticky, and this is the (shared) autograd thread: state it on EVERY call, so
            # that a backward without a hook — a second model, a removed hook — ends with the join it relies on
            if defer:
                want_defer = begin > 0
            else:
                defer_only = bool(getattr(hook, "defer_wgrad_join", False))
                want_defer = defer_only
            C.uniter_encoder_defer_side_join(1 if want_defer else 0)
            # gradient buckets of a data-parallel reducer (one call, flags per bucket): per thread and sticky, so stated every time
            gb = getattr(hook, "grad_buckets", None)
            C.uniter_encoder_set_grad_buckets(int(gb()) if callable(gb) and begin == 0 and end == n else 0)
            if defer or defer_only:
                # the call returns before the deferred launch (which reads the saved activations, the range's input and this
                # call's dy) has run: keep them alive until the weight-gradient stream is joined (_lib.join_wgrads)
                _lib.hold_until_wgrad_join(ctx.acts, xc, dy)
            # deferred weight gradients (include/uniter_hip.h): one set of dy buffers per layer of this call; twice that when
            # the stack is cut into several calls, so that consecutive ranges alternate halves of the stage
            n_call = end - begin
            st_bytes = C.uniter_encoder_wgrad_stage_bytes(ctypes.byref(s), n_call) * (1 if (begin == 0 and end == n) else 2)
            if st_bytes > 0 and _WGRAD_STAGE:
                stage = _scratch(("enc_wgrad_stage", xc.device.index), st_bytes, xc.device)
                C.uniter_encoder_set_wgrad_stage(ptr(stage), st_bytes)
            else:
                C.uniter_encoder_set_wgrad_stage(None, 0)       # (the registration is per thread and outlives a call)
            C.uniter_encoder_set_grad_overwrite(1 if overwrite else 0)       # (per call: the library consumes it)
            # folded gradient norm: only a whole stack in one call without a reducer hook leaves usable per-tile sums
            want_sq = _FOLD_NORM and (hook is None or type(hook) is DeferWgradJoin) and begin == 0 and end == n
            C.uniter_encoder_set_grad_sq(1 if want_sq else 0)                # (per thread and sticky: stated every time)
            C.uniter_encoder_backward(ctypes.byref(s), table, begin, end, x_in,
                                      None if ctx.packed is not None else ptr(mask_bias), ptr(dy), ptr(dx),
                                      ptr(ctx.acts), ptr(scratch), ctx.seed, ctx.off, st)
            if want_sq:
                sq_ptr, sq_n = ctypes.c_void_p(), ctypes.c_int32()
                C.uniter_encoder_last_grad_sq(ctypes.byref(sq_ptr), ctypes.byref(sq_n))
                if sq_n.value > 0 and sq_ptr.value:
                    wts = [gs[k] for _, gs in keep for k in (0, 2, 6, 8)]           # wqkv, wo, w1, w2 gradients (LayerView.NAMES)
                    _lib.sq_state = dict(ptr=sq_ptr.value, n=sq_n.value, tensors=wts, versions=[t._version for t in wts],
                                         ranges=frozenset((t.data_ptr(), t.numel() * t.element_size()) for t in wts))
            if hook is not None:
                for l in range(end - 1, begin - 1, -1):
                    hook(l)
            end = begin
            if end > 0:
                # dx is the gradient w.r.t. the output of layer end-1
                if (end - 1) in cuts:
                    dx = dx + extra[end - 1]
                dy, dx = dx, torch.empty_like(dx)
        del keep
        return (dx if ctx.needs_input_grad[0] else None, None, None, None, None, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 8)


def encoder_forward(layers, x, mask_bias, cfg_like, training, need_all=False, hook=None):
    """Run a list of BertLayer modules on x [B,L,H] bf16.  mask_bias: [B,L] fp32 additive key mask."""
    _check_dev(x, "hidden_states")
    _check_dev(mask_bias, "attention_mask", torch.float32)
    if x.dim() != 3:
        raise _lib.UniterHipError("hidden_states must be [B, L, H]")
    B, L, H = x.shape
    if mask_bias.numel() != B * L:
        raise _lib.UniterHipError("attention_mask must have B*L elements (got %d for B=%d L=%d)" % (mask_bias.numel(), B, L))
    mask_bias = mask_bias.reshape(B, L).contiguous()
    track = torch.is_grad_enabled() and training
    if not track:
        with torch.no_grad():
            return _EncoderFn.apply(x, mask_bias, list(layers), cfg_like, False, need_all, None, None)
    # one parameter is enough to make the output require grad (gradients are written straight into .grad)
    anchor = layers[0].output.dense.weight
    if not anchor.requires_grad:
        anchor = next((p for lay in layers for p in lay.parameters() if p.requires_grad), None)
    extra = () if anchor is None else (anchor,)
    return _EncoderFn.apply(x, mask_bias, list(layers), cfg_like, True, need_all, hook, None, *extra)


def encoder_forward_packed(layers, x, cu_seqlens, n_examples, max_len, cfg_like, training, need_all=False, hook=None):
    """Padding-free form: x [T, H] bf16 holds only real tokens, example b = rows
    cu_seqlens[b] .. cu_seqlens[b+1]-1 (int32 [n_examples+1] on the device), max_len = longest example."""
    _check_dev(x, "hidden_states")
    _check_dev(cu_seqlens, "cu_seqlens", torch.int32)
    if x.dim() != 2 or cu_seqlens.numel() != n_examples + 1:
        raise _lib.UniterHipError("packed hidden_states must be [T, H] and cu_seqlens [n_examples + 1]")
    packed = (cu_seqlens.contiguous(), int(n_examples), int(max_len))
    dummy_mask = cu_seqlens                      # placeholder argument of the autograd node; never handed to a kernel
    track = torch.is_grad_enabled() and training
    if not track:
        with torch.no_grad():
            return _EncoderFn.apply(x.contiguous(), dummy_mask, list(layers), cfg_like, False, need_all, None, packed)
    anchor = layers[0].output.dense.weight
    if not anchor.requires_grad:
        anchor = next((p for lay in layers for p in lay.parameters() if p.requires_grad), None)
    extra = () if anchor is None else (anchor,)
    return _EncoderFn.apply(x.contiguous(), dummy_mask, list(layers), cfg_like, True, need_all, hook, packed, *extra)


# ----------------------------------------------------------------------------------------------------
# additive attention mask (model/model.py:342-345)
# ----------------------------------------------------------------------------------------------------
def mask_bias(attention_mask):
    m = attention_mask
    if not m.is_cuda:
        raise _lib.UniterHipError("attention_mask must be a CUDA tensor")
    m = m.to(torch.int64).contiguous()
    out = torch.empty(m.shape, dtype=torch.float32, device=m.device)
    C.uniter_mask_bias(ptr(m), ptr(out), m.numel(), _lib.stream_ptr())
    return out


def wgrad_group(dys, lddys, xs_, ldxs, dws, dbs, M, Ns, Ks, training=True):
    """Up to four weight gradients dw_q[N_q, K_q] += dy_q[M, N_q]^T x_q[M, K_q] (and db_q += column sums of dy_q) over the
    same M rows in one launch (uniter_gemm_wgrad_group).  All arguments are raw device addresses / row strides."""
    n = len(dys)
    PA, IA = ctypes.c_void_p * n, ctypes.c_int64 * n
    Na, Ka = IA(*Ns), IA(*Ks)
    st = _lib.stream_ptr()
    key = (3, int(M), int(sum(Ns)), int(sum(Ks)))
    if key not in _aux_keys:
        _aux_keys.append(key)
        out = (ctypes.c_int32 * 2)()
        C.uniter_gemm_tuned_choice(key[0], key[1], key[2], key[3], out)
        if (out[0] < 0 and training and os.environ.get("UNITER_AMD_AUTOTUNE", "1") != "0"
                and not torch.cuda.is_current_stream_capturing()):
            C.uniter_gemm_wgrad_group_autotune(n, M, Na, Ka, st)          # one-off, times the legal tiles of this group
            cache = os.environ.get("UNITER_AMD_TUNE_CACHE", "")
            if cache:
                _save_tune_cache(cache, None)
    C.uniter_gemm_wgrad_group(n, PA(*dys), IA(*lddys), PA(*xs_), IA(*ldxs), PA(*dws), PA(*dbs), M, Na, Ka, 1, st)


_HEAD_GROUP = True          # (False: one launch per problem — the A/B of round 4, kept as a module attribute for tests)


def fwd_group(xs_, ldxs, ws, biases, ys, ldys, M, Ns, K):
    """Up to four y_q[M, N_q] = x_q[M, K] w_q^T + b_q over the same rows in one launch (uniter_gemm_bias_fwd_group).  Raw device
    addresses / row strides (0 = dense)."""
    n = len(xs_)
    if not _HEAD_GROUP:                                    # (A/B switch: one launch per problem)
        for q in range(n):
            C.uniter_gemm_bias_fwd_ld(xs_[q], ldxs[q] or K, ws[q], biases[q], ys[q], ldys[q] or Ns[q], M, Ns[q], K, _lib.stream_ptr())
        return
    PA, IA = ctypes.c_void_p * n, ctypes.c_int64 * n
    C.uniter_gemm_bias_fwd_group(n, PA(*xs_), IA(*ldxs), PA(*ws), PA(*biases), PA(*ys), IA(*ldys), M, IA(*Ns), K, _lib.stream_ptr())


def dgrad_group(dys, lddys, ws, resids, dxs, M, Ns, K):
    """Up to four dx_q[M, K] = dy_q[M, N_q] w_q[N_q, K] (+ resid_q) in one launch (uniter_gemm_dgrad_group)."""
    n = len(dys)
    if not _HEAD_GROUP:
        for q in range(n):
            C.uniter_gemm_dgrad_ld(dys[q], lddys[q] or Ns[q], ws[q], resids[q], dxs[q], M, Ns[q], K, _lib.stream_ptr())
        return
    PA, IA = ctypes.c_void_p * n, ctypes.c_int64 * n
    C.uniter_gemm_dgrad_group(n, PA(*dys), IA(*lddys), PA(*ws), PA(*resids), PA(*dxs), M, IA(*Ns), K, _lib.stream_ptr())


# ----------------------------------------------------------------------------------------------------
# The sub-modules of a BertLayer called on their own (model/layer.py:47-156).  UniterEncoder / BertLayer run the fused stack
# (uniter_encoder_forward); third-party code that calls BertSelfAttention, BertSelfOutput, BertAttention, BertIntermediate or
# BertOutput directly gets the same kernels one operation at a time through these autograd nodes.  Parameter gradients are
# accumulated into `.grad` as everywhere in this package.
# ----------------------------------------------------------------------------------------------------
def _grad_or_dummy(p, pool):
    if p is None:
        return None
    return ensure_grad(p) if p.requires_grad else _dummy_grad_like(p, pool)


def _wgrad_into(dy, x, gw, gb, M, N, K):
    """gw[N, K] += dy[M, N]^T x[M, K]; gb[N] += column sums of dy (gb may be None)."""
    wsb = C.uniter_gemm_wgrad_workspace_bytes(M, N, K)
    ws = _scratch(("sub_wgrad", dy.device.index), wsb, dy.device)
    C.uniter_gemm_wgrad(ptr(dy), ptr(x), ptr(gw), ptr(gb), M, N, K, 1, ptr(ws), wsb, _lib.stream_ptr())


class _SelfAttentionFn(torch.autograd.Function):
    """BertSelfAttention.forward (model/layer.py:75-101): fused [3H, H] projection, then softmax(QK^T / sqrt(dh) + mask) V per head."""

    @staticmethod
    def forward(ctx, x, mask_bias, att, p_attn, *anchor):
        B, L, H = x.shape
        T = B * L
        heads = att.num_attention_heads
        st = _lib.stream_ptr()
        xc = x.contiguous()
        wqkv, bqkv = att.fused_qkv()
        qkv = torch.empty(T, 3 * H, dtype=_BF16, device=x.device)
        out = torch.empty(B, L, H, dtype=_BF16, device=x.device)
        lse = torch.empty(B * heads * L, dtype=torch.float32, device=x.device)
        C.uniter_gemm_bias_fwd(ptr(xc), ptr(wqkv), ptr(bqkv), ptr(qkv), T, 3 * H, H, st)
        seed, off = _next_offsets(1) if p_attn > 0.0 else (0, 0)
        C.uniter_attention_fwd(ptr(qkv), ptr(mask_bias), ptr(out), ptr(lse), B, L, heads, p_attn, seed, off, st)
        ctx.att, ctx.p, ctx.seed, ctx.off = att, p_attn, seed, off
        ctx.save_for_backward(xc, mask_bias, qkv, out, lse)
        return out

    @staticmethod
    def backward(ctx, dout):
        xc, mask_bias, qkv, out, lse = ctx.saved_tensors
        att = ctx.att
        B, L, H = xc.shape
        T = B * L
        heads = att.num_attention_heads
        st = _lib.stream_ptr()
        dout = dout.contiguous()
        dqkv = torch.empty(T, 3 * H, dtype=_BF16, device=xc.device)
        awb = C.uniter_attention_bwd_workspace_bytes(B, L, heads)
        aws = _scratch(("sub_attn", xc.device.index), max(awb, 16), xc.device)
        C.uniter_attention_bwd_ws(ptr(qkv), ptr(mask_bias), None, ptr(out), ptr(lse), ptr(dout), ptr(dqkv), B, L, heads,
                                  ctx.p, ctx.seed, ctx.off, ptr(aws), awb, st)
        wqkv, _ = att.fused_qkv()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(xc)
            C.uniter_gemm_dgrad(ptr(dqkv), ptr(wqkv), None, ptr(dx), T, 3 * H, H, st)
        if any(m.weight.requires_grad for m in (att.query, att.key, att.value)):
            gw, gb = att.fused_qkv_grad()
            _wgrad_into(dqkv, xc, gw, gb, T, 3 * H, H)
        return (dx, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 4)


class _DenseDropResLnFn(torch.autograd.Function):
    """BertSelfOutput / BertOutput.forward (model/layer.py:111-115, 152-156): LayerNorm(dropout(dense(h)) + input)."""

    @staticmethod
    def forward(ctx, h, inp, mod, p, *anchor):
        K = h.shape[-1]
        H = inp.shape[-1]
        hc = h.contiguous().view(-1, K)
        ic = inp.contiguous().view(-1, H)
        T = hc.shape[0]
        st = _lib.stream_ptr()
        z = torch.empty(T, H, dtype=_BF16, device=h.device)
        seed, off = _next_offsets(1) if p > 0.0 else (0, 0)
        C.uniter_gemm_bias_dropout_residual_fwd(ptr(hc), ptr(mod.dense.weight), ptr(mod.dense.bias), ptr(ic), ptr(z), T, H, K, p, seed, off, st)
        y, mean, rstd = _ln_fwd(z, mod.LayerNorm.weight, mod.LayerNorm.bias, 1e-12, 0.0, 0, 0)
        ctx.mod, ctx.p, ctx.seed, ctx.off, ctx.shape = mod, p, seed, off, tuple(inp.shape)
        ctx.save_for_backward(hc, z, mean, rstd)
        return y.view(inp.shape)

    @staticmethod
    def backward(ctx, dy):
        hc, z, mean, rstd = ctx.saved_tensors
        mod = ctx.mod
        T, K = hc.shape
        H = z.shape[1]
        st = _lib.stream_ptr()
        dy = dy.contiguous().view(T, H)
        dz = torch.empty_like(z)                      # gradient of the residual sum: goes to `input` as it is
        dd = torch.empty_like(z)                      # ... and through the dropout mask to the dense branch
        pool = {}
        ln = mod.LayerNorm
        wsb = C.uniter_layernorm_bwd_workspace_bytes(T, H)
        ws = _scratch(("ln", z.device.index), wsb, z.device)
        C.uniter_layernorm_bwd(ptr(dy), None, ptr(z), ptr(mean), ptr(rstd), ptr(ln.weight), ptr(dz), ptr(dd),
                               ptr(_grad_or_dummy(ln.weight, pool)), ptr(_grad_or_dummy(ln.bias, pool)),
                               ptr(_grad_or_dummy(mod.dense.bias, pool)), T, H, 1, ctx.p, ctx.seed, ctx.off, 0, ptr(ws), wsb, st)
        dh = None
        if ctx.needs_input_grad[0]:
            dh = torch.empty_like(hc)
            C.uniter_gemm_dgrad(ptr(dd), ptr(mod.dense.weight), None, ptr(dh), T, H, K, st)
            dh = dh.view(ctx.shape[:-1] + (K,))
        if mod.dense.weight.requires_grad:
            _wgrad_into(dd, hc, ensure_grad(mod.dense.weight), None, T, H, K)
        dinp = dz.view(ctx.shape) if ctx.needs_input_grad[1] else None
        return (dh, dinp, None, None) + (None,) * (len(ctx.needs_input_grad) - 4)


class _DenseGeluFn(torch.autograd.Function):
    """BertIntermediate.forward (model/layer.py:139-142) with the exact erf GELU: one GEMM with a fused epilogue."""

    @staticmethod
    def forward(ctx, x, mod, *anchor):
        H = x.shape[-1]
        I = mod.dense.weight.shape[0]
        xc = x.contiguous().view(-1, H)
        T = xc.shape[0]
        u = torch.empty(T, I, dtype=_BF16, device=x.device)
        g = torch.empty(T, I, dtype=_BF16, device=x.device)
        C.uniter_gemm_bias_gelu_fwd(ptr(xc), ptr(mod.dense.weight), ptr(mod.dense.bias), ptr(u), ptr(g), T, I, H, _lib.stream_ptr())
        ctx.mod, ctx.shape = mod, tuple(x.shape)
        ctx.save_for_backward(xc, u)
        return g.view(x.shape[:-1] + (I,))

    @staticmethod
    def backward(ctx, dg):
        xc, u = ctx.saved_tensors
        mod = ctx.mod
        T, H = xc.shape
        I = u.shape[1]
        st = _lib.stream_ptr()
        dg = dg.contiguous().view(T, I)
        du = torch.empty_like(u)
        C.uniter_gelu_bwd(ptr(dg), ptr(u), ptr(du), u.numel(), st)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(xc)
            C.uniter_gemm_dgrad(ptr(du), ptr(mod.dense.weight), None, ptr(dx), T, I, H, st)
            dx = dx.view(ctx.shape)
        pool = {}
        if mod.dense.weight.requires_grad or mod.dense.bias.requires_grad:
            _wgrad_into(du, xc, _grad_or_dummy(mod.dense.weight, pool), _grad_or_dummy(mod.dense.bias, pool), T, I, H)
        return (dx, None) + (None,) * (len(ctx.needs_input_grad) - 2)


def _anchor(params):
    if not torch.is_grad_enabled():
        return ()
    a = next((q for q in params if q is not None and q.requires_grad), None)
    return () if a is None else (a,)


def _check_mod(params, what):
    for q in params:
        _check_dev(q, what)
        if not q.is_contiguous():
            raise _lib.UniterHipError("%s must be contiguous" % what)


def self_attention(att, hidden_states, attention_mask):
    """BertSelfAttention.forward: hidden_states [B, L, H] bf16, attention_mask additive ([B, 1, 1, L] or [B, L], (1 - m) * -10000)."""
    _check_dev(hidden_states, "hidden_states")
    if hidden_states.dim() != 3:
        raise _lib.UniterHipError("hidden_states must be [B, L, H]")
    B, L, H = hidden_states.shape
    if att.attention_head_size != 64 or L > 512:
        raise _lib.UniterHipError("the attention kernels need 64-wide heads and L <= 512")
    lin = (att.query, att.key, att.value)
    _check_mod([m.weight for m in lin] + [m.bias for m in lin], "attention parameter")
    params_ready(*[m.weight for m in lin])
    mb = attention_mask.float().reshape(B, L).contiguous()
    p = float(att.dropout.p) if (att.training and torch.is_grad_enabled()) else 0.0
    return _SelfAttentionFn.apply(hidden_states, mb, att, p, *_anchor([m.weight for m in lin] + [m.bias for m in lin]))


def dense_dropout_residual_layernorm(mod, hidden_states, input_tensor):
    """BertSelfOutput / BertOutput.forward."""
    _check_dev(hidden_states, "hidden_states")
    _check_dev(input_tensor, "input_tensor")
    K, H = mod.dense.weight.shape[1], mod.dense.weight.shape[0]
    if hidden_states.shape[-1] != K or input_tensor.shape[-1] != H or hidden_states.shape[:-1] != input_tensor.shape[:-1]:
        raise _lib.UniterHipError("dense + residual: shapes %s / %s do not fit Linear(%d, %d)" % (tuple(hidden_states.shape), tuple(input_tensor.shape), K, H))
    prm = [mod.dense.weight, mod.dense.bias, mod.LayerNorm.weight, mod.LayerNorm.bias]
    _check_mod(prm, "output-block parameter")
    params_ready(*prm)
    p = float(mod.dropout.p) if (mod.training and torch.is_grad_enabled()) else 0.0
    return _DenseDropResLnFn.apply(hidden_states, input_tensor, mod, p, *_anchor(prm))


def dense_gelu(mod, hidden_states):
    """BertIntermediate.forward for hidden_act = "gelu" (the exact erf form, model/layer.py:31-37)."""
    _check_dev(hidden_states, "hidden_states")
    prm = [mod.dense.weight, mod.dense.bias]
    _check_mod(prm, "intermediate parameter")
    params_ready(*prm)
    return _DenseGeluFn.apply(hidden_states, mod, *_anchor(prm))


# ----------------------------------------------------------------------------------------------------
# LayerNorm (+ dropout on the output) as used by the embedding blocks
# ----------------------------------------------------------------------------------------------------
def _ln_fwd(z, weight, bias, eps, p, seed, off):
    rows, H = z.shape
    y = torch.empty_like(z)
    mean = torch.empty(rows, dtype=torch.float32, device=z.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=z.device)
    C.uniter_layernorm_fwd(ptr(z), ptr(weight), ptr(bias), ptr(y), ptr(mean), ptr(rstd), rows, H, eps, p, seed, off,
                           _lib.stream_ptr())
    return y, mean, rstd


def _ln_bwd(dy, z, mean, rstd, weight, bias_param, p, seed, off, post, dbias=None):
    rows, H = z.shape
    dz = torch.empty_like(z)
    wsb = C.uniter_layernorm_bwd_workspace_bytes(rows, H)
    ws = _scratch(("ln", z.device.index), wsb, z.device)
    pool = {}
    gw = ensure_grad(weight) if weight.requires_grad else _dummy_grad_like(weight, pool)
    gb = ensure_grad(bias_param) if bias_param.requires_grad else _dummy_grad_like(bias_param, pool)
    C.uniter_layernorm_bwd(ptr(dy), None, ptr(z), ptr(mean), ptr(rstd), ptr(weight), ptr(dz), None, ptr(gw), ptr(gb),
                           ptr(dbias), rows, H, 1, p, seed, off, 1 if post else 0, ptr(ws), wsb, _lib.stream_ptr())
    return dz


class _TxtEmbedFn(torch.autograd.Function):
    """UniterTextEmbeddings.forward (model/model.py:232-245)."""

    @staticmethod
    def forward(ctx, input_ids, position_ids, token_type_ids, mod, p, word, pos, typ, ln_w, ln_b):
        for t, n in ((word, "word_embeddings"), (pos, "position_embeddings"), (typ, "token_type_embeddings"),
                     (ln_w, "LayerNorm.weight"), (ln_b, "LayerNorm.bias")):
            _check_dev(t, n)
        B, Lt = input_ids.shape
        H = word.shape[1]
        ids = input_ids.to(torch.int64).contiguous()
        pids = position_ids.to(torch.int64).reshape(-1).contiguous()
        if pids.numel() != Lt:
            raise _lib.UniterHipError("position_ids must have Lt=%d entries (shape [1, Lt], data/mlm.py:115-116)" % Lt)
        tids = None if token_type_ids is None else token_type_ids.to(torch.int64).contiguous()
        z = torch.empty(B * Lt, H, dtype=_BF16, device=word.device)
        st = _lib.stream_ptr()
        C.uniter_embed_txt_fwd(ptr(ids), ptr(pids), ptr(tids), ptr(word), ptr(pos), ptr(typ), ptr(z), B, Lt, H,
                               word.shape[0], pos.shape[0], typ.shape[0], st)
        seed, off = _next_offsets(1) if p > 0 else (0, 0)
        y, mean, rstd = _ln_fwd(z, ln_w, ln_b, 1e-12, p, seed, off)
        ctx.mod, ctx.p, ctx.seed, ctx.off = mod, p, seed, off
        ctx.ids, ctx.pids, ctx.tids = ids, pids, tids
        ctx.save_for_backward(z, mean, rstd)
        return y.view(B, Lt, H)

    @staticmethod
    def backward(ctx, dy):
        z, mean, rstd = ctx.saved_tensors
        mod = ctx.mod
        B, Lt = ctx.ids.shape
        H = z.shape[1]
        dy = dy.contiguous().view(B * Lt, H)
        dz = _ln_bwd(dy, z, mean, rstd, mod.LayerNorm.weight, mod.LayerNorm.bias, ctx.p, ctx.seed, ctx.off, post=True)
        st = _lib.stream_ptr()
        word, pos, typ = mod.word_embeddings.weight, mod.position_embeddings.weight, mod.token_type_embeddings.weight
        gword = ensure_grad(word) if word.requires_grad else None
        gpos = ensure_grad(pos) if pos.requires_grad else None
        C.uniter_embed_txt_bwd(ptr(ctx.ids), ptr(ctx.pids), ptr(ctx.tids), ptr(dz), ptr(gword), ptr(gpos), None, B, Lt, H,
                               word.shape[0], pos.shape[0], typ.shape[0], st)
        if typ.requires_grad:
            gtyp = ensure_grad(typ)
            wsb = C.uniter_embed_ws_bytes(B * Lt, H)
            ws = _scratch(("emb", z.device.index), wsb, z.device)
            C.uniter_embed_type_bwd(ptr(dz), ptr(ctx.tids), ptr(gtyp), B * Lt, H, typ.shape[0], 0, ptr(ws), wsb, st)
        return (None,) * 10


def txt_embeddings(mod, input_ids, position_ids, token_type_ids):
    params_ready(mod.word_embeddings.weight, mod.position_embeddings.weight, mod.token_type_embeddings.weight,
                 mod.LayerNorm.weight, mod.LayerNorm.bias)
    p = float(mod.dropout.p) if (mod.training and torch.is_grad_enabled()) else 0.0
    args = (input_ids, position_ids, token_type_ids, mod, p, mod.word_embeddings.weight, mod.position_embeddings.weight,
            mod.token_type_embeddings.weight, mod.LayerNorm.weight, mod.LayerNorm.bias)
    return _TxtEmbedFn.apply(*args)


class _ImgEmbedFn(torch.autograd.Function):
    """UniterImageEmbeddings.forward (model/model.py:261-272)."""

    @staticmethod
    def forward(ctx, img_feat, img_pos_feat, type_ids, img_masks, mod, type_table, p, *params):
        dev = type_table.device
        B, Li, D = img_feat.shape
        rows = B * Li
        H = type_table.shape[-1]
        st = _lib.stream_ptr()
        feat = img_feat.contiguous()
        if feat.dtype not in (torch.float32, _BF16):
            feat = feat.float()
        posf = img_pos_feat.contiguous()
        if posf.dtype not in (torch.float32, _BF16):
            posf = posf.float()
        masks = None
        mask_row = None
        if img_masks is not None:
            # model/model.py:262-265: row 0 of mask_embedding is forced to zero, masked regions add row 1
            mod.mask_embedding.weight.data[0, :].fill_(0)
            masks = img_masks.to(torch.uint8).contiguous().view(-1)
            mask_row = mod.mask_embedding.weight.data[1]
        tids = None if type_ids is None else type_ids.to(torch.int64).contiguous().view(-1)
        # the reference's call convention (model/model.py:261-272): `type_embeddings` already looked up, [B, Li, H] — every row is
        # its own "type": the fused combine kernel reads row r of the tensor, and the gradient w.r.t. it is dz itself
        dense_type = type_table.dim() == 3
        if dense_type:
            if tuple(type_table.shape) != (B, Li, H) or type_ids is not None:
                raise _lib.UniterHipError("dense type_embeddings must be [B, Li, H] = %s (and come without ids)" % ((B, Li, H),))
            type_table = type_table.contiguous().view(rows, H)
            tids = torch.arange(rows, dtype=torch.int64, device=dev)
        f = torch.empty(rows, D, dtype=_BF16, device=dev)
        C.uniter_embed_img_prep(ptr(feat), 1 if feat.dtype == torch.float32 else 0, ptr(masks), ptr(mask_row), ptr(f),
                                rows, D, st)
        lin = torch.empty(rows, H, dtype=_BF16, device=dev)
        C.uniter_gemm_bias_fwd(ptr(f), ptr(mod.img_linear.weight), ptr(mod.img_linear.bias), ptr(lin), rows, H, D, st)
        t_im, mean_i, rstd_i = _ln_fwd(lin, mod.img_layer_norm.weight, mod.img_layer_norm.bias, 1e-12, 0.0, 0, 0)
        pl = torch.empty(rows, H, dtype=_BF16, device=dev)
        C.uniter_embed_pos_linear_fwd(ptr(posf), 1 if posf.dtype == torch.float32 else 0, ptr(mod.pos_linear.weight),
                                      ptr(mod.pos_linear.bias), ptr(pl), rows, H, st)
        t_pos, mean_p, rstd_p = _ln_fwd(pl, mod.pos_layer_norm.weight, mod.pos_layer_norm.bias, 1e-12, 0.0, 0, 0)
        z = torch.empty(rows, H, dtype=_BF16, device=dev)
        C.uniter_embed_img_combine_fwd(ptr(t_im), ptr(t_pos), ptr(tids), ptr(type_table), ptr(z), rows, H,
                                       type_table.shape[0], st)
        seed, off = _next_offsets(1) if p > 0 else (0, 0)
        y, mean_z, rstd_z = _ln_fwd(z, mod.LayerNorm.weight, mod.LayerNorm.bias, 1e-12, p, seed, off)
        ctx.mod, ctx.type_table, ctx.p, ctx.seed, ctx.off = mod, type_table, p, seed, off
        ctx.dense_type = dense_type
        ctx.tids, ctx.masks, ctx.posf, ctx.shape = tids, masks, posf, (B, Li, D, H)
        ctx.save_for_backward(f, lin, mean_i, rstd_i, pl, mean_p, rstd_p, z, mean_z, rstd_z)
        return y.view(B, Li, H)

    @staticmethod
    def backward(ctx, dy):
        f, lin, mean_i, rstd_i, pl, mean_p, rstd_p, z, mean_z, rstd_z = ctx.saved_tensors
        mod, type_table = ctx.mod, ctx.type_table
        B, Li, D, H = ctx.shape
        rows = B * Li
        st = _lib.stream_ptr()
        dev = z.device
        dy = dy.contiguous().view(rows, H)
        dz = _ln_bwd(dy, z, mean_z, rstd_z, mod.LayerNorm.weight, mod.LayerNorm.bias, ctx.p, ctx.seed, ctx.off, post=True)
        wsb = max(C.uniter_embed_ws_bytes(rows, max(H, D)), C.uniter_gemm_wgrad_workspace_bytes(rows, H, D))
        ws = _scratch(("emb", dev.index), wsb, dev)
        if type_table.requires_grad and not ctx.dense_type:
            C.uniter_embed_type_bwd(ptr(dz), ptr(ctx.tids), ptr(ensure_grad(type_table)), rows, H, type_table.shape[0], 1,
                                    ptr(ws), wsb, st)
        # position branch
        dpl = _ln_bwd(dz, pl, mean_p, rstd_p, mod.pos_layer_norm.weight, mod.pos_layer_norm.bias, 0.0, 0, 0, post=False)
        gwp = ensure_grad(mod.pos_linear.weight) if mod.pos_linear.weight.requires_grad else None
        gbp = ensure_grad(mod.pos_linear.bias) if mod.pos_linear.bias.requires_grad else None
        C.uniter_embed_pos_linear_bwd(ptr(ctx.posf), 1 if ctx.posf.dtype == torch.float32 else 0, ptr(dpl), ptr(gwp),
                                      ptr(gbp), rows, H, ptr(ws), wsb, st)
        # feature branch
        gbi = ensure_grad(mod.img_linear.bias) if mod.img_linear.bias.requires_grad else None
        dlin = _ln_bwd(dz, lin, mean_i, rstd_i, mod.img_layer_norm.weight, mod.img_layer_norm.bias, 0.0, 0, 0, post=False,
                       dbias=gbi)
        w = mod.img_linear.weight
        if w.requires_grad:
            C.uniter_gemm_wgrad(ptr(dlin), ptr(f), ptr(ensure_grad(w)), None, rows, H, D, 1, ptr(ws), wsb, st)
        if ctx.masks is not None and mod.mask_embedding.weight.requires_grad:
            df = torch.empty(rows, D, dtype=_BF16, device=dev)
            C.uniter_gemm_dgrad(ptr(dlin), ptr(w), None, ptr(df), rows, H, D, st)
            gme = ensure_grad(mod.mask_embedding.weight)
            C.uniter_embed_mask_bwd(ptr(df), ptr(ctx.masks), ptr(gme[1]), rows, D, ptr(ws), wsb, st)
        d_type = dz.view(B, Li, H) if (ctx.dense_type and ctx.needs_input_grad[5]) else None
        return (None,) * 5 + (d_type,) + (None,) * (1 + 11)


def img_embeddings(mod, img_feat, img_pos_feat, type_table, type_ids, img_masks):
    for t, n in ((mod.img_linear.weight, "img_linear.weight"), (type_table, "token_type_embeddings.weight")):
        _check_dev(t, n)
    _check_dev(img_feat, "img_feat", None)
    _check_dev(img_pos_feat, "img_pos_feat", None)
    p = float(mod.dropout.p) if (mod.training and torch.is_grad_enabled()) else 0.0
    params = (mod.img_linear.weight, mod.img_linear.bias, mod.img_layer_norm.weight, mod.img_layer_norm.bias,
              mod.pos_layer_norm.weight, mod.pos_layer_norm.bias, mod.pos_linear.weight, mod.pos_linear.bias,
              mod.mask_embedding.weight, mod.LayerNorm.weight, mod.LayerNorm.bias)
    params_ready(type_table, *params)
    return _ImgEmbedFn.apply(img_feat, img_pos_feat, type_ids, img_masks, mod, type_table, p, *params)


class _GatherFn(torch.autograd.Function):
    """torch.gather(cat([txt, img], 1), 1, gather_index) (model/model.py:330-333)."""

    @staticmethod
    def forward(ctx, txt, img, gather_index):
        B, Lt, H = txt.shape
        Li = img.shape[1]
        gi = gather_index.to(torch.int64).contiguous()
        Lout = gi.shape[1]
        out = torch.empty(B, Lout, H, dtype=_BF16, device=txt.device)
        C.uniter_embed_gather_fwd(ptr(txt.contiguous()), ptr(img.contiguous()), ptr(gi), ptr(out), B, Lt, Li, Lout, H,
                                  _lib.stream_ptr())
        ctx.gi, ctx.dims = gi, (B, Lt, Li, Lout, H)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, Lt, Li, Lout, H = ctx.dims
        dtxt = torch.empty(B, Lt, H, dtype=_BF16, device=dout.device)
        dimg = torch.empty(B, Li, H, dtype=_BF16, device=dout.device)
        C.uniter_embed_gather_bwd(ptr(dout.contiguous()), ptr(ctx.gi), ptr(dtxt), ptr(dimg), B, Lt, Li, Lout, H,
                                  _lib.stream_ptr())
        return dtxt, dimg, None


def gather_embeddings(txt, img, gather_index):
    _check_dev(txt, "txt_emb")
    _check_dev(img, "img_emb")
    return _GatherFn.apply(txt, img, gather_index)


# ----------------------------------------------------------------------------------------------------
# NLVR2 paired cross attention (model/nlvr2.py:170-189 through model/attention.py:13-265)
# ----------------------------------------------------------------------------------------------------
class _PairedCrossAttnFn(torch.autograd.Function):
    """attn1(left, right, right) and attn2(right, left, left) of UniterForNlvr2PairedAttn in one autograd node.

    xs [2, n, L, H]: xs[0] = the "left" sequences of the n pairs, xs[1] = the "right" ones.  The four input
    projections write column blocks of ONE packed [2nL, 3H] buffer laid out so that "attention instance" i
    (rows i*L..) holds [q_i | k_partner(i) | v_partner(i)]: rows 0..nL = attn1 (queries from left, keys/values from
    right), rows nL..2nL = attn2 (queries from right, keys/values from left).  The encoder's fused attention kernel
    then runs unchanged on 2n instances, and each module's in_proj bias gradient is a plain column sum of its half.
    mask_bias_p [2n, L] fp32 = additive key mask of the PARTNER sequence of every instance.
    Gradients of the module parameters are accumulated straight into `.grad` (like the encoder stack).
    """

    @staticmethod
    def forward(ctx, xs, mask_bias_p, attn1, attn2, p_drop, training, *anchor):
        two, n, L, H = xs.shape
        heads = attn1.num_heads
        T2 = n * L
        T = 2 * T2
        dev = xs.device
        st = _lib.stream_ptr()
        P = torch.empty(T, 3 * H, dtype=_BF16, device=dev)
        cx = torch.empty(T, H, dtype=_BF16, device=dev)
        lse = torch.empty(2 * n * heads * L, dtype=torch.float32, device=dev)
        out = torch.empty(2, n, L, H, dtype=_BF16, device=dev)
        es = 2                                           # bytes per bf16 element
        x_l, x_r = xs.data_ptr(), xs.data_ptr() + T2 * H * es
        w1, w2 = attn1.in_proj_weight, attn2.in_proj_weight
        b1, b2 = attn1.in_proj_bias, attn2.in_proj_bias
        bp = lambda b, o: None if b is None else b.data_ptr() + o * es
        p0, p1 = P.data_ptr(), P.data_ptr() + T2 * 3 * H * es
        # rows 0..T2: attn1 — q from left, k|v from right;  rows T2..T: attn2 — q from right, k|v from left.
        # The four projections are ONE grouped launch (1 536-row problems: a launch each filled a third of the chip)
        fwd_group([x_l, x_r, x_r, x_l], [H] * 4,
                  [w1.data_ptr(), w1.data_ptr() + H * H * es, w2.data_ptr(), w2.data_ptr() + H * H * es],
                  [bp(b1, 0), bp(b1, H), bp(b2, 0), bp(b2, H)],
                  [p0, p0 + H * es, p1, p1 + H * es], [3 * H] * 4, T2, [H, 2 * H, H, 2 * H], H)
        p = float(p_drop) if training else 0.0
        seed, off = _next_offsets(1) if p > 0.0 else (0, 0)
        C.uniter_attention_fwd(ptr(P), ptr(mask_bias_p), ptr(cx), ptr(lse), 2 * n, L, heads, p, seed, off, st)
        o0, o1 = out.data_ptr(), out.data_ptr() + T2 * H * es
        c0, c1 = cx.data_ptr(), cx.data_ptr() + T2 * H * es
        fwd_group([c0, c1], [0, 0], [ptr(attn1.out_proj.weight), ptr(attn2.out_proj.weight)],
                  [ptr(attn1.out_proj.bias), ptr(attn2.out_proj.bias)], [o0, o1], [0, 0], T2, [H, H], H)
        ctx.mods = (attn1, attn2)
        ctx.p, ctx.seed, ctx.off = p, seed, off
        ctx.save_for_backward(xs, mask_bias_p, P, cx, lse)
        return out

    @staticmethod
    def backward(ctx, dout):
        xs, mask_bias_p, P, cx, lse = ctx.saved_tensors
        attn1, attn2 = ctx.mods
        two, n, L, H = xs.shape
        heads = attn1.num_heads
        T2 = n * L
        T = 2 * T2
        dev = xs.device
        st = _lib.stream_ptr()
        es = 2
        dout = dout.contiguous()
        dcx = torch.empty(T, H, dtype=_BF16, device=dev)
        dP = torch.empty(T, 3 * H, dtype=_BF16, device=dev)
        dxs = torch.empty_like(xs)
        wsb = max(C.uniter_gemm_wgrad_workspace_bytes(T2, 2 * H, H), C.uniter_colsum_workspace_bytes(T2, 3 * H))
        ws = _scratch(("pca", dev.index), wsb, dev)
        pool = {}

        def grad_of(prm):
            if prm is None:
                return None
            return ensure_grad(prm) if prm.requires_grad else _dummy_grad_like(prm, pool)

        half = T2 * H * es
        # ---- out_proj (model/attention.py:257) ----
        dgrad_group([dout.data_ptr(), dout.data_ptr() + half], [0, 0], [ptr(attn1.out_proj.weight), ptr(attn2.out_proj.weight)],
                    [None, None], [dcx.data_ptr(), dcx.data_ptr() + half], T2, [H, H], H)
        # both out_proj weight + bias gradients in one launch
        wgrad_group([dout.data_ptr(), dout.data_ptr() + half], [0, 0], [cx.data_ptr(), cx.data_ptr() + half], [0, 0],
                    [ptr(grad_of(attn1.out_proj.weight)), ptr(grad_of(attn2.out_proj.weight))],
                    [ptr(grad_of(attn1.out_proj.bias)), ptr(grad_of(attn2.out_proj.bias))], T2, [H, H], [H, H])
        # ---- attention core ----
        # (beyond 256 tokens the attention backward is two launches and needs its small workspace)
        awb = C.uniter_attention_bwd_workspace_bytes(2 * n, L, heads)
        aws = _scratch(("pca_attn", dev.index), max(awb, 16), dev)
        C.uniter_attention_bwd_ws(ptr(P), ptr(mask_bias_p), None, ptr(cx), ptr(lse), ptr(dcx), ptr(dP), 2 * n, L, heads,
                                  ctx.p, ctx.seed, ctx.off, ptr(aws), awb, st)
        # ---- in_proj (model/attention.py:103-127, the kv_same branch) ----
        x_l, x_r = xs.data_ptr(), xs.data_ptr() + half
        dx_l, dx_r = dxs.data_ptr(), dxs.data_ptr() + half
        d0, d1 = dP.data_ptr(), dP.data_ptr() + T2 * 3 * H * es
        w1, w2 = attn1.in_proj_weight, attn2.in_proj_weight
        wq1, wkv1 = w1.data_ptr(), w1.data_ptr() + H * H * es
        wq2, wkv2 = w2.data_ptr(), w2.data_ptr() + H * H * es
        # d_left = dq(attn1) Wq1 + dkv(attn2) Wkv2 ; d_right = dq(attn2) Wq2 + dkv(attn1) Wkv1
        # (two grouped launches: the q halves of both sides, then the k|v halves added on top)
        dgrad_group([d0, d1], [3 * H, 3 * H], [wq1, wq2], [None, None], [dx_l, dx_r], T2, [H, H], H)
        dgrad_group([d1 + H * es, d0 + H * es], [3 * H, 3 * H], [wkv2, wkv1], [dx_l, dx_r], [dx_l, dx_r], T2, [2 * H, 2 * H], H)
        # the four in_proj weight gradients (q and k|v blocks of both modules) and their bias gradients in one launch
        g1, g2 = grad_of(w1), grad_of(w2)
        gb1, gb2 = grad_of(attn1.in_proj_bias), grad_of(attn2.in_proj_bias)
        bq1, bkv1 = (gb1.data_ptr(), gb1.data_ptr() + H * es) if gb1 is not None else (None, None)
        bq2, bkv2 = (gb2.data_ptr(), gb2.data_ptr() + H * es) if gb2 is not None else (None, None)
        wgrad_group([d0, d0 + H * es, d1, d1 + H * es], [3 * H] * 4, [x_l, x_r, x_r, x_l], [H] * 4,
                    [g1.data_ptr(), g1.data_ptr() + H * H * es, g2.data_ptr(), g2.data_ptr() + H * H * es],
                    [bq1, bkv1, bq2, bkv2], T2, [H, 2 * H, H, 2 * H], [H, H, H, H])
        return (dxs if ctx.needs_input_grad[0] else None, None, None, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 6)


class _PairedCrossAttnCatFn(torch.autograd.Function):
    """cat([attn_i(own, partner, partner), own], -1) of UniterForNlvr2PairedAttn (model/nlvr2.py:172-184) as ONE autograd node.

    seq [2n, L, H] is the encoder output in (pair, side) row order.  The node regroups it to [side, pair] order straight into the
    right half of the result `cat` [2, n, L, 2H] (one strided copy), runs _PairedCrossAttnFn's launches with that half as their
    input (row stride 2H) and lets the two output projections write the left half (row stride 2H): the torch.cat launch and its
    backward (slice copy, add, regroup copy) disappear — the backward reads d cat through row strides and ends in one elementwise
    kernel that adds the direct path and undoes the regrouping.
    """

    @staticmethod
    def forward(ctx, seq, mask_bias_p, attn1, attn2, p_drop, training, *anchor):
        bs, L, H = seq.shape
        n = bs // 2
        heads = attn1.num_heads
        T2 = n * L
        T = 2 * T2
        dev = seq.device
        st = _lib.stream_ptr()
        es = 2
        cat = torch.empty(2, n, L, 2 * H, dtype=_BF16, device=dev)
        cat[..., H:].copy_(seq.view(n, 2, L, H).transpose(0, 1))        # rows 2i / 2i+1 -> [left block; right block]
        P = torch.empty(T, 3 * H, dtype=_BF16, device=dev)
        cx = torch.empty(T, H, dtype=_BF16, device=dev)
        lse = torch.empty(2 * n * heads * L, dtype=torch.float32, device=dev)
        side = T2 * 2 * H * es                                             # bytes of one side block of cat
        x_l, x_r = cat.data_ptr() + H * es, cat.data_ptr() + side + H * es
        w1, w2 = attn1.in_proj_weight, attn2.in_proj_weight
        b1, b2 = attn1.in_proj_bias, attn2.in_proj_bias
        bp = lambda b, o: None if b is None else b.data_ptr() + o * es
        p0, p1 = P.data_ptr(), P.data_ptr() + T2 * 3 * H * es
        fwd_group([x_l, x_r, x_r, x_l], [2 * H] * 4,
                  [w1.data_ptr(), w1.data_ptr() + H * H * es, w2.data_ptr(), w2.data_ptr() + H * H * es],
                  [bp(b1, 0), bp(b1, H), bp(b2, 0), bp(b2, H)],
                  [p0, p0 + H * es, p1, p1 + H * es], [3 * H] * 4, T2, [H, 2 * H, H, 2 * H], H)
        p = float(p_drop) if training else 0.0
        seed, off = _next_offsets(1) if p > 0.0 else (0, 0)
        C.uniter_attention_fwd(ptr(P), ptr(mask_bias_p), ptr(cx), ptr(lse), 2 * n, L, heads, p, seed, off, st)
        c0, c1 = cx.data_ptr(), cx.data_ptr() + T2 * H * es
        fwd_group([c0, c1], [0, 0], [ptr(attn1.out_proj.weight), ptr(attn2.out_proj.weight)],
                  [ptr(attn1.out_proj.bias), ptr(attn2.out_proj.bias)], [cat.data_ptr(), cat.data_ptr() + side], [2 * H] * 2,
                  T2, [H, H], H)
        ctx.mods = (attn1, attn2)
        ctx.p, ctx.seed, ctx.off = p, seed, off
        ctx.save_for_backward(cat, mask_bias_p, P, cx, lse)
        return cat

    @staticmethod
    def backward(ctx, dcat):
        cat, mask_bias_p, P, cx, lse = ctx.saved_tensors
        attn1, attn2 = ctx.mods
        two, n, L, H2 = cat.shape
        H = H2 // 2
        heads = attn1.num_heads
        T2 = n * L
        T = 2 * T2
        dev = cat.device
        st = _lib.stream_ptr()
        es = 2
        dcat = dcat.contiguous()
        dcx = torch.empty(T, H, dtype=_BF16, device=dev)
        dP = torch.empty(T, 3 * H, dtype=_BF16, device=dev)
        dxs = torch.empty(2, n, L, H, dtype=_BF16, device=dev)
        wsb = max(C.uniter_gemm_wgrad_workspace_bytes(T2, 2 * H, H), C.uniter_colsum_workspace_bytes(T2, 3 * H))
        _scratch(("pca", dev.index), wsb, dev)
        pool = {}

        def grad_of(prm):
            if prm is None:
                return None
            return ensure_grad(prm) if prm.requires_grad else _dummy_grad_like(prm, pool)

        half = T2 * H * es
        side = T2 * 2 * H * es
        do_l, do_r = dcat.data_ptr(), dcat.data_ptr() + side               # d(attended) = left half of d cat, row stride 2H
        # ---- out_proj (model/attention.py:257) ----
        dgrad_group([do_l, do_r], [2 * H] * 2, [ptr(attn1.out_proj.weight), ptr(attn2.out_proj.weight)],
                    [None, None], [dcx.data_ptr(), dcx.data_ptr() + half], T2, [H, H], H)
        wgrad_group([do_l, do_r], [2 * H] * 2, [cx.data_ptr(), cx.data_ptr() + half], [0, 0],
                    [ptr(grad_of(attn1.out_proj.weight)), ptr(grad_of(attn2.out_proj.weight))],
                    [ptr(grad_of(attn1.out_proj.bias)), ptr(grad_of(attn2.out_proj.bias))], T2, [H, H], [H, H])
        # ---- attention core ----
        awb = C.uniter_attention_bwd_workspace_bytes(2 * n, L, heads)
        aws = _scratch(("pca_attn", dev.index), max(awb, 16), dev)
        C.uniter_attention_bwd_ws(ptr(P), ptr(mask_bias_p), None, ptr(cx), ptr(lse), ptr(dcx), ptr(dP), 2 * n, L, heads,
                                  ctx.p, ctx.seed, ctx.off, ptr(aws), awb, st)
        # ---- in_proj (model/attention.py:103-127, the kv_same branch) ----
        x_l, x_r = cat.data_ptr() + H * es, cat.data_ptr() + side + H * es
        dx_l, dx_r = dxs.data_ptr(), dxs.data_ptr() + half
        d0, d1 = dP.data_ptr(), dP.data_ptr() + T2 * 3 * H * es
        w1, w2 = attn1.in_proj_weight, attn2.in_proj_weight
        wq1, wkv1 = w1.data_ptr(), w1.data_ptr() + H * H * es
        wq2, wkv2 = w2.data_ptr(), w2.data_ptr() + H * H * es
        dgrad_group([d0, d1], [3 * H, 3 * H], [wq1, wq2], [None, None], [dx_l, dx_r], T2, [H, H], H)
        dgrad_group([d1 + H * es, d0 + H * es], [3 * H, 3 * H], [wkv2, wkv1], [dx_l, dx_r], [dx_l, dx_r], T2, [2 * H, 2 * H], H)
        g1, g2 = grad_of(w1), grad_of(w2)
        gb1, gb2 = grad_of(attn1.in_proj_bias), grad_of(attn2.in_proj_bias)
        bq1, bkv1 = (gb1.data_ptr(), gb1.data_ptr() + H * es) if gb1 is not None else (None, None)
        bq2, bkv2 = (gb2.data_ptr(), gb2.data_ptr() + H * es) if gb2 is not None else (None, None)
        wgrad_group([d0, d0 + H * es, d1, d1 + H * es], [3 * H] * 4, [x_l, x_r, x_r, x_l], [2 * H] * 4,
                    [g1.data_ptr(), g1.data_ptr() + H * H * es, g2.data_ptr(), g2.data_ptr() + H * H * es],
                    [bq1, bkv1, bq2, bkv2], T2, [H, 2 * H, H, 2 * H], [H, H, H, H])
        dseq = None
        if ctx.needs_input_grad[0]:
            # d seq = regroup^-1(d(own via the projections) + d(own via the right half of cat)), one elementwise launch
            # (explicit contiguous output: an elementwise op on two equally permuted inputs would keep their permuted layout)
            dseq = torch.empty(n, 2, L, H, dtype=_BF16, device=dev)
            torch.add(dxs.transpose(0, 1), dcat[..., H:].transpose(0, 1), out=dseq)
            dseq = dseq.view(2 * n, L, H)
        return (dseq, None, None, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 6)


def paired_cross_attention_cat(seq, partner_bias, attn1, attn2, p_drop, training):
    """seq [2n, L, H] bf16, rows 2i / 2i+1 = the two sequences of pair i; partner_bias [2n, L] fp32 from nlvr2_pair_masks.
    Returns cat([attended, own], -1) as [2, n, L, 2H] (left block, right block): the input of UniterForNlvr2PairedAttn.fc."""
    _check_dev(seq, "paired sequences")
    if seq.dim() != 3 or seq.size(0) % 2:
        raise _lib.UniterHipError("paired sequences must be [2n, L, H]")
    H = seq.size(2)
    for mod in (attn1, attn2):
        if mod.embed_dim != H or mod.head_dim != 64 or mod.out_proj.bias is None:
            raise _lib.UniterHipError("fused paired attention needs embed_dim == H, head_dim 64 and an out_proj bias")
        for prm in (mod.in_proj_weight, mod.in_proj_bias, mod.out_proj.weight, mod.out_proj.bias):
            if prm is not None:
                _check_dev(prm, "attention parameter")
    seq = seq.contiguous()
    if not torch.is_grad_enabled():
        return _PairedCrossAttnCatFn.apply(seq, partner_bias, attn1, attn2, p_drop, training)
    anchor = next((p for m in (attn1, attn2) for p in m.parameters() if p.requires_grad), None)
    extra = () if anchor is None else (anchor,)
    return _PairedCrossAttnCatFn.apply(seq, partner_bias, attn1, attn2, p_drop, training, *extra)


def nlvr2_pair_masks(attn_masks):
    """attn_masks [2n, L] int64, rows in (pair, side) order -> (pad [2n, L] uint8, rows regrouped [left block; right block];
    partner key-mask bias [2n, L] fp32) in one launch (uniter_nlvr2_pair_masks)."""
    _check_dev(attn_masks, "attn_masks", torch.int64)
    bs, L = attn_masks.shape
    m = attn_masks.contiguous()
    pad = torch.empty(bs, L, dtype=torch.uint8, device=m.device)
    bias = torch.empty(bs, L, dtype=torch.float32, device=m.device)
    C.uniter_nlvr2_pair_masks(ptr(m), ptr(pad), ptr(bias), bs // 2, L, _lib.stream_ptr())
    return pad, bias


def paired_cross_attention(xs, key_valid_partner, attn1, attn2, p_drop, training, partner_bias=None):
    """xs [2, n, L, H] bf16 (left block, right block); key_valid_partner [2n, L] = attention mask (1 = real token) of
    the sequence each instance attends TO (or partner_bias: its additive fp32 form, already computed).
    Returns [2, n, L, H]: attn1(left->right) block, attn2(right->left) block."""
    _check_dev(xs, "paired sequences")
    if xs.dim() != 4 or xs.size(0) != 2:
        raise _lib.UniterHipError("paired sequences must be [2, n, L, H]")
    H = xs.size(3)
    for mod in (attn1, attn2):
        if mod.embed_dim != H or mod.head_dim != 64 or mod.out_proj.bias is None:
            raise _lib.UniterHipError("fused paired attention needs embed_dim == H, head_dim 64 and an out_proj bias")
        for prm in (mod.in_proj_weight, mod.in_proj_bias, mod.out_proj.weight, mod.out_proj.bias):
            if prm is not None:
                _check_dev(prm, "attention parameter")
    xs = xs.contiguous()
    mb = partner_bias if partner_bias is not None else mask_bias(key_valid_partner)
    track = torch.is_grad_enabled()
    if not track:
        with torch.no_grad():
            return _PairedCrossAttnFn.apply(xs, mb, attn1, attn2, p_drop, training)
    anchor = next((p for m in (attn1, attn2) for p in m.parameters() if p.requires_grad), None)
    extra = () if anchor is None else (anchor,)
    return _PairedCrossAttnFn.apply(xs, mb, attn1, attn2, p_drop, training, *extra)


# ----------------------------------------------------------------------------------------------------
# word-region alignment: IPOT optimal-transport distance (model/ot.py:11-85, model/pretrain.py:166-188)
# ----------------------------------------------------------------------------------------------------
class _OtDistFn(torch.autograd.Function):
    """dist [B] fp32 = optimal_transport_dist(text slots, image slots) of the compact joint sequence `seq`."""

    @staticmethod
    def forward(ctx, seq, scatter, txt_pad, img_pad, beta, iteration, k):
        B, L, H = seq.shape
        tl, il = txt_pad.size(1), img_pad.size(1)
        dist = torch.empty(B, dtype=torch.float32, device=seq.device)
        plan = torch.empty(B, il, tl, dtype=torch.float32, device=seq.device)
        C.uniter_ot_fwd(ptr(seq), ptr(scatter), ptr(txt_pad), ptr(img_pad), ptr(dist), ptr(plan), B, L, H, tl, il,
                        float(beta), int(iteration), int(k), _lib.stream_ptr())
        ctx.save_for_backward(seq, scatter, txt_pad, img_pad, plan)
        return dist

    @staticmethod
    def backward(ctx, gdist):
        seq, scatter, txt_pad, img_pad, plan = ctx.saved_tensors
        B, L, H = seq.shape
        tl, il = txt_pad.size(1), img_pad.size(1)
        g = gdist.contiguous().to(torch.float32)
        dseq = torch.empty_like(seq)
        C.uniter_ot_bwd(ptr(seq), ptr(scatter), ptr(txt_pad), ptr(img_pad), ptr(plan), ptr(g), ptr(dseq), B, L, H, tl, il,
                        _lib.stream_ptr())
        return dseq, None, None, None, None, None, None


def optimal_transport_dist(seq, ot_scatter, txt_pad, img_pad, beta=0.5, iteration=50, k=1, return_plan=False):
    """seq [B, L, H] bf16 encoder output, ot_scatter [B, L] int64, txt_pad [B, tl] / img_pad [B, il] bool (True = pad).
    Returns the transport distances [B] in fp32 (autograd flows into `seq` through the cosine cost only)."""
    _check_dev(seq, "sequence_output")
    if seq.dim() != 3 or ot_scatter.shape != seq.shape[:2]:
        raise _lib.UniterHipError("sequence_output must be [B, L, H] and ot_scatter [B, L]")
    if txt_pad.size(0) != seq.size(0) or img_pad.size(0) != seq.size(0):
        raise _lib.UniterHipError("padding masks must have one row per example")
    dev = seq.device
    sc = ot_scatter.to(device=dev, dtype=torch.int64).contiguous()
    tp = txt_pad.to(device=dev, dtype=torch.uint8).contiguous()
    ip = img_pad.to(device=dev, dtype=torch.uint8).contiguous()
    return _OtDistFn.apply(seq.contiguous(), sc, tp, ip, beta, iteration, k)


# ----------------------------------------------------------------------------------------------------
# pre-training output heads: dense -> GELU -> LayerNorm -> projection to V classes -> cross entropy
# (MLM: model/layer.py:188-222 + model/pretrain.py:129-133, V = 28996 tied to the word embeddings;
#  MRC with hard labels: model/pretrain.py:36-47,206-229, V = 1601)                      SURVEY.md section 8 f-2
# ----------------------------------------------------------------------------------------------------
class _HeadLossFn(torch.autograd.Function):
    """One C-ABI call each way (uniter_head_ce_* / uniter_head_kl_*): fp32 logits never exist — the projection GEMM writes
    bf16 logits into a buffer whose row stride is V rounded up to 64, one kernel turns them into the loss (+ log-sum-exp),
    the backward kernel overwrites them with d loss / d logits, and those feed the weight-gradient GEMM in place.
    `target` is int64 [n] (cross entropy, negative = ignored) or fp32 [n, V] (element-wise KL divergence)."""

    @staticmethod
    def forward(ctx, x, target, dense, ln, w, b, *anchor):
        n, H = x.shape
        V = w.size(0)
        dev = x.device
        hp = _lib.UniterHeadParams()
        hp.dense_w, hp.dense_b = dense.weight.data_ptr(), dense.bias.data_ptr()
        hp.ln_g, hp.ln_b = ln.weight.data_ptr(), ln.bias.data_ptr()
        hp.proj_w, hp.proj_b = w.data_ptr(), (b.data_ptr() if b is not None else None)
        save = torch.empty(C.uniter_head_ce_save_bytes(n, H, V), dtype=torch.uint8, device=dev)
        kl = target.dtype == torch.float32
        loss = torch.empty((n, V) if kl else (n,), dtype=torch.float32, device=dev)
        fwd = C.uniter_head_kl_fwd if kl else C.uniter_head_ce_fwd
        fwd(ctypes.byref(hp), ptr(x), ptr(target), ptr(loss), ptr(save), n, H, V, float(ln.eps), _lib.stream_ptr())
        ctx.mods = (dense, ln, w, b)
        ctx.hp = hp
        ctx.kl = kl
        ctx.save_for_backward(x, target, save)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        x, target, save = ctx.saved_tensors
        dense, ln, w, b = ctx.mods
        hp = ctx.hp
        n, H = x.shape
        V = w.size(0)
        dev = x.device
        pool = {}

        def grad_of(p):
            return (ensure_grad(p) if p.requires_grad else _dummy_grad_like(p, pool)).data_ptr()
        hp.g_dense_w, hp.g_dense_b = grad_of(dense.weight), grad_of(dense.bias)
        hp.g_ln_g, hp.g_ln_b = grad_of(ln.weight), grad_of(ln.bias)
        hp.g_proj_w = grad_of(w)
        hp.g_proj_b = grad_of(b) if b is not None else None
        wsb = C.uniter_head_ce_workspace_bytes(n, H, V)
        ws = _scratch(("headce", dev.index), wsb, dev)
        gl = gloss.to(torch.float32).contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        bwd = C.uniter_head_kl_bwd if ctx.kl else C.uniter_head_ce_bwd
        bwd(ctypes.byref(hp), ptr(x), ptr(target), ptr(gl), ptr(dx), ptr(save), ptr(ws), wsb, n, H, V, _lib.stream_ptr())
        return (dx, None, None, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 6)


def _head_loss(x, target, dense, ln, weight, bias):
    _check_dev(x, "head input")
    _check_dev(weight, "projection weight")
    _check_dev(dense.weight, "transform weight")
    if bias is not None:
        _check_dev(bias, "projection bias")
    if x.dim() != 2 or x.size(1) % 64 != 0 or dense.weight.shape != (x.size(1), x.size(1)) or weight.size(1) != x.size(1):
        raise _lib.UniterHipError("head loss: x must be [n, H] with H % 64 == 0, dense H->H, weight [V, H]")
    params = [p for p in (dense.weight, dense.bias, ln.weight, ln.bias, weight, bias) if p is not None and p.requires_grad]
    extra = tuple(params[:1]) if torch.is_grad_enabled() else ()
    return _HeadLossFn.apply(x.contiguous(), target, dense, ln, weight, bias, *extra)


def head_cross_entropy(x, labels, dense, ln, weight, bias):
    """loss[n] = cross_entropy(LN(gelu(dense(x))) @ weight^T + bias, labels) with reduction 'none'; rows whose label is
    negative are ignored (loss 0, no gradient).  x [n, H] bf16, labels [n] int64, dense = nn.Linear(H, H),
    ln = nn.LayerNorm(H), weight [V, H] bf16, bias [V] bf16.  Parameter gradients are accumulated into `.grad`.
    The backward pass reuses the logits buffer for d loss / d logits, so a graph can be back-propagated once."""
    if x.size(0) == 0:
        return x.float().sum(1)
    return _head_loss(x, labels.to(device=x.device, dtype=torch.int64).contiguous(), dense, ln, weight, bias)


def head_kl_div(x, soft_targets, dense, ln, weight, bias):
    """loss[n, V] = F.kl_div(log_softmax(LN(gelu(dense(x))) @ weight^T + bias), soft_targets, reduction='none')."""
    if x.size(0) == 0:
        return x.new_zeros(0, weight.size(0), dtype=torch.float32)
    if soft_targets.shape != (x.size(0), weight.size(0)):
        raise _lib.UniterHipError("head_kl_div: soft targets must be [n, V]")
    return _head_loss(x, soft_targets.to(device=x.device, dtype=torch.float32).contiguous(), dense, ln, weight, bias)


def mlm_head_loss(x, labels, predictions):
    """BertLMPredictionHead + F.cross_entropy(reduction='none') on the masked rows (model/pretrain.py:129-133)."""
    tr = predictions.transform
    return head_cross_entropy(x, labels, tr.dense, tr.LayerNorm, predictions.decoder.weight, predictions.bias)


# ----------------------------------------------------------------------------------------------------
# attention pooling of the NLVR2 paired-attention head (model/nlvr2.py:110-125)
# ----------------------------------------------------------------------------------------------------
class _AttnPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pad, lin, p_drop, training, *anchor):
        B, L, H = x.shape
        dev = x.device
        out = torch.empty(B, H, dtype=_BF16, device=dev)
        raw = torch.empty(B, L, dtype=torch.float32, device=dev)
        sm = torch.empty_like(raw)
        pw = torch.empty_like(raw)
        p = float(p_drop) if training else 0.0
        seed, off = _next_offsets(1) if p > 0.0 else (0, 0)
        C.uniter_attn_pool_fwd(ptr(x), ptr(pad), ptr(lin.weight), ptr(lin.bias), ptr(out), ptr(raw), ptr(sm), ptr(pw),
                               B, L, H, p, seed, off, _lib.stream_ptr())
        ctx.lin = lin
        ctx.save_for_backward(x, raw, sm, pw)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, raw, sm, pw = ctx.saved_tensors
        lin = ctx.lin
        B, L, H = x.shape
        dx = torch.empty_like(x)
        wsb = C.uniter_attn_pool_workspace_bytes(B, H)
        ws = _scratch(("pool", x.device.index), wsb, x.device)
        gw = ensure_grad(lin.weight) if lin.weight.requires_grad else None
        gb = ensure_grad(lin.bias) if (lin.bias is not None and lin.bias.requires_grad) else None
        C.uniter_attn_pool_bwd(ptr(x), ptr(lin.weight), ptr(raw), ptr(sm), ptr(pw), ptr(dout.contiguous()), ptr(dx),
                               ptr(gw), ptr(gb), B, L, H, ptr(ws), wsb, _lib.stream_ptr())
        return (dx if ctx.needs_input_grad[0] else None, None, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 5)


def attention_pool(x, pad, lin, p_drop, training):
    """x [B, L, H] bf16, pad [B, L] bool (True = padding) or None, lin = the pool's nn.Linear(H, 1).  Returns [B, H]."""
    _check_dev(x, "pool input")
    _check_dev(lin.weight, "pool weight")
    x = x.contiguous()
    pad_u8 = None if pad is None else pad.to(device=x.device, dtype=torch.uint8).contiguous()
    if not torch.is_grad_enabled():
        return _AttnPoolFn.apply(x, pad_u8, lin, p_drop, training)
    anchor = next((p for p in lin.parameters() if p.requires_grad), None)
    extra = () if anchor is None else (anchor,)
    return _AttnPoolFn.apply(x, pad_u8, lin, p_drop, training, *extra)


# ----------------------------------------------------------------------------------------------------
# dense pieces of the NLVR2 paired-attention head (model/nlvr2.py:150-153, 196-204)
# ----------------------------------------------------------------------------------------------------
class _LinearReluDropoutFn(torch.autograd.Function):
    """y = dropout(relu(x W^T + b)) as ONE GEMM with a fused epilogue; backward = mask kernel + dgrad + grouped wgrad."""

    @staticmethod
    def forward(ctx, x, lin, p, *anchor):
        T, K = x.shape
        N = lin.weight.size(0)
        y = torch.empty(T, N, dtype=_BF16, device=x.device)
        seed, off = _next_offsets(1) if p > 0.0 else (0, 0)
        C.uniter_gemm_bias_relu_dropout_fwd(ptr(x), ptr(lin.weight), ptr(lin.bias), ptr(y), T, N, K, p, seed, off,
                                            _lib.stream_ptr())
        ctx.lin, ctx.p = lin, p
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        lin = ctx.lin
        T, K = x.shape
        N = lin.weight.size(0)
        st = _lib.stream_ptr()
        dy = dy.contiguous()
        dpre = torch.empty_like(y)
        C.uniter_relu_dropout_bwd(ptr(dy), ptr(y), ptr(dpre), y.numel(), ctx.p, st)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            C.uniter_gemm_dgrad(ptr(dpre), ptr(lin.weight), None, ptr(dx), T, N, K, st)
        pool = {}
        gw = ensure_grad(lin.weight) if lin.weight.requires_grad else _dummy_grad_like(lin.weight, pool)
        gb = None
        if lin.bias is not None:
            gb = ensure_grad(lin.bias) if lin.bias.requires_grad else _dummy_grad_like(lin.bias, pool)
        wgrad_group([dpre.data_ptr()], [0], [x.data_ptr()], [0], [gw.data_ptr()], [ptr(gb)], T, [N], [K])
        return (dx, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)


def linear_relu_dropout(x, lin, p_drop, training):
    """x [T, K] bf16, lin = nn.Linear(K, N) (bf16, N % 64 == 0, K % 64 == 0) -> dropout(relu(lin(x))) [T, N]."""
    _check_dev(x, "fc input")
    _check_dev(lin.weight, "fc weight")
    if x.dim() != 2 or lin.weight.size(1) != x.size(1) or x.size(1) % 64 or lin.weight.size(0) % 64:
        raise _lib.UniterHipError("linear_relu_dropout: x must be [T, K] with K and N multiples of 64")
    p = float(p_drop) if training else 0.0
    x = x.contiguous()
    anchor = next((q for q in lin.parameters() if q.requires_grad), None) if torch.is_grad_enabled() else None
    return _LinearReluDropoutFn.apply(x, lin, p, *(() if anchor is None else (anchor,)))


class _LinearCrossEntropyFn(torch.autograd.Function):
    """F.cross_entropy(lin(x).float(), targets, reduction='none') for a handful of classes: one small kernel each way."""

    @staticmethod
    def forward(ctx, x, lin, targets, *anchor):
        n, D = x.shape
        Cn = lin.weight.size(0)
        loss = torch.empty(n, dtype=torch.float32, device=x.device)
        probs = torch.empty(n, Cn, dtype=torch.float32, device=x.device)
        C.uniter_cls_ce_fwd(ptr(x), ptr(lin.weight), ptr(lin.bias), ptr(targets), ptr(loss), ptr(probs), None, n, D, Cn,
                            _lib.stream_ptr())
        ctx.lin = lin
        ctx.save_for_backward(x, targets, probs)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        x, targets, probs = ctx.saved_tensors
        lin = ctx.lin
        n, D = x.shape
        Cn = lin.weight.size(0)
        pool = {}
        gw = ensure_grad(lin.weight) if lin.weight.requires_grad else _dummy_grad_like(lin.weight, pool)
        gb = None
        if lin.bias is not None:
            gb = ensure_grad(lin.bias) if lin.bias.requires_grad else _dummy_grad_like(lin.bias, pool)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        C.uniter_cls_ce_bwd(ptr(x), ptr(lin.weight), ptr(probs), ptr(targets), ptr(gloss.to(torch.float32).contiguous()), ptr(dx),
                            ptr(gw), ptr(gb), n, D, Cn, _lib.stream_ptr())
        return (dx, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)


def linear_cross_entropy(x, lin, targets):
    """x [n, D] bf16, lin = nn.Linear(D, C <= 8) bf16, targets [n] int64 -> per-row cross entropy [n] fp32."""
    _check_dev(x, "classifier input")
    _check_dev(lin.weight, "classifier weight")
    if (x.dim() != 2 or lin.weight.size(1) != x.size(1) or x.size(1) % 8 or lin.weight.size(0) > 8 or x.size(0) > 4096
            or x.size(0) * lin.weight.size(0) * 4 > 64 * 1024):
        raise _lib.UniterHipError("linear_cross_entropy: x must be [n <= 4096, D % 8 == 0], the classifier at most 8-way and "
                                  "n * classes * 4 bytes at most 64 KiB (the backward kernel stages the probabilities in LDS)")
    t = targets.to(device=x.device, dtype=torch.int64).contiguous()
    anchor = next((q for q in lin.parameters() if q.requires_grad), None) if torch.is_grad_enabled() else None
    return _LinearCrossEntropyFn.apply(x.contiguous(), lin, t, *(() if anchor is None else (anchor,)))
