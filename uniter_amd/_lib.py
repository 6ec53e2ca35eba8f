"""ctypes binding of libuniter_hip.so (the C ABI declared in include/uniter_hip.h).

The product path has NO fallback: if the shared library is missing or a call returns a non-zero
status, an exception is raised.  The library is built in-tree by ``uniter_amd/csrc/build.py``
(``__graft_entry__.build()``) into ``uniter_amd/csrc/build/libuniter_hip.so``.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint8,
                    c_uint64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UNITER_AMD_LIB") or os.path.join(_HERE, "csrc", "build", "libuniter_hip.so")   # (UNITER_AMD_LIB: a variant build, A/B runs)
ABI_VERSION = 8          # UNITER_HIP_ABI_VERSION of include/uniter_hip.h (struct layouts below must match it)


class UniterHipError(RuntimeError):
    pass


def head_torch_path(what, why):
    """The task heads run on HIP kernels (include/uniter_hip.h head / pool / OT entry points).  Their plain PyTorch module path
    (the reference's op sequence) is a comparison path for tests and scripts: it is taken only when UNITER_AMD_HEAD_TORCH=1 is set.
    Otherwise an input the kernels do not cover RAISES, like the encoder does — a mis-typed run (fp32 weights, CPU tensors, an
    unsupported width) must not degrade silently to eager PyTorch."""
    if os.environ.get("UNITER_AMD_HEAD_TORCH") == "1":
        return True
    raise UniterHipError("%s: %s — the HIP head path does not cover this input and UNITER_AMD_HEAD_TORCH=1 (PyTorch module path, "
                         "tests / comparisons only) is not set" % (what, why))


class UniterLayerParams(Structure):
    _fields_ = [(n, c_void_p) for n in (
        "wqkv", "bqkv", "wo", "bo", "ln1_g", "ln1_b", "w1", "b1", "w2", "b2", "ln2_g", "ln2_b",
        "g_wqkv", "g_bqkv", "g_wo", "g_bo", "g_ln1_g", "g_ln1_b", "g_w1", "g_b1", "g_w2", "g_b2",
        "g_ln2_g", "g_ln2_b")]


class UniterHeadParams(Structure):
    _fields_ = [(n, c_void_p) for n in (
        "dense_w", "dense_b", "ln_g", "ln_b", "proj_w", "proj_b",
        "g_dense_w", "g_dense_b", "g_ln_g", "g_ln_b", "g_proj_w", "g_proj_b")]


class UniterEncoderShape(Structure):
    _fields_ = [("B", c_int64), ("L", c_int64), ("H", c_int64), ("heads", c_int64), ("I", c_int64),
                ("p_hidden", c_float), ("p_attn", c_float), ("ln_eps", c_float), ("training", c_int32),
                ("total_tokens", c_int64), ("cu_seqlens", c_void_p), ("hidden_act", c_int32)]


class UniterAdamTensor(Structure):
    _fields_ = [("param", c_void_p), ("grad", c_void_p), ("master", c_void_p), ("exp_avg", c_void_p),
                ("exp_avg_sq", c_void_p), ("numel", c_int64), ("group", c_int32), ("param_is_bf16", c_int32)]


class UniterTimingRecord(Structure):
    _fields_ = [("kind", c_int32), ("calls", c_int32), ("M", c_int64), ("N", c_int64), ("K", c_int64),
                ("total_us", ctypes.c_double)]


class UniterAdamGroup(Structure):
    _fields_ = [("lr", c_float), ("beta1", c_float), ("beta2", c_float), ("eps", c_float),
                ("weight_decay", c_float), ("correct_bias", c_int32), ("step", c_int32)]


_P = c_void_p
_I = c_int64
# name -> (restype, argtypes).  Must list EVERY symbol of include/uniter_hip.h (checked by tests).
SIGNATURES = {
    "uniter_hip_abi_version": (c_int, []),
    "uniter_hip_last_error": (c_char_p, []),
    "uniter_hip_device_info": (c_int, [POINTER(c_int32)]),
    "uniter_hip_set_dropout_offset_ptr": (c_int, [_P]),
    "uniter_hip_counter_add": (c_int, [_P, c_uint64, _P]),
    "uniter_hip_timing_begin": (c_int, []),
    "uniter_hip_timing_end": (c_int, [POINTER(UniterTimingRecord), c_int32, POINTER(c_int32)]),
    "uniter_gemm_debug_force": (c_int, [c_int, c_int]),
    "uniter_gemm_debug_act_flags": (c_int, [c_int]),
    "uniter_gemm_autotune": (c_int, [c_int, _I, _I, _I, _P]),
    "uniter_gemm_set_tuned": (c_int, [c_int, _I, _I, _I, c_int32, c_int32]),
    "uniter_gemm_tuned_choice": (c_int, [c_int, _I, _I, _I, POINTER(c_int32)]),
    "uniter_gemm_bias_fwd": (c_int, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "uniter_gemm_bias_gelu_fwd": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "uniter_gemm_bias_dropout_residual_fwd": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, c_float, c_uint64, c_uint64, _P]),
    "uniter_gemm_dgrad": (c_int, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "uniter_gemm_dgrad_gelu": (c_int, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "uniter_gemm_wgrad_workspace_bytes": (c_size_t, [_I, _I, _I]),
    "uniter_gemm_wgrad": (c_int, [_P, _P, _P, _P, _I, _I, _I, c_int, _P, c_size_t, _P]),
    "uniter_gemm_dgrad_splitk_workspace_bytes": (c_size_t, [_I, _I, _I]),
    "uniter_gemm_dgrad_splitk": (c_int, [_P, _I, _P, _P, _I, _I, _I, _P, c_size_t, _P]),
    "uniter_head_ce_save_bytes": (c_size_t, [_I, _I, _I]),
    "uniter_head_ce_workspace_bytes": (c_size_t, [_I, _I, _I]),
    "uniter_head_ce_fwd": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, c_float, _P]),
    "uniter_head_ce_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_size_t, _I, _I, _I, _P]),
    "uniter_ce_fwd": (c_int, [_P, _I, _P, _P, _P, _I, _I, _P]),
    "uniter_ce_bwd": (c_int, [_P, _I, _P, _P, _P, _I, _I, _P]),
    "uniter_head_kl_fwd": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, c_float, _P]),
    "uniter_head_kl_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_size_t, _I, _I, _I, _P]),
    "uniter_gelu_bwd": (c_int, [_P, _P, _P, _I, _P]),
    "uniter_gemm_tile_count": (c_int, []),
    "uniter_gemm_bias_fwd_ld": (c_int, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "uniter_gemm_dgrad_ld": (c_int, [_P, _I, _P, _P, _P, _I, _I, _I, _P]),
    "uniter_gemm_wgrad_ld": (c_int, [_P, _I, _P, _I, _P, _P, _I, _I, _I, c_int, _P, c_size_t, _P]),
    "uniter_gemm_wgrad_group": (c_int, [c_int32, _P, _P, _P, _P, _P, _P, _I, _P, _P, c_int, _P]),
    "uniter_gemm_wgrad_group_autotune": (c_int, [c_int32, _I, _P, _P, _P]),
    "uniter_gemm_wgrad_group_workspace_bytes": (c_size_t, [c_int32, _P, _P]),
    "uniter_encoder_wgrad_stage_bytes": (c_size_t, [_P, c_int32]),
    "uniter_encoder_set_wgrad_stage": (c_int, [_P, c_size_t]),
    "uniter_encoder_side_join_all": (c_int, [_P]),
    "uniter_gemm_wgrad_group_ws": (c_int, [c_int32, _P, _P, _P, _P, _P, _P, _I, _P, _P, c_int, _P, c_size_t, c_int, c_int, _P]),
    "uniter_attention_fwd": (c_int, [_P, _P, _P, _P, _I, _I, _I, c_float, c_uint64, c_uint64, _P]),
    "uniter_qkv_attention_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, c_float, c_uint64, c_uint64, _P]),
    "uniter_attention_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, c_float, c_uint64, c_uint64, _P]),
    "uniter_attention_bwd_workspace_bytes": (c_size_t, [_I, _I, _I]),
    "uniter_attention_bwd_ws": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, c_float, c_uint64, c_uint64, _P, c_size_t, _P]),
    "uniter_attention_fwd_packed": (c_int, [_P, _P, _P, _P, _I, _I, _I, c_float, c_uint64, c_uint64, _P]),
    "uniter_attention_bwd_packed": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, c_float, c_uint64, c_uint64, _P]),
    "uniter_layernorm_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, c_float, c_float, c_uint64, c_uint64, _P]),
    "uniter_layernorm_bwd_workspace_bytes": (c_size_t, [_I, _I]),
    "uniter_layernorm_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, c_int,
                                     c_float, c_uint64, c_uint64, c_int, _P, c_size_t, _P]),
    "uniter_colsum_workspace_bytes": (c_size_t, [_I, _I]),
    "uniter_colsum": (c_int, [_P, _P, _I, _I, c_int, _P, c_size_t, _P]),
    "uniter_embed_ws_bytes": (c_size_t, [_I, _I]),
    "uniter_embed_txt_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "uniter_embed_txt_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "uniter_embed_img_prep": (c_int, [_P, c_int, _P, _P, _P, _I, _I, _P]),
    "uniter_embed_pos_linear_fwd": (c_int, [_P, c_int, _P, _P, _P, _I, _I, _P]),
    "uniter_embed_pos_linear_bwd": (c_int, [_P, c_int, _P, _P, _P, _I, _I, _P, c_size_t, _P]),
    "uniter_embed_img_combine_fwd": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "uniter_embed_type_bwd": (c_int, [_P, _P, _P, _I, _I, _I, c_int, _P, c_size_t, _P]),
    "uniter_embed_mask_bwd": (c_int, [_P, _P, _P, _I, _I, _P, c_size_t, _P]),
    "uniter_embed_gather_fwd": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "uniter_embed_gather_bwd": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "uniter_mask_bias": (c_int, [_P, _P, _I, _P]),
    "uniter_encoder_layer_act_bytes": (c_size_t, [POINTER(UniterEncoderShape)]),
    "uniter_encoder_scratch_bytes": (c_size_t, [POINTER(UniterEncoderShape)]),
    "uniter_encoder_layer_out_offset": (c_size_t, [POINTER(UniterEncoderShape)]),
    "uniter_encoder_forward": (c_int, [POINTER(UniterEncoderShape), POINTER(UniterLayerParams), c_int32, c_int32,
                                       _P, _P, _P, _P, c_uint64, c_uint64, _P]),
    "uniter_encoder_backward": (c_int, [POINTER(UniterEncoderShape), POINTER(UniterLayerParams), c_int32, c_int32,
                                        _P, _P, _P, _P, _P, _P, c_uint64, c_uint64, _P]),
    "uniter_encoder_autotune": (c_int, [POINTER(UniterEncoderShape), _P]),
    "uniter_encoder_debug_side_stream": (c_int, [c_int]),
    "uniter_encoder_defer_side_join": (c_int, [c_int]),
    "uniter_encoder_set_grad_overwrite": (c_int, [c_int32]),
    "uniter_encoder_set_grad_sq": (c_int, [c_int32]),
    "uniter_encoder_last_grad_sq": (c_int, [POINTER(c_void_p), POINTER(c_int32)]),
    "uniter_encoder_side_join": (c_int, [c_void_p]),
    "uniter_encoder_side_stream": (c_int, [POINTER(c_void_p)]),
    "uniter_encoder_set_grad_buckets": (c_int, [c_int32]),
    "uniter_encoder_grad_bucket_count": (c_int, [POINTER(c_int32)]),
    "uniter_encoder_bucket_wait": (c_int, [c_int32, c_void_p]),
    "uniter_encoder_bucket_token": (c_int, [c_int32, POINTER(c_void_p), POINTER(ctypes.c_uint32)]),
    "uniter_hip_stream_wait_value32": (c_int, [c_void_p, c_void_p, ctypes.c_uint32]),
    "uniter_encoder_debug_chain": (c_int, [c_int]),
    "uniter_encoder_chain_status": (c_int, [POINTER(UniterEncoderShape), c_void_p, POINTER(c_int32)]),
    "uniter_encoder_debug_tune_in_situ": (c_int, [c_int]),
    "uniter_gemm_bias_relu_dropout_fwd": (c_int, [_P, _P, _P, _P, _I, _I, _I, c_float, c_uint64, c_uint64, _P]),
    "uniter_relu_dropout_bwd": (c_int, [_P, _P, _P, _I, c_float, _P]),
    "uniter_cls_ce_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "uniter_cls_ce_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "uniter_adamw_plan_create": (c_int, [POINTER(UniterAdamTensor), _I, POINTER(c_void_p)]),
    "uniter_adamw_plan_destroy": (c_int, [_P]),
    "uniter_adamw_plan_set_flags": (c_int, [_P, _P, c_int64]),
    "uniter_adamw_grad_norm_ex": (c_int, [_P, c_float, c_float, _P, _P, c_int32, _P]),
    "uniter_adamw_grad_norm": (c_int, [_P, c_float, c_float, _P, _P]),
    "uniter_adamw_step": (c_int, [_P, POINTER(UniterAdamGroup), c_int32, _P, _P]),
    "uniter_adamw_step_dev": (c_int, [_P, _P, c_int32, _P, _P]),
    "uniter_adamw_step_zero": (c_int, [_P, POINTER(UniterAdamGroup), c_int32, _P, _P]),
    "uniter_adamw_step_async": (c_int, [_P, POINTER(UniterAdamGroup), c_int32, _P, POINTER(c_void_p), c_int32, c_int32, _P]),
    "uniter_params_wait": (c_int, [_P, _P]),
    "uniter_params_wait_all": (c_int, [_P]),
    "uniter_attn_pool_workspace_bytes": (c_size_t, [_I, _I]),
    "uniter_attn_pool_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, c_float, c_uint64, c_uint64, _P]),
    "uniter_attn_pool_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, c_size_t, _P]),
    "uniter_ot_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, c_float, c_int32, c_int32, _P]),
    "uniter_ot_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "uniter_nlvr2_pair_masks": (c_int, [_P, _P, _P, _I, _I, _P]),
    "uniter_gemm_bias_fwd_group": (c_int, [c_int32, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P]),
    "uniter_gemm_dgrad_group": (c_int, [c_int32, _P, _P, _P, _P, _P, _I, _P, _I, _P]),
    "uniter_comm_unique_id": (c_int, [POINTER(c_uint8)]),
    "uniter_comm_init": (c_int, [POINTER(c_uint8), c_int32, c_int32, POINTER(c_void_p)]),
    "uniter_comm_destroy": (c_int, [_P]),
    "uniter_comm_allreduce": (c_int, [_P, _P, _I, c_int32, _P]),
    "uniter_comm_broadcast": (c_int, [_P, _P, _I, c_int32, _P]),
    "uniter_comm_allgather": (c_int, [_P, _P, _P, _I, _P]),
}

# functions that return a size / pointer rather than a status code
_NO_STATUS = {"uniter_hip_abi_version", "uniter_gemm_tile_count", "uniter_hip_last_error", "uniter_gemm_wgrad_workspace_bytes",
              "uniter_gemm_wgrad_group_workspace_bytes", "uniter_encoder_wgrad_stage_bytes",
              "uniter_gemm_dgrad_splitk_workspace_bytes", "uniter_head_ce_save_bytes", "uniter_head_ce_workspace_bytes",
              "uniter_layernorm_bwd_workspace_bytes", "uniter_colsum_workspace_bytes", "uniter_embed_ws_bytes",
              "uniter_attn_pool_workspace_bytes",
              "uniter_encoder_layer_act_bytes", "uniter_encoder_scratch_bytes", "uniter_encoder_layer_out_offset"}
# every size query returns a byte count, not a status (a name missing from the list above must not turn a size into an "error")
_NO_STATUS |= {name for name, (res, _args) in SIGNATURES.items() if res is c_size_t}

_lib = None


def load():
    """Load the shared library (once).  Raises UniterHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UniterHipError(
            "libuniter_hip.so not found at %s — build it with `python uniter_amd/csrc/build.py` "
            "(there is no CPU / PyTorch fallback for the encoder hot path)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.uniter_hip_abi_version() != ABI_VERSION:
        raise UniterHipError("libuniter_hip.so has ABI version %d, this package needs %d — rebuild it "
                             "(python uniter_amd/csrc/build.py --force)" % (lib.uniter_hip_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


class _Checked:
    """Attribute access returns a wrapper that raises on a non-zero status."""

    def __getattr__(self, name):
        lib = load()
        fn = getattr(lib, name)
        if name in _NO_STATUS:
            wrapper = fn
        else:
            def wrapper(*args, _fn=fn, _name=name):
                rc = _fn(*args)
                if rc != 0:
                    msg = lib.uniter_hip_last_error()
                    raise UniterHipError("%s failed with status %d: %s" % (
                        _name, rc, msg.decode("utf-8", "replace") if msg else ""))
                return 0
        setattr(self, name, wrapper)
        return wrapper


C = _Checked()


def ptr(t):
    """Device pointer of a (contiguous) torch tensor, or None."""
    if t is None:
        return None
    return t.data_ptr()


_async_pending = False


def async_pending():
    """True while an asynchronous optimizer step (AdamW.enable_overlap) may still be writing parameters."""
    return _async_pending


def set_async_pending(flag):
    global _async_pending
    _async_pending = bool(flag)


_raw_stream = None


def stream_ptr():
    """hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream() builds a Stream object
    through several Python layers (~12 us, a few hundred calls per step); the raw getter is one C call."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        return _raw_stream(torch._C._cuda_getDevice())
    return torch.cuda.current_stream().cuda_stream


TIMING_KINDS = ("gemm fwd +bias", "gemm fwd +bias+gelu", "gemm fwd +bias+dropout+residual", "gemm dgrad",
                "gemm dgrad x gelu'", "gemm wgrad", "attention fwd", "attention bwd", "layernorm fwd", "layernorm bwd",
                "column sum", "adamw", "layernorm bwd column sums", "gemm wgrad grouped")


def timing_begin():
    C.uniter_hip_timing_begin()


def timing_end(cap=256):
    """Per-(kind, M, N, K) launch counts and summed HIP-event durations since timing_begin()."""
    recs = (UniterTimingRecord * cap)()
    n = c_int32(0)
    C.uniter_hip_timing_end(recs, cap, ctypes.byref(n))
    return [{"kind": TIMING_KINDS[r.kind] if 0 <= r.kind < len(TIMING_KINDS) else str(r.kind), "kind_id": r.kind,
             "M": r.M, "N": r.N, "K": r.K, "calls": r.calls, "total_us": r.total_us} for r in recs[:min(n.value, cap)]]


def device_info():
    out = (c_int32 * 4)()
    C.uniter_hip_device_info(out)
    return {"cus": out[0], "wave": out[1], "lds_per_cu": out[2], "gfx": out[3]}


# ---- gradient attachment epoch ------------------------------------------------------------------------------------------
# Bumped whenever a parameter gets a `.grad` tensor it did not have before (first use of an arena slot, a fresh zeros
# tensor, a gradient adopted from autograd).  The optimizer compares it with the value it saw when it last validated its
# device plan, so a gradient attached between grad_norm() and step() (or between two calls) is never missed.
_grad_attach_epoch = 0


def note_grad_attached():
    global _grad_attach_epoch
    _grad_attach_epoch += 1


def grad_attach_epoch():
    return _grad_attach_epoch


# ---- weight gradients still in flight on the library's side stream ------------------------------------------------------------
# A training loop may let its backward call return without joining the weight-gradient stream (ops.defer_wgrad_join): the
# embedding backward then overlaps the deferred weight-gradient launch.  Everything that reads or writes a weight gradient
# afterwards (grad_norm / step / zero_grad of uniter_amd.optim.AdamW) calls join_wgrads() first.  The tensors that launch
# still reads (the saved activations, the encoder input, the incoming gradient) are handed to the caching allocator with
# record_stream on that stream: their memory is not given out again before the launch is through, but no Python reference
# outlives the backward call (round 3 parked the tensors in a list until the join — with gradient accumulation 4 on
# UNITER-large that was four micro-steps' activation arenas alive at once).
# Lazy zero_grad (AdamW.lazy_zero, round 6): the encoder's backward REPLACES the parameter gradients of its layers when they are
# marked undefined, so the fused optimizer step need not zero them and the deferred launch need not read them.
#   lazy_ranges      {(first byte, byte length)} of the gradient storages the last encoder backward wrote (per layer, fused q|k|v once)
#   lazy_undefined   True between a fused step that skipped those storages and the backward that overwrites them
# Folded gradient norm (AdamW.fold_norm, round 6): the deferred weight-gradient launch of the last encoder backward left per-tile sums of
# squares of the gradients it stored; grad_norm() adds them instead of re-reading those tensors — as long as nothing has touched the
# tensors since (torch's version counters: an in-place op, an allreduce) and the optimizer's plan holds exactly these tensors.
#   sq_state   None or dict(ptr, n, ranges = frozenset of (first byte, byte length) of the weight gradients covered, tensors, versions)
sq_state = None
lazy_ranges = frozenset()
lazy_tensors = []        # the gradient tensors behind lazy_ranges (zeroed explicitly when an undefined state has to be resolved)
lazy_undefined = False


def lazy_resolve():
    """Give the undefined gradient storages the zeros a zero_grad() promises (someone is about to read them before a backward
    pass has replaced them: a gradient norm, an optimizer step, a set_to_none)."""
    global lazy_undefined
    if lazy_undefined:
        for t in lazy_tensors:
            t.zero_()
        lazy_undefined = False

_wgrads_pending = False
_side_streams = {}


def wgrads_in_flight():
    return _wgrads_pending


def hold_until_wgrad_join(*tensors):
    """Called on the thread that ran uniter_encoder_backward (the side stream is per thread), right before the call."""
    global _wgrads_pending
    import torch
    h = ctypes.c_void_p()
    C.uniter_encoder_side_stream(ctypes.byref(h))
    dev = tensors[0].device
    key = (h.value, dev.index)
    ext = _side_streams.get(key)
    if ext is None:
        ext = _side_streams[key] = torch.cuda.ExternalStream(h.value, device=dev)
    for t in tensors:
        t.record_stream(ext)
    _wgrads_pending = True


def join_wgrads():
    """Make the current stream wait for every un-joined weight-gradient launch; no-op when none is outstanding."""
    global _wgrads_pending
    if _wgrads_pending:
        C.uniter_encoder_side_join_all(stream_ptr())
        _wgrads_pending = False
