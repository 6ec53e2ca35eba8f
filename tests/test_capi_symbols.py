"""The C-ABI library builds, loads, and exports every symbol include/uniter_hip.h declares — and the test hooks of
include/uniter_hip_test.h, which are deliberately NOT part of that boundary (no GPU needed)."""
import ctypes
import os
import re

from uniter_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="uniter_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(uniter_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_and_loads():
    assert os.path.exists(_lib.LIB_PATH), "run `python -c 'import __graft_entry__ as g; g.build()'` first"
    lib = _lib.load()
    assert lib.uniter_hip_abi_version() == _lib.ABI_VERSION == 8


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared()
    hooks = _declared("uniter_hip_test.h")
    assert len(names) >= 40
    assert not [n for n in names if "_debug_" in n], "test hooks belong in include/uniter_hip_test.h"
    assert hooks and all("_debug_" in n for n in hooks), hooks
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names + hooks:
        assert hasattr(raw, n), "libuniter_hip.so lacks %s" % n
        assert n in _lib.SIGNATURES, "uniter_amd/_lib.py has no binding for %s" % n
    for n in _lib.SIGNATURES:
        assert n in names or n in hooks, "binding %s is declared in neither header" % n


def test_argument_errors_are_reported_not_thrown():
    lib = _lib.load()
    # null pointers -> negative status + message, no crash, no GPU touched
    rc = lib.uniter_gemm_bias_fwd(None, None, None, None, 1, 64, 64, None)
    assert rc < 0
    assert b"null pointer" in lib.uniter_hip_last_error()
    s = _lib.UniterEncoderShape()
    s.B, s.L, s.H, s.heads, s.I = 2, 600, 128, 2, 128
    assert lib.uniter_encoder_layer_act_bytes(ctypes.byref(s)) == 0          # L > 512 (max_position_embeddings) is rejected
    assert b"512" in lib.uniter_hip_last_error()
    s.L = 96
    assert lib.uniter_encoder_layer_act_bytes(ctypes.byref(s)) > 0
    assert lib.uniter_encoder_scratch_bytes(ctypes.byref(s)) > 0


def test_checked_wrapper_raises():
    import pytest
    with pytest.raises(_lib.UniterHipError):
        _lib.C.uniter_attention_fwd(None, None, None, None, 1, 1, 1, 0.0, 0, 0, None)


def test_size_queries_are_not_status_checked():
    """A C-ABI function that returns a byte count must never go through the status-code wrapper (a non-zero size would
    read as an error)."""
    import ctypes
    from uniter_amd import _lib
    sized = [n for n, (res, _a) in _lib.SIGNATURES.items() if res is ctypes.c_size_t]
    assert "uniter_attention_bwd_workspace_bytes" in sized
    assert all(n in _lib._NO_STATUS for n in sized)
