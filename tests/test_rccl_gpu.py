"""RCCL on the one GPU a test box has: (1) the gradient reducer's real backward-hook path over a forced one-rank RCCL
group must reproduce the collective-free step bit for bit and must actually issue its bucket allreduces; (2) the raw
C-ABI communicator (uniter_comm_*: unique id -> init -> allreduce / broadcast / allgather -> destroy) on a one-rank
communicator.  Multi-rank behaviour is covered over gloo in tests/test_distributed_gloo.py; an N > 1 RCCL measurement needs
the driver's 8-GPU node.  Reference: utils/distributed.py:16-43,100-148."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gradient_reducer_over_one_rank_rccl_matches_plain_step():
    env = dict(os.environ, UNITER_DIST_FORCE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_script.py")], capture_output=True, text=True,
                       timeout=600, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, (r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads(lines[-1])
    assert out["backend"] == "nccl"
    assert out["identical"], out
    # per step: one bucket per encoder layer (layers_per_bucket=1) + the non-encoder remainder ranges
    assert out["allreduce_calls"] >= 2 * (out["encoder_layers"] + 1), out
    assert out["elements_reduced_per_step"] == out["arena_elements"], out     # every gradient element exactly once


def test_raw_rccl_communicator_one_rank():
    code = r'''
import ctypes, sys, torch
sys.path.insert(0, %r)
from uniter_amd import _lib
from uniter_amd._lib import C
torch.cuda.set_device(0)
uid = (ctypes.c_uint8 * 128)()
C.uniter_comm_unique_id(uid)
comm = ctypes.c_void_p()
C.uniter_comm_init(uid, 0, 1, ctypes.byref(comm))
st = _lib.stream_ptr()
a = torch.randn(100003, device="cuda").to(torch.bfloat16); a0 = a.clone()
C.uniter_comm_allreduce(comm, a.data_ptr(), a.numel(), 0, st)
f = torch.randn(4099, device="cuda"); f0 = f.clone()
C.uniter_comm_allreduce(comm, f.data_ptr(), f.numel(), 1, st)
b = torch.arange(1000, device="cuda", dtype=torch.int32); b0 = b.clone()
C.uniter_comm_broadcast(comm, b.data_ptr(), b.numel() * 4, 0, st)
src = torch.arange(64, device="cuda", dtype=torch.uint8); dst = torch.zeros(64, device="cuda", dtype=torch.uint8)
C.uniter_comm_allgather(comm, src.data_ptr(), dst.data_ptr(), 64, st)
torch.cuda.synchronize()
assert torch.equal(a, a0) and torch.equal(f, f0) and torch.equal(b, b0) and torch.equal(src, dst)
C.uniter_comm_destroy(comm)
print("RCCL_OK")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), cwd=ROOT)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
