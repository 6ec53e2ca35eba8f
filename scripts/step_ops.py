"""Which ATen operators (and which Python lines of uniter_amd) launch the small copy / cast / fill kernels of one optimizer
step of the headline workload.  torch.profiler with stacks; prints, per step: kernel launches by name, and for the copy-like
operators the innermost uniter_amd frame that issued them."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

from uniter_amd.train import StepRunner

name = sys.argv[1] if len(sys.argv) > 1 else 'c2'
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
r = StepRunner(name, dev)
for _ in range(6):
    r.train_step()
torch.cuda.synchronize()
STEPS = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(STEPS):
        r.train_step()
    torch.cuda.synchronize()
ev = prof.events()
kern = collections.Counter()
ktime = collections.Counter()
for e in ev:
    if e.device_type is not None and str(e.device_type).endswith("CUDA"):
        kern[e.name[:70]] += 1
        ktime[e.name[:70]] += e.device_time if hasattr(e, "device_time") else e.cuda_time
print("== device kernels per step (count, us) ==")
for k, c in sorted(kern.items(), key=lambda kv: -ktime[kv[0]]):
    print("%6.1f %9.1f  %s" % (c / STEPS, ktime[k] / STEPS, k))
print("== copy-like operators by issuing uniter_amd frame (per step) ==")
ops = collections.Counter()
for e in ev:
    if e.name in ("aten::copy_", "aten::_to_copy", "aten::cat", "aten::contiguous", "aten::clone", "aten::fill_", "aten::zero_",
                  "aten::zeros", "aten::zeros_like", "aten::empty_like", "aten::index", "aten::flip", "aten::mean", "aten::sum",
                  "aten::mul", "aten::add", "aten::eq", "aten::masked_fill", "aten::masked_fill_"):
        frame = "?"
        for s in (e.stack or []):
            if "uniter_amd" in s or "bench.py" in s:
                frame = s.strip()[-110:]
                break
        shape = str(e.input_shapes)[:60] if e.input_shapes else ""
        ops[(e.name, frame, shape)] += 1
for (n, f, sh), c in sorted(ops.items(), key=lambda kv: -kv[1])[:70]:
    print("%5.1f  %-18s %-62s %s" % (c / STEPS, n, sh, f))
