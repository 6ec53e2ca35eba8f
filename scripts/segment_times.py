"""GPU time of the segments of one optimizer step of the headline workload, from HIP events recorded between the Python
calls (forward | backward | reducer.finish | clip_grad_norm_ | optimizer.step + zero_grad): a segment's elapsed time minus
the kernels it contains is GPU idle time caused by host-side latency at that point of the step.  With UNITER_DIST_FORCE=1
(a one-rank RCCL group) the data-parallel path — bucketed backward, gradient reducer — is the one that runs."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uniter_amd.optim import clip_grad_norm_
from uniter_amd.train import StepRunner
from uniter_amd.utils import distributed as D

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
D.init()
r = StepRunner('c2', dev)
batch = r.batches['nlvr2']
for _ in range(8):
    r.train_step()
N = 30
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(N)]
host = [0.0] * 5
torch.cuda.synchronize()
for i in range(N):
    r._schedule_lr()
    t0 = time.perf_counter()
    ev[i][0].record()
    loss = r.model(batch, compute_loss=True).mean()
    t1 = time.perf_counter()
    ev[i][1].record()
    if r.reducer is not None:
        r.reducer.begin()
    loss.backward()
    t2 = time.perf_counter()
    ev[i][2].record()
    scale = r.reducer.finish() if r.reducer is not None else 1.0
    t3 = time.perf_counter()
    ev[i][3].record()
    clip_grad_norm_(r.optimizer, r.opts.grad_norm, grad_scale=scale)
    t4 = time.perf_counter()
    ev[i][4].record()
    r.optimizer.step()
    r.optimizer.zero_grad()
    t5 = time.perf_counter()
    ev[i][5].record()
    if i >= 5:
        for k, (a, b) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5))):
            host[k] += (b - a) * 1e3 / (N - 5)
torch.cuda.synchronize()
names = ["forward", "backward", "reducer.finish", "clip_grad_norm_", "step+zero_grad"]
tot = 0.0
for k in range(5):
    t = sum(ev[i][k].elapsed_time(ev[i][k + 1]) for i in range(5, N)) / (N - 5)
    tot += t
    print("%-16s GPU %8.3f ms | host issue %6.3f ms" % (names[k], t, host[k]))
gap = sum(ev[i][5].elapsed_time(ev[i + 1][0]) for i in range(5, N - 1)) / (N - 6)
print("%-16s %8.3f ms   (between steps)" % ("inter-step", gap))
print("sum %.3f ms (reducer %s)" % (tot + gap, "on" if r.reducer is not None else "off"))
