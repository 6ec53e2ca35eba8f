"""How far ahead of the GPU is the host at the segment boundaries of an optimizer step (c2)?

At every boundary (step start, after forward issue, after backward issue, after clip, after step) the host records an event and its
own clock.  After the run, the GPU completion time of every event (elapsed from a base event recorded right after a device
synchronize, when both clocks are aligned to within the synchronize latency) minus the host time at which the event was ISSUED is
the host's lead at that boundary: a lead near zero means the GPU had nothing queued when the host got there (GPU idle = host
latency), a large lead means the GPU was still busy with earlier work.  Also prints the GPU-side gap between consecutive boundary
events that the kernels of the segment do not explain (segment GPU time is printed for reference)."""
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
N = 24
names = ["step start", "forward issued", "backward issued", "clip issued", "step+zero issued"]
ev = [[torch.cuda.Event(enable_timing=True) for _ in names] for _ in range(N)]
host = [[0.0] * len(names) for _ in range(N)]
base = torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
base.record()
torch.cuda.synchronize()
t_base = time.perf_counter()
for i in range(N):
    r._schedule_lr()
    host[i][0] = time.perf_counter(); ev[i][0].record()
    loss = r.model(batch, compute_loss=True).mean()
    host[i][1] = time.perf_counter(); ev[i][1].record()
    loss.backward()
    host[i][2] = time.perf_counter(); ev[i][2].record()
    clip_grad_norm_(r.optimizer, r.opts.grad_norm, grad_scale=1.0)
    host[i][3] = time.perf_counter(); ev[i][3].record()
    r.optimizer.step()
    r.optimizer.zero_grad()
    host[i][4] = time.perf_counter(); ev[i][4].record()
torch.cuda.synchronize()
print("boundary            host issue (ms into step) | GPU reaches it (ms into step) | host lead (ms)   [mean of steps 8..%d]" % (N - 1))
for k, nm in enumerate(names):
    hs, gs, ld = [], [], []
    for i in range(8, N):
        h0 = (host[i][0] - t_base) * 1e3
        g0 = base.elapsed_time(ev[i][0])
        h = (host[i][k] - t_base) * 1e3
        g = base.elapsed_time(ev[i][k])
        hs.append(h - h0); gs.append(g - g0); ld.append(g - h)
    print("%-18s %10.3f %28.3f %22.3f (min %.3f)" % (nm, sum(hs) / len(hs), sum(gs) / len(gs), sum(ld) / len(ld), min(ld)))
per = (base.elapsed_time(ev[N - 1][4]) - base.elapsed_time(ev[8][4])) / (N - 1 - 8)
print("ms per step (GPU clock) %.3f" % per)
