"""GPU time of the segments of one optimizer step of the headline workload, from HIP events recorded between the Python
calls (forward | backward | clip_grad_norm_ | optimizer.step + zero_grad): a segment's elapsed time minus the kernels it
contains is GPU idle time caused by host-side latency at that point of the step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uniter_amd.optim import clip_grad_norm_
from uniter_amd.train import StepRunner

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
r = StepRunner('c2', dev)
batch = r.batches['nlvr2']
for _ in range(8):
    r.train_step()
N = 30
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(N)]
torch.cuda.synchronize()
for i in range(N):
    r._schedule_lr()
    ev[i][0].record()
    loss = r.model(batch, compute_loss=True).mean()
    ev[i][1].record()
    loss.backward()
    ev[i][2].record()
    clip_grad_norm_(r.optimizer, r.opts.grad_norm)
    ev[i][3].record()
    r.optimizer.step()
    r.optimizer.zero_grad()
    ev[i][4].record()
torch.cuda.synchronize()
names = ["forward", "backward", "clip_grad_norm_", "step+zero_grad"]
tot = 0.0
for k in range(4):
    t = sum(ev[i][k].elapsed_time(ev[i][k + 1]) for i in range(5, N)) / (N - 5)
    tot += t
    print("%-16s %8.3f ms" % (names[k], t))
gap = sum(ev[i][4].elapsed_time(ev[i + 1][0]) for i in range(5, N - 1)) / (N - 6)
print("%-16s %8.3f ms   (between steps)" % ("inter-step", gap))
print("sum %.3f ms" % (tot + gap))
