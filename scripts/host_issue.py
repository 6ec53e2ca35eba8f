"""Pure host cost of issuing one optimizer step of the headline workload: the GPU is idle (synchronised) when a step starts, so
nothing the host does can be waiting for queue space; the time until train_step() returns is what the Python + C-ABI side costs.
The step is GPU-bound as long as this stays below the GPU time of a step (bench.py ms_per_step)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uniter_amd.train import StepRunner

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
r = StepRunner(sys.argv[1] if len(sys.argv) > 1 else 'c2', dev)
for _ in range(8):
    r.train_step()
torch.cuda.synchronize()
issue, wall = [], []
for _ in range(30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.train_step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    issue.append((t1 - t0) * 1e3)
    wall.append((t2 - t0) * 1e3)
issue.sort(); wall.sort()
print("host issue per step: median %.3f ms, min %.3f ms | step from an idle GPU: median %.3f ms" % (issue[len(issue) // 2], issue[0], wall[len(wall) // 2]))
