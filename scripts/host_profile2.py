"""Host-side cost of one optimizer step of the headline workload (bench.py c2), by Python function (tottime)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uniter_amd.train import StepRunner

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
r = StepRunner('c2', dev)
for _ in range(8):
    r.train_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    r.train_step()
ti = time.perf_counter() - t0
torch.cuda.synchronize()
tw = time.perf_counter() - t0
print("issue %.3f ms/step, wall %.3f ms/step" % (ti / 30 * 1e3, tw / 30 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    r.train_step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
