import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import bench
from uniter_amd.optim import build_optimizer, clip_grad_norm_
from uniter_amd.utils.arena import flatten_model
from uniter_amd.utils.misc import Struct
dev = torch.device("cuda", 0)
cfg = "/tmp/ta_cfg.json"; bench.write_cfg(cfg)
model = bench.build_model(dev, cfg, 77); arena = flatten_model(model)
opt = build_optimizer(model, Struct(bench.TRAIN))
arena.grad.normal_(0, 0.01)
for _ in range(3): opt.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): opt.step()
e1.record(); e1.synchronize()
n = arena.numel
us = e0.elapsed_time(e1) * 1000 / 20
print("adamw step %.1f us, %.2f TB/s (28 B/param, %d params)" % (us, 28.0 * n / us * 1e-6, n))
