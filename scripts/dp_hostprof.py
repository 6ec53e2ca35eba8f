import os, sys, time, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("UNITER_DIST_FORCE", "1"); os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29547")
os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1"); os.environ.setdefault("LOCAL_RANK","0")
import torch, torch.distributed as dist
from uniter_amd.utils import distributed as D
from uniter_amd.train import StepRunner
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
D.init("nccl")
r = StepRunner('c2', dev, rank=0, world=1, reducer_layers_per_bucket=4)
for _ in range(5): r.train_step()
torch.cuda.synchronize()
# wrap
lib = __import__('uniter_amd._lib', fromlist=['x']).load()
T = {}
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); out = f(*a, **k); T.setdefault(key, []).append((time.perf_counter()-t)*1e6); return out
    setattr(obj, name, g)
wrap(dist, 'all_reduce', 'all_reduce')
red = r.reducer
orig_rr = red._reduce_range
def rr(lo, hi, bucket=None):
    t=time.perf_counter(); orig_rr(lo, hi, bucket=bucket); T.setdefault('reduce_range', []).append((time.perf_counter()-t)*1e6)
red._reduce_range = rr
orig_fin = red.finish
def fin(word_ids=None):
    t=time.perf_counter(); out = orig_fin(word_ids=None); T.setdefault('finish', []).append((time.perf_counter()-t)*1e6); return out
red.finish = fin
for _ in range(3):
    T.clear()
    t=time.perf_counter(); r.train_step(); T['step_host']=[(time.perf_counter()-t)*1e6]
    torch.cuda.synchronize()
print({k: [round(x) for x in v] for k, v in T.items()})
dist.destroy_process_group()
