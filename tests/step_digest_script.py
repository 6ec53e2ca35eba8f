"""Helper of tests/test_experiments_gpu.py, run as a subprocess on the GPU box: N optimizer steps of a bench workload
(uniter_amd.train.StepRunner, dropout on, fixed seeds) and one JSON line with SHA-256 digests of the loss of every step and
of every parameter after the last one.  Two runs of this script under different library switches (environment variables
read once at load time) must print the same digests when the switch only re-orders work that is order-independent.

    python tests/step_digest_script.py [workload=c2] [steps=2]
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def _digest(t):
    return hashlib.sha256(t.detach().contiguous().cpu().reshape(-1).view(torch.uint8).numpy().tobytes()).hexdigest()[:16]


def main():
    from uniter_amd import ops
    from uniter_amd.train import StepRunner
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    runner = StepRunner(name, dev, seed=77)
    ops.manual_seed(4321)
    losses = []
    for _ in range(steps):
        losses.append(_digest(runner.train_step().float()))
    torch.cuda.synchronize()
    params = {n: _digest(p) for n, p in runner.model.named_parameters()}
    whole = hashlib.sha256("".join(params[n] for n in sorted(params)).encode()).hexdigest()[:16]
    print(json.dumps({"workload": name, "steps": steps, "losses": losses, "params": whole, "n_params": len(params),
                      "per_param": params}))


if __name__ == "__main__":
    main()
