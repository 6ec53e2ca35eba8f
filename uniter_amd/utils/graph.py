"""Whole-step hipGraph capture (host launch overhead -> one graph launch per optimizer step).

A UNITER-base step is ~550 kernel launches (our ~350 + the PyTorch head / glue ops); issued eagerly the host
needs ~7.6 ms for them — longer than the GPU needs to execute them.  `GraphedStep` captures one complete
optimizer step (forward, backward incl. the side-stream weight-gradient GEMMs, clip, fused AdamW, zero_grad) with
`torch.cuda.graph` and replays it.  What changes from step to step lives in device memory instead of kernel
arguments: the dropout counter (ops.enable_graph_rng) and the AdamW hyper-parameter table
(AdamW.enable_graph_mode / graph_prepare), so LR schedules and fresh dropout masks keep working under replay.

Inputs must be static tensors: copy every new batch INTO the tensors the step closure reads (`copy_`), exactly as
with any CUDA-graph training loop.
"""
import torch

from .. import ops


class GraphedStep(object):
    def __init__(self, step_fn, optimizer, device, warmup=3, pre_step=None):
        """step_fn(): one full eager optimizer step returning the loss tensor.  pre_step(): host-side work before every
        step (e.g. writing the scheduled lr into optimizer.param_groups)."""
        self.step_fn, self.optimizer, self.pre_step = step_fn, optimizer, pre_step
        self.device = torch.device(device)
        self.graph = None
        self.loss = None
        ops.enable_graph_rng(self.device)
        # eager warm-up on a side stream (allocators, lazy plans, caches), as torch.cuda.graph requires
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            self._eager_once()                       # creates gradients / optimizer plan
            optimizer.enable_graph_mode()
            for _ in range(max(int(warmup) - 1, 1)):
                self._eager_once()
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)

    def _host_prepare(self):
        if self.pre_step is not None:
            self.pre_step()
        if self.optimizer._graph is not None:
            self.optimizer.graph_prepare()

    def _eager_once(self):
        self._host_prepare()
        ops.graph_rng_step()
        return self.step_fn()

    def capture(self):
        self._host_prepare()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            ops.graph_rng_step()
            self.loss = self.step_fn()
        self.graph = g
        return self

    def __call__(self):
        if self.graph is None:
            return self._eager_once()
        self._host_prepare()
        self.graph.replay()
        return self.loss
