"""Loaders around a PyTorch DataLoader.  Reference: data/loader.py:17-142.

`PrefetchLoader` overlaps the host-to-device copy of batch k+1 with the training step on batch k: the copy is issued on
its own HIP stream from pinned host memory, the consumer stream waits on it with an event (no host sync), and the
caching allocator is told which stream uses the tensors (`record_stream`).  Two things the MI355X path adds:
  * `float_dtype` — floating-point inputs (the 2048-d region features are 9.4 MB per 32x36 batch in fp32) can be cast
    to the model's bf16 on the copy stream, so the encoder's input kernels read half the bytes;
  * `batch['seq_lens']` — the per-example token counts are taken from the attention mask while it is still on the host,
    which lets the padding-free encoder path run without ever reading a length back from the GPU.
`MetaLoader` draws the task of each optimisation step from the configured mix; with several ranks the draw of rank 0 is
broadcast through torch.distributed (RCCL or gloo) instead of Horovod."""
import random

import torch

from .collate import sequence_lengths


def move_to_cuda(batch, device=None, float_dtype=None):
    """Recursively issue non-blocking copies of every tensor in a batch (data/loader.py:59-70)."""
    if isinstance(batch, torch.Tensor):
        dtype = float_dtype if (float_dtype is not None and batch.is_floating_point()) else None
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        return batch.to(device=device, dtype=dtype, non_blocking=True)
    if isinstance(batch, list):
        return [move_to_cuda(t, device, float_dtype) for t in batch]
    if isinstance(batch, tuple):
        return tuple(move_to_cuda(t, device, float_dtype) for t in batch)
    if isinstance(batch, dict):
        return {k: (v if k == 'seq_lens' else move_to_cuda(v, device, float_dtype)) for k, v in batch.items()}
    return batch


def record_cuda_stream(batch, stream=None):
    """Tell the caching allocator that the current stream uses these tensors (data/loader.py:73-84)."""
    if isinstance(batch, torch.Tensor):
        if batch.is_cuda:
            batch.record_stream(stream if stream is not None else torch.cuda.current_stream(batch.device))
    elif isinstance(batch, (list, tuple)):
        for t in batch:
            record_cuda_stream(t, stream)
    elif isinstance(batch, dict):
        for t in batch.values():
            record_cuda_stream(t, stream)


def _pin(batch):
    if isinstance(batch, torch.Tensor):
        return batch if (batch.is_cuda or batch.is_pinned()) else batch.pin_memory()
    if isinstance(batch, list):
        return [_pin(t) for t in batch]
    if isinstance(batch, tuple):
        return tuple(_pin(t) for t in batch)
    if isinstance(batch, dict):
        return {k: _pin(v) for k, v in batch.items()}
    return batch


class PrefetchLoader(object):
    """Iterates like the wrapped loader but hands out device-resident batches whose copy was started one step earlier.
    Without a GPU (unit tests, CPU tools) it degrades to a pass-through that still adds `seq_lens`."""

    def __init__(self, loader, device=None, float_dtype=None, add_seq_lens=True):
        self.loader = loader
        self.float_dtype = float_dtype
        self.add_seq_lens = add_seq_lens
        self.use_gpu = torch.cuda.is_available()
        self.device = torch.device(device) if device is not None else (
            torch.device("cuda", torch.cuda.current_device()) if self.use_gpu else torch.device("cpu"))
        self.stream = torch.cuda.Stream(device=self.device) if self.use_gpu else None
        self.batch = None

    def __iter__(self):
        it = iter(self.loader)
        self.preload(it)
        batch = self.next(it)
        while batch is not None:
            yield batch
            batch = self.next(it)

    def __len__(self):
        return len(self.loader)

    def _annotate(self, batch):
        body = batch[1] if (isinstance(batch, tuple) and len(batch) == 2 and isinstance(batch[1], dict)) else batch
        if (self.add_seq_lens and isinstance(body, dict) and 'seq_lens' not in body
                and isinstance(body.get('attn_masks'), torch.Tensor) and not body['attn_masks'].is_cuda):
            body['seq_lens'] = sequence_lengths(body['attn_masks'])
        return batch

    def preload(self, it):
        try:
            batch = next(it)
        except StopIteration:
            self.batch = None
            return
        batch = self._annotate(batch)
        if not self.use_gpu:
            self.batch = batch
            return
        with torch.cuda.stream(self.stream):
            self.batch = move_to_cuda(_pin(batch), self.device, self.float_dtype)

    def next(self, it):
        if self.use_gpu:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
        batch = self.batch
        if batch is not None and self.use_gpu:
            record_cuda_stream(batch)
        self.preload(it)
        return batch

    def __getattr__(self, name):
        return getattr(self.__dict__['loader'], name)


class MetaLoader(object):
    """Infinite iterator over (task name, batch): `loaders` maps a task name to a loader or a (loader, ratio) pair; a task
    is drawn (proportionally to its ratio) once per `accum_steps` batches, so that all micro-batches of one optimiser step
    come from one task; exhausted loaders restart (data/loader.py:17-56).

    merge_micro_batches=True hands out ONE batch per optimiser step instead: the step's `accum_steps` batches concatenated by
    data/merge.py (`batch['micro']` carries what `accumulated_loss` needs to reproduce the loop's per-micro-batch means); the
    training loop then runs with an accumulation count of 1.  Done on the host, before PrefetchLoader's copy."""

    def __init__(self, loaders, accum_steps=1, distributed=False, rng=None, merge_micro_batches=False):
        if not isinstance(loaders, dict) or not loaders:
            raise ValueError("loaders must be a non-empty dict")
        self.name2loader, self.name2iter, self.sampling_pools = {}, {}, []
        for name, entry in loaders.items():
            loader, ratio = entry if isinstance(entry, tuple) else (entry, 1)
            self.name2loader[name] = loader
            self.name2iter[name] = iter(loader)
            self.sampling_pools.extend([name] * int(ratio))
        self.accum_steps = accum_steps
        self.distributed = distributed
        self.step = 0
        self._rng = rng if rng is not None else random
        self.merge_micro_batches = bool(merge_micro_batches) and int(accum_steps) > 1

    def _agree(self, task):
        if not (self.distributed and torch.distributed.is_available() and torch.distributed.is_initialized()
                and torch.distributed.get_world_size() > 1):
            return task
        box = [task]
        torch.distributed.broadcast_object_list(box, src=0)      # every process trains the same task
        return box[0]

    def __iter__(self):
        if not self.merge_micro_batches:
            yield from self._micro_batches()
            return
        from .merge import merge_batches
        group = []
        for task, batch in self._micro_batches():
            group.append(batch)
            if len(group) == self.accum_steps:
                yield task, merge_batches(group)
                group = []

    def _micro_batches(self):
        task = self.sampling_pools[0]
        while True:
            if self.step % self.accum_steps == 0:
                task = self._agree(self._rng.choice(self.sampling_pools))
            self.step += 1
            try:
                batch = next(self.name2iter[task])
            except StopIteration:
                self.name2iter[task] = iter(self.name2loader[task])
                batch = next(self.name2iter[task])
            yield task, batch
