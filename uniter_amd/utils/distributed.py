"""Data-parallel helpers on RCCL (torch.distributed backend "nccl" IS RCCL on ROCm) — one process per GPU.

Reference: utils/distributed.py:16-209, which wraps Horovod 0.16 (`hvd.allreduce_` = average over ranks,
`hvd.broadcast_`, `hvd.allgather`).  The function names, argument meaning and in-place behaviour are kept:

    all_reduce_and_rescale_tensors(tensors, rescale_denom)      average over ranks, then / rescale_denom
    all_reduce_and_rescale_tensors_chunked(tensors, rescale_denom, buffer_size)
    broadcast_tensors(tensors, root_rank, buffer_size)
    all_gather_list(data) -> list (rank order)                  any picklable object
    any_broadcast(data, root_rank)

MI355X design notes (xGMI is point-to-point, 7 links x ~153 GB/s per GPU):
  * when the tensors are views of one flat arena (utils/arena.py) the collective runs IN PLACE on the
    covering range — no flatten / unflatten copies (the reference does 2 x 228 copy kernels per step);
  * `GradientReducer` splits the gradient arena into per-layer buckets and starts each bucket's allreduce
    on a side HIP stream as soon as backward has produced it (hook from UniterEncoder), overlapping
    communication with the remaining backward compute; the 1/world averaging is folded into the fused
    gradient-norm / AdamW kernels instead of a separate pass over the gradients.
The CPU tests run the same code over gloo with world_size 2.
"""
import ctypes
import math
import os

import torch
import torch.distributed as dist

# ------------------------------------------------------------------------------------------------------
# process-group plumbing (hvd.init / rank / size / local_rank)
# ------------------------------------------------------------------------------------------------------


def init(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run).
    Single-process runs need no initialisation."""
    if dist.is_available() and dist.is_initialized():
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and not _FORCE:
        return
    if world <= 1:                     # UNITER_DIST_FORCE=1: a one-rank group, to exercise the collective path on one GPU
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_PORT", "29517")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank())
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend=backend)


_FORCE = os.environ.get("UNITER_DIST_FORCE", "0") == "1"


def _on():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE)


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else int(os.environ.get("RANK", "0"))


def size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def local_rank():
    return int(os.environ.get("LOCAL_RANK", "0"))


# ------------------------------------------------------------------------------------------------------
# flat-range detection
# ------------------------------------------------------------------------------------------------------
def _covering_flat(tensors):
    """If every tensor is a contiguous window of ONE storage (views of a flat arena, also after `.data`),
    return a 1-D tensor over that storage spanning all of them (gaps are the arena's zero padding); else None.
    Only used when the windows are dense enough that reducing the whole span is cheaper than copying."""
    first = tensors[0]
    try:
        st = first.untyped_storage()
    except Exception:
        return None
    sptr = st.data_ptr()
    lo, hi, used = None, None, 0
    for t in tensors:
        if t.dtype != first.dtype or t.device != first.device or not t.is_contiguous():
            return None
        if t.untyped_storage().data_ptr() != sptr:
            return None
        start = t.storage_offset()
        end = start + t.numel()
        used += t.numel()
        lo = start if lo is None else min(lo, start)
        hi = end if hi is None else max(hi, end)
    if len(tensors) < 2 or (hi - lo) > used * 1.05 + 4096:
        return None        # a single tensor, or too sparse a cover: take the generic path
    return torch.empty(0, dtype=first.dtype, device=first.device).set_(st, lo, (hi - lo,))


def _flatten(tensors):
    flat = tensors[0].new_zeros(sum(t.numel() for t in tensors))
    o = 0
    for t in tensors:
        flat[o:o + t.numel()].copy_(t.reshape(-1))
        o += t.numel()
    return flat


def _unflatten(flat, tensors):
    o = 0
    for t in tensors:
        t.copy_(flat[o:o + t.numel()].view_as(t))
        o += t.numel()


# ------------------------------------------------------------------------------------------------------
# reference API
# ------------------------------------------------------------------------------------------------------
def all_reduce_and_rescale_tensors(tensors, rescale_denom):
    """In-place: every tensor becomes mean-over-ranks(tensor) / rescale_denom, all tensors in ONE collective."""
    tensors = list(tensors)
    if not tensors:
        return
    world = size()
    scale = 1.0 / (float(world) * float(rescale_denom))
    flat = _covering_flat(tensors)
    if flat is not None:
        if _on():
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if scale != 1.0:
            flat.mul_(scale)
        return
    buf = _flatten(tensors)
    if _on():
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    if scale != 1.0:
        buf.mul_(scale)
    _unflatten(buf, tensors)


def _chunks(tensors, buffer_size):
    """Greedy packing of tensors into groups of at most buffer_size bytes; oversized tensors travel alone."""
    group, filled = [], 0
    for t in tensors:
        nbytes = t.numel() * t.element_size()
        if nbytes > buffer_size:
            yield [t], True
            continue
        if filled + nbytes > buffer_size and group:
            yield group, False
            group, filled = [], 0
        group.append(t)
        filled += nbytes
    if group:
        yield group, False


def all_reduce_and_rescale_tensors_chunked(tensors, rescale_denom, buffer_size=10485760):
    """Same result as all_reduce_and_rescale_tensors, in collectives of at most buffer_size bytes."""
    world = size()
    scale = 1.0 / (float(world) * float(rescale_denom))
    for group, alone in _chunks(list(tensors), buffer_size):
        if alone:
            t = group[0]
            if _on():
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
            if scale != 1.0:
                t.mul_(scale)
        else:
            all_reduce_and_rescale_tensors(group, rescale_denom)


def broadcast_tensors(tensors, root_rank, buffer_size=10485760):
    """In-place broadcast of every tensor from root_rank (parameters at start-up, pretrain.py:225)."""
    tensors = list(tensors)
    if not tensors or not _on():
        return
    flat = _covering_flat(tensors)
    if flat is not None:
        dist.broadcast(flat, src=root_rank)
        return
    for group, alone in _chunks(tensors, buffer_size):
        if alone:
            dist.broadcast(group[0], src=root_rank)
        else:
            buf = _flatten(group)
            dist.broadcast(buf, src=root_rank)
            _unflatten(buf, group)


def all_gather_list(data):
    """Gathers arbitrary picklable data from all ranks into a list ordered by rank."""
    if not _on():
        return [data]
    out = [None] * size()
    dist.all_gather_object(out, data)
    return out


def any_broadcast(data, root_rank):
    """Broadcast arbitrary picklable data from root_rank to all ranks."""
    if not _on():
        return data
    box = [data if rank() == root_rank else None]
    dist.broadcast_object_list(box, src=root_rank)
    return box[0]


# ------------------------------------------------------------------------------------------------------
# overlapped, bucketed gradient allreduce
# ------------------------------------------------------------------------------------------------------
class _LayerHook(object):
    """callable(layer_index) + the set of layers at which it actually has work to do."""

    def __init__(self, fn, ready_layers, joins_side_stream=False, defer_wgrad_join=False, grad_buckets=None):
        self.fn = fn
        self.ready_layers = ready_layers
        self.joins_side_stream = joins_side_stream      # ops._EncoderFn.backward: see uniter_encoder_defer_side_join
        self.defer_wgrad_join = defer_wgrad_join        # one backward call, returns without joining the weight-gradient stream
        self.grad_buckets = grad_buckets                # callable -> layers per bucket for uniter_encoder_set_grad_buckets (0 = off)

    def __call__(self, layer_index):
        return self.fn(layer_index)


class GradientReducer(object):
    """Bucketed in-place sum-allreduce of a gradient arena, overlapped with backward.

    Usage (one optimizer step, accumulation boundary only — earlier micro-steps just accumulate):

        reducer = GradientReducer(arena, model.uniter.encoder)        # once
        reducer.begin()                     # before loss.backward() of the LAST micro-step
        loss.backward()                     # encoder hook fires per layer -> bucket allreduce on a side stream
        scale = reducer.finish()            # joins the side stream; returns 1/world
        clip_grad_norm_(optimizer, max_norm, grad_scale=scale); optimizer.step()

    Buckets are contiguous arena ranges: one per `layers_per_bucket` encoder layers (reverse order, as backward
    produces them) plus one for everything outside the encoder (embeddings, pooler, heads), reduced at finish()
    (heads finish first in backward, embeddings last: both are small next to the encoder).

    On the GPU the encoder's backward stays ONE call with ONE deferred weight-gradient launch (round 4; include/uniter_hip.h
    "Gradient buckets"): the launch completes the buckets in order and raises a flag per bucket, and a bucket's allreduce is
    enqueued on the communication stream behind a wait for that flag — rounds 1-3 cut the backward into one call per bucket,
    whose shorter launches filled the chip less well (+12 % per step before a byte crossed xGMI).  UNITER_AMD_DP_SINGLE_LAUNCH=0
    keeps the per-bucket calls.

    finish(word_ids=...) with UNITER_AMD_DP_SPARSE_WORD=1 (opt-in) exchanges the word-embedding gradient as touched rows when the
    step's loss reaches that table only through the input lookup (every task but MLM, whose tied decoder makes the gradient dense): each rank contributes the rows of
    the tokens in its batch (all-gather of ids and rows, <= B*Lt x H per rank instead of a vocab x H allreduce).
    """

    def __init__(self, arena, encoder=None, layers_per_bucket=3, word_embeddings=None, word_ids_cap=None, single_launch=None):
        self.arena = arena
        self.word_compact = False             # tests: segment sums on the gathered rows even for a small vocabulary
        self.word_ids_cap = word_ids_cap      # most word ids (batch x padded text length) any rank hands to finish(); None: agreed per step
        self.encoder = encoder
        self.buckets = []          # (lo, hi) element ranges
        self.layer_bucket = {}
        self.rest = []
        self._armed = False
        self._stream = None
        self._pending = []
        self.layers_per_bucket = int(layers_per_bucket)
        if single_launch is None:              # default: on, unless the environment (or bench.py's start-up self-check) says otherwise
            single_launch = os.environ.get("UNITER_AMD_DP_SINGLE_LAUNCH", "1") != "0"
        self.single_launch = (encoder is not None and bool(arena.grad.is_cuda) and bool(single_launch)
                              and (len(list(encoder.layer)) + self.layers_per_bucket - 1) // self.layers_per_bucket <= 24)
        covered = []
        if encoder is not None:
            layers = list(encoder.layer)
            n = len(layers)
            for hi_l in range(n, 0, -layers_per_bucket):
                lo_l = max(0, hi_l - layers_per_bucket)
                params = [p for l in range(lo_l, hi_l) for p in layers[l].parameters()]
                span = arena.span(params)
                self.layer_bucket[lo_l] = len(self.buckets)     # ready once layer lo_l's backward is enqueued
                self.buckets.append(span)
                covered.append(span)
            if self.single_launch:
                # no cut points: the whole stack is one backward call; the hook still fires per layer (after the call)
                encoder.grad_ready_hook = _LayerHook(self._on_layer, set(), joins_side_stream=False, defer_wgrad_join=True,
                                                     grad_buckets=lambda: self.layers_per_bucket if self._armed and _on() else 0)
            else:
                # backward only has to hand control back at the layers that complete a bucket (ops._EncoderFn.backward cuts
                # the stack there instead of after every layer)
                encoder.grad_ready_hook = _LayerHook(self._on_layer, set(self.layer_bucket.keys()),
                                                     joins_side_stream=bool(arena.grad.is_cuda))
        covered.sort()
        pos = 0
        for lo, hi in covered:
            if lo > pos:
                self.rest.append((pos, lo))
            pos = max(pos, hi)
        if pos < arena.numel:
            self.rest.append((pos, arena.numel))
        # Parameters that sit behind the encoder in the arena (pooler, task heads) are downstream of it in the model: their
        # gradients are final before the encoder's backward starts, so their ranges go out with the first bucket instead
        # of at finish(), where only the embeddings (final last) are left.
        enc_hi = covered[-1][1] if covered else 0
        self.rest_early = [r for r in self.rest if covered and r[0] >= enc_hi]
        self.rest = [r for r in self.rest if r not in self.rest_early]
        self._early_done = False
        # the word-embedding table's range (sparse exchange in finish): must lie inside one `rest` range
        self.word_span = None
        if word_embeddings is not None:
            lo, hi = arena.span([word_embeddings])
            if any(r[0] <= lo and hi <= r[1] for r in self.rest):
                self.word_span = (lo, hi, int(word_embeddings.shape[0]), int(word_embeddings.shape[1]))

    def begin(self):
        self._armed = True
        self.last_flag_waits = 0        # buckets of this step whose allreduce went behind a flag wait (tests)
        self._tokens = {}               # single-launch mode: bucket -> (flag address, value) noted by the backward hook
        self._pending = []
        self._early_done = False
        self._chain_done = None         # single-launch mode: event on the compute stream at the end of the encoder's backward chain
        if _on() and self.arena.grad.is_cuda and self._stream is None:
            self._stream = torch.cuda.Stream()

    def _reduce_range(self, lo, hi, token=None, after=None):
        """Sum-allreduce of arena.grad[lo:hi] on the communication stream.  token = (flag address, value) of a gradient bucket
        of the single deferred launch: the collective goes behind a wait for that flag; else behind `after` (an event recorded
        earlier on the compute stream) or, without one, behind the compute stream as it stands now (and, for encoder ranges of the
        per-bucket path, the library's weight-gradient stream)."""
        if not _on() or hi <= lo:
            return
        g = self.arena.grad[lo:hi]
        if g.is_cuda:
            from .. import _lib
            lib = _lib.load()
            if token is not None:
                lib.uniter_hip_stream_wait_value32(ctypes.c_void_p(self._stream.cuda_stream), ctypes.c_void_p(token[0]),
                                                   ctypes.c_uint32(token[1]))
                self.last_flag_waits += 1
            else:
                ev = after
                if ev is None:
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream())
                self._stream.wait_event(ev)
                if not self.single_launch:
                    # the weight gradients of the bucket come from the library's side stream, which the current stream has not
                    # been made to wait for between layer ranges (uniter_encoder_defer_side_join)
                    lib.uniter_encoder_side_join(ctypes.c_void_p(self._stream.cuda_stream))
            with torch.cuda.stream(self._stream):
                self._pending.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True))
        else:
            self._pending.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True))

    def _exchange_word_rows(self, word_ids):
        """Sum of the ranks' word-embedding gradients when each is non-zero only in the rows of its own tokens: all-gather of
        (ids, rows of first occurrences) and a local fp32 row sum, written back in place.  Same result as the dense sum-allreduce
        up to the summation order (exactly the same for two ranks); no host synchronisation, fixed shapes."""
        lo, hi, V, H = self.word_span
        g = self.arena.grad[lo:hi].view(V, H)
        ids = word_ids.reshape(-1).to(g.device, torch.int64)
        # ranks pad their text to their own batch's longest sentence: agree on one length (the configured cap, or the largest of
        # this step — one scalar max-allreduce and a host read) and fill up with id 0, whose extra copies are duplicates below
        # The count is validated COLLECTIVELY: every rank learns the largest count of the step before any rank can raise, so an
        # overflow of word_ids_cap is an error on all ranks at the same point instead of one rank raising while the others sit
        # in all_gather.
        nmax = torch.tensor([ids.numel()], dtype=torch.int64, device=g.device)
        dist.all_reduce(nmax, op=dist.ReduceOp.MAX)
        nmax = int(nmax.item())
        n = self.word_ids_cap if self.word_ids_cap is not None else nmax
        if nmax > n:
            raise ValueError("GradientReducer: %d word ids on some rank in this step exceed word_ids_cap=%d" % (nmax, n))
        if ids.numel() < n:
            ids = torch.cat([ids, ids.new_zeros(n - ids.numel())])
        # first occurrence of every id in this rank's batch (the others contribute a zero row): sort, compare neighbours
        sid, order = torch.sort(ids, stable=True)
        first = torch.ones(n, dtype=torch.bool, device=g.device)
        first[1:] = sid[1:] != sid[:-1]
        rows = g.index_select(0, sid) * first.unsqueeze(1).to(g.dtype)
        world = size()
        all_ids = torch.empty(world * n, dtype=torch.int64, device=g.device)
        all_rows = torch.empty(world * n, H, dtype=g.dtype, device=g.device)
        dist.all_gather_into_tensor(all_ids, sid)
        dist.all_gather_into_tensor(all_rows, rows)
        # rows of the union, summed over ranks in fp32 and rounded once; every touched row is rewritten by every rank alike
        uniq_ids = all_ids                       # (duplicates carry zero rows or the same id from another rank: index_add sums them)
        compact = V * H > (1 << 22) or self.word_compact
        acc = None if compact else torch.zeros(V, H, dtype=torch.float32, device=g.device)
        if acc is not None:
            acc.index_add_(0, uniq_ids, all_rows.float())
            touched = torch.zeros(V, dtype=torch.bool, device=g.device)
            touched[uniq_ids] = True
            g.copy_(torch.where(touched.unsqueeze(1), acc.to(g.dtype), g))
        else:
            # large vocabulary: accumulate on the compact list instead of a vocab-sized fp32 buffer.  Sort the gathered ids,
            # segment-sum equal ids, scatter the segment sums back.
            s2, o2 = torch.sort(uniq_ids, stable=True)
            r2 = all_rows.index_select(0, o2).float()
            head = torch.ones(s2.numel(), dtype=torch.bool, device=g.device)
            head[1:] = s2[1:] != s2[:-1]
            seg = torch.cumsum(head.to(torch.int64), 0) - 1                   # segment index of every gathered row
            sums = torch.zeros(s2.numel(), H, dtype=torch.float32, device=g.device)
            sums.index_add_(0, seg, r2)
            # every row of a segment receives the segment's sum (same id -> same value: the scatter below is race-free in value)
            g.index_copy_(0, s2, sums.index_select(0, seg).to(g.dtype))

    def _on_layer(self, layer_index):
        if not self._armed:
            return
        b = self.layer_bucket.get(layer_index)
        if b is None:
            return
        if self.single_launch and self.arena.grad.is_cuda and _on():
            # Called on autograd's thread right after the one backward call of the stack.  Only note the bucket's flag here; the
            # waits and collectives are enqueued by finish(), after backward() has returned: the ~0.3 ms of host work they cost
            # would otherwise sit between the encoder's backward and the embedding backward, whose kernels then reach the GPU
            # when the deferred launch already owns every CU and do not run until it ends (rocprofv3 trace, EXPERIMENTS.md section 5).
            from .. import _lib
            lib = _lib.load()
            if self._chain_done is None:
                # the compute stream as it stands right after the encoder's backward call: the heads' gradients (final before that
                # call) may be reduced from here on, and the communication stream must not be held back by what the compute stream
                # does later (the embedding backward runs beside the deferred launch whose buckets it waits for)
                self._chain_done = torch.cuda.Event()
                self._chain_done.record(torch.cuda.current_stream())
            n = ctypes.c_int32(0)
            lib.uniter_encoder_grad_bucket_count(ctypes.byref(n))
            if b < n.value:
                flag, val = ctypes.c_void_p(), ctypes.c_uint32(0)
                lib.uniter_encoder_bucket_token(ctypes.c_int32(b), ctypes.byref(flag), ctypes.byref(val))
                self._tokens[b] = (flag.value, val.value)
            else:
                self._tokens[b] = None                   # the call ran without buckets: finish() joins the weight-gradient stream
            return
        if not self._early_done:
            self._early_done = True
            for lo, hi in self.rest_early:
                self._reduce_range(lo, hi)
        self._reduce_range(*self.buckets[b])

    def finish(self, word_ids=None):
        """Reduce what is left (non-encoder parameters), wait for every bucket, return the averaging factor.
        word_ids: EVERY input id that contributed to the gradients since they were last zeroed (under gradient accumulation: the
        micro-batches' input_ids concatenated), when the word-embedding gradient is non-zero only in those rows (see the class
        docstring); rows of ids that are not listed would keep this rank's local value."""
        # rows instead of the dense table: OPT-IN (UNITER_AMD_DP_SPARSE_WORD=1).  The exchange costs ~0.2 ms of small kernels on
        # the communication stream (measured on one rank) against the ~0.26 ms a 44.5 MB ring allreduce is modelled to take at
        # 8 ranks, and it is only correct when `word_ids` lists EVERY id that contributed since zero_grad (rows of omitted ids
        # would silently keep the rank-local value): unmeasured on real links, so the dense allreduce stays the default
        sparse = self._armed and _on() and word_ids is not None and self.word_span is not None \
            and os.environ.get("UNITER_AMD_DP_SPARSE_WORD") == "1"
        if self._armed:
            if self.encoder is None:
                self._reduce_range(0, self.arena.numel)
            else:
                if not self._early_done:
                    for lo, hi in self.rest_early:
                        self._reduce_range(lo, hi, after=self._chain_done if self.single_launch else None)
                if self.single_launch and self._tokens:
                    # the buckets the backward hook noted, in the order the deferred launch completes them
                    if any(t is None for t in self._tokens.values()):
                        from .. import _lib
                        _lib.load().uniter_encoder_side_join_all(ctypes.c_void_p(self._stream.cuda_stream))
                    for b in sorted(self._tokens):
                        self._reduce_range(*self.buckets[b], token=self._tokens[b])
                    self._tokens = {}
                for lo, hi in self.rest:
                    if sparse and lo <= self.word_span[0] and self.word_span[1] <= hi:
                        # dense around the word table, rows inside it
                        self._reduce_range(lo, self.word_span[0])
                        self._reduce_range(self.word_span[1], hi)
                    else:
                        self._reduce_range(lo, hi)
        if sparse:
            if self.arena.grad.is_cuda:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                self._stream.wait_event(ev)
                with torch.cuda.stream(self._stream):
                    self._exchange_word_rows(word_ids)
            else:
                self._exchange_word_rows(word_ids)
        for w in self._pending:
            w.wait()
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)
        if self.arena.grad.is_cuda:
            from .. import _lib
            _lib.join_wgrads()           # backward ranges returned without joining the weight-gradient stream (and kept the
                                         # tensors its deferred launches read alive until now)
        self._pending = []
        self._armed = False
        return 1.0 / float(size())
