"""Data-parallel helpers over gloo with world_size 2 on CPU (the N>1 path of utils/distributed.py).

Semantics checked against the oracle's statement of Horovod 0.16 behaviour (utils/distributed.py:16-43,100-209):
average over ranks then divide, in-place; broadcast from root; object gather in rank order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, fn_name, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from uniter_amd.utils import distributed as D
    D.init(backend="gloo")
    try:
        globals()[fn_name](rank, world, D)
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def _spawn(fn_name, tmp_path, world=2):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, fn_name, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / ("ok%d" % r)).exists()


def _case_allreduce(rank, world, D):
    from oracle import uniter_oracle as O
    assert D.rank() == rank and D.size() == world
    gens = [torch.Generator().manual_seed(100 + r) for r in range(world)]
    shapes = [(7, 5), (3,), (64, 9), (1,)]
    per_rank = [[torch.randn(s, generator=g) for s in shapes] for g in gens]
    # separate storages -> flatten path
    mine = [t.clone() for t in per_rank[rank]]
    D.all_reduce_and_rescale_tensors(mine, 2.0)
    for i, t in enumerate(mine):
        torch.testing.assert_close(t, O.allreduce_average([per_rank[r][i] for r in range(world)], 2.0))
    # views of one flat arena (also through .data) -> in-place path, padding stays zero
    flat = torch.zeros(1024)
    views, o = [], 0
    for t in per_rank[rank]:
        n = t.numel()
        v = flat[o:o + n].view(t.shape)
        v.copy_(t)
        views.append(v.data)
        o += (n + 127) // 128 * 128
    assert D._covering_flat(views) is not None
    D.all_reduce_and_rescale_tensors(views, 1.0)
    for i, v in enumerate(views):
        torch.testing.assert_close(v, O.allreduce_average([per_rank[r][i] for r in range(world)]))
    # chunked variant, tiny buffer so that one tensor travels alone
    mine = [t.clone() for t in per_rank[rank]]
    D.all_reduce_and_rescale_tensors_chunked(mine, 1.0, buffer_size=256)
    for i, t in enumerate(mine):
        torch.testing.assert_close(t, O.allreduce_average([per_rank[r][i] for r in range(world)]))


def _case_broadcast_and_objects(rank, world, D):
    t = [torch.full((5,), float(rank)), torch.full((300,), float(rank) + 0.5), torch.full((2, 2), float(rank) - 1)]
    D.broadcast_tensors(t, 1, buffer_size=64)
    assert float(t[0][0]) == 1.0 and float(t[1][7]) == 1.5 and float(t[2][1, 1]) == 0.0
    got = D.all_gather_list({"rank": rank, "n": [rank] * (rank + 1)})
    assert [g["rank"] for g in got] == list(range(world)) and got[1]["n"] == [1, 1]
    assert D.any_broadcast("task_%d" % rank, 0) == "task_0"
    assert D.any_broadcast(("mlm", rank), 1) == ("mlm", 1)


def _case_gradient_reducer(rank, world, D):
    """Bucketed overlap path on CPU tensors: per-layer hooks + finish() == one big averaged allreduce."""
    from tests.common import IMG_DIM, LABEL_DIM, TINY_CONFIG
    from uniter_amd.model.pretrain import UniterForPretraining
    from uniter_amd.utils.arena import flatten_model
    torch.manual_seed(0)
    model = UniterForPretraining.from_pretrained(TINY_CONFIG, {}, img_dim=IMG_DIM, img_label_dim=LABEL_DIM)
    arena = flatten_model(model)
    reducer = D.GradientReducer(arena, model.uniter.encoder, layers_per_bucket=1)
    covered = sorted(reducer.buckets + reducer.rest + reducer.rest_early)
    assert covered[0][0] == 0 and covered[-1][1] == arena.numel
    for (a, b), (c, d) in zip(covered, covered[1:]):
        assert b == c                                         # buckets tile the arena exactly once
    g = torch.Generator().manual_seed(50 + rank)
    arena.grad.copy_(torch.randn(arena.numel, generator=g))
    mine = arena.grad.clone()
    reducer.begin()
    n_layers = len(model.uniter.encoder.layer)
    for l in reversed(range(n_layers)):                       # what _EncoderFn.backward does through the hook
        model.uniter.encoder.grad_ready_hook(l)
    scale = reducer.finish()
    assert scale == 1.0 / world
    both = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    torch.testing.assert_close(arena.grad * scale, sum(both) / world)
    # an un-armed reducer (earlier micro-steps of an accumulation window) must not communicate
    before = arena.grad.clone()
    model.uniter.encoder.grad_ready_hook(0)
    assert torch.equal(arena.grad, before)


def _case_sparse_word_rows(rank, world, D):
    """finish(word_ids=...): the word-embedding gradient exchanged as touched rows (all-gather of ids + rows) equals the dense
    sum-allreduce of the same gradients — both forms of the row sum (vocabulary-sized fp32 buffer / compact segment sums), bf16
    and fp32 arenas, duplicate tokens inside a rank and tokens shared between ranks."""
    from tests.common import IMG_DIM, LABEL_DIM, TINY_CONFIG
    from uniter_amd.model.pretrain import UniterForPretraining
    from uniter_amd.utils.arena import flatten_model
    for dtype, big_vocab in ((torch.float32, False), (torch.bfloat16, False), (torch.bfloat16, True)):
        torch.manual_seed(0)
        model = UniterForPretraining.from_pretrained(TINY_CONFIG, {}, img_dim=IMG_DIM, img_label_dim=LABEL_DIM).to(dtype)
        arena = flatten_model(model)
        word = model.uniter.embeddings.word_embeddings.weight
        V, H = word.shape
        g = torch.Generator().manual_seed(60 + rank)
        ids = torch.randint(0, V, (3, 7 + 2 * rank), generator=g)     # (ranks pad their text to different lengths)
        ids[0, :3] = 5                                            # duplicates inside the rank, and id 5 on every rank
        ids[1, 0] = 7 + rank                                      # an id only this rank has

        def fill():
            arena.grad.copy_(torch.randn(arena.numel, generator=torch.Generator().manual_seed(70 + rank)).to(dtype))
            lo, hi = arena.span([word])
            wg = arena.grad[lo:hi].view(V, H)
            keep = torch.zeros(V, dtype=torch.bool)
            keep[ids.reshape(-1)] = True
            wg[~keep] = 0                                         # a lookup-only table: gradient rows of absent tokens are zero
        results = []
        for sparse in (False, True):
            reducer = D.GradientReducer(arena, model.uniter.encoder, layers_per_bucket=1, word_embeddings=word,
                                        word_ids_cap=None if dtype == torch.float32 else 40)
            assert reducer.word_span is not None
            if big_vocab:
                reducer.word_span = reducer.word_span[:2] + (V, H)
            fill()
            reducer.begin()
            for l in reversed(range(len(model.uniter.encoder.layer))):
                model.uniter.encoder.grad_ready_hook(l)
            reducer.word_compact = bool(sparse and big_vocab)
            os.environ["UNITER_AMD_DP_SPARSE_WORD"] = "1"         # the row exchange is opt-in at every world size
            scale = reducer.finish(word_ids=ids if sparse else None)
            os.environ.pop("UNITER_AMD_DP_SPARSE_WORD", None)
            assert scale == 1.0 / world
            results.append(arena.grad.clone())
        if world == 2:                                            # two addends: the order cannot matter
            assert torch.equal(results[0], results[1]), (dtype, big_vocab, float((results[0].float() - results[1].float()).abs().max()))
        else:
            # more ranks: the dense collective adds in its ring order (rounding to the arena's dtype at every hop), the row exchange
            # adds in rank order in fp32 and rounds once — equal up to that
            # (the ranges around the table are also cut differently, which moves the ring's chunk boundaries)
            a, b = results[0].float(), results[1].float()
            tol = 1e-5 if dtype == torch.float32 else 2.0 ** -6
            assert float((a - b).abs().max()) <= tol * float(a.abs().max()), (dtype, big_vocab, float((a - b).abs().max()), float(a.abs().max()))


def _case_task_mix_and_retrieval_gather(rank, world, D):
    """MetaLoader: every rank trains the task rank 0 drew.  itm_eval.evaluate: ranks own different numbers of texts, the
    score rows are gathered in rank order and only rank 0 reports (reference utils/itm_eval.py:68-90)."""
    import random
    from uniter_amd.data import MetaLoader
    from uniter_amd.utils import itm_eval as IE

    class Listy(list):
        pass
    loaders = {"mlm": (Listy([{"t": "mlm"}] * 2), 1), "itm": (Listy([{"t": "itm"}] * 2), 1), "mrfr": (Listy([{"t": "mrfr"}]), 1)}
    meta = MetaLoader(loaders, accum_steps=1, distributed=True, rng=random.Random(1234 + 17 * rank))   # different local draws
    it = iter(meta)
    mine = [next(it)[0] for _ in range(24)]
    both = D.all_gather_list(mine)
    assert both[0] == both[1] and len(set(both[0])) > 1

    n_img = 9
    img_ids = ["i%d" % j for j in range(n_img)]
    all_txt = ["t%d" % k for k in range(13)]
    txt2img = {t: img_ids[k % n_img] for k, t in enumerate(all_txt)}
    img2txts = {i: [t for t in all_txt if txt2img[t] == i] for i in img_ids}
    g = torch.Generator().manual_seed(3)
    full = torch.randn(len(all_txt), n_img, generator=g)
    own = all_txt[:8] if rank == 0 else all_txt[8:]                     # 8 + 5 rows
    rows = [all_txt.index(t) for t in own]

    class Dset:
        ids, all_img_ids = own, img_ids

        def __len__(self):
            return len(own)
    Dset.txt2img, Dset.img2txts = txt2img, img2txts

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

        def forward(self, batch, compute_loss=False):
            return full[batch['row'], batch['cols']].unsqueeze(1)
    loader = Listy([[{'row': r, 'cols': torch.arange(0, 4)}, {'row': r, 'cols': torch.arange(4, n_img)}] for r in rows])
    loader.dataset = Dset()
    log = IE.evaluate(Model(), loader)
    if rank == 0:
        ref = IE.itm_eval(full.to(torch.bfloat16).float(), all_txt, img_ids, txt2img, img2txts)
        assert all(abs(log[k] - ref[k]) < 1e-12 for k in ref)
    else:
        assert log == {}


# Bound for the bf16 SUM over 8 ranks the reducer issues (distributed.py: in-place allreduce of the bf16 gradient arena, the
# 1/world factor folded into clip / AdamW afterwards) against the mean of the same 8 gradients in fp32.  Whatever order the
# collective adds in, every partial sum is rounded to bf16 once per hop (relative 2^-9 each, at most 7 hops), so the error of
# an element is bounded by 7 * 2^-9 * sum_r |g_r| and behaves like a random walk of ~sqrt(7) such roundings in practice.
BF16_SUM8_REL_L2 = 4e-3            # whole-vector relative L2 (measured over gloo: ~2.5e-3) — one bf16 epsilon (2^-8)
BF16_SUM8_ABS_FACTOR = 7 * 2.0 ** -9


def _case_bf16_sum_over_8_ranks(rank, world, D):
    assert world == 8
    n = 1 << 16
    gens = [torch.Generator().manual_seed(900 + r) for r in range(world)]
    # gradient-like values: heavy dynamic range per element across ranks (log-normal scale x normal)
    per_rank = [(torch.randn(n, generator=g) * torch.exp(2.0 * torch.randn(n, generator=g))).to(torch.bfloat16) for g in gens]
    mine = per_rank[rank].clone()
    dist.all_reduce(mine, op=dist.ReduceOp.SUM)                      # what GradientReducer._reduce_range issues
    got = mine.float() * (1.0 / world)                               # the averaging factor is applied in fp32 later
    ref = sum(t.double() for t in per_rank) / world
    err = (got.double() - ref)
    rel = float(err.norm() / ref.norm())
    assert rel <= BF16_SUM8_REL_L2, rel
    bound = BF16_SUM8_ABS_FACTOR * sum(t.double().abs() for t in per_rank) / world + 1e-30
    assert bool((err.abs() <= bound * 1.0001 + ref.abs() * 2.0 ** -9).all())   # + the final rounding of the sum itself
    # and through the reducer itself (CPU arena in bf16, no encoder: one range)
    class _Arena:
        pass
    a = _Arena()
    a.grad = per_rank[rank].clone()
    a.numel = n
    red = D.GradientReducer(a, None)
    red.begin()
    scale = red.finish()
    assert scale == 1.0 / world
    torch.testing.assert_close(a.grad.float() * scale, got, rtol=0, atol=0)


@pytest.mark.parametrize("case", ["_case_allreduce", "_case_broadcast_and_objects", "_case_gradient_reducer", "_case_sparse_word_rows",
                                  "_case_task_mix_and_retrieval_gather"])
def test_world_size_2(case, tmp_path):
    _spawn(case, tmp_path)


def test_sparse_word_rows_are_the_default_from_four_ranks(tmp_path):
    _spawn("_case_sparse_word_rows", tmp_path, world=4)


def test_bf16_sum_allreduce_over_8_ranks_stays_within_the_stated_bound(tmp_path):
    _spawn("_case_bf16_sum_over_8_ranks", tmp_path, world=8)


def test_single_process_is_a_noop():
    from uniter_amd.utils import distributed as D
    t = [torch.ones(3), torch.ones(4) * 2]
    D.all_reduce_and_rescale_tensors(t, 2.0)
    assert float(t[0][0]) == 0.5 and float(t[1][0]) == 1.0
    D.broadcast_tensors(t, 0)
    assert D.all_gather_list(5) == [5] and D.any_broadcast("x", 0) == "x"
    assert D.rank() == 0 and D.size() == 1
