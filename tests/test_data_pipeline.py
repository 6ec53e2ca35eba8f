"""SURVEY.md section 8 row f-4, host side: loaders, token-bucket batching and collate helpers (reference data/loader.py,
data/sampler.py, data/data.py:250-279).  CPU tests cover the iteration contracts; the GPU test covers the side-stream copy."""
import random

import pytest
import torch

from uniter_amd.data import (MetaLoader, PrefetchLoader, TokenBucketSampler, get_gather_index, pad_tensors,
                             sequence_lengths)
from uniter_amd.utils.synthetic import make_batch


def test_token_bucket_sampler_covers_every_example_within_budget():
    rng = random.Random(3)
    lens = [rng.randint(12, 96) for _ in range(1003)]
    sampler = TokenBucketSampler(lens, bucket_size=256, batch_size=3072, droplast=False, size_multiple=8,
                                 rng=random.Random(5))
    batches = list(iter(sampler))
    seen = sorted(i for b in batches for i in b)
    assert seen == list(range(len(lens)))                                   # every example exactly once
    for b in batches:
        assert max(lens[i] for i in b) * len(b) <= 3072                     # padded tokens within the budget
    assert sum(1 for b in batches if len(b) % 8) <= (1003 + 255) // 256     # only bucket tails are not multiples of 8
    # a second epoch is shuffled differently
    assert [tuple(b) for b in iter(sampler)] != [tuple(b) for b in batches]
    with pytest.raises(ValueError):
        list(iter(TokenBucketSampler([500] * 16, 16, 3072, size_multiple=8)))
    with pytest.raises(ValueError):
        len(sampler)


def test_token_bucket_sampler_keeps_bucket_maximum_like_upstream():
    # one bucket, lengths sorted 100, 10, 10, ...: upstream sizes every batch of the bucket by its longest example
    lens = [100] + [10] * 31
    batches = list(iter(TokenBucketSampler(lens, bucket_size=32, batch_size=800, size_multiple=8, rng=random.Random(0))))
    assert sorted(len(b) for b in batches) == [8, 8, 8, 8]


def test_pad_tensors_and_gather_index():
    feats = [torch.arange(6.).view(3, 2), torch.ones(1, 2), torch.full((2, 2), 7.)]
    out = pad_tensors(feats)
    assert out.shape == (3, 3, 2) and out[1, 1:].abs().sum() == 0 and torch.equal(out[2, :2], feats[2])
    assert pad_tensors(feats, pad=-1)[1, 2, 0] == -1
    gi = get_gather_index([3, 2], [2, 1], 2, 4, 6)
    assert gi[0].tolist() == [0, 1, 2, 4, 5, 5] and gi[1].tolist() == [0, 1, 4, 3, 4, 5]


class _ListLoader:
    dataset = "marker"

    def __init__(self, batches):
        self.batches = batches

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


def test_meta_loader_mix_and_accumulation():
    a = _ListLoader([{"id": i} for i in range(3)])
    b = _ListLoader([{"id": 100 + i} for i in range(2)])
    meta = MetaLoader({"mlm": (a, 3), "itm": (b, 1)}, accum_steps=2, rng=random.Random(1))
    it = iter(meta)
    got = [next(it) for _ in range(400)]
    tasks = [t for t, _ in got]
    assert all(tasks[i] == tasks[i + 1] for i in range(0, 400, 2))          # both micro-batches of a step: one task
    frac = tasks.count("mlm") / 400.0
    assert 0.65 < frac < 0.85                                               # 3 : 1 mix
    ids = [bt["id"] for t, bt in got if t == "itm"]
    assert ids[:4] == [100, 101, 100, 101]                                  # exhausted loaders restart


def test_prefetch_loader_passthrough_on_cpu_adds_seq_lens():
    batches = [make_batch('mlm', 4, seed=s, ragged=True) for s in range(3)]
    for bt in batches:
        bt.pop('seq_lens', None)
    loader = PrefetchLoader(_ListLoader(batches))
    assert len(loader) == 3 and loader.dataset == "marker"
    out = list(loader)
    assert len(out) == 3
    for src, got in zip(batches, out):
        assert got['seq_lens'] == sequence_lengths(src['attn_masks'])
        assert torch.equal(got['input_ids'].cpu(), src['input_ids'])
    assert list(loader) and len(list(loader)) == 3                          # re-iterable


@pytest.mark.gpu
def test_prefetch_loader_side_stream_copy_matches_source():
    assert torch.cuda.is_available()
    batches = [make_batch('itm', 8, seed=10 + s, ragged=True) for s in range(4)]
    for bt in batches:
        bt.pop('seq_lens', None)
    loader = PrefetchLoader(_ListLoader([dict(b) for b in batches]), float_dtype=torch.bfloat16)
    n = 0
    for src, got in zip(batches, loader):
        n += 1
        assert got['img_feat'].is_cuda and got['img_feat'].dtype == torch.bfloat16
        assert got['input_ids'].is_cuda and got['input_ids'].dtype == torch.int64
        assert got['seq_lens'] == sequence_lengths(src['attn_masks'])
        # consume on the current stream right away: the loader's event ordering must make the data visible
        assert torch.equal(got['input_ids'].cpu(), src['input_ids'])
        torch.testing.assert_close(got['img_feat'].float().cpu(), src['img_feat'].to(torch.bfloat16).float())
        assert float((got['attn_masks'].sum(1).cpu() - torch.tensor(got['seq_lens'])).abs().max()) == 0.0
    assert n == 4
