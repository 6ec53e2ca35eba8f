"""Length-bucketed batching by token budget.  Reference: data/sampler.py:16-60 (same batching rule; no cytoolz /
horovod dependency, and the shuffles take an explicit `random.Random` so that every rank can be given its own stream)."""
import random

from torch.utils.data import Sampler


class TokenBucketSampler(Sampler):
    """Yields lists of example indices.  Examples are shuffled, cut into buckets of `bucket_size`, sorted by length inside
    a bucket (longest first) and packed `size_multiple` at a time while  longest length x examples  stays within
    `batch_size` tokens (padding included); the batches are shuffled again."""

    def __init__(self, lens, bucket_size, batch_size, droplast=False, size_multiple=8, rng=None):
        self._lens = lens
        self._max_tok = batch_size
        self._bucket_size = bucket_size
        self._droplast = droplast
        self._size_mul = size_multiple
        self._rng = rng if rng is not None else random

    def _create_ids(self):
        return list(range(len(self._lens)))

    def _sort_fn(self, i):
        return self._lens[i]

    def __iter__(self):
        ids = self._create_ids()
        self._rng.shuffle(ids)
        batches = []
        for start in range(0, len(ids), self._bucket_size):
            bucket = sorted(ids[start:start + self._bucket_size], key=self._sort_fn, reverse=True)
            current, longest = [], 0
            for g in range(0, len(bucket), self._size_mul):
                group = bucket[g:g + self._size_mul]
                longest = max(longest, max(self._lens[i] for i in group))
                if longest * (len(current) + self._size_mul) > self._max_tok:
                    if not current:
                        raise ValueError("max_tokens too small / max_seq_len too long")
                    batches.append(current)
                    current = list(group)          # (the running maximum is kept for the rest of the bucket, as upstream)
                else:
                    current.extend(group)
            if current and not self._droplast:
                batches.append(current)
        self._rng.shuffle(batches)
        return iter(batches)

    def __len__(self):
        raise ValueError("NOT supported. This has some randomness across epochs")
