"""Host-side input pipeline pieces that do not depend on the LMDB / lz4 / msgpack stack:
the prefetching loader (H2D on a side HIP stream), the multi-task loader, token-bucket batching and the collate helpers."""
from .loader import MetaLoader, PrefetchLoader, move_to_cuda, record_cuda_stream  # noqa: F401
from .sampler import TokenBucketSampler  # noqa: F401
from .collate import get_gather_index, pad_tensors, sequence_lengths  # noqa: F401
