"""Collate helpers shared by every task's batch builder.  Reference: data/data.py:250-279."""
import torch

from ..utils.synthetic import get_gather_index  # noqa: F401  (data/data.py:271-279; one implementation in the package)


def pad_tensors(tensors, lens=None, pad=0):
    """List of B tensors [T_i, D] -> one [B, max T, D] tensor, rows beyond T_i filled with `pad` (data/data.py:250-263)."""
    sizes = [int(t.size(0)) for t in tensors]
    if lens is None or list(lens) == sizes:                      # the usual case: one C++ loop instead of a Python loop of row copies
        return torch.nn.utils.rnn.pad_sequence(list(tensors), batch_first=True, padding_value=pad)
    out = tensors[0].new_full((len(tensors), max(lens), tensors[0].size(-1)), pad)
    for row, (t, n) in enumerate(zip(tensors, lens)):
        out[row, :n] = t[:n]
    return out


def sequence_lengths(attn_masks):
    """Real tokens per example as a Python list, read from the (still host-resident) attention mask.  The packed encoder
    path (UniterModel.pack_padding) takes this as `batch['seq_lens']` and then never synchronises with the GPU."""
    if attn_masks.is_cuda:
        raise ValueError("sequence_lengths is a collate-time helper: call it before the batch leaves the host")
    return [int(v) for v in attn_masks.ne(0).sum(dim=1).tolist()]
