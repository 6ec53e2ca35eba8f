"""Gradient accumulation as ONE large batch: the micro-batches of an optimizer step concatenated along the batch axis.

The reference accumulates because a 16 / 32 GB GPU cannot hold the batch (`gradient_accumulation_steps` 2 / 4 in
config/pretrain-*.json, config/train-vqa-large-8gpu.json; loop: pretrain.py:264-312, train_vqa.py:183-206): the micro-steps of a
step draw the same task (data/loader.py:42-47), each does `loss.mean().backward()` and the gradients add up.  On a 288 GB
MI355X the micro-batches fit side by side, and one forward / backward over 4 x 32 sequences runs the same arithmetic in GEMMs
four times as tall — the shape this chip is efficient at (EXPERIMENTS.md section 11).  Two pieces:

  merge_batches(batches)         collated micro-batches of one task -> the batch the task's collate function would have built
                                 from all their examples at once (same keys, padding values, gather / scatter indices)
  accumulated_loss(loss, ...)    the un-reduced loss of the merged batch -> sum over micro-batches of the reference's per-micro-
                                 batch reduction, i.e. exactly the scalar whose gradient the accumulation loop produces

Token-bucket micro-batches (data/sampler.py:31-57) of one step can have very different padded widths — 64 sequences of 40
tokens and 16 of 160 merge into 80 rows padded to 160, 2.5x the tokens of the two batches run separately.  With real (ragged)
data the merged batch therefore belongs on the padding-free path (`UniterModel.pack_padding`), which
computes real tokens only; `padding_overhead(batch)` reports what dense execution of a merged batch would waste.

`tests/test_merge_accumulation.py` checks the first against the collate functions of `uniter_amd.data.tasks` (bit-exact) and
the second against the oracle's accumulated gradients.
"""
import torch

# how a key of the batch dict is extended when the merged batch is wider than a micro-batch: (axis-1 length it follows, fill value)
_TXT, _IMG, _JOINT = 'txt', 'img', 'joint'
_PADDED = {'input_ids': (_TXT, 0), 'txt_labels': (_TXT, -1), 'txt_type_ids': (_TXT, 0),
           'img_feat': (_IMG, 0), 'img_pos_feat': (_IMG, 0), 'img_masks': (_IMG, 0), 'img_type_ids': (_IMG, 0),
           'attn_masks': (_JOINT, 0), 'img_mask_tgt': (_JOINT, 0)}
# keys with one row per example / per masked position / per pair: plain concatenation
_ROWS = ('targets', 'feat_targets', 'label_targets', 'qids')
_REBUILT = ('position_ids', 'gather_index', 'ot_inputs', 'seq_lens', 'micro')


def _widen(t, width, fill):
    if t.size(1) == width:
        return t
    out = t.new_full((t.size(0), width) + tuple(t.shape[2:]), fill)
    out[:, :t.size(1)] = t
    return out


def text_lengths(input_ids):
    """Tokens per row of a padded id matrix: the collate functions pad with id 0 ([PAD]), which no tokenised text contains
    (data/mlm.py:104, data/itm.py:104 ...: pad_sequence(..., padding_value=0))."""
    return input_ids.ne(0).sum(dim=1)


def merge_batches(batches):
    """Concatenate collated batches of ONE task along the batch axis.  The result is what the task's collate function returns
    for the concatenated example list: every padded key is widened to the largest width with its own fill value, the
    compaction index (`gather_index`, data/data.py:271-279) and the OT scatter index / pads (data/itm.py:128-142) are rebuilt for
    the new padded text width, row-wise targets are concatenated.  Works on host or device tensors (no Python loop over
    examples); adds `batch['micro']` = {'rows': examples per micro-batch, 'loss_rows': loss rows per micro-batch} for
    `accumulated_loss`."""
    batches = list(batches)
    if not batches:
        raise ValueError("merge_batches: no batches")
    first = batches[0]
    keys = [k for k in first if k not in _REBUILT]
    for b in batches[1:]:
        if sorted(k for k in b if k not in _REBUILT) != sorted(keys):
            raise ValueError("merge_batches: the micro-batches of a step come from one task and carry the same keys "
                             "(%s vs %s)" % (sorted(keys), sorted(k for k in b if k not in _REBUILT)))
    if first.get('input_ids') is None or first.get('img_feat') is None:
        raise ValueError("merge_batches: text-only / image-only batches are not merged")
    width = {_TXT: max(int(b['input_ids'].size(1)) for b in batches),
             _IMG: max(int(b['img_feat'].size(1)) for b in batches),
             _JOINT: max(int(b['attn_masks'].size(1)) for b in batches)}
    out = {}
    for k in keys:
        vals = [b[k] for b in batches]
        if all(v is None for v in vals):
            out[k] = None
        elif any(v is None for v in vals):
            raise ValueError("merge_batches: %r is None in some micro-batches only" % k)
        elif k in _PADDED:
            kind, fill = _PADDED[k]
            out[k] = torch.cat([_widen(v, width[kind], fill) for v in vals], dim=0)
        elif k in _ROWS or (isinstance(vals[0], torch.Tensor) and vals[0].dim() >= 1
                            and all(v.shape[1:] == vals[0].shape[1:] for v in vals)):
            out[k] = torch.cat(vals, dim=0) if isinstance(vals[0], torch.Tensor) else [x for v in vals for x in v]
        else:
            raise ValueError("merge_batches: do not know how to merge key %r" % k)
    ids, masks = out['input_ids'], out['attn_masks']
    B, Lt, Lj = int(ids.size(0)), width[_TXT], width[_JOINT]
    dev = ids.device
    tl = text_lengths(ids).unsqueeze(1)                                           # [B, 1]
    nbb = masks.ne(0).sum(dim=1, keepdim=True) - tl                               # regions per row
    pos = torch.arange(Lj, dtype=torch.long, device=dev).unsqueeze(0)
    out['position_ids'] = torch.arange(0, Lt, dtype=torch.long, device=dev).unsqueeze(0)
    # data/data.py:271-279: identity, except that the region slots [tl, tl + nbb) point behind the padded text
    out['gather_index'] = torch.where((pos >= tl) & (pos < tl + nbb), pos - tl + Lt, pos)
    if 'ot_inputs' in first:
        if any(b['ot_inputs'] is None for b in batches):
            if not all(b['ot_inputs'] is None for b in batches):
                raise ValueError("merge_batches: ot_inputs is None in some micro-batches only")
            out['ot_inputs'] = None
        else:
            o0 = first['ot_inputs']
            scatter = torch.where(pos < tl, pos, pos - tl + Lt)                   # data/itm.py:128-135
            tpos = torch.arange(Lt, dtype=torch.long, device=dev).unsqueeze(0)
            ipos = torch.arange(width[_IMG], dtype=torch.long, device=dev).unsqueeze(0)
            out['ot_inputs'] = {'ot_scatter': scatter, 'scatter_max': int(scatter.max().item()),
                                'txt_pad': (tpos >= tl).to(o0['txt_pad'].dtype),  # data/itm.py:138-142 (uint8 there, bool in synthetic batches)
                                'img_pad': (ipos >= nbb).to(o0['img_pad'].dtype)}
    if 'seq_lens' in first:
        out['seq_lens'] = [n for b in batches for n in b['seq_lens']]
    rows = [int(b['input_ids'].size(0)) for b in batches]
    out['micro'] = {'rows': rows, 'loss_rows': [_loss_rows(b) for b in batches]}
    return out


def padding_overhead(batch):
    """Tokens a dense forward pass computes for `batch` divided by its real tokens (1.0 = no padding).  For a merged batch compare
    with the micro-batches' own figures: merging batches of different widths raises it, the packed path is unaffected."""
    masks = batch['attn_masks']
    real = int(masks.ne(0).sum().item())
    return float(masks.numel()) / float(max(real, 1))


def _loss_rows(b):
    """Rows of the un-reduced loss a model returns for batch `b` (its `.mean()` divides by this times the row width)."""
    if b.get('txt_labels') is not None:                       # MLM: one row per masked token (model/pretrain.py:115-118)
        return int(b['txt_labels'].ne(-1).sum().item())
    if b.get('img_mask_tgt') is not None:                     # MRFR / MRC: one row per masked region (model/pretrain.py:145)
        return int(b['img_mask_tgt'].ne(0).sum().item())
    t = b.get('targets')
    if t is not None:                                         # ITM / VQA: one row per example; NLVR2: one per pair
        return int(t.size(0))
    return int(b['input_ids'].size(0))


def _segment_weights(micro, name, device):
    """[sum(counts)] fp32 tensor holding 1 / counts[i] in segment i, counts = micro[name] (Python ints: built on the host, moved
    once per device and cached on `micro`, so a resident batch pays for it once)."""
    cache = micro.setdefault('_weights', {})
    key = (name, str(device))
    if key not in cache:
        counts = micro[name]
        w = torch.cat([torch.full((n,), 1.0 / n if n else 0.0, dtype=torch.float32) for n in counts]) if counts else torch.zeros(0)
        cache[key] = w.to(device)
    return cache[key]


def accumulated_loss(loss, micro, reduce='mean', scale=1.0):
    """Sum over micro-batches of the reference's per-micro-batch reduction of an un-reduced loss.

    loss   [n] or [n, C]: rows in batch order, micro-batch i owns `micro['loss_rows'][i]` consecutive rows.
    reduce 'mean': each micro-batch contributes mean(its elements) * scale — pretrain.py:295 (`loss.mean()`), train_nlvr2.py:165,
           train_vqa.py:188 (`loss.mean() * targets.size(1)`: pass scale = C).
    """
    if reduce != 'mean':
        raise ValueError("accumulated_loss: only the mean reduction of the reference's loops is implemented")
    counts = micro['loss_rows']
    if int(loss.size(0)) != sum(counts):
        raise ValueError("accumulated_loss: %d loss rows for micro-batches of %s rows" % (int(loss.size(0)), counts))
    w = _segment_weights(micro, 'loss_rows', loss.device)
    per_row = loss.float().reshape(loss.size(0), -1)
    return (per_row.sum(dim=1) * w).sum() * (float(scale) / per_row.size(1))


def accumulated_itm_ot_loss(itm_loss, ot_loss, targets, micro, ot_lambda):
    """pretrain.py:270-290 per micro-batch, summed: itm.mean() + lambda * (sum(pos) - sum(neg)) / (n_pos + n_neg).
    ot_loss = (distances of the examples with target 1, with target 0), each in batch order (model/pretrain.py:188-193)."""
    total = accumulated_loss(itm_loss, micro)
    if ot_loss is None:
        return total
    ot_pos, ot_neg = ot_loss
    w = _segment_weights(micro, 'rows', targets.device)       # 1 / (n_pos + n_neg) of the example's micro-batch
    pos_w = w[targets == 1]
    neg_w = w[targets == 0]
    ot = (ot_pos.float().reshape(-1) * pos_w).sum() - (ot_neg.float().reshape(-1) * neg_w).sum()
    return total + float(ot_lambda) * ot
