"""Synthetic batches with the reference's batch-dict contract.

Keys / dtypes / shapes follow the collate functions of the reference: data/mlm.py:96-136 (`mlm_collate`),
data/mrm.py:75-121,165-200 (`mrfr_collate`, `mrc_collate`), data/itm.py:98-124 (`itm_collate`),
data/vqa.py:44-71 (`vqa_collate`), data/nlvr2.py:61-95 (`nlvr2_paired_collate`) and
data/data.py:255-279 (`pad_tensors`, `get_gather_index`).  There is no dataset access in this environment,
so throughput and parity runs use these seeded batches; `ragged=True` gives per-example text / region counts
with real padding and a non-trivial `gather_index`.
"""
import torch


def get_gather_index(txt_lens, num_bbs, batch_size, max_len, out_size):
    """Positions of the compact [txt_i ; img_i ; pad] sequence inside cat([txt_padded, img_padded])."""
    gather_index = torch.arange(0, out_size, dtype=torch.long).unsqueeze(0).repeat(batch_size, 1)
    for i, (tl, nbb) in enumerate(zip(txt_lens, num_bbs)):
        gather_index[i, tl:tl + nbb] = torch.arange(max_len, max_len + nbb, dtype=torch.long)
    return gather_index


def _boxes(gen, n):
    """Normalised (x1, y1, x2, y2, w, h, w*h) rows (data/data.py:250)."""
    xy = torch.rand(n, 2, generator=gen) * 0.6
    wh = torch.rand(n, 2, generator=gen) * 0.35 + 0.05
    x2y2 = xy + wh
    return torch.cat([xy, x2y2, wh, wh[:, :1] * wh[:, 1:]], dim=-1)


def make_batch(task, batch_size, max_txt_len=60, num_bb=36, img_dim=2048, vocab_size=28996, seed=0, ragged=False,
               img_label_dim=1601, num_answer=3129, min_txt_len=10, min_bb=10, mask_prob=0.15, with_ot=False):
    """Returns a dict of CPU tensors.  task in {mlm, mrfr, mrc, mrckl, itm, vqa, nlvr2}.

    nlvr2: `batch_size` rows = batch_size/2 pairs; rows 2i / 2i+1 share the text and carry img_type_ids 1 / 2."""
    gen = torch.Generator().manual_seed(int(seed))
    B = int(batch_size)

    def rint(lo, hi, n):
        return torch.randint(lo, hi, (n,), generator=gen)

    if ragged:
        txt_lens = rint(min_txt_len, max_txt_len + 1, B).tolist()
        num_bbs = rint(min_bb, num_bb + 1, B).tolist()
        txt_lens[0] = max_txt_len         # keep the padded sizes deterministic
        num_bbs[-1] = num_bb
    else:
        txt_lens = [max_txt_len] * B
        num_bbs = [num_bb] * B
    if task == 'nlvr2':
        if B % 2:
            raise ValueError("nlvr2 batches hold pairs: batch_size must be even")
        for i in range(0, B, 2):
            txt_lens[i + 1] = txt_lens[i]
    Lt, Li = max(txt_lens), max(num_bbs)

    low = min(1000, max(3, vocab_size // 8))
    input_ids = torch.zeros(B, Lt, dtype=torch.long)
    for i, tl in enumerate(txt_lens):
        ids = rint(low, vocab_size, tl)
        ids[0] = min(101, vocab_size - 2)          # [CLS]
        ids[-1] = min(102, vocab_size - 1)         # [SEP]
        input_ids[i, :tl] = ids
    if task == 'nlvr2':
        for i in range(0, B, 2):
            input_ids[i + 1] = input_ids[i]
    position_ids = torch.arange(0, Lt, dtype=torch.long).unsqueeze(0)

    img_feat = torch.zeros(B, Li, img_dim)
    img_pos_feat = torch.zeros(B, Li, 7)
    for i, nbb in enumerate(num_bbs):
        img_feat[i, :nbb] = torch.randn(nbb, img_dim, generator=gen).abs()      # ReLU-like detector features
        img_pos_feat[i, :nbb] = _boxes(gen, nbb)

    out_size = max(tl + nbb for tl, nbb in zip(txt_lens, num_bbs))
    attn_masks = torch.zeros(B, out_size, dtype=torch.long)
    for i, (tl, nbb) in enumerate(zip(txt_lens, num_bbs)):
        attn_masks[i, :tl + nbb] = 1
    gather_index = get_gather_index(txt_lens, num_bbs, B, Lt, out_size)

    batch = {'input_ids': input_ids, 'position_ids': position_ids, 'img_feat': img_feat,
             'img_pos_feat': img_pos_feat, 'attn_masks': attn_masks, 'gather_index': gather_index,
             # host-side real lengths (a python list, so to_device leaves it on the host): optional hint for the
             # padding-free encoder path
             'seq_lens': [tl + nbb for tl, nbb in zip(txt_lens, num_bbs)]}

    if task == 'mlm':
        txt_labels = torch.full((B, Lt), -1, dtype=torch.long)
        for i, tl in enumerate(txt_lens):
            pick = torch.rand(tl, generator=gen) < mask_prob
            pick[0] = False
            if not pick.any():
                pick[1 + int(rint(0, max(tl - 1, 1), 1))] = True
            txt_labels[i, :tl][pick] = input_ids[i, :tl][pick]
            input_ids[i, :tl][pick] = min(103, vocab_size - 1)                  # [MASK]
        batch['txt_labels'] = txt_labels
    elif task in ('mrfr', 'mrc', 'mrckl'):
        img_masks = torch.zeros(B, Li, dtype=torch.bool)
        for i, nbb in enumerate(num_bbs):
            pick = torch.rand(nbb, generator=gen) < mask_prob
            if not pick.any():
                pick[int(rint(0, nbb, 1))] = True
            img_masks[i, :nbb] = pick
        # position of the masked regions inside the compact joint sequence (data/mrm.py:24-33)
        img_mask_tgt = torch.zeros(B, out_size, dtype=torch.bool)
        for i, (tl, nbb) in enumerate(zip(txt_lens, num_bbs)):
            img_mask_tgt[i, tl:tl + nbb] = img_masks[i, :nbb]
        if task == 'mrfr':
            batch['feat_targets'] = img_feat[img_masks].clone()                  # data/mrm.py:36-39
        else:
            n_mask = int(img_masks.sum())
            batch['label_targets'] = torch.softmax(torch.randn(n_mask, img_label_dim, generator=gen), dim=-1)
        img_feat[img_masks] = 0                                                  # masked regions are zeroed
        batch['img_masks'] = img_masks
        batch['img_mask_tgt'] = img_mask_tgt
    elif task == 'itm':
        batch['targets'] = rint(0, 2, B)
        batch['ot_inputs'] = None
        if with_ot:
            # word-region alignment inputs of itm_ot_collate (data/itm.py:128-184), bool pads
            joint_len = batch['attn_masks'].size(1)
            ot_scatter = torch.arange(0, joint_len, dtype=torch.long).unsqueeze(0).repeat(B, 1)
            for i, tl in enumerate(txt_lens):
                ot_scatter[i, tl:] = torch.arange(Lt, Lt + (joint_len - tl), dtype=torch.long)
            txt_pad = torch.zeros(B, Lt, dtype=torch.bool)
            img_pad = torch.zeros(B, Li, dtype=torch.bool)
            for i in range(B):
                txt_pad[i, txt_lens[i]:] = True
                img_pad[i, num_bbs[i]:] = True
            batch['ot_inputs'] = {'ot_scatter': ot_scatter, 'scatter_max': int(ot_scatter.max()), 'txt_pad': txt_pad,
                                  'img_pad': img_pad}
    elif task == 'vqa':
        targets = torch.zeros(B, num_answer)
        scores = torch.tensor([0.3, 0.6, 0.9, 1.0])
        for i in range(B):
            k = int(rint(1, 11, 1))
            idx = torch.randperm(num_answer, generator=gen)[:k]
            targets[i, idx] = scores[rint(0, 4, k)]
        batch['targets'] = targets
    elif task == 'nlvr2':
        batch['targets'] = rint(0, 2, B // 2)
        type_ids = torch.ones(B, Li, dtype=torch.long)
        type_ids[1::2] = 2
        batch['img_type_ids'] = type_ids
    else:
        raise ValueError("unknown task %r" % (task,))
    return batch


def to_device(batch, device, float_dtype=None):
    """Move a batch dict to `device`; floating tensors optionally cast (e.g. torch.bfloat16 features)."""
    out = {}
    for k, v in batch.items():
        if isinstance(v, torch.Tensor):
            if v.is_floating_point() and float_dtype is not None:
                v = v.to(float_dtype)
            out[k] = v.to(device)
        elif isinstance(v, dict):
            out[k] = to_device(v, device, float_dtype)
        else:
            out[k] = v
    return out
