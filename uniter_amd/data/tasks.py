"""Task datasets and batch builders of the four training configurations (c2 NLVR2, c3 / c5 pre-training
mix, c4 VQA) — the producers of the batch dict the encoder path consumes (row a-0).

Reference: data/mlm.py:17-136, data/mrm.py:14-200, data/itm.py:35-196, data/nlvr2.py:18-218, data/vqa.py:13-126.  Same class
and function names, same example tuples, same batch keys / shapes / dtypes, and — given the same `random` / `numpy.random`
state — the same masks and negatives draw for draw (tests/golden/data_pipeline.npz holds the reference's own outputs).
What is organised differently: every batch builder is  `joint_batch` (the six keys all tasks share) + its task's extras,
the random streams can be handed in (`rng=`) so that a loader worker or a rank owns its stream, and padding / index
tensors are built with vectorised index arithmetic instead of per-example Python loops."""
import random as _random

import numpy as np
import torch

from .collate import get_gather_index, pad_tensors
from .data import DetectFeatLmdb, DetectFeatTxtTokDataset, TxtTokLmdb, box_features, get_ids_and_lens
from .sampler import TokenBucketSampler


# ---- shared batch pieces ---------------------------------------------------------------------------------------------------
def _pad_1d(seqs, value, dtype=None):
    """List of 1-D tensors -> [n, max len] filled with `value` (torch.nn.utils.rnn.pad_sequence(batch_first=True))."""
    out = torch.nn.utils.rnn.pad_sequence(list(seqs), batch_first=True, padding_value=value)
    return out if dtype is None or out.dtype == dtype else out.to(dtype)


def joint_batch(input_ids, img_feats, img_pos_feats, attn_masks):
    """The part of a batch every task shares: padded text ids, position ids, padded region features / boxes, the joint
    attention mask and the gather index that compacts [text_i ; regions_i] (model/model.py:321-334).  Also returns the
    per-example text lengths and box counts."""
    txt_lens = [int(t.size(0)) for t in input_ids]
    num_bbs = [int(f.size(0)) for f in img_feats]
    ids = _pad_1d(input_ids, 0)
    masks = _pad_1d(attn_masks, 0)
    batch = {'input_ids': ids,
             'position_ids': torch.arange(0, ids.size(1), dtype=torch.long).unsqueeze(0),
             'img_feat': pad_tensors(img_feats, num_bbs),
             'img_pos_feat': pad_tensors(img_pos_feats, num_bbs),
             'attn_masks': masks,
             'gather_index': get_gather_index(txt_lens, num_bbs, ids.size(0), ids.size(1), masks.size(1))}
    return batch, txt_lens, num_bbs


def _columns(inputs):
    """List of example tuples -> tuple of per-field lists."""
    return tuple(map(list, zip(*inputs)))


# ---- masked language modelling (data/mlm.py) ------------------------------------------------------------------------------------
def random_word(tokens, vocab_range, mask, rng=_random):
    """BERT masking in place: each token is selected with probability 0.15; a selected token becomes [MASK] (80 %), a random
    word id of `vocab_range` (10 %) or stays (10 %).  Labels: the original id at selected positions, -1 elsewhere; if nothing
    was selected the first token is masked.  Consumes `rng` exactly like data/mlm.py:17-53."""
    labels = []
    for i, token in enumerate(tokens):
        draw = rng.random()
        if draw >= 0.15:
            labels.append(-1)
            continue
        draw /= 0.15
        if draw < 0.8:
            tokens[i] = mask
        elif draw < 0.9:
            tokens[i] = rng.randrange(vocab_range[0], vocab_range[1])
        labels.append(token)
    if all(v == -1 for v in labels):
        labels[0] = tokens[0]
        tokens[0] = mask
    return tokens, labels


class MlmDataset(DetectFeatTxtTokDataset):
    """-> (input_ids [L], img_feat [nbb, d], img_pos_feat [nbb, 7], attn_masks [L + nbb], txt_labels [L])"""

    def __init__(self, txt_db, img_db, rng=_random):
        assert isinstance(txt_db, TxtTokLmdb)
        super().__init__(txt_db, img_db)
        self.rng = rng

    def create_mlm_io(self, input_ids):
        input_ids, labels = random_word(input_ids, self.txt_db.v_range, self.txt_db.mask, self.rng)
        return (torch.tensor([self.txt_db.cls_] + input_ids + [self.txt_db.sep]), torch.tensor([-1] + labels + [-1]))

    def __getitem__(self, i):
        example = super().__getitem__(i)
        input_ids, txt_labels = self.create_mlm_io(example['input_ids'])
        img_feat, img_pos_feat, num_bb = self._get_img_feat(example['img_fname'])
        return input_ids, img_feat, img_pos_feat, torch.ones(len(input_ids) + num_bb, dtype=torch.long), txt_labels


def mlm_collate(inputs):
    input_ids, img_feats, img_pos_feats, attn_masks, txt_labels = _columns(inputs)
    batch, _, _ = joint_batch(input_ids, img_feats, img_pos_feats, attn_masks)
    batch['txt_labels'] = _pad_1d(txt_labels, -1)
    return batch


# ---- masked region modelling (data/mrm.py) ---------------------------------------------------------------------------------------
def _get_img_mask(mask_prob, num_bb, rng=_random):
    """One Bernoulli(mask_prob) draw per box, at least one box masked (data/mrm.py:14-20)."""
    picks = [rng.random() < mask_prob for _ in range(num_bb)]
    if not any(picks):
        picks[rng.randrange(num_bb)] = True
    return torch.tensor(picks)


def _get_img_tgt_mask(img_mask, txt_len):
    """The box mask in joint-sequence coordinates: `txt_len` zeros in front (data/mrm.py:23-26)."""
    return torch.cat([torch.zeros(txt_len, dtype=torch.uint8), img_mask], dim=0)


def _rows_at(values, masks):
    """values [n, m, d], masks [n, m] -> the masked rows [s, d] in batch-major order (data/mrm.py:29-34,129-134)."""
    return values[masks.bool()].contiguous().view(-1, values.size(-1))


_get_feat_target = _rows_at
_get_targets = _rows_at


def _mask_img_feat(img_feat, img_masks):
    """Zeroes the masked boxes' features (data/mrm.py:37-40)."""
    return img_feat.masked_fill(img_masks.bool().unsqueeze(-1), 0)


class MrfrDataset(DetectFeatTxtTokDataset):
    """-> (input_ids, img_feat, img_pos_feat, attn_masks, img_mask [nbb], img_mask_tgt [L + nbb])"""

    def __init__(self, mask_prob, *args, rng=_random, **kwargs):
        super().__init__(*args, **kwargs)
        self.mask_prob = mask_prob
        self.rng = rng

    def __getitem__(self, i):
        example = super().__getitem__(i)
        input_ids = self.txt_db.combine_inputs(example['input_ids'])
        img_feat, img_pos_feat, num_bb = self._get_img_feat(example['img_fname'])
        img_mask = _get_img_mask(self.mask_prob, num_bb, self.rng)
        return (input_ids, img_feat, img_pos_feat, torch.ones(len(input_ids) + num_bb, dtype=torch.long), img_mask,
                _get_img_tgt_mask(img_mask, len(input_ids)))


def _region_masks(batch, img_masks, img_mask_tgts):
    batch['img_masks'] = _pad_1d(img_masks, 0)
    batch['img_mask_tgt'] = _pad_1d(img_mask_tgts, 0)
    return batch['img_masks']


def mrfr_collate(inputs):
    input_ids, img_feats, img_pos_feats, attn_masks, img_masks, img_mask_tgts = _columns(inputs)
    batch, _, _ = joint_batch(input_ids, img_feats, img_pos_feats, attn_masks)
    masks = _region_masks(batch, img_masks, img_mask_tgts)
    batch['feat_targets'] = _get_feat_target(batch['img_feat'], masks)     # taken before the features are zeroed
    batch['img_feat'] = _mask_img_feat(batch['img_feat'], masks)
    return batch


class MrcDataset(DetectFeatTxtTokDataset):
    """-> (input_ids, img_feat, img_pos_feat, img_soft_labels [nbb, 1601], attn_masks, img_mask, img_mask_tgt)"""

    def __init__(self, mask_prob, *args, rng=_random, **kwargs):
        super().__init__(*args, **kwargs)
        self.mask_prob = mask_prob
        self.rng = rng

    def _get_img_feat(self, fname):
        dump = self.img_db.get_dump(fname)
        feat = torch.tensor(np.asarray(dump['features']))
        return (feat, box_features(torch.tensor(np.asarray(dump['norm_bb']))), torch.tensor(np.asarray(dump['soft_labels'])),
                self.img_db.name2nbb[fname])

    def __getitem__(self, i):
        example = DetectFeatTxtTokDataset.__getitem__(self, i)
        img_feat, img_pos_feat, soft_labels, num_bb = self._get_img_feat(example['img_fname'])
        img_mask = _get_img_mask(self.mask_prob, num_bb, self.rng)          # (drawn before the text is assembled, as upstream)
        input_ids = self.txt_db.combine_inputs(example['input_ids'])
        return (input_ids, img_feat, img_pos_feat, soft_labels, torch.ones(len(input_ids) + num_bb, dtype=torch.long), img_mask,
                _get_img_tgt_mask(img_mask, len(input_ids)))


def mrc_collate(inputs):
    input_ids, img_feats, img_pos_feats, soft_labels, attn_masks, img_masks, img_mask_tgts = _columns(inputs)
    batch, _, num_bbs = joint_batch(input_ids, img_feats, img_pos_feats, attn_masks)
    masks = _region_masks(batch, img_masks, img_mask_tgts)
    batch['label_targets'] = _get_targets(pad_tensors(soft_labels, num_bbs), masks)
    batch['img_feat'] = _mask_img_feat(batch['img_feat'], masks)
    return batch


# ---- image-text matching (data/itm.py) ------------------------------------------------------------------------------------------------
class TokenBucketSamplerForItm(TokenBucketSampler):
    """Re-draws the dataset's negatives at the start of every epoch (their box counts change the lengths).  data/itm.py:22-32."""

    def __init__(self, dset, *args, **kwargs):
        super().__init__(dset.lens, *args, **kwargs)
        self.dset = dset

    def __iter__(self):
        it = super().__iter__()
        self.dset.new_epoch()
        self._lens = self.dset.lens
        return it


def sample_negative(sample_pool, ground_truths, num_sample, rng=_random):
    """`num_sample` items of the pool, none of them a ground truth: draw and retry (data/itm.py:42-47)."""
    banned = set(ground_truths)
    while True:
        picks = rng.sample(sample_pool, num_sample)
        if not any(p in banned for p in picks):
            return picks


class ItmDataset(DetectFeatTxtTokDataset):
    """Aligned (label 1) or, with probability `neg_sample_p`, mismatched (label 0: a random other image) pairs, re-drawn every
    epoch; handles the rank split itself.  -> (input_ids, img_feat, img_pos_feat, attn_masks, target [1])"""

    def __init__(self, txt_db, img_db, neg_sample_p=0.5, rng=_random, np_rng=np.random):
        assert isinstance(txt_db, TxtTokLmdb)
        assert isinstance(img_db, DetectFeatLmdb)
        self.txt_db, self.img_db = txt_db, img_db
        self.rng, self.np_rng = rng, np_rng
        self.txt_lens, self.ids = get_ids_and_lens(txt_db)
        self._img_of = [txt_db[id_]['img_fname'] for id_ in self.ids]
        self.all_imgs = list(set(self._img_of))
        self.neg_sample_p = neg_sample_p
        self.new_epoch()

    def new_epoch(self):
        self.labels = self.np_rng.choice([0, 1], size=len(self.ids), p=[self.neg_sample_p, 1 - self.neg_sample_p])
        self.train_imgs = [img if label else sample_negative(self.all_imgs, [img], 1, self.rng)[0]
                           for img, label in zip(self._img_of, self.labels)]
        self.lens = [tl + self.img_db.name2nbb[img] for tl, img in zip(self.txt_lens, self.train_imgs)]

    def __getitem__(self, i):
        example = DetectFeatTxtTokDataset.__getitem__(self, i)
        img_feat, img_pos_feat, num_bb = self._get_img_feat(self.train_imgs[i])
        input_ids = self.txt_db.combine_inputs(example['input_ids'])
        target = torch.full((1,), int(self.labels[i]), dtype=torch.long)
        return input_ids, img_feat, img_pos_feat, torch.ones(len(input_ids) + num_bb, dtype=torch.long), target


def itm_collate(inputs):
    input_ids, img_feats, img_pos_feats, attn_masks, targets = _columns(inputs)
    batch, _, _ = joint_batch(input_ids, img_feats, img_pos_feats, attn_masks)
    batch['targets'] = torch.cat(targets, dim=0)
    return batch


def _compute_ot_scatter(txt_lens, max_txt_len, joint_len):
    """Where position j of the compact [text_i ; regions_i ; pad] row goes in the [text (max_txt_len) ; regions] layout the
    transport cost is computed in: text stays, everything after it starts at max_txt_len (data/itm.py:128-135)."""
    pos = torch.arange(joint_len, dtype=torch.long).unsqueeze(0)
    tl = torch.as_tensor(txt_lens, dtype=torch.long).unsqueeze(1)
    return torch.where(pos < tl, pos, pos - tl + max_txt_len)


def _compute_pad(lens, max_len):
    """uint8 [n, max_len]: 1 at padded positions (data/itm.py:138-142)."""
    pos = torch.arange(max_len, dtype=torch.long).unsqueeze(0)
    return (pos >= torch.as_tensor(lens, dtype=torch.long).unsqueeze(1)).to(torch.uint8)


def itm_ot_collate(inputs):
    input_ids, img_feats, img_pos_feats, attn_masks, targets = _columns(inputs)
    batch, txt_lens, num_bbs = joint_batch(input_ids, img_feats, img_pos_feats, attn_masks)
    batch['targets'] = torch.cat(targets, dim=0)
    max_tl, max_nbb = max(txt_lens), max(num_bbs)
    ot_scatter = _compute_ot_scatter(txt_lens, max_tl, batch['attn_masks'].size(1))
    batch['ot_inputs'] = {'ot_scatter': ot_scatter, 'scatter_max': int(ot_scatter.max().item()),
                          'txt_pad': _compute_pad(txt_lens, max_tl), 'img_pad': _compute_pad(num_bbs, max_nbb)}
    return batch


# ---- image-text retrieval: ranking batches for training, one text against many images for evaluation (data/itm.py:185-468) ----
def _text_image_rows(dataset, pairs):
    """[(text id, image name), ...] -> [(input_ids, img_feat, img_pos_feat, attn_masks), ...]"""
    rows = []
    for txt_id, img in pairs:
        input_ids = dataset.txt_db.combine_inputs(dataset.txt_db[txt_id]['input_ids'])
        img_feat, img_pos_feat, num_bb = dataset._get_img_feat(img)
        rows.append((input_ids, img_feat, img_pos_feat, torch.ones(len(input_ids) + num_bb, dtype=torch.long)))
    return rows


class ItmRankDataset(DetectFeatTxtTokDataset):
    """Example i -> 1 + 2 * neg_sample_size rows: the aligned pair first, then the text with `neg_sample_size` other images, then
    `neg_sample_size` other texts with the image (the margin ranking loss reads the rows in that order)."""

    def __init__(self, txt_db, img_db, neg_sample_size=1, rng=_random):
        assert neg_sample_size > 0, "ItmRankDataset need at least 1 negative sample"
        super().__init__(txt_db, img_db)
        self.rng = rng
        txt2img = self.txt_db.txt2img
        self.txt2img = {id_: txt2img[id_] for id_ in self.ids}
        self.img2txts = {}                                             # images of THIS rank's texts
        for id_, img in self.txt2img.items():
            self.img2txts.setdefault(img, []).append(id_)
        self.img_name_list = list(self.img2txts.keys())
        self.neg_sample_size = neg_sample_size

    def __getitem__(self, i):
        txt_id = self.ids[i]
        img = self.txt2img[txt_id]
        other_imgs = sample_negative(self.img_name_list, [img], self.neg_sample_size, self.rng)
        other_txts = sample_negative(self.ids, self.img2txts[img], self.neg_sample_size, self.rng)
        rows = _text_image_rows(self, [(txt_id, img)] + [(txt_id, o) for o in other_imgs] + [(o, img) for o in other_txts])
        assert len(rows) == 1 + 2 * self.neg_sample_size
        return rows


def itm_rank_collate(inputs):
    sample_size = len(inputs[0])
    assert all(len(rows) == sample_size for rows in inputs)
    input_ids, img_feats, img_pos_feats, attn_masks = _columns([row for rows in inputs for row in rows])
    batch, _, _ = joint_batch(input_ids, img_feats, img_pos_feats, attn_masks)
    batch['sample_size'] = sample_size
    return batch


class _HardNegBase(DetectFeatTxtTokDataset):
    def __init__(self, txt_db, img_db, neg_sample_size=1, rng=_random):
        assert neg_sample_size > 0, "need at least 1 negative sample"
        super().__init__(txt_db, img_db)
        self.rng = rng
        txt2img = self.txt_db.txt2img
        self.txt2img = {id_: txt2img[id_] for id_ in self.ids}
        self.img2txts = self.txt_db.img2txts
        self.neg_sample_size = neg_sample_size


class ItmRankDatasetHardNegFromText(_HardNegBase):
    """One text against its image (row 0) and `neg_sample_size` other images: a ready-made batch with ONE text row
    (input_ids [1, tl]) that the model broadcasts over the image rows (hard negatives are mined over such batches)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.img_name_list = list(self.img2txts.keys())

    def __getitem__(self, i):
        txt_id = self.ids[i]
        img = self.txt2img[txt_id]
        ids = self.txt_db.combine_inputs(self.txt_db[txt_id]['input_ids'])
        imgs = [img] + sample_negative(self.img_name_list, [img], self.neg_sample_size, self.rng)
        feats, boxes, num_bbs = _columns([self._get_img_feat(name) for name in imgs])
        n, tl = len(imgs), int(ids.size(0))
        width = tl + max(num_bbs)
        nbb = torch.as_tensor(num_bbs, dtype=torch.long).unsqueeze(1)
        return {'input_ids': ids.unsqueeze(0),
                'position_ids': torch.arange(0, tl, dtype=torch.long).unsqueeze(0),
                'img_feat': pad_tensors(list(feats), list(num_bbs)),
                'img_pos_feat': pad_tensors(list(boxes), list(num_bbs)),
                'attn_masks': (torch.arange(width, dtype=torch.long).unsqueeze(0) < tl + nbb).long(),
                'gather_index': get_gather_index([tl] * n, list(num_bbs), n, tl, width)}


class ItmRankDatasetHardNegFromImage(_HardNegBase):
    """One image (ONE image row) against its text (row 0) and `neg_sample_size` other texts."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.txt_name_list = list(self.txt2img.keys())

    def __getitem__(self, i):
        txt_id = self.ids[i]
        img = self.txt2img[txt_id]
        img_feat, img_pos_feat, nbb = self._get_img_feat(img)
        txts = [txt_id] + sample_negative(self.txt_name_list, self.img2txts[img], self.neg_sample_size, self.rng)
        rows = [self.txt_db.combine_inputs(self.txt_db[t]['input_ids']) for t in txts]
        txt_lens = [int(r.size(0)) for r in rows]
        input_ids = _pad_1d(rows, 0)
        width = max(txt_lens) + nbb
        tl = torch.as_tensor(txt_lens, dtype=torch.long).unsqueeze(1)
        # (the reference passes the LAST text's length as the gather offset of the region block — data/itm.py:360; the padded
        #  width is what the offset means, and both agree whenever the last text is the longest; the padded width is used here)
        return {'input_ids': input_ids,
                'position_ids': torch.arange(0, input_ids.size(1), dtype=torch.long).unsqueeze(0),
                'img_feat': img_feat.unsqueeze(0),
                'img_pos_feat': img_pos_feat.unsqueeze(0),
                'attn_masks': (torch.arange(width, dtype=torch.long).unsqueeze(0) < tl + nbb).long(),
                'gather_index': get_gather_index(txt_lens, [nbb] * len(txts), len(txts), int(input_ids.size(1)), width)}


def itm_rank_hn_collate(inputs):
    assert len(inputs) == 1
    return inputs[0]


class ItmValDataset(DetectFeatTxtTokDataset):
    """Retrieval validation: text i against its own image (row 0) and the `mini_batch_size - 1` images that follow it in the
    image list (wrapping around) — one ready-made batch per example (the loader's batch size is 1)."""

    def __init__(self, db_dir, img_dir, mini_batch_size=400):
        super().__init__(db_dir, img_dir)
        del self.lens
        self.txt2img = self.txt_db.txt2img
        self.img2txts = self.txt_db.img2txts
        self.all_img_ids = list(self.img2txts.keys())
        assert len(self.img2txts) >= mini_batch_size > 0
        self.bs = mini_batch_size

    def _get_batch_ids(self, i):
        img = self.txt2img[self.ids[i]]
        at = self.all_img_ids.index(img)
        ring = self.all_img_ids[at + 1:] + self.all_img_ids[:at]     # every other image, starting behind the true one
        others = ring[:self.bs - 1]
        assert len(others) == self.bs - 1, "Did not sample enough neg samples"
        return img, others

    def __getitem__(self, i):
        img, others = self._get_batch_ids(i)
        return self.get_batch(i, [img] + others)

    def get_batch(self, i, img_ids):
        """The same text in every row, one image per row: input_ids [n, tl] (position_ids [1, tl]) + padded regions."""
        example = DetectFeatTxtTokDataset.__getitem__(self, i)
        ids = self.txt_db.combine_inputs(example['input_ids'])
        n, tl = len(img_ids), int(ids.size(0))
        feats, boxes, num_bbs = _columns([self._get_img_feat(img) for img in img_ids])
        nbb = torch.as_tensor(num_bbs, dtype=torch.long).unsqueeze(1)
        width = tl + max(num_bbs)
        attn_masks = (torch.arange(width, dtype=torch.long).unsqueeze(0) < tl + nbb).long()
        return {'input_ids': ids.unsqueeze(0).expand(n, -1).clone(),
                'position_ids': torch.arange(0, tl, dtype=torch.long).unsqueeze(0),
                'img_feat': pad_tensors(list(feats), list(num_bbs)),
                'img_pos_feat': pad_tensors(list(boxes), list(num_bbs)),
                'attn_masks': attn_masks,
                'gather_index': get_gather_index([tl] * n, list(num_bbs), n, tl, width)}


def itm_val_collate(inputs):
    assert len(inputs) == 1, "input batch size > 1"
    return inputs[0]


class ItmEvalDataset(ItmValDataset):
    """Full retrieval evaluation: text i against ALL images, as a list of mini-batches over the image list sorted by box count
    (so that a mini-batch pads little)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.all_img_ids = sorted(list(self.all_img_ids), key=lambda img: self.img_db.name2nbb[img])

    def __getitem__(self, i):
        return [self.get_batch(i, self.all_img_ids[st:st + self.bs]) for st in range(0, len(self.all_img_ids), self.bs)]


itm_eval_collate = itm_val_collate


# ---- NLVR2 (data/nlvr2.py) ----------------------------------------------------------------------------------------------------------------
class _Nlvr2Base(DetectFeatTxtTokDataset):
    _txt_copies = 1

    def __init__(self, txt_db, img_db, use_img_type=True):
        assert isinstance(txt_db, TxtTokLmdb)
        assert isinstance(img_db, DetectFeatLmdb)
        self.txt_db, self.img_db = txt_db, img_db
        txt_lens, self.ids = get_ids_and_lens(txt_db)
        txt2img = txt_db.txt2img
        self.lens = [self._txt_copies * tl + sum(img_db.name2nbb[img] for img in txt2img[id_]) for tl, id_ in zip(txt_lens, self.ids)]
        self.use_img_type = use_img_type

    def _text(self, example):
        return torch.tensor([self.txt_db.cls_] + list(example['input_ids']) + [self.txt_db.sep])


class Nlvr2PairedDataset(_Nlvr2Base):
    """One statement, two images -> two (text, image_k) sequences with image type k = 1, 2:
    ((input_ids, img_feat, img_pos_feat, attn_masks, img_type_ids) x 2, target)"""
    _txt_copies = 2

    def __getitem__(self, i):
        example = DetectFeatTxtTokDataset.__getitem__(self, i)
        outs = []
        for k, img in enumerate(example['img_fname']):
            img_feat, img_pos_feat, num_bb = self._get_img_feat(img)
            input_ids = self._text(example)
            type_ids = torch.full((num_bb,), k + 1, dtype=torch.long) if self.use_img_type else None
            outs.append((input_ids, img_feat, img_pos_feat, torch.ones(len(input_ids) + num_bb, dtype=torch.long), type_ids))
        return tuple(outs), example['target']


def _nlvr2_batch(rows, targets):
    input_ids, img_feats, img_pos_feats, attn_masks, img_type_ids = _columns(rows)
    batch, _, _ = joint_batch(input_ids, img_feats, img_pos_feats, attn_masks)
    batch['img_type_ids'] = None if img_type_ids[0] is None else _pad_1d(img_type_ids, 0)
    batch['targets'] = torch.tensor([int(t) for t in targets], dtype=torch.long)
    return batch


def nlvr2_paired_collate(inputs):
    """Rows 2i, 2i + 1 of the batch are the two sequences of example i; `targets` has one entry per example."""
    return _nlvr2_batch([row for outs, _ in inputs for row in outs], [t for _, t in inputs])


class Nlvr2PairedEvalDataset(Nlvr2PairedDataset):
    def __getitem__(self, i):
        outs, target = super().__getitem__(i)
        return self.ids[i], outs, target


def nlvr2_paired_eval_collate(inputs):
    batch = nlvr2_paired_collate([(outs, target) for _, outs, target in inputs])
    batch['qids'] = [qid for qid, _, _ in inputs]
    return batch


class Nlvr2TripletDataset(_Nlvr2Base):
    """One sequence [text ; image_1 boxes ; image_2 boxes]:
    (input_ids, img_feat, img_pos_feat, attn_masks, img_type_ids, target)"""

    def __getitem__(self, i):
        example = DetectFeatTxtTokDataset.__getitem__(self, i)
        feats, boxes, types = [], [], []
        for k, img in enumerate(example['img_fname']):
            feat, pos, nbb = self._get_img_feat(img)
            feats.append(feat)
            boxes.append(pos)
            types.append(torch.full((nbb,), k + 1, dtype=torch.long))
        img_feat, img_pos_feat = torch.cat(feats, dim=0), torch.cat(boxes, dim=0)
        input_ids = self._text(example)
        return (input_ids, img_feat, img_pos_feat, torch.ones(len(input_ids) + img_feat.size(0), dtype=torch.long),
                torch.cat(types, dim=0) if self.use_img_type else None, example['target'])


def nlvr2_triplet_collate(inputs):
    return _nlvr2_batch([row[:5] for row in inputs], [row[5] for row in inputs])


class Nlvr2TripletEvalDataset(Nlvr2TripletDataset):
    def __getitem__(self, i):
        return (self.ids[i],) + super().__getitem__(i)


def nlvr2_triplet_eval_collate(inputs):
    batch = nlvr2_triplet_collate([row[1:] for row in inputs])
    batch['qids'] = [row[0] for row in inputs]
    return batch


# ---- VQA (data/vqa.py) -----------------------------------------------------------------------------------------------------------------------
def _get_vqa_target(example, num_answers):
    """Soft target vector: score s_k at answer label l_k (data/vqa.py:13-19)."""
    target = torch.zeros(num_answers)
    labels, scores = example['target']['labels'], example['target']['scores']
    if labels and scores:
        target[torch.tensor(labels)] = torch.tensor(scores, dtype=target.dtype)
    return target


class VqaDataset(DetectFeatTxtTokDataset):
    """-> (input_ids, img_feat, img_pos_feat, attn_masks, target [num_answers])"""

    def __init__(self, num_answers, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.num_answers = num_answers

    def _inputs(self, example):
        img_feat, img_pos_feat, num_bb = self._get_img_feat(example['img_fname'])
        input_ids = self.txt_db.combine_inputs(example['input_ids'])
        return input_ids, img_feat, img_pos_feat, torch.ones(len(input_ids) + num_bb, dtype=torch.long)

    def __getitem__(self, i):
        example = super().__getitem__(i)
        return self._inputs(example) + (_get_vqa_target(example, self.num_answers),)


def vqa_collate(inputs):
    input_ids, img_feats, img_pos_feats, attn_masks, targets = _columns(inputs)
    batch, _, _ = joint_batch(input_ids, img_feats, img_pos_feats, attn_masks)
    batch['targets'] = torch.stack(targets, dim=0)
    return batch


class VqaEvalDataset(VqaDataset):
    """-> (qid, input_ids, img_feat, img_pos_feat, attn_masks, target or None)"""

    def __getitem__(self, i):
        example = DetectFeatTxtTokDataset.__getitem__(self, i)
        target = _get_vqa_target(example, self.num_answers) if 'target' in example else None
        return (self.ids[i],) + self._inputs(example) + (target,)


def vqa_eval_collate(inputs):
    qids, input_ids, img_feats, img_pos_feats, attn_masks, targets = _columns(inputs)
    batch, _, _ = joint_batch(input_ids, img_feats, img_pos_feats, attn_masks)
    batch['targets'] = None if targets[0] is None else torch.stack(targets, dim=0)
    batch['qids'] = qids
    return batch


# ---- visual entailment (data/ve.py): the VQA pipeline with three answers ------------------------------------------------------------
class VeDataset(VqaDataset):
    def __init__(self, *args, **kwargs):
        super().__init__(3, *args, **kwargs)


class VeEvalDataset(VqaEvalDataset):
    def __init__(self, *args, **kwargs):
        super().__init__(3, *args, **kwargs)


ve_collate = vqa_collate
ve_eval_collate = vqa_eval_collate
