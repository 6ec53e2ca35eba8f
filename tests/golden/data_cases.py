"""Deterministic inputs of the data-pipeline golden vectors: shared by the generator (make_golden_data.py, which feeds them to
the REFERENCE's functions) and by tests/test_data_records.py (which feeds them to uniter_amd.data)."""
import numpy as np
import torch

VOCAB_RANGE = (106, 28996)
MASK_ID, CLS_ID, SEP_ID = 103, 101, 102
IMG_DIM, LABEL_DIM, NUM_ANSWERS = 16, 11, 13


def token_lists(seed, n=6):
    r = np.random.RandomState(seed)
    return [[int(t) for t in r.randint(VOCAB_RANGE[0], 2000, size=int(r.randint(1, 14)))] for _ in range(n)]


def _boxes(r, nbb):
    xy = r.rand(nbb, 2) * 0.6
    wh = r.rand(nbb, 2) * 0.35 + 0.05
    return torch.from_numpy(np.concatenate([xy, xy + wh, wh], axis=1).astype(np.float32))


def _pos7(bb):
    return torch.cat([bb, bb[:, 4:5] * bb[:, 5:]], dim=-1)


def _common(r, tl, nbb):
    input_ids = torch.from_numpy(np.concatenate([[CLS_ID], r.randint(VOCAB_RANGE[0], 2000, size=tl - 2), [SEP_ID]]).astype(np.int64))
    img_feat = torch.from_numpy(r.randn(nbb, IMG_DIM).astype(np.float32))
    return input_ids, img_feat, _pos7(_boxes(r, nbb)), torch.ones(tl + nbb, dtype=torch.long)


def example_tuples(task, seed, n=5):
    """What the task's Dataset.__getitem__ returns, for n ragged examples."""
    r = np.random.RandomState(1000 + seed)
    out = []
    for _ in range(n):
        tl, nbb = int(r.randint(3, 12)), int(r.randint(2, 9))
        input_ids, img_feat, pos, attn = _common(r, tl, nbb)
        if task == 'mlm':
            labels = torch.from_numpy(np.where(r.rand(tl) < 0.3, r.randint(106, 2000, size=tl), -1).astype(np.int64))
            out.append((input_ids, img_feat, pos, attn, labels))
        elif task in ('mrfr', 'mrc'):
            mask = torch.from_numpy(r.rand(nbb) < 0.4)
            if not mask.any():
                mask[0] = True
            tgt = torch.cat([torch.zeros(tl, dtype=torch.uint8), mask], dim=0)
            if task == 'mrfr':
                out.append((input_ids, img_feat, pos, attn, mask, tgt))
            else:
                soft = torch.from_numpy(r.rand(nbb, LABEL_DIM).astype(np.float32))
                out.append((input_ids, img_feat, pos, soft, attn, mask, tgt))
        elif task in ('itm', 'itm_ot'):
            out.append((input_ids, img_feat, pos, attn, torch.full((1,), int(r.randint(0, 2)), dtype=torch.long)))
        elif task == 'vqa':
            target = torch.zeros(NUM_ANSWERS)
            target[int(r.randint(0, NUM_ANSWERS))] = float(np.float32(r.rand()))
            out.append((input_ids, img_feat, pos, attn, target))
        elif task == 'nlvr2_paired':
            rows = []
            for k in range(2):
                nb = int(r.randint(2, 9))
                feat = torch.from_numpy(r.randn(nb, IMG_DIM).astype(np.float32))
                rows.append((input_ids.clone(), feat, _pos7(_boxes(r, nb)), torch.ones(tl + nb, dtype=torch.long),
                             torch.full((nb,), k + 1, dtype=torch.long)))
            out.append((tuple(rows), int(r.randint(0, 2))))
        elif task == 'nlvr2_triplet':
            types = torch.from_numpy(np.sort(r.randint(1, 3, size=nbb)).astype(np.int64))
            out.append((input_ids, img_feat, pos, attn, types, int(r.randint(0, 2))))
        else:
            raise ValueError(task)
    return out


COLLATE_TASKS = ['mlm', 'mrfr', 'mrc', 'itm', 'itm_ot', 'vqa', 'nlvr2_paired', 'nlvr2_triplet']


def flatten(batch, prefix=''):
    """dict of tensors / nested dicts / scalars -> {flat key: ndarray}"""
    flat = {}
    for k, v in batch.items():
        if isinstance(v, dict):
            flat.update(flatten(v, prefix + k + '.'))
        elif isinstance(v, torch.Tensor):
            flat[prefix + k] = v.numpy().astype(np.uint8) if v.dtype == torch.bool else v.numpy()
            flat[prefix + k + '#dtype'] = np.array(str(v.dtype))
        elif v is None:
            flat[prefix + k + '#none'] = np.array(1)
        else:
            flat[prefix + k] = np.asarray(v)
    return flat


# ---- a small retrieval corpus behind duck-typed databases (both the reference's and this package's dataset methods only call
#      txt_db[id], txt_db.combine_inputs, img_db[name] and img_db.name2nbb) -------------------------------------------------------
class FakeTxtDb(object):
    cls_, sep = CLS_ID, SEP_ID

    def __init__(self, examples):
        self.examples = examples

    def __getitem__(self, id_):
        return {k: (list(v) if isinstance(v, list) else v) for k, v in self.examples[id_].items()}

    def combine_inputs(self, *inputs):
        ids = [self.cls_]
        for part in inputs:
            ids.extend(list(part) + [self.sep])
        return torch.tensor(ids)


class FakeImgDb(object):
    def __init__(self, names, seed):
        r = np.random.RandomState(seed)
        self.name2nbb = {n: int(r.randint(2, 9)) for n in names}
        self.feats = {n: (torch.from_numpy(r.randn(self.name2nbb[n], IMG_DIM).astype(np.float32)), _boxes(r, self.name2nbb[n])) for n in names}

    def __getitem__(self, name):
        return self.feats[name]


def retrieval_corpus(seed, n_img=7, n_txt=15):
    r = np.random.RandomState(500 + seed)
    names = ['im%02d' % k for k in range(n_img)]
    examples, txt2img, img2txts = {}, {}, {}
    for k in range(n_txt):
        img = names[k % n_img] if k < n_img else names[int(r.randint(0, n_img))]
        tid = 't%02d' % k
        examples[tid] = {'input_ids': [int(t) for t in r.randint(VOCAB_RANGE[0], 2000, size=int(r.randint(1, 9)))], 'img_fname': img}
        txt2img[tid] = img
        img2txts.setdefault(img, []).append(tid)
    return FakeTxtDb(examples), FakeImgDb(names, 900 + seed), list(examples.keys()), txt2img, img2txts


def bare(cls, **attrs):
    """An instance of a dataset class without its constructor (which opens databases), with the attributes its methods read."""
    obj = object.__new__(cls)
    for k, v in attrs.items():
        setattr(obj, k, v)
    return obj
