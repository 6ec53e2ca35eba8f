"""Dataset interfaces over the reference's text / image databases.  Reference: data/data.py:30-312 (same class names,
constructor arguments, database naming rules and return values; Horovod's rank / size come from torch.distributed, the
record store is `store.open_store` — LMDB when the package is there, a memory-mapped pack directory otherwise — and the
record decoding lives in `codec`).

MI355X side of it: `DetectFeatLmdb` also accepts a `FeaturePack` next to (or instead of) the record database —
`<img_dir>/<db_name>.pack/feature_pack.json` — in which case an image is two zero-copy fp16 row ranges of memory-mapped
files and nothing is inflated or unpickled per example."""
import json
import os
from collections import defaultdict
from contextlib import contextmanager

import numpy as np
import torch
from torch.utils.data import ConcatDataset, Dataset

from . import codec
from .collate import get_gather_index, pad_tensors  # noqa: F401  (re-exported like the reference module does)
from .store import FeaturePack, open_store


def _rank_and_world():
    dist = torch.distributed
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _multi_node():
    """data/data.py:37-43: read-ahead only on single-node runs (one page cache shared by all readers)."""
    world = _rank_and_world()[1]
    local = int(os.environ.get('LOCAL_WORLD_SIZE', world))
    return world != local


def compute_num_bb(confs, conf_th, min_bb, max_bb):
    """Boxes kept for an image: those above the confidence threshold, clamped to [min_bb, max_bb] (data/data.py:30-33)."""
    return int(min(max_bb, max(min_bb, int((np.asarray(confs) > conf_th).sum()))))


class DetectFeatLmdb(object):
    """Region features of one image directory.  data/data.py:46-126.

    db name: feat_numbb{num_bb} (conf_th == -1) or feat_th{conf_th}_max{max_bb}_min{min_bb}, + '_compressed'; when the
    per-image box counts (nbb_th...json) are not precomputed: 'all' / 'all_compressed' and the counts are computed from the
    stored confidences on opening (the reference reads self.conf_th there without ever setting it; here it is set)."""

    def __init__(self, img_dir, conf_th=0.2, max_bb=100, min_bb=10, num_bb=36, compress=True, keep_half=False):
        self.img_dir = img_dir
        # keep_half: hand out the stored fp16 region features as fp16 tensors (the reference converts to fp32 per image).  The batch then
        # travels to the GPU at half the bytes and PrefetchLoader(float_dtype=bf16) casts it there — fp16 -> bf16 rounds exactly as
        # fp16 -> fp32 -> bf16 does, so the model sees the same inputs
        self.keep_half = keep_half
        self.conf_th, self.max_bb, self.min_bb, self.num_bb = conf_th, max_bb, min_bb, num_bb
        if conf_th == -1:
            db_name = 'feat_numbb%d' % num_bb
            self.name2nbb = defaultdict(lambda: num_bb)
        else:
            db_name = 'feat_th%s_max%d_min%d' % (conf_th, max_bb, min_bb)
            nbb = os.path.join(img_dir, 'nbb_th%s_max%d_min%d.json' % (conf_th, max_bb, min_bb))
            self.name2nbb = json.load(open(nbb)) if os.path.exists(nbb) else None
        self.compress = compress
        if compress:
            db_name += '_compressed'
        if self.name2nbb is None:
            db_name = 'all_compressed' if compress else 'all'
        self.db_name = db_name
        pack_dir = os.path.join(img_dir, db_name + '.pack')
        self.pack = FeaturePack(pack_dir) if os.path.isfile(os.path.join(pack_dir, 'feature_pack.json')) else None
        self.env = None
        if self.pack is None or os.path.isdir(os.path.join(img_dir, db_name)):
            self.env = open_store(os.path.join(img_dir, db_name), readonly=True, readahead=not _multi_node())
        if self.name2nbb is None:
            self.name2nbb = self._compute_nbb()

    def _record(self, file_name, fields=None):
        if self.env is None:
            raise KeyError("%s is not in the feature pack of %s and there is no record database beside it" % (file_name, self.img_dir))
        blob = self.env.get(file_name)
        if blob is None:
            raise KeyError(file_name)
        return codec.decode_img_record(blob, self.compress, fields)

    def _compute_nbb(self):
        if self.pack is not None and 'conf' in self.pack.fields:
            return {f: compute_num_bb(self.pack.get(f, 'conf'), self.conf_th, self.min_bb, self.max_bb) for f in self.pack.images}
        blob = self.env.get('__keys__')
        fnames = json.loads(bytes(blob).decode('utf-8')) if blob is not None else [k for k in self.env.keys() if k != '__keys__']
        return {f: compute_num_bb(self._record(f, ['conf'])['conf'], self.conf_th, self.min_bb, self.max_bb) for f in fnames}

    def get_dump(self, file_name):
        """Every stored array of the image as fp32, cut to its box count (the MRC task reads 'soft_labels' this way)."""
        nbb = self.name2nbb[file_name]
        if self.pack is not None and file_name in self.pack:
            rec = {k: self.pack.get(file_name, k, nbb) for k in self.pack.fields}
        else:
            rec = self._record(file_name)
        return {k: (np.asarray(a, dtype=np.float32) if np.asarray(a).dtype == np.float16 else np.asarray(a))[:nbb, ...]
                for k, a in rec.items()}

    def __getitem__(self, file_name):
        nbb = self.name2nbb[file_name]
        if self.pack is not None and file_name in self.pack:
            feat, bb = self.pack.get(file_name, 'features', nbb), self.pack.get(file_name, 'norm_bb', nbb)
        else:
            rec = self._record(file_name, ['features', 'norm_bb'])
            feat, bb = rec['features'][:nbb, :], rec['norm_bb'][:nbb, :]
        dtype = np.float16 if (self.keep_half and np.asarray(feat).dtype == np.float16) else np.float32
        # (the boxes stay fp32: the area w*h appended to them is then computed exactly as the reference computes it)
        return (torch.from_numpy(np.array(feat, dtype=dtype)), torch.from_numpy(np.array(bb, dtype=np.float32)))

    def close(self):
        if self.env is not None:
            self.env.close()
            self.env = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TxtLmdb(object):
    """Text examples, one msgpack + LZ4-frame record per id.  data/data.py:138-176."""

    def __init__(self, db_dir, readonly=True):
        self.readonly = readonly
        self.env = open_store(db_dir, readonly=readonly, readahead=not _multi_node())

    def __getitem__(self, key):
        blob = self.env.get(key)
        if blob is None:
            raise KeyError(key)
        return codec.decode_txt_record(blob)

    def __setitem__(self, key, value):
        if self.readonly:
            raise ValueError('readonly text DB')
        self.env.put(key, codec.encode_txt_record(value))

    def __del__(self):
        try:
            self.env.close()
        except Exception:
            pass


@contextmanager
def open_lmdb(db_dir, readonly=False):
    db = TxtLmdb(db_dir, readonly)
    try:
        yield db
    finally:
        del db


class TxtTokLmdb(object):
    """Tokenised text database + its side files (id2len.json, meta.json, txt2img.json, img2txts.json).  data/data.py:179-215."""

    def __init__(self, db_dir, max_txt_len=60):
        id2len = json.load(open(os.path.join(db_dir, 'id2len.json')))
        self.id2len = id2len if max_txt_len == -1 else {i: n for i, n in id2len.items() if n <= max_txt_len}
        self.db_dir = db_dir
        self.db = TxtLmdb(db_dir, readonly=True)
        meta = json.load(open(os.path.join(db_dir, 'meta.json')))
        self.cls_, self.sep, self.mask, self.v_range = meta['CLS'], meta['SEP'], meta['MASK'], meta['v_range']

    def __getitem__(self, id_):
        return self.db[id_]

    def combine_inputs(self, *inputs):
        ids = [self.cls_]
        for part in inputs:
            ids.extend(list(part) + [self.sep])
        return torch.tensor(ids)

    @property
    def txt2img(self):
        return json.load(open(os.path.join(self.db_dir, 'txt2img.json')))

    @property
    def img2txts(self):
        return json.load(open(os.path.join(self.db_dir, 'img2txts.json')))


def get_ids_and_lens(db):
    """This rank's share of the examples (every world-th id starting at the rank) and their text lengths.  data/data.py:218-225."""
    assert isinstance(db, TxtTokLmdb)
    rank, world = _rank_and_world()
    ids = list(db.id2len.keys())[rank::world]
    return [db.id2len[i] for i in ids], ids


def box_features(bb):
    """[x1, y1, x2, y2, w, h] -> the 7-d position feature with the area w*h appended (data/data.py:245)."""
    out = bb.new_empty(bb.size(0), 7)
    out[:, :6] = bb
    torch.mul(bb[:, 4], bb[:, 5], out=out[:, 6])
    return out


class DetectFeatTxtTokDataset(Dataset):
    """Text example i + the region features of its image; `lens` = text + box count per example (for the token-bucket
    sampler).  data/data.py:228-247."""

    def __init__(self, txt_db, img_db):
        assert isinstance(txt_db, TxtTokLmdb)
        assert isinstance(img_db, DetectFeatLmdb)
        self.txt_db = txt_db
        self.img_db = img_db
        txt_lens, self.ids = get_ids_and_lens(txt_db)
        txt2img = txt_db.txt2img
        self.lens = [tl + self.img_db.name2nbb[txt2img[id_]] for tl, id_ in zip(txt_lens, self.ids)]

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, i):
        return self.txt_db[self.ids[i]]

    def _get_img_feat(self, fname):
        img_feat, bb = self.img_db[fname]
        return img_feat, box_features(bb), img_feat.size(0)


class ConcatDatasetWithLens(ConcatDataset):
    """ConcatDataset that also concatenates `lens` and forwards unknown methods to every member.  data/data.py:282-296."""

    def __init__(self, datasets):
        super().__init__(datasets)
        self.lens = [n for d in datasets for n in d.lens]

    def __getattr__(self, name):
        def run_all(*args, **kwargs):
            return [getattr(d, name)(*args, **kwargs) for d in self.datasets]
        return run_all


class ImageLmdbGroup(object):
    """Opens image databases on demand with one set of box-count parameters.  data/data.py:299-312 (which re-opens a path on
    every lookup because it never stores the handle; here it is cached)."""

    def __init__(self, conf_th, max_bb, min_bb, num_bb, compress, keep_half=False):
        self.path2imgdb = {}
        self.conf_th, self.max_bb, self.min_bb, self.num_bb, self.compress = conf_th, max_bb, min_bb, num_bb, compress
        self.keep_half = keep_half

    def __getitem__(self, path):
        if path not in self.path2imgdb:
            self.path2imgdb[path] = DetectFeatLmdb(path, self.conf_th, self.max_bb, self.min_bb, self.num_bb, self.compress, self.keep_half)
        return self.path2imgdb[path]
