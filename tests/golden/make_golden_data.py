"""Golden vectors of the data pipeline, produced by the REFERENCE's own functions.

Imports /root/reference/data/{data,mlm,mrm,itm,nlvr2,vqa}.py — under stand-ins for the packages this image lacks (horovod,
lmdb, lz4, msgpack_numpy, toolz, cytoolz: none of them takes part in the functions exercised here except toolz.sandbox.unzip /
cytoolz.concat, whose two-line definitions are given below) — and records, for the seeded inputs of data_cases.py, what the
reference returns:  random_word, _get_img_mask, sample_negative, compute_num_bb, _get_vqa_target, get_gather_index, pad_tensors,
and the eight batch builders.  Output: tests/golden/data_pipeline.npz.   python tests/golden/make_golden_data.py [out.npz]"""
import importlib.util
import itertools
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import data_cases as dc  # noqa: E402

REF = os.environ.get('UNITER_REFERENCE', '/root/reference')


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def load_reference_data_modules():
    hvd = _stub('horovod.torch', rank=lambda: 0, size=lambda: 1, local_size=lambda: 1)
    _stub('horovod', torch=hvd)
    _stub('lmdb')
    frame = _stub('lz4.frame', compress=lambda b: b, decompress=lambda b: b)
    _stub('lz4', frame=frame)
    _stub('msgpack_numpy', patch=lambda: None)
    sandbox = _stub('toolz.sandbox', unzip=lambda seq: zip(*seq))
    _stub('toolz', sandbox=sandbox)
    _stub('cytoolz', concat=itertools.chain.from_iterable, partition_all=None)
    pkg = types.ModuleType('refdata')
    pkg.__path__ = [os.path.join(REF, 'data')]
    sys.modules['refdata'] = pkg
    mods = {}
    for name in ('data', 'sampler', 'mlm', 'mrm', 'itm', 'nlvr2', 'vqa'):
        spec = importlib.util.spec_from_file_location('refdata.' + name, os.path.join(REF, 'data', name + '.py'))
        mod = importlib.util.module_from_spec(spec)
        sys.modules['refdata.' + name] = mod
        spec.loader.exec_module(mod)
        mods[name] = mod
    return mods


def main(out_path):
    ref = load_reference_data_modules()
    g = {}
    # random_word: five seeded streams over six token lists each
    for seed in range(5):
        random.seed(seed)
        for k, toks in enumerate(dc.token_lists(seed)):
            out, labels = ref['mlm'].random_word(list(toks), dc.VOCAB_RANGE, dc.MASK_ID)
            g['random_word/%d/%d/tokens' % (seed, k)] = np.asarray(out, dtype=np.int64)
            g['random_word/%d/%d/labels' % (seed, k)] = np.asarray(labels, dtype=np.int64)
    # _get_img_mask and sample_negative
    for seed in range(6):
        random.seed(100 + seed)
        for k, nbb in enumerate((1, 3, 10, 36, 100)):
            g['img_mask/%d/%d' % (seed, k)] = ref['mrm']._get_img_mask(0.15, nbb).numpy().astype(np.uint8)
        pool = ['img%03d' % i for i in range(17)]
        g['negatives/%d' % seed] = np.asarray([ref['itm'].sample_negative(pool, [pool[(3 * j) % 17], pool[(5 * j + 1) % 17]], 3)
                                               for j in range(8)])
    # compute_num_bb
    r = np.random.RandomState(7)
    confs = [r.rand(int(n)).astype(np.float16) for n in (5, 20, 64, 100, 150)]
    for k, c in enumerate(confs):
        g['num_bb/%d/conf' % k] = c
        g['num_bb/%d/out' % k] = np.asarray([ref['data'].compute_num_bb(c, th, mn, mx) for th, mn, mx in ((0.2, 10, 100), (0.5, 1, 36), (0.9, 10, 12))])
    # VQA soft targets
    for k, (labels, scores) in enumerate((([2, 7], [0.3, 1.0]), ([], []), ([0], [0.6]))):
        g['vqa_target/%d' % k] = ref['vqa']._get_vqa_target({'target': {'labels': labels, 'scores': scores}}, dc.NUM_ANSWERS).numpy()
    # batch builders
    collates = {'mlm': ref['mlm'].mlm_collate, 'mrfr': ref['mrm'].mrfr_collate, 'mrc': ref['mrm'].mrc_collate,
                'itm': ref['itm'].itm_collate, 'itm_ot': ref['itm'].itm_ot_collate, 'vqa': ref['vqa'].vqa_collate,
                'nlvr2_paired': ref['nlvr2'].nlvr2_paired_collate, 'nlvr2_triplet': ref['nlvr2'].nlvr2_triplet_collate}
    for task in dc.COLLATE_TASKS:
        for seed in range(3):
            batch = collates[task](dc.example_tuples(task, seed))
            for key, arr in dc.flatten(batch).items():
                g['collate/%s/%d/%s' % (task, seed, key)] = arr
    # retrieval datasets over the duck-typed corpus of data_cases.retrieval_corpus
    from collections import defaultdict
    itm = ref['itm']
    for seed in range(2):
        txt_db, img_db, ids, txt2img, img2txts = dc.retrieval_corpus(seed)
        val = dc.bare(itm.ItmValDataset, txt_db=txt_db, img_db=img_db, ids=ids, txt2img=txt2img, img2txts=img2txts,
                      all_img_ids=list(img2txts.keys()), bs=4)
        for i in (0, 5, len(ids) - 1):
            for key, arr in dc.flatten(itm.itm_val_collate([val[i]])).items():
                g['itm_val/%d/%d/%s' % (seed, i, key)] = arr
        ev = dc.bare(itm.ItmEvalDataset, txt_db=txt_db, img_db=img_db, ids=ids, txt2img=txt2img, img2txts=img2txts, bs=3,
                     all_img_ids=sorted(list(img2txts.keys()), key=lambda n: img_db.name2nbb[n]))
        for m, mb in enumerate(itm.itm_eval_collate([ev[2]])):
            for key, arr in dc.flatten(mb).items():
                g['itm_eval/%d/%d/%s' % (seed, m, key)] = arr
        i2t = defaultdict(list)
        for t, im in txt2img.items():
            i2t[im].append(t)
        rk = dc.bare(itm.ItmRankDataset, txt_db=txt_db, img_db=img_db, ids=ids, txt2img=txt2img, img2txts=i2t,
                     img_name_list=list(i2t.keys()), neg_sample_size=2)
        random.seed(40 + seed)
        for key, arr in dc.flatten(itm.itm_rank_collate([rk[i] for i in (0, 4, 9)])).items():
            g['itm_rank/%d/%s' % (seed, key)] = arr
        # hard-negative ranking batches (the image-side class passes the LAST text's length as the region offset of the gather
        # index, data/itm.py:360 — recorded as the reference computes it; the test knows)
        hn_t = dc.bare(itm.ItmRankDatasetHardNegFromText, txt_db=txt_db, img_db=img_db, ids=ids, txt2img=txt2img, img2txts=img2txts,
                       img_name_list=list(img2txts.keys()), neg_sample_size=3)
        hn_i = dc.bare(itm.ItmRankDatasetHardNegFromImage, txt_db=txt_db, img_db=img_db, ids=ids, txt2img=txt2img, img2txts=img2txts,
                       txt_name_list=list(txt2img.keys()), neg_sample_size=3)
        random.seed(70 + seed)
        for i in (1, 6, 11):
            for key, arr in dc.flatten(itm.itm_rank_hn_collate([hn_t[i]])).items():
                g['itm_hn_text/%d/%d/%s' % (seed, i, key)] = arr
            for key, arr in dc.flatten(itm.itm_rank_hn_collate([hn_i[i]])).items():
                g['itm_hn_image/%d/%d/%s' % (seed, i, key)] = arr
    np.savez_compressed(out_path, **g)
    print("wrote %s: %d arrays" % (out_path, len(g)))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, 'data_pipeline.npz'))
