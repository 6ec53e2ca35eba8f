#!/usr/bin/env python
"""Host-side throughput of the image-feature read path: images/s of `DetectFeatLmdb.__getitem__`
from npz-compressed records (the reference's `compress=True` databases), msgpack records (`compress=False`) and the
memory-mapped `FeaturePack`, and NLVR2 batches/s through dataset + collate, on ONE host core.
usage: python scripts/bench_data_pipeline.py [n_images] [out.txt]"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uniter_amd.data import (DetectFeatLmdb, FeaturePack, Nlvr2PairedDataset, PackWriter, TxtTokLmdb, codec,  # noqa: E402
                             nlvr2_paired_collate)


def build(root, n_img, compress, with_pack):
    r = np.random.RandomState(0)
    img_dir, txt_dir = os.path.join(root, 'img'), os.path.join(root, 'txt')
    os.makedirs(img_dir), os.makedirs(txt_dir)
    db_name = 'feat_numbb36' + ('_compressed' if compress else '')
    names = ['img%05d.npz' % k for k in range(n_img)]

    def arrays():
        xy, wh = r.rand(36, 2) * 0.6, r.rand(36, 2) * 0.35 + 0.05
        return {'features': r.randn(36, 2048).astype(np.float16), 'norm_bb': np.concatenate([xy, xy + wh, wh], axis=1).astype(np.float16),
                'conf': r.rand(36).astype(np.float16), 'soft_labels': r.rand(36, 1601).astype(np.float16)}
    images = [(n, arrays()) for n in names]
    with PackWriter(os.path.join(img_dir, db_name)) as w:
        for n, a in images:
            w.put(n, codec.encode_img_record(a, compress))
    if with_pack:
        FeaturePack.build(os.path.join(img_dir, db_name + '.pack'), images)
    id2len, txt2img = {}, {}
    writer = PackWriter(txt_dir)
    for k in range(n_img):
        ids = [int(t) for t in r.randint(1000, 28000, size=int(r.randint(8, 40)))]
        pair = [names[k], names[(k + 7) % n_img]]
        writer.put('q%d' % k, codec.encode_txt_record({'input_ids': ids, 'img_fname': pair, 'target': k % 2}))
        id2len['q%d' % k], txt2img['q%d' % k] = len(ids), pair
    writer.close()
    json.dump(id2len, open(os.path.join(txt_dir, 'id2len.json'), 'w'))
    json.dump(txt2img, open(os.path.join(txt_dir, 'txt2img.json'), 'w'))
    json.dump({'CLS': 101, 'SEP': 102, 'MASK': 103, 'v_range': [106, 28996]}, open(os.path.join(txt_dir, 'meta.json'), 'w'))
    return txt_dir, img_dir, names


def rate(fn, n, min_s=2.0):
    fn(0)
    t0, done = time.perf_counter(), 0
    while time.perf_counter() - t0 < min_s:
        fn(done % n)
        done += 1
    return done / (time.perf_counter() - t0)


def main():
    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    out = sys.argv[2] if len(sys.argv) > 2 else None
    torch.set_num_threads(1)
    lines = ["# scripts/bench_data_pipeline.py %d : one host core, %d images of 36 boxes x 2048 fp16 features (+ 1601 soft labels), warm page cache" % (n_img, n_img)]
    for label, compress, pack, half in (("npz-compressed records (reference compress=True)", True, False, False),
                                        ("msgpack records (reference compress=False)", False, False, False),
                                        ("memory-mapped FeaturePack", True, True, False),
                                        ("memory-mapped FeaturePack, fp16 kept (keep_half)", True, True, True)):
        with tempfile.TemporaryDirectory() as root:
            txt_dir, img_dir, names = build(root, n_img, compress, pack)
            db = DetectFeatLmdb(img_dir, -1, 100, 10, 36, compress, keep_half=half)
            per_img = rate(lambda k: db[names[k]], n_img)
            data = Nlvr2PairedDataset(TxtTokLmdb(txt_dir, -1), db)
            per_batch = rate(lambda k: nlvr2_paired_collate([data[(16 * k + j) % n_img] for j in range(16)]), n_img // 16)
            lines.append("%-52s %8.0f images/s   %6.1f NLVR2 batches/s (16 pairs = 32 sequences) = %7.0f sequences/s" %
                         (label, per_img, per_batch, per_batch * 32))
            db.close()
    lines.append("# the GPU consumes ~6 900 sequences/s per MI355X at the headline configuration (bench.py): cores needed = that / sequences/s above")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, 'w').write(text + "\n")


if __name__ == '__main__':
    main()
