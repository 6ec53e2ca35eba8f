"""SURVEY.md section 8 row f-4, decode half: record codecs and stores, dataset classes and batch builders
(reference data/data.py, mlm.py, mrm.py, itm.py, nlvr2.py, vqa.py).

Parity bar for index / integer work: bit-exact.  `tests/golden/data_pipeline.npz` holds what the REFERENCE's functions
return for the seeded inputs of tests/golden/data_cases.py (generator: tests/golden/make_golden_data.py)."""
import json
import os
import random
import struct
import sys

import numpy as np
import pytest
import torch

from uniter_amd.data import codec, tasks
from uniter_amd.data import (ConcatDatasetWithLens, DetectFeatLmdb, FeaturePack, ImageLmdbGroup, ItmDataset, LmdbStore,
                             MlmDataset, MrcDataset, MrfrDataset, Nlvr2PairedDataset, Nlvr2TripletDataset, PackStore,
                             PackWriter, PrefetchLoader, TokenBucketSampler, TxtLmdb, TxtTokLmdb, VqaDataset, VqaEvalDataset,
                             compute_num_bb, convert_store, itm_ot_collate, mlm_collate, mrc_collate, mrfr_collate,
                             nlvr2_paired_collate, nlvr2_triplet_collate, vqa_collate, vqa_eval_collate)

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import data_cases as dc  # noqa: E402

GOLDEN = os.path.join(HERE, 'golden', 'data_pipeline.npz')


@pytest.fixture(scope='module')
def gold():
    return np.load(GOLDEN, allow_pickle=False)


# ---- the reference's own outputs ---------------------------------------------------------------------------------------------
def test_random_word_draw_for_draw_like_reference(gold):
    for seed in range(5):
        rng = random.Random(seed)                  # the reference seeded the module-level generator the same way
        for k, toks in enumerate(dc.token_lists(seed)):
            out, labels = tasks.random_word(list(toks), dc.VOCAB_RANGE, dc.MASK_ID, rng)
            assert out == gold['random_word/%d/%d/tokens' % (seed, k)].tolist()
            assert labels == gold['random_word/%d/%d/labels' % (seed, k)].tolist()


def test_region_masks_and_negatives_like_reference(gold):
    for seed in range(6):
        rng = random.Random(100 + seed)
        for k, nbb in enumerate((1, 3, 10, 36, 100)):
            mask = tasks._get_img_mask(0.15, nbb, rng)
            assert mask.dtype == torch.bool and mask.any()
            assert mask.numpy().astype(np.uint8).tolist() == gold['img_mask/%d/%d' % (seed, k)].tolist()
        pool = ['img%03d' % i for i in range(17)]
        got = [tasks.sample_negative(pool, [pool[(3 * j) % 17], pool[(5 * j + 1) % 17]], 3, rng) for j in range(8)]
        assert got == gold['negatives/%d' % seed].tolist()


def test_box_counts_and_vqa_targets_like_reference(gold):
    for k in range(5):
        conf = gold['num_bb/%d/conf' % k]
        got = [compute_num_bb(conf, th, mn, mx) for th, mn, mx in ((0.2, 10, 100), (0.5, 1, 36), (0.9, 10, 12))]
        assert got == gold['num_bb/%d/out' % k].tolist()
    for k, (labels, scores) in enumerate((([2, 7], [0.3, 1.0]), ([], []), ([0], [0.6]))):
        t = tasks._get_vqa_target({'target': {'labels': labels, 'scores': scores}}, dc.NUM_ANSWERS)
        assert np.array_equal(t.numpy(), gold['vqa_target/%d' % k])


_COLLATES = {'mlm': mlm_collate, 'mrfr': mrfr_collate, 'mrc': mrc_collate, 'itm': tasks.itm_collate, 'itm_ot': itm_ot_collate,
             'vqa': vqa_collate, 'nlvr2_paired': nlvr2_paired_collate, 'nlvr2_triplet': nlvr2_triplet_collate}


@pytest.mark.parametrize('task', dc.COLLATE_TASKS)
def test_batch_builders_bit_exact_vs_reference(gold, task):
    for seed in range(3):
        flat = dc.flatten(_COLLATES[task](dc.example_tuples(task, seed)))
        prefix = 'collate/%s/%d/' % (task, seed)
        want = {k[len(prefix):]: gold[k] for k in gold.files if k.startswith(prefix)}
        assert sorted(flat) == sorted(want), (task, sorted(set(flat) ^ set(want)))
        for key, ref in want.items():
            got = flat[key]
            assert got.dtype == ref.dtype and got.shape == ref.shape, (task, key, got.dtype, ref.dtype, got.shape, ref.shape)
            assert np.array_equal(got, ref), (task, seed, key)


def test_retrieval_datasets_bit_exact_vs_reference(gold):
    """ItmValDataset / ItmEvalDataset / ItmRankDataset (+ their collates) over the duck-typed corpus the golden generator gave the
    reference's classes: the same batches, the same negatives draw for draw."""
    from uniter_amd.data import ItmEvalDataset, ItmRankDataset, ItmValDataset, itm_eval_collate, itm_rank_collate, itm_val_collate

    def same(prefix, batch):
        flat = dc.flatten(batch)
        want = {k[len(prefix):]: gold[k] for k in gold.files if k.startswith(prefix)}
        assert sorted(flat) == sorted(want), (prefix, sorted(set(flat) ^ set(want)))
        for key, ref in want.items():
            assert flat[key].dtype == ref.dtype and np.array_equal(flat[key], ref), (prefix, key)
    for seed in range(2):
        txt_db, img_db, ids, txt2img, img2txts = dc.retrieval_corpus(seed)
        val = dc.bare(ItmValDataset, txt_db=txt_db, img_db=img_db, ids=ids, txt2img=txt2img, img2txts=img2txts,
                      all_img_ids=list(img2txts.keys()), bs=4)
        for i in (0, 5, len(ids) - 1):
            same('itm_val/%d/%d/' % (seed, i), itm_val_collate([val[i]]))
        ev = dc.bare(ItmEvalDataset, txt_db=txt_db, img_db=img_db, ids=ids, txt2img=txt2img, img2txts=img2txts, bs=3,
                     all_img_ids=sorted(list(img2txts.keys()), key=lambda n: img_db.name2nbb[n]))
        mbs = itm_eval_collate([ev[2]])
        assert len(mbs) == 3
        for m, mb in enumerate(mbs):
            same('itm_eval/%d/%d/' % (seed, m), mb)
        i2t = {}
        for t, im in txt2img.items():
            i2t.setdefault(im, []).append(t)
        rk = dc.bare(ItmRankDataset, txt_db=txt_db, img_db=img_db, ids=ids, txt2img=txt2img, img2txts=i2t, img_name_list=list(i2t.keys()),
                     neg_sample_size=2, rng=random.Random(40 + seed))
        same('itm_rank/%d/' % seed, itm_rank_collate([rk[i] for i in (0, 4, 9)]))
        from uniter_amd.data import ItmRankDatasetHardNegFromImage, ItmRankDatasetHardNegFromText, get_gather_index, itm_rank_hn_collate
        rng = random.Random(70 + seed)
        hn_t = dc.bare(ItmRankDatasetHardNegFromText, txt_db=txt_db, img_db=img_db, ids=ids, txt2img=txt2img, img2txts=img2txts,
                       img_name_list=list(img2txts.keys()), neg_sample_size=3, rng=rng)
        hn_i = dc.bare(ItmRankDatasetHardNegFromImage, txt_db=txt_db, img_db=img_db, ids=ids, txt2img=txt2img, img2txts=img2txts,
                       txt_name_list=list(txt2img.keys()), neg_sample_size=3, rng=rng)
        for i in (1, 6, 11):
            same('itm_hn_text/%d/%d/' % (seed, i), itm_rank_hn_collate([hn_t[i]]))
            got = itm_rank_hn_collate([hn_i[i]])
            # one deliberate difference: the reference offsets the region block of the gather index by the LAST text's length
            # (a loop variable left over, data/itm.py:360) instead of the padded text width the index is defined against
            ref_gi = gold['itm_hn_image/%d/%d/gather_index' % (seed, i)]
            lens = (got['input_ids'] != 0).sum(1).tolist()
            nbb = int(got['img_feat'].size(1))
            if lens[-1] == max(lens):
                assert np.array_equal(got['gather_index'].numpy(), ref_gi)
            else:
                assert np.array_equal(get_gather_index(lens, [nbb] * 4, 4, lens[-1], ref_gi.shape[1]).numpy(), ref_gi)      # what the reference did
                assert np.array_equal(got['gather_index'].numpy(), get_gather_index(lens, [nbb] * 4, 4, max(lens), ref_gi.shape[1]).numpy())
            got.pop('gather_index')
            prefix = 'itm_hn_image/%d/%d/' % (seed, i)
            flat = dc.flatten(got)
            want = {k[len(prefix):]: gold[k] for k in gold.files if k.startswith(prefix) and 'gather_index' not in k}
            assert sorted(flat) == sorted(want)
            for key, ref in want.items():
                assert flat[key].dtype == ref.dtype and np.array_equal(flat[key], ref), (prefix, key)
    with pytest.raises(AssertionError):
        itm_val_collate([1, 2])


def test_golden_recipe_regenerates_data_fixture(tmp_path):
    if not os.path.isdir('/root/reference/data'):
        pytest.skip("the reference checkout is not on this machine")
    import subprocess
    out = str(tmp_path / 'again.npz')
    subprocess.run([sys.executable, os.path.join(HERE, 'golden', 'make_golden_data.py'), out], check=True, capture_output=True)
    a, b = np.load(GOLDEN), np.load(out)
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), k


# ---- record codecs ----------------------------------------------------------------------------------------------------------
def test_xxh32_matches_the_xxhash_package():
    xxhash = pytest.importorskip('xxhash')
    r = random.Random(1)
    for n in (0, 1, 3, 4, 15, 16, 17, 31, 32, 100, 1000):
        data = bytes(r.randrange(256) for _ in range(n))
        for seed in (0, 1, 0x9E3779B1):
            assert codec.xxh32(data, seed) == xxhash.xxh32(data, seed=seed).intdigest()
    assert codec.xxh32(b'') == 0x02CC5D05                    # the specification's own vector for the empty input


def _greedy_lz4_block(data):
    """An independent block COMPRESSOR written from lz4_Block_format.md (hash of 4-byte windows, greedy matches; the last 5
    bytes are literals, the last match starts 12 bytes before the end) — test-side only, to give the decoder real matches."""
    out = bytearray()
    n, i, anchor, table = len(data), 0, 0, {}

    def emit(lit, mlen, offset):
        token_l = min(len(lit), 15)
        token_m = 0 if mlen is None else min(mlen - 4, 15)
        out.append((token_l << 4) | token_m)
        if len(lit) >= 15:
            rest = len(lit) - 15
            while rest >= 255:
                out.append(255)
                rest -= 255
            out.append(rest)
        out.extend(lit)
        if mlen is None:
            return
        out.extend(struct.pack('<H', offset))
        if mlen - 4 >= 15:
            rest = mlen - 4 - 15
            while rest >= 255:
                out.append(255)
                rest -= 255
            out.append(rest)

    while i + 12 < n:
        key = data[i:i + 4]
        cand = table.get(key)
        table[key] = i
        if cand is not None and i - cand <= 65535:
            m = 4
            while i + m < n - 5 and data[cand + m] == data[i + m]:
                m += 1
            emit(data[anchor:i], m, i - cand)
            i += m
            anchor = i
        else:
            i += 1
    emit(data[anchor:], None, 0)
    return bytes(out)


def test_lz4_block_decoder_against_an_independent_compressor():
    r = random.Random(2)
    cases = [b'', b'a', b'abcabcabcabcabcabcabcabcabcabcabcabc', bytes(300), b'0123456789' * 100,
             bytes(r.randrange(4) for _ in range(5000)), bytes(r.randrange(256) for _ in range(2000)),
             ('{"input_ids": [101, 2023, 2003, 102], "img_fname": "nlvr2_dev_0001.npz"}' * 7).encode()]
    for data in cases:
        block = _greedy_lz4_block(data)
        assert bytes(codec.lz4_block_decode(block, bytearray())) == data
        if len(data) > 64:
            assert len(block) < len(data) or data == cases[6]          # (really compressed, except the random bytes)
    # hand-assembled from the block document: 2 literals "ab", match offset 2 length 4+6 (overlapping), final literals "xyz12"
    block = bytes([0x26]) + b'ab' + struct.pack('<H', 2) + bytes([0x50]) + b'xyz12'
    assert bytes(codec.lz4_block_decode(block, bytearray())) == b'ab' + b'ab' * 5 + b'xyz12'
    for bad in (bytes([0x10]), bytes([0x04, 0x01, 0x00]), bytes([0x00, 0x05, 0x00])):
        with pytest.raises(ValueError):
            codec.lz4_block_decode(bad, bytearray())


def _frame(blocks, content=None, independent=True, block_sum=False, content_sum=False, with_size=False):
    flg = 0x40 | (0x20 if independent else 0) | (0x10 if block_sum else 0) | (0x08 if with_size else 0) | (0x04 if content_sum else 0)
    desc = bytes([flg, 0x40]) + (struct.pack('<Q', len(content)) if with_size else b'')
    out = struct.pack('<I', 0x184D2204) + desc + bytes([(codec.xxh32(desc) >> 8) & 0xFF])
    for payload, stored in blocks:
        out += struct.pack('<I', len(payload) | (0x80000000 if stored else 0)) + payload
        if block_sum:
            out += struct.pack('<I', codec.xxh32(payload))
    out += struct.pack('<I', 0)
    if content_sum:
        out += struct.pack('<I', codec.xxh32(content))
    return out


@pytest.mark.skipif(codec._lz4frame is not None, reason="the real lz4 package decodes here")
def test_lz4_frame_decoder_on_assembled_frames():
    part1, part2 = b'hello hello hello hello hello hello ', b'hello hello world world world world'
    content = part1 + part2
    # independent blocks, one compressed one stored, every optional field present
    f = _frame([(_greedy_lz4_block(part1), False), (part2, True)], content, True, True, True, True)
    assert codec.lz4_frame_decode(f) == content
    # linked blocks: the second block's matches reach into the first block's output
    linked2 = bytes([0x00]) + struct.pack('<H', 6) + bytes([0x50]) + b'world'       # match(4) from 6 back, then 5 literals
    f = _frame([(_greedy_lz4_block(part1), False), (linked2, False)], None, independent=False)
    assert codec.lz4_frame_decode(f) == part1 + part1[-6:-2] + b'world'
    # what the encoder writes is a frame the decoder reads, with the content size recorded
    for data in (b'', b'x', bytes(range(256)) * 50):
        enc = codec.lz4_frame_encode(data)
        assert enc[:4] == b'\x04\x22\x4d\x18' and codec.lz4_frame_decode(enc) == data
    broken = bytearray(_frame([(part2, True)], content, content_sum=True))
    broken[-1] ^= 1
    with pytest.raises(ValueError):
        codec.lz4_frame_decode(bytes(broken))
    with pytest.raises(ValueError):
        codec.lz4_frame_decode(b'\x00' * 16)


def test_msgpack_numpy_records_round_trip_and_foreign_keys():
    import msgpack
    feats = np.random.RandomState(0).randn(5, 8).astype(np.float16)
    rec = {'features': feats, 'norm_bb': np.arange(30, dtype=np.float16).reshape(5, 6), 'conf': np.float32(0.5), 'n': 3}
    back = codec.unpackb(codec.packb(rec))
    assert back['n'] == 3 and back['conf'] == np.float32(0.5) and back['conf'].dtype == np.float32
    assert back['features'].dtype == np.float16 and np.array_equal(back['features'], feats)
    # a writer that packed the map keys as str (use_bin_type=False era) is read as well
    foreign = msgpack.packb({'a': {'nd': True, 'type': '<f2', 'kind': '', 'shape': [5, 8], 'data': feats.tobytes()}}, use_bin_type=True)
    assert np.array_equal(codec.unpackb(foreign)['a'], feats)
    for compress in (True, False):
        blob = codec.encode_img_record({'features': feats, 'conf': np.ones(5, np.float16)}, compress)
        assert np.array_equal(codec.decode_img_record(blob, compress, ['features'])['features'], feats)
    example = {'input_ids': [5, 6, 7], 'img_fname': 'a.npz', 'target': {'labels': [1], 'scores': [0.9]}}
    assert codec.decode_txt_record(codec.encode_txt_record(example)) == example


# ---- stores and datasets on a small database ------------------------------------------------------------------------------------
def _build_db(root, compress, n_img=9, n_txt=23, with_pack=False, precomputed_nbb=True, two_images=False):
    """Text + image databases in the reference's layout (pack stores instead of LMDB environments)."""
    r = np.random.RandomState(5)
    img_dir, txt_dir = os.path.join(root, 'img'), os.path.join(root, 'txt')
    os.makedirs(img_dir), os.makedirs(txt_dir)
    images, name2nbb = {}, {}
    for k in range(n_img):
        boxes = int(r.randint(12, 19))
        xy, wh = r.rand(boxes, 2) * 0.6, r.rand(boxes, 2) * 0.35 + 0.05
        conf = r.rand(boxes).astype(np.float16)
        conf[:4] = 0.9                                                      # a few confident boxes per image
        images['img%02d.npz' % k] = {'features': r.randn(boxes, 64).astype(np.float16),
                                     'norm_bb': np.concatenate([xy, xy + wh, wh], axis=1).astype(np.float16), 'conf': conf,
                                     'soft_labels': r.rand(boxes, 11).astype(np.float16)}
        name2nbb['img%02d.npz' % k] = compute_num_bb(conf, 0.2, 4, 10)
    db_name = ('feat_th0.2_max10_min4' if precomputed_nbb else 'all') + ('_compressed' if compress else '')
    if precomputed_nbb:
        json.dump(name2nbb, open(os.path.join(img_dir, 'nbb_th0.2_max10_min4.json'), 'w'))
    with PackWriter(os.path.join(img_dir, db_name)) as w:
        for fname, arrays in images.items():
            w.put(fname, codec.encode_img_record(arrays, compress))
        w.put('__keys__', json.dumps(list(images)).encode())
    if with_pack:
        FeaturePack.build(os.path.join(img_dir, db_name + '.pack'), images.items())
    id2len, txt2img, examples = {}, {}, {}
    writer = PackWriter(txt_dir)
    for k in range(n_txt):
        ids = [int(t) for t in r.randint(10, 90, size=int(r.randint(2, 9)))]
        fname = 'img%02d.npz' % (k % n_img)
        ex = {'input_ids': ids, 'img_fname': [fname, 'img%02d.npz' % ((k + 3) % n_img)] if two_images else fname,
              'target': int(k % 2) if two_images else {'labels': [int(k % 13)], 'scores': [1.0]}}
        examples['q%d' % k] = ex
        writer.put('q%d' % k, codec.encode_txt_record(ex))
        id2len['q%d' % k] = len(ids)
        txt2img['q%d' % k] = ex['img_fname']
    writer.close()
    json.dump(id2len, open(os.path.join(txt_dir, 'id2len.json'), 'w'))
    json.dump(txt2img, open(os.path.join(txt_dir, 'txt2img.json'), 'w'))
    if not two_images:
        img2txts = {}
        for tid, fname in txt2img.items():
            img2txts.setdefault(fname, []).append(tid)
        json.dump(img2txts, open(os.path.join(txt_dir, 'img2txts.json'), 'w'))
    json.dump({'CLS': 1, 'SEP': 2, 'MASK': 3, 'v_range': [10, 96]}, open(os.path.join(txt_dir, 'meta.json'), 'w'))
    return txt_dir, img_dir, images, name2nbb, examples


def test_pack_store_and_feature_pack(tmp_path):
    with PackWriter(str(tmp_path / 'p')) as w:
        w.put('a', b'123')
        w.put('b', b'')
        w.put('c', bytes(range(200)))
    s = PackStore(str(tmp_path / 'p'))
    assert sorted(s.keys()) == ['a', 'b', 'c'] and bytes(s.get('a')) == b'123' and bytes(s.get('c')) == bytes(range(200))
    assert s.get('zz') is None and len(s) == 3
    assert convert_store(s, str(tmp_path / 'q'), keys=['c', 'a']) == 2
    assert bytes(PackStore(str(tmp_path / 'q')).get('c')) == bytes(range(200))
    s.close()
    try:
        import lmdb  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match='lmdb'):
            LmdbStore(str(tmp_path / 'nothing'))
    arrays = [('x', {'features': np.arange(12, dtype=np.float32).reshape(3, 4), 'conf': np.array([.1, .5, .9])}),
              ('y', {'features': np.ones((2, 4)), 'conf': np.array([.3, .3])})]
    fp = FeaturePack.build(str(tmp_path / 'f'), arrays)
    assert fp.rows('y') == 2 and fp.get('x', 'features').dtype == np.float16 and fp.get('x', 'features', 2).shape == (2, 4)
    assert np.array_equal(np.asarray(fp.get('y', 'conf')), np.array([.3, .3], np.float16)) and 'x' in fp and 'q' not in fp


@pytest.mark.parametrize('compress', [True, False])
def test_image_database_record_path_and_decode_free_path_agree(tmp_path, compress):
    txt_dir, img_dir, images, name2nbb, _ = _build_db(str(tmp_path / 'a'), compress, with_pack=True)
    packed = DetectFeatLmdb(img_dir, 0.2, 10, 4, 36, compress)
    assert packed.pack is not None and packed.name2nbb == name2nbb
    os.rename(os.path.join(img_dir, packed.db_name + '.pack'), os.path.join(img_dir, 'moved'))
    records = DetectFeatLmdb(img_dir, 0.2, 10, 4, 36, compress)
    assert records.pack is None
    for fname, arrays in images.items():
        nbb = name2nbb[fname]
        for db in (packed, records):
            feat, bb = db[fname]
            assert feat.dtype == torch.float32 and feat.shape == (nbb, 64) and bb.shape == (nbb, 6)
            assert np.array_equal(feat.numpy(), arrays['features'][:nbb].astype(np.float32))
            assert np.array_equal(bb.numpy(), arrays['norm_bb'][:nbb].astype(np.float32))
            dump = db.get_dump(fname)
            assert dump['soft_labels'].dtype == np.float32 and dump['soft_labels'].shape == (nbb, 11)
    # keep_half: the stored fp16 features as they are (the GPU casts to bf16 from either form with the same result), boxes in fp32
    os.rename(os.path.join(img_dir, 'moved'), os.path.join(img_dir, packed.db_name + '.pack'))
    half = DetectFeatLmdb(img_dir, 0.2, 10, 4, 36, compress, keep_half=True)
    for fname in list(images)[:3]:
        f16, b32 = half[fname]
        f32, _ = records[fname]
        assert f16.dtype == torch.float16 and b32.dtype == torch.float32 and torch.equal(f16.float(), f32)
        assert torch.equal(f16.to(torch.bfloat16), f32.to(torch.bfloat16))
    # box counts computed from the stored confidences when they are not precomputed ('all' database)
    _, img_dir2, _, name2nbb2, _ = _build_db(str(tmp_path / 'b'), compress, precomputed_nbb=False)
    assert DetectFeatLmdb(img_dir2, 0.2, 10, 4, 36, compress).name2nbb == name2nbb2


def _check_joint(batch, n_rows):
    ids, masks, gi = batch['input_ids'], batch['attn_masks'], batch['gather_index']
    assert ids.dtype == torch.int64 and ids.size(0) == n_rows and batch['position_ids'].shape == (1, ids.size(1))
    assert batch['img_feat'].dtype == torch.float32 and batch['img_pos_feat'].shape[-1] == 7
    assert masks.shape == gi.shape and masks.dtype == torch.int64
    lens = masks.sum(1)
    assert int(lens.max()) == masks.size(1)
    # the gather index maps the compact row onto [text ; regions]: position tl + j reads region j
    for b in range(n_rows):
        tl = int((ids[b] != 0).sum())
        nbb = int(lens[b]) - tl
        assert gi[b, :tl].tolist() == list(range(tl)) and gi[b, tl:tl + nbb].tolist() == list(range(ids.size(1), ids.size(1) + nbb))


def test_datasets_and_loaders_over_a_database(tmp_path):
    txt_dir, img_dir, images, name2nbb, examples = _build_db(str(tmp_path / 'single'), True, with_pack=True)
    txt_db = TxtTokLmdb(txt_dir, max_txt_len=7)
    assert all(n <= 7 for n in txt_db.id2len.values()) and len(txt_db.id2len) < 23
    assert txt_db['q0'] == examples['q0'] and txt_db.combine_inputs([5, 6], [7]).tolist() == [1, 5, 6, 2, 7, 2]
    txt_db = TxtTokLmdb(txt_dir, max_txt_len=-1)
    img_db = ImageLmdbGroup(0.2, 10, 4, 36, True)[img_dir]
    assert ImageLmdbGroup(0.2, 10, 4, 36, True).path2imgdb == {}
    rng = random.Random(0)
    mlm = MlmDataset(txt_db, img_db, rng=rng)
    assert len(mlm) == 23 and mlm.lens[0] == len(examples['q0']['input_ids']) + name2nbb['img00.npz']
    input_ids, img_feat, pos, attn, labels = mlm[3]
    assert input_ids[0] == 1 and input_ids[-1] == 2 and labels[0] == -1 and labels[-1] == -1 and (labels != -1).any()
    assert attn.numel() == input_ids.numel() + img_feat.size(0) and pos.shape == (img_feat.size(0), 7)
    assert torch.allclose(pos[:, 6], pos[:, 4] * pos[:, 5])
    orig = torch.tensor(examples['q3']['input_ids'])
    picked = labels[1:-1] != -1
    assert torch.equal(labels[1:-1][picked], orig[picked]) and torch.equal(input_ids[1:-1][~picked], orig[~picked])
    sampler = TokenBucketSampler(mlm.lens, bucket_size=16, batch_size=160, size_multiple=4, rng=random.Random(1))
    loader = torch.utils.data.DataLoader(mlm, batch_sampler=sampler, collate_fn=mlm_collate, num_workers=0)
    seen = 0
    for batch in PrefetchLoader(loader):
        _check_joint(batch, batch['input_ids'].size(0))
        assert batch['txt_labels'].shape == batch['input_ids'].shape and 'seq_lens' in batch
        assert batch['attn_masks'].size(1) * batch['input_ids'].size(0) <= 160
        seen += batch['input_ids'].size(0)
    assert seen == 23
    # region tasks
    mrfr, mrc = MrfrDataset(0.15, txt_db, img_db, rng=rng), MrcDataset(0.15, txt_db, img_db, rng=rng)
    b = mrfr_collate([mrfr[i] for i in range(6)])
    _check_joint(b, 6)
    n_masked = int(b['img_masks'].sum())
    assert b['feat_targets'].shape == (n_masked, 64) and float(b['img_feat'][b['img_masks']].abs().sum()) == 0.0
    assert b['img_mask_tgt'].shape == b['attn_masks'].shape and int(b['img_mask_tgt'].sum()) == n_masked
    b = mrc_collate([mrc[i] for i in range(6)])
    assert b['label_targets'].shape == (int(b['img_masks'].sum()), 11) and b['label_targets'].dtype == torch.float32
    # ITM: labels and negatives are re-drawn per epoch, a negative is never the true image
    itm = ItmDataset(txt_db, img_db, neg_sample_p=0.5, rng=rng, np_rng=np.random.RandomState(3))
    first = list(itm.train_imgs)
    for i, (img, label) in enumerate(zip(itm.train_imgs, itm.labels)):
        assert (img == examples[itm.ids[i]]['img_fname']) == bool(label)
    assert 0 < int(itm.labels.sum()) < 23
    itm.new_epoch()
    assert itm.train_imgs != first and itm.lens == [tl + name2nbb[img] for tl, img in zip(itm.txt_lens, itm.train_imgs)]
    b = itm_ot_collate([itm[i] for i in range(5)])
    _check_joint(b, 5)
    ot = b['ot_inputs']
    assert b['targets'].tolist() == [int(v) for v in itm.labels[:5]] and ot['scatter_max'] == int(ot['ot_scatter'].max())
    assert ot['txt_pad'].dtype == torch.uint8 and ot['img_pad'].shape == (5, b['img_feat'].size(1))
    # VQA
    vqa = VqaDataset(13, txt_db, img_db)
    b = vqa_collate([vqa[i] for i in range(4)])
    assert b['targets'].shape == (4, 13) and b['targets'].sum(1).tolist() == [1.0] * 4
    ev = vqa_eval_collate([VqaEvalDataset(13, txt_db, img_db)[i] for i in range(3)])
    assert ev['qids'] == ['q0', 'q1', 'q2'] and ev['targets'].shape == (3, 13)
    # retrieval evaluation through the real constructors: one text against its image + the next three, then against all images
    from uniter_amd.data import ItmEvalDataset, ItmValDataset, VeDataset, itm_val_collate, ve_collate
    val = ItmValDataset(txt_db, img_db, mini_batch_size=4)
    vb = itm_val_collate([val[5]])
    assert vb['input_ids'].shape[0] == 4 and torch.equal(vb['input_ids'][0], vb['input_ids'][3]) and not hasattr(val, 'lens')
    assert int(vb['attn_masks'][0].sum()) == vb['input_ids'].size(1) + name2nbb[examples['q5']['img_fname']]
    mbs = ItmEvalDataset(txt_db, img_db, mini_batch_size=4)[5]
    assert [m['img_feat'].size(0) for m in mbs] == [4, 4, 1]
    counts = [int(m['attn_masks'][r].sum()) - m['input_ids'].size(1) for m in mbs for r in range(m['attn_masks'].size(0))]
    assert counts == sorted(counts) and sorted(counts) == sorted(name2nbb.values())       # all images, by box count
    assert VeDataset.__mro__[1] is VqaDataset and ve_collate is vqa_collate
    both = ConcatDatasetWithLens([vqa, vqa])
    assert len(both) == 46 and both.lens == vqa.lens * 2 and both.__len__() == 46


def test_nlvr2_datasets_over_a_database(tmp_path):
    txt_dir, img_dir, images, name2nbb, examples = _build_db(str(tmp_path / 'pair'), False, two_images=True)
    txt_db, img_db = TxtTokLmdb(txt_dir, -1), DetectFeatLmdb(img_dir, 0.2, 10, 4, 36, False)
    paired = Nlvr2PairedDataset(txt_db, img_db)
    ex = examples['q0']
    assert paired.lens[0] == 2 * len(ex['input_ids']) + sum(name2nbb[f] for f in ex['img_fname'])
    b = nlvr2_paired_collate([paired[i] for i in range(4)])
    _check_joint(b, 8)
    assert b['targets'].tolist() == [0, 1, 0, 1] and torch.equal(b['input_ids'][0], b['input_ids'][1])
    assert set(b['img_type_ids'][0].tolist()) <= {0, 1} and set(b['img_type_ids'][1].tolist()) <= {0, 2}
    trip = Nlvr2TripletDataset(txt_db, img_db, use_img_type=False)
    row = trip[1]
    assert row[4] is None and row[1].size(0) == sum(name2nbb[f] for f in examples['q1']['img_fname'])
    assert nlvr2_triplet_collate([trip[i] for i in range(3)])['img_type_ids'] is None
    # write path of the text database (prepro tooling): records written through TxtLmdb are read back by it
    class _Mem(object):
        def __init__(self):
            self.d = {}

        def get(self, k):
            return self.d.get(k)

        def put(self, k, v):
            self.d[k] = v

        def close(self):
            pass
    db = TxtLmdb.__new__(TxtLmdb)
    db.readonly, db.env = False, _Mem()
    db['k'] = ex
    assert db['k'] == ex
    db.readonly = True
    with pytest.raises(ValueError):
        db['k2'] = ex


# ---- the pipeline feeding the model on the GPU ---------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_database_to_gpu_step_equals_direct_batches(tmp_path):
    """Database -> dataset -> token buckets -> collate -> PrefetchLoader (side-stream copy, bf16 cast) -> model on the MI355X:
    the loss of every batch equals the loss of the same collated batch handed over directly."""
    from tests.common import TINY_CONFIG
    from uniter_amd.model.nlvr2 import UniterForNlvr2PairedAttn
    from uniter_amd.utils.misc import set_dropout
    assert torch.cuda.is_available()
    txt_dir, img_dir, _, _, _ = _build_db(str(tmp_path / 'gpu'), True, n_txt=37, with_pack=True, two_images=True)
    data = Nlvr2PairedDataset(TxtTokLmdb(txt_dir, -1), DetectFeatLmdb(img_dir, 0.2, 10, 4, 36, True))
    torch.manual_seed(0)
    model = UniterForNlvr2PairedAttn.from_pretrained(TINY_CONFIG, {}, img_dim=64)
    model.init_type_embedding()
    model.cuda().bfloat16()
    set_dropout(model, 0.0)
    for m in model.modules():
        if hasattr(m, 'dropout') and isinstance(m.dropout, float):
            m.dropout = 0.0
    model.eval()
    batches = [list(b) for b in TokenBucketSampler(data.lens, bucket_size=16, batch_size=300, size_multiple=2, rng=random.Random(4))]
    loader = torch.utils.data.DataLoader(data, batch_sampler=batches, collate_fn=nlvr2_paired_collate, num_workers=0)
    n = 0
    with torch.no_grad():
        for idx, got in zip(batches, PrefetchLoader(loader, float_dtype=torch.bfloat16)):
            direct = nlvr2_paired_collate([data[i] for i in idx])
            assert got['img_feat'].is_cuda and got['img_feat'].dtype == torch.bfloat16 and len(got['seq_lens']) == 2 * len(idx)
            dev = {k: (v.cuda().to(torch.bfloat16) if (torch.is_tensor(v) and v.is_floating_point()) else (v.cuda() if torch.is_tensor(v) else v))
                   for k, v in direct.items()}
            a = model(got, compute_loss=True)
            b = model(dev, compute_loss=True)
            assert a.shape == (len(idx),) and torch.equal(a, b) and torch.isfinite(a).all()
            n += len(idx)
    assert n == 37


def test_token_bucket_batches_merge_into_one_batch_per_optimizer_step(tmp_path):
    """Database -> MlmDataset / ItmDataset -> TokenBucketSampler -> DataLoader -> MetaLoader(accum_steps=2, merge_micro_batches=True):
    every batch the loop receives equals the collate of the examples of its two token-bucket micro-batches (whose padded widths
    differ), and `micro` carries the per-micro-batch row counts the loss reduction needs (uniter_amd/data/merge.py)."""
    from uniter_amd.data import MetaLoader
    txt_dir, img_dir, _, _, _ = _build_db(str(tmp_path / 'merge'), False)
    txt_db = TxtTokLmdb(txt_dir, max_txt_len=-1)
    img_db = ImageLmdbGroup(0.2, 10, 4, 36, False)[img_dir]

    class _Fixed(MlmDataset):                      # deterministic masking, so the same example can be drawn twice
        def create_mlm_io(self, input_ids):
            ids = torch.tensor([self.txt_db.cls_] + list(input_ids) + [self.txt_db.sep])
            labels = torch.full_like(ids, -1)
            labels[1] = ids[1]
            ids = ids.clone()
            ids[1] = self.txt_db.mask
            return ids, labels
    mlm = _Fixed(txt_db, img_db)
    itm = ItmDataset(txt_db, img_db, neg_sample_p=0.5, rng=random.Random(2), np_rng=np.random.RandomState(4))

    def loader(ds, collate, seed):
        sampler = TokenBucketSampler(ds.lens, bucket_size=8, batch_size=96, size_multiple=2, rng=random.Random(seed))
        batches = [b for b in sampler]                                     # (the sampler refuses len(), as upstream)
        return torch.utils.data.DataLoader(ds, batch_sampler=batches, collate_fn=collate, num_workers=0), batches
    l_mlm, b_mlm = loader(mlm, mlm_collate, 1)
    l_itm, b_itm = loader(itm, itm_ot_collate, 2)
    assert len(b_mlm) >= 4 and len({len(b) for b in b_mlm}) > 1               # ragged micro-batches
    meta = MetaLoader({'mlm': (l_mlm, 1), 'itm': (l_itm, 1)}, accum_steps=2, rng=random.Random(9), merge_micro_batches=True)
    cursor = {'mlm': 0, 'itm': 0}
    plan = {'mlm': (mlm, mlm_collate, b_mlm), 'itm': (itm, itm_ot_collate, b_itm)}
    it = iter(meta)
    widths = set()
    for _ in range(5):
        task, batch = next(it)
        ds, collate, batches = plan[task]
        k = cursor[task]
        if k + 2 > len(batches):                                               # (an exhausted loader restarts: data/loader.py:50-54)
            break
        cursor[task] = k + 2
        group = batches[k] + batches[k + 1]
        whole = collate([ds[i] for i in group])
        info = batch.pop('micro')
        assert info['rows'] == [len(batches[k]), len(batches[k + 1])]
        assert sorted(batch) == sorted(whole)
        for key, v in whole.items():
            if isinstance(v, dict):
                assert all(torch.equal(v[q], batch[key][q]) if isinstance(v[q], torch.Tensor) else v[q] == batch[key][q] for q in v), key
            else:
                assert torch.equal(v, batch[key]), (task, key)
        widths.add((collate([ds[i] for i in batches[k]])['input_ids'].size(1), collate([ds[i] for i in batches[k + 1]])['input_ids'].size(1)))
    assert any(a != b for a, b in widths)                                      # at least one step merged micro-batches of different text widths
