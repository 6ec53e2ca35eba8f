"""SURVEY.md section 8 row f-4, eval half: retrieval recalls and the NLVR2 prediction loop (reference utils/itm_eval.py,
inf_nlvr2.py:84-106).  The vectorised recall computation is checked against a literal restatement of the reference's
loops on random score matrices, including its row-index counting quirk."""
import random

import torch

from uniter_amd.utils import itm_eval as IE
from uniter_amd.utils import nlvr2_eval as NE


def _upstream_itm_eval(score_matrix, txt_ids, img_ids, txt2img, img2txts):
    # the reference's algorithm, loop for loop (utils/itm_eval.py:17-64)
    img2j = {i: j for j, i in enumerate(img_ids)}
    _, rank_txt = score_matrix.topk(10, dim=1)
    gt_img_j = torch.LongTensor([img2j[txt2img[t]] for t in txt_ids]).unsqueeze(1).expand_as(rank_txt)
    rank = (rank_txt == gt_img_j).nonzero()
    ir = [(rank < c).sum().item() / len(txt_ids) for c in (1, 5, 10)] if rank.numel() else [0, 0, 0]
    txt2i = {t: i for i, t in enumerate(txt_ids)}
    _, rank_img = score_matrix.topk(10, dim=0)
    tr = [0, 0, 0]
    for j, img_id in enumerate(img_ids):
        gt_is = [txt2i[t] for t in img2txts[img_id]]
        ranks = [(rank_img[:, j] == i).nonzero() for i in gt_is]
        r = min([10] + [x.item() for x in ranks if x.numel()])
        for k, c in enumerate((1, 5, 10)):
            tr[k] += int(r < c)
    tr = [v / len(img_ids) for v in tr]
    return tr, ir


def _retrieval_problem(seed, n_img=23, caps=3):
    rng = random.Random(seed)
    img_ids = ["img%d" % j for j in range(n_img)]
    txt_ids, txt2img, img2txts = [], {}, {i: [] for i in img_ids}
    for j, img in enumerate(img_ids):
        for c in range(rng.randint(1, caps)):
            t = "t%d_%d" % (j, c)
            txt_ids.append(t)
            txt2img[t] = img
            img2txts[img].append(t)
    rng.shuffle(txt_ids)
    g = torch.Generator().manual_seed(seed)
    score = torch.randn(len(txt_ids), n_img, generator=g)
    img2j = {i: j for j, i in enumerate(img_ids)}
    for r, t in enumerate(txt_ids):                      # make the ground truth likely but not certain
        score[r, img2j[txt2img[t]]] += 1.5
    return score, txt_ids, img_ids, txt2img, img2txts


def test_itm_eval_matches_the_reference_loops():
    for seed in range(6):
        score, txt_ids, img_ids, txt2img, img2txts = _retrieval_problem(seed)
        tr, ir = _upstream_itm_eval(score, txt_ids, img_ids, txt2img, img2txts)
        got = IE.itm_eval(score, txt_ids, img_ids, txt2img, img2txts)
        want = {'txt_r1': tr[0], 'txt_r5': tr[1], 'txt_r10': tr[2], 'img_r1': ir[0], 'img_r5': ir[1], 'img_r10': ir[2]}
        for k, v in want.items():
            assert abs(got[k] - v) < 1e-12, (seed, k, got[k], v)
        assert abs(got['txt_r_mean'] - sum(tr) / 3) < 1e-12 and abs(got['img_r_mean'] - sum(ir) / 3) < 1e-12
        assert abs(got['r_mean'] - (sum(tr) / 3 + sum(ir) / 3) / 2) < 1e-12
        plain = IE.itm_eval(score, txt_ids, img_ids, txt2img, img2txts, upstream_counting=False)
        assert plain['img_r1'] <= got['img_r1'] and plain['img_r10'] <= got['img_r10']
        assert all(0.0 <= plain[k] <= 1.0 for k in plain if k != 'r_mean')


class _Dset:
    def __init__(self, ids, all_img_ids, txt2img, img2txts):
        self.ids, self.all_img_ids, self.txt2img, self.img2txts = ids, all_img_ids, txt2img, img2txts

    def __len__(self):
        return len(self.ids)


class _Loader(list):
    dataset = None


class _RankModel(torch.nn.Module):
    def __init__(self, score):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))
        self.score = score

    def forward(self, batch, compute_loss=False):
        return self.score[batch['row'], batch['cols']].unsqueeze(1)


def test_inference_and_evaluate_single_process():
    score, txt_ids, img_ids, txt2img, img2txts = _retrieval_problem(11, n_img=12)
    loader = _Loader([[{'row': i, 'cols': torch.arange(0, 7)}, {'row': i, 'cols': torch.arange(7, 12)}]
                      for i in range(len(txt_ids))])
    loader.dataset = _Dset(txt_ids, img_ids, txt2img, img2txts)
    model = _RankModel(score)
    model.train()
    m = IE.inference(model, loader, device=torch.device("cpu"), dtype=torch.float32)
    assert torch.equal(m, score) and model.training                   # model put back into training mode
    log = IE.evaluate(model, loader)
    ref = IE.itm_eval(score.to(torch.bfloat16).float(), txt_ids, img_ids, txt2img, img2txts)
    assert all(abs(log[k] - ref[k]) < 1e-12 for k in ref)


class _Nlvr2Model(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))

    def forward(self, batch, compute_loss=False):
        assert 'targets' not in batch and 'qids' not in batch
        return batch['logits']


def test_nlvr2_evaluate_and_results_file(tmp_path):
    g = torch.Generator().manual_seed(2)
    batches = []
    for b in range(3):
        logits = torch.randn(4, 2, generator=g)
        batches.append({'qids': ["q%d_%d" % (b, i) for i in range(4)], 'targets': torch.zeros(4), 'logits': logits})
    model = _Nlvr2Model()
    model.train()
    lines = []
    res = NE.evaluate(model, batches, log=lines.append)
    assert model.training and len(res) == 12 and len(lines) == 1
    for (qid, ans), (b, i) in zip(res, [(b, i) for b in range(3) for i in range(4)]):
        assert qid == "q%d_%d" % (b, i)
        assert ans == ('True' if batches[b]['logits'][i, 1] > batches[b]['logits'][i, 0] else 'False')
    assert 'qids' in batches[0] and 'targets' in batches[0]            # the caller's batch dicts are left intact
    out = tmp_path / "results.csv"
    NE.write_results(res, str(out))
    assert out.read_text().splitlines()[0] == "%s,%s" % res[0]
