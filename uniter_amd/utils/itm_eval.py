"""Image-text retrieval evaluation (eval half).  Reference: utils/itm_eval.py:17-113.

`inference` fills one row of the [texts x images] score matrix per loader item from `model(batch, compute_loss=False)`
(bf16 on the device, where the reference keeps fp16), `evaluate` gathers the rows of every rank through torch.distributed
(RCCL / gloo, instead of Horovod) and lets rank 0 compute the recalls.  `itm_eval` computes the same nine numbers as the
reference, but from two top-10 index tables in a handful of tensor operations: the reference walks the images in a Python
loop with one `.nonzero()` (a device-to-host synchronisation) per ground-truth caption, i.e. ~5 syncs per image."""
from time import time

import torch

from . import distributed as D


@torch.no_grad()
def itm_eval(score_matrix, txt_ids, img_ids, txt2img, img2txts, upstream_counting=True):
    """score_matrix [n_txt, n_img]; returns the dict of utils/itm_eval.py:55-64 (recall@1/5/10 both ways + means).

    upstream_counting: the reference's image-retrieval recalls are computed as `(rank < c).sum()` over the [hits, 2]
    output of `.nonzero()`, which also counts the ROW index of every hit (utils/itm_eval.py:26-30): a text whose index is
    below c and whose image is in its top 10 adds one extra count.  True reproduces those published numbers digit for
    digit; False gives the plain recall."""
    n_txt, n_img = len(txt_ids), len(img_ids)
    k = min(10, n_img)
    dev = score_matrix.device
    # image retrieval: rank of the ground-truth image among each text's top-10 images
    img2j = {i: j for j, i in enumerate(img_ids)}
    gt_img = torch.tensor([img2j[txt2img[t]] for t in txt_ids], dtype=torch.long, device=dev)
    rank_txt = score_matrix.topk(k, dim=1)[1]                                    # [n_txt, k]
    hit = rank_txt == gt_img.unsqueeze(1)
    pos = torch.where(hit.any(dim=1), hit.float().argmax(dim=1), torch.full((n_txt,), 10, device=dev))
    ir = []
    for c in (1, 5, 10):
        count = int((pos < c).sum().item())
        if upstream_counting:
            count += int((pos[:c] < 10).sum().item())
        ir.append(float(count) / n_txt)
    # text retrieval: best rank of any ground-truth caption among each image's top-10 texts
    txt2i = {t: i for i, t in enumerate(txt_ids)}
    kt = min(10, n_txt)
    rank_img = score_matrix.topk(kt, dim=0)[1]                                   # [kt, n_img]
    is_gt = torch.zeros(n_txt, n_img, dtype=torch.bool, device=dev)
    rows = [txt2i[t] for img_id in img_ids for t in img2txts[img_id]]
    cols = [j for j, img_id in enumerate(img_ids) for _ in img2txts[img_id]]
    if rows:
        is_gt[torch.tensor(rows, device=dev), torch.tensor(cols, device=dev)] = True
    hit_t = is_gt.gather(0, rank_img)                                            # [kt, n_img]: is the r-th text a caption?
    best = torch.where(hit_t.any(dim=0), hit_t.float().argmax(dim=0), torch.full((n_img,), 10, device=dev))
    tr = [float((best < c).sum().item()) / n_img for c in (1, 5, 10)]
    tr_mean, ir_mean = sum(tr) / 3, sum(ir) / 3
    return {'txt_r1': tr[0], 'txt_r5': tr[1], 'txt_r10': tr[2], 'txt_r_mean': tr_mean,
            'img_r1': ir[0], 'img_r5': ir[1], 'img_r10': ir[2], 'img_r_mean': ir_mean,
            'r_mean': (tr_mean + ir_mean) / 2}


@torch.no_grad()
def inference(model, eval_loader, device=None, dtype=torch.bfloat16):
    """One row of scores per loader item: the item is a list of mini-batches that together cover every image."""
    was_training = model.training
    model.eval()
    if device is None:
        device = next(model.parameters()).device
    dset = eval_loader.dataset
    score_matrix = torch.zeros(len(dset), len(dset.all_img_ids), device=device, dtype=dtype)
    for i, mini_batches in enumerate(eval_loader):
        j = 0
        for batch in mini_batches:
            scores = model(batch, compute_loss=False)
            bs = scores.size(0)
            score_matrix[i, j:j + bs] = scores.reshape(bs).to(dtype)        # rank scores [bs, 1] (model/itm.py)
            j += bs
        assert j == score_matrix.size(1), "the mini-batches of an item must cover every image exactly once"
    model.train(was_training)
    return score_matrix


def _gather_rows(score_matrix):
    """Concatenate the per-rank row blocks (ranks may own different numbers of texts)."""
    if not D._on():
        return score_matrix
    import torch.distributed as dist
    counts = D.all_gather_list(int(score_matrix.size(0)))
    most = max(counts)
    padded = score_matrix.new_zeros(most, score_matrix.size(1))
    padded[:score_matrix.size(0)] = score_matrix
    parts = [torch.empty_like(padded) for _ in counts]
    dist.all_gather(parts, padded)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


@torch.no_grad()
def evaluate(model, eval_loader):
    """utils/itm_eval.py:68-90: every rank scores its shard of the texts, rank 0 reports (other ranks return {})."""
    st = time()
    score_matrix = inference(model, eval_loader)
    dset = eval_loader.dataset
    all_score = _gather_rows(score_matrix)
    all_txt_ids = [i for ids in D.all_gather_list(list(dset.ids)) for i in ids] if D._on() else list(dset.ids)
    all_img_ids = dset.all_img_ids
    assert all_score.size() == (len(all_txt_ids), len(all_img_ids))
    if D.rank() != 0:
        return {}
    eval_log = itm_eval(all_score.float(), all_txt_ids, all_img_ids, dset.txt2img, dset.img2txts)
    eval_log['eval_seconds'] = time() - st
    return eval_log
