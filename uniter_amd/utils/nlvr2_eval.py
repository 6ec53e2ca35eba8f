"""NLVR2 inference loop (eval half).  Reference: inf_nlvr2.py:84-106 (and the validation
loop of train_nlvr2.py): arg-max over the two classes per example, 'True' / 'False' strings keyed by question id."""
from time import time

import torch


@torch.no_grad()
def evaluate(model, eval_loader, device=None, log=None):
    """Returns [(qid, 'True' | 'False'), ...] for every example of the loader; the model is put back into its mode.
    The arg-max indices of a batch come back to the host in one transfer after the whole loop has been issued, so the
    GPU is never idle waiting for Python (the reference calls `.cpu().tolist()` once per batch)."""
    was_training = model.training
    model.eval()
    st = time()
    qids, picks = [], []
    for batch in eval_loader:
        batch = dict(batch)
        qids.extend(batch.pop('qids'))
        batch.pop('targets', None)
        scores = model(batch, compute_loss=False)
        picks.append(scores.max(dim=-1, keepdim=False)[1])
    answers = torch.cat(picks).cpu().tolist() if picks else []
    results = [(q, 'True' if a == 1 else 'False') for q, a in zip(qids, answers)]
    model.train(was_training)
    if log is not None:
        tot = max(time() - st, 1e-9)
        log("evaluation finished in %d seconds at %d examples per second" % (int(tot), int(len(results) / tot)))
    return results


def write_results(results, path):
    """results.csv of inf_nlvr2.py:76-79: one `qid,answer` line per example."""
    with open(path, 'w') as f:
        for id_, ans in results:
            f.write('%s,%s\n' % (id_, ans))
