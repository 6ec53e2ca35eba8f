"""Optimizer construction (reference optim/misc.py:12-35)."""
from torch.optim import Adam, Adamax

from .adamw import AdamW

# substrings of parameter NAMES that switch weight decay off.  The match is by case-sensitive substring, as in
# the reference, so e.g. `img_layer_norm.weight` and LayerNorms inside nn.Sequential (`vqa_output.2.weight`)
# ARE decayed — a quirk kept on purpose (SURVEY.md §8 checklist 13).
NO_DECAY = ('bias', 'LayerNorm.bias', 'LayerNorm.weight')


def split_decay(named_parameters, weight_decay):
    """Two param groups: decayed / not decayed, each in named_parameters order."""
    named = list(named_parameters)
    decayed = [p for n, p in named if not any(tag in n for tag in NO_DECAY)]
    plain = [p for n, p in named if any(tag in n for tag in NO_DECAY)]
    return [{'params': decayed, 'weight_decay': weight_decay}, {'params': plain, 'weight_decay': 0.0}]


def build_optimizer(model, opts):
    groups = split_decay(model.named_parameters(), opts.weight_decay)
    classes = {'adam': Adam, 'adamax': Adamax, 'adamw': AdamW}
    if opts.optim not in classes:
        raise ValueError('invalid optimizer')
    return classes[opts.optim](groups, lr=opts.learning_rate, betas=opts.betas)


def build_vqa_optimizer(model, opts):
    """Four param groups: the `vqa_output` head (decayed / not decayed) then the rest (decayed / not decayed) — the head's
    groups get `lr * opts.lr_mul` written into them every step by the training loop (train_vqa.py:51-86,208-214)."""
    named = list(model.named_parameters())
    top = [(n, p) for n, p in named if 'vqa_output' in n]
    rest = [(n, p) for n, p in named if 'vqa_output' not in n]
    groups = []
    for part, with_lr in ((top, True), (rest, False)):
        for g in split_decay(part, opts.weight_decay):
            if with_lr:
                g['lr'] = opts.learning_rate
            groups.append(g)
    classes = {'adam': Adam, 'adamax': Adamax, 'adamw': AdamW}
    if opts.optim not in classes:
        raise ValueError('invalid optimizer')
    return classes[opts.optim](groups, lr=opts.learning_rate, betas=opts.betas)
