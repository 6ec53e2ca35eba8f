"""Small helpers the training loops use (reference utils/misc.py:17-70)."""
import random

import numpy as np
import torch

from .. import ops


class Struct(object):
    def __init__(self, dict_):
        self.__dict__.update(dict_)


def set_dropout(model, drop_p):
    """Overwrite p of every nn.Dropout (the HIP encoder reads these values at each call)."""
    for _, module in model.named_modules():
        if isinstance(module, torch.nn.Dropout) and module.p != drop_p:
            module.p = drop_p


def set_random_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    ops.manual_seed(seed)      # Philox stream of the in-kernel dropout
