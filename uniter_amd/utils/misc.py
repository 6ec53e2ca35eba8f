"""Small helpers the training loops use (reference utils/misc.py:17-70)."""
import json
import random
import sys

import numpy as np
import torch

from .. import ops


class NoOp(object):
    """Swallows every method call; stands in for loggers / savers on non-zero ranks."""

    def __getattr__(self, name):
        return self.noop

    def noop(self, *args, **kwargs):
        return


def parse_with_config(parser):
    """argparse defaults < JSON --config < flags given explicitly on the command line."""
    args = parser.parse_args()
    if args.config is not None:
        with open(args.config) as f:
            config_args = json.load(f)
        explicit = {arg[2:].split('=')[0] for arg in sys.argv[1:] if arg.startswith('--')}
        for key, value in config_args.items():
            if key not in explicit:
                setattr(args, key, value)
    del args.config
    return args


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
