"""Small helpers the training loops use (reference utils/misc.py:17-70)."""
import json
import random
import sys

import numpy as np
import torch

from .. import ops


class NoOp(object):
    """Stand-in for a logger / progress bar / writer on ranks > 0: every attribute is a callable that does nothing and returns
    None (reference utils/misc.py:17-23; used by every train_*.py as `pbar = NoOp()` etc.)."""

    def __getattr__(self, name):
        return self._nothing

    def _nothing(self, *args, **kwargs):
        return None


def parse_with_config(parser, argv=None):
    """parser.parse_args() with a JSON file below the command line in precedence (reference utils/misc.py:26-36): every key of
    the file named by --config becomes an attribute unless the SAME option was also spelled on the command line (`--key value`
    or `--key=value`); the `config` attribute itself is removed.  `argv` defaults to sys.argv[1:]."""
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parser.parse_args(argv)
    if args.config is not None:
        with open(args.config) as f:
            from_file = json.load(f)
        on_cli = {a[2:].split('=')[0] for a in argv if a.startswith('--')}
        for key, value in from_file.items():
            if key not in on_cli:
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
