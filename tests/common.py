"""Shared test helpers: golden-fixture loading and error metrics."""
import json
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden", "uniter_tiny.npz")
TINY_CONFIG = os.path.join(HERE, "golden", "uniter_tiny_config.json")
IMG_DIM, LABEL_DIM, N_ANS = 64, 11, 13


def _bf16_bits_to_f32(a):
    return torch.from_numpy(a.astype(np.int16)).view(torch.bfloat16).float()


class Golden(object):
    def __init__(self, z):
        self.cfg = json.loads(str(z["config_json"]))
        self.groups = {}
        for k in z.files:
            if "/" not in k:
                continue
            g, name = k.split("/", 1)
            self.groups.setdefault(g, {})[name] = z[k]

    def weights(self, group):
        """fp32 tensors of a bf16-stored weight group ('pre', 'vqa', 'nlvr2')."""
        return {k: _bf16_bits_to_f32(v) for k, v in self.groups[group].items()}

    def pretrain_sd(self):
        sd = self.weights("pre")
        sd["cls.predictions.decoder.weight"] = sd["uniter.embeddings.word_embeddings.weight"]
        return sd

    def batch(self, task):
        return {k: torch.from_numpy(np.array(v)) for k, v in self.groups["batch_" + task].items()}

    def out(self, task):
        return {k: torch.from_numpy(np.array(v)) for k, v in self.groups["out_" + task].items()}

    def grads(self, task):
        return {k: torch.from_numpy(np.array(v)) for k, v in self.groups["grad_" + task].items()}

    def adamw(self):
        g = self.groups["adamw"]
        p = {k[2:]: torch.from_numpy(np.array(v)) for k, v in g.items() if k.startswith("p/")}
        v = {k[2:]: torch.from_numpy(np.array(val)) for k, val in g.items() if k.startswith("v/")}
        return float(g["grad_norm"]), p, v


def load_golden():
    return Golden(np.load(GOLDEN))


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
