"""Run the STAGED reference (oracle/_ref/reference, see oracle/make_ref.py) on the CPU: the headline NLVR2 paired-attention
training step exactly as train_nlvr2.py:153-195 drives it — model(batch, compute_loss=True).mean().backward(), clip_grad_norm_,
lr from get_lr_sched, the reference's own AdamW.step(), zero_grad — with apex's FusedLayerNorm shimmed by torch.nn.LayerNorm
(apex is CUDA-only; same arithmetic) and amp / horovod left out (single process, fp32).

TEST INFRASTRUCTURE: imported by bench.py's cpu_baseline leg and tests only; never by uniter_amd/."""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
STAGED = os.path.join(HERE, "_ref", "reference")


def available():
    return os.path.exists(os.path.join(STAGED, "model", "nlvr2.py")) and os.path.exists(os.path.join(STAGED, "optim", "adamw.py"))


def _import_reference():
    """The reference's packages are called `model` and `optim` (no top-level package): import them from the staged tree under
    private names so they cannot shadow or be shadowed by anything else on sys.path."""
    import importlib.util
    for name in ('apex', 'apex.normalization', 'apex.normalization.fused_layer_norm'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['apex.normalization.fused_layer_norm'].FusedLayerNorm = torch.nn.LayerNorm
    mods = {}
    for pkg in ("model", "optim"):
        init = os.path.join(STAGED, pkg, "__init__.py")
        spec = importlib.util.spec_from_file_location(
            "uniter_reference_" + pkg, init if os.path.exists(init) else None,
            submodule_search_locations=[os.path.join(STAGED, pkg)])
        if spec is None or spec.loader is None:          # namespace package (model/ has no __init__.py)
            m = types.ModuleType("uniter_reference_" + pkg)
            m.__path__ = [os.path.join(STAGED, pkg)]
            sys.modules[m.__name__] = m
        else:
            m = importlib.util.module_from_spec(spec)
            sys.modules[m.__name__] = m
            spec.loader.exec_module(m)
        mods[pkg] = m
    import importlib
    nlvr2 = importlib.import_module("uniter_reference_model.nlvr2")
    adamw = importlib.import_module("uniter_reference_optim.adamw")
    sched = importlib.import_module("uniter_reference_optim.sched")
    return nlvr2, adamw, sched


class ReferenceNlvr2Step:
    """UniterForNlvr2PairedAttn + AdamW of the reference, on given fp32 weights and a given batch."""

    def __init__(self, cfg_path, state_dict, train, img_dim=2048):
        nlvr2, adamw, sched = _import_reference()
        self.sched = sched
        self.train = train
        model = nlvr2.UniterForNlvr2PairedAttn.from_pretrained(cfg_path, {}, img_dim=img_dim)
        model.init_type_embedding()
        missing, unexpected = model.load_state_dict(state_dict, strict=False)
        assert not missing and not unexpected, (missing, unexpected)
        model.train()                                   # dropout on, as in training
        self.model = model
        # optim/misc.py:12-35 build_optimizer: no weight decay for bias / LayerNorm.{bias,weight}
        no_decay = ['bias', 'LayerNorm.bias', 'LayerNorm.weight']
        named = list(model.named_parameters())
        groups = [{'params': [p for n, p in named if not any(nd in n for nd in no_decay)], 'weight_decay': train['weight_decay']},
                  {'params': [p for n, p in named if any(nd in n for nd in no_decay)], 'weight_decay': 0.0}]
        self.opt = adamw.AdamW(groups, lr=train['learning_rate'], betas=tuple(train['betas']))

    def step(self, batch, global_step):
        import warnings
        t = self.train
        loss = self.model(batch, compute_loss=True).mean()
        loss.backward()
        lr = self.sched.get_lr_sched(global_step, types.SimpleNamespace(
            learning_rate=t['learning_rate'], warmup_steps=t['warmup_steps'], num_train_steps=t['num_train_steps'],
            decay='linear'))
        for g in self.opt.param_groups:
            g['lr'] = lr
        torch.nn.utils.clip_grad_norm_(self.model.parameters(), t['grad_norm'])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")             # the reference's add_(Number, Tensor) overloads are deprecated
            self.opt.step()
        self.opt.zero_grad()
        return float(loss.detach())
