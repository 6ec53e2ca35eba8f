"""Step loops of the reference training scripts on synthetic, HBM-resident batches.

One `StepRunner.train_step()` = one OPTIMIZER step of the corresponding reference loop, including its gradient
accumulation, loss reduction, LR schedule, clipping and parameter-group handling:

  c2  train_nlvr2.py:153-195   UNITER-base, UniterForNlvr2PairedAttn, accumulation 1, lr 3e-5, clip 2.0
                               (config/train-nlvr2-base-1gpu.json)
  c3  pretrain.py:264-335      UNITER-base, UniterForPretraining, task drawn per optimizer step from the pool
                               itm:mlm:mrfr:mrckl = 2:2:1:1 (both datasets of config/pretrain-indomain-base-8gpu.json
                               carry the same ratios), accumulation 2, lr 5e-5, clip 5.0, ITM + 0.1 * OT
  c4  train_vqa.py:183-229     UNITER-large, UniterForVisualQuestionAnswering, accumulation 4, four parameter groups
                               with lr_mul 10 on vqa_output, loss.mean() * 3129, clip 2.0
                               (config/train-vqa-large-8gpu.json)
  c5  pretrain.py:264-335      UNITER-large, pool itm:mlm:mrfr:mrckl = 6:8:4:4 + 6:8:4:4 ... of
                               config/pretrain-alldata-large-16gpu.json, long sequences 128 txt + 50 img, the 16-GPU
                               accumulation 4 collapsed to 8 GPUs x accumulation 2 (BASELINE.json configs[4])

Gradients of the micro-steps are SUMMED (no division by the accumulation count), the allreduce (N > 1) happens once,
on the last micro-step, before clipping, and the LR is written into the parameter groups before `step()` — the
reference's order (SURVEY.md §8 checklist 11-12).
"""
import random

import os
import torch

BASE = dict(vocab_size=28996, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
            hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
            max_position_embeddings=512, type_vocab_size=2, initializer_range=0.02)
LARGE = dict(BASE, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)

IMG_DIM, IMG_LABEL_DIM, NUM_ANSWER = 2048, 1601, 3129

WORKLOADS = {
    'c2': dict(model='nlvr2', cfg=BASE, batch=32, max_txt_len=60, num_bb=36, accum=1, learning_rate=3e-5, betas=(0.9, 0.98),
               weight_decay=0.01, grad_norm=2.0, warmup_steps=800, num_train_steps=8000, dropout=0.1, optim='adamw',
               desc="UNITER-base NLVR2 paired-attn finetune step (config/train-nlvr2-base-1gpu.json shapes): "
                    "fwd+bwd+clip+fused AdamW, dropout 0.1, random-init weights"),
    'c3': dict(model='pretrain', cfg=BASE, batch=32, max_txt_len=60, num_bb=36, accum=2, learning_rate=5e-5,
               betas=(0.9, 0.98), weight_decay=0.01, grad_norm=5.0, warmup_steps=10000, num_train_steps=200000, dropout=0.1,
               optim='adamw', itm_ot_lambda=0.1, mix=(('itm', 2), ('mlm', 2), ('mrfr', 1), ('mrckl', 1)),
               desc="UNITER-base in-domain pretrain step (config/pretrain-indomain-base-8gpu.json): task drawn per optimizer "
                    "step from itm:mlm:mrfr:mrckl = 2:2:1:1, 2 accumulated micro-batches of 32, ITM + 0.1*OT, clip 5.0"),
    'c4': dict(model='vqa', cfg=LARGE, batch=32, max_txt_len=60, num_bb=36, accum=4, learning_rate=5e-5, lr_mul=10.0,
               betas=(0.9, 0.98), weight_decay=0.01, grad_norm=2.0, warmup_steps=500, num_train_steps=5000, dropout=0.1,
               optim='adamw',
               desc="UNITER-large VQA finetune step (config/train-vqa-large-8gpu.json): 4 accumulated micro-batches of 32, "
                    "4 parameter groups (lr_mul 10 on vqa_output), loss.mean()*3129, clip 2.0"),
    'c5': dict(model='pretrain', cfg=LARGE, batch=32, max_txt_len=128, num_bb=50, accum=2, learning_rate=5e-5,
               betas=(0.9, 0.98), weight_decay=0.01, grad_norm=5.0, warmup_steps=10000, num_train_steps=500000, dropout=0.1,
               optim='adamw', itm_ot_lambda=0.1, mix=(('itm', 6), ('mlm', 8), ('mrfr', 4), ('mrckl', 4)),
               desc="UNITER-large all-data pretrain step (config/pretrain-alldata-large-16gpu.json collapsed to accumulation 2): "
                    "long sequences 128 txt + 50 img, task mix itm:mlm:mrfr:mrckl = 6:8:4:4, ITM + 0.1*OT, clip 5.0"),
}


class _Opts(object):
    def __init__(self, d):
        self.__dict__.update(d)


def encoder_flops(B, L, H, I, n_layers):
    """Algorithmic FLOP of one encoder forward: n_layers * (24*T*H^2 + 4*T*L*H) for I = 4H."""
    T = B * L
    return n_layers * (2.0 * T * H * (3 * H + H + 2 * I) + 4.0 * T * L * H)


def build_model(kind, cfg, device, seed, cfg_path):
    """Random-initialised model of the reference architecture (`from_pretrained(cfg, {})`, pretrain.py:215-221), bf16."""
    import json

    from .utils.misc import set_random_seed
    with open(cfg_path, "w") as f:
        json.dump(cfg, f)
    set_random_seed(seed)
    if kind == 'nlvr2':
        from .model.nlvr2 import UniterForNlvr2PairedAttn
        model = UniterForNlvr2PairedAttn.from_pretrained(cfg_path, {}, img_dim=IMG_DIM)
        model.init_type_embedding()                                   # use_img_type (train_nlvr2.py:117)
    elif kind == 'pretrain':
        from .model.pretrain import UniterForPretraining
        model = UniterForPretraining.from_pretrained(cfg_path, {}, img_dim=IMG_DIM, img_label_dim=IMG_LABEL_DIM)
    elif kind == 'vqa':
        from .model.vqa import UniterForVisualQuestionAnswering
        model = UniterForVisualQuestionAnswering.from_pretrained(cfg_path, {}, img_dim=IMG_DIM, num_answer=NUM_ANSWER)
    else:
        raise ValueError(kind)
    model.to(device).bfloat16()
    return model


class StepRunner(object):
    """Model + optimizer + resident synthetic batches of one workload; `train_step()` runs one optimizer step."""

    def __init__(self, name, device, rank=0, world=1, seed=77, ragged=False, pack=False, overlap=False, cfg_path=None,
                 reducer_layers_per_bucket=None, merge_accum=False):
        from .optim import build_optimizer, build_vqa_optimizer, overlap_boundaries
        from .utils import distributed as D
        from .utils.arena import flatten_model
        from .utils.misc import set_dropout
        from .utils.synthetic import make_batch, to_device
        if name not in WORKLOADS:
            raise ValueError("unknown workload %r (have %s)" % (name, sorted(WORKLOADS)))
        w = dict(WORKLOADS[name])
        self.name, self.w, self.opts = name, w, _Opts(w)
        # merge_accum: the `accum` micro-batches of an optimizer step run as ONE batch of accum x batch sequences (data/merge.py:
        # same examples, same loss, same gradients as the accumulation loop; GEMMs accum times as tall).  Default off.
        self.merge_accum = bool(merge_accum) and int(w['accum']) > 1
        self.device, self.rank, self.world = device, rank, world
        self.model = build_model(w['model'], w['cfg'], device, seed, cfg_path or "/tmp/uniter_cfg_%s_%d.json" % (name, rank))
        set_dropout(self.model, w['dropout'])
        self.model.train()
        self.arena = flatten_model(self.model)
        D.broadcast_tensors([p.data for p in self.model.parameters()], 0)            # pretrain.py:225
        self.optimizer = (build_vqa_optimizer if w['model'] == 'vqa' else build_optimizer)(self.model, self.opts)
        if overlap:
            self.optimizer.enable_overlap(overlap_boundaries(self.model))
        else:
            self.optimizer.fuse_zero_grad = True
        if reducer_layers_per_bucket is None:
            # 4-layer buckets: a backward range's deferred weight-gradient launch is 432 tiles (1.7 rounds of 256 CUs); with 3
            # layers it is 324 (1.3 rounds) and the one-rank RCCL step measures 5.49 ms against 5.26 (EXPERIMENTS.md section 5)
            reducer_layers_per_bucket = int(os.environ.get("UNITER_AMD_LAYERS_PER_BUCKET", "4"))
        self._reducer_args = dict(layers_per_bucket=reducer_layers_per_bucket,
                                  word_embeddings=self.model.uniter.embeddings.word_embeddings.weight,
                                  word_ids_cap=int(w['batch']) * int(w['max_txt_len']) * (int(w['accum']) if self.merge_accum else 1))     # (every rank's text is padded to at most this)
        self.reducer = (D.GradientReducer(self.arena, self.model.uniter.encoder, **self._reducer_args)
                        if (world > 1 or D._on()) else None)
        self.model.uniter.pack_padding = bool(pack)
        # each rank trains on its own shard (data/data.py:222): different synthetic batches per rank, resident in HBM
        tasks = [t for t, _ in w['mix']] if w['model'] == 'pretrain' else [w['model']]
        self.batches = {}
        for i, t in enumerate(tasks):
            b = make_batch(t, w['batch'], w['max_txt_len'], w['num_bb'], seed=1000 + 17 * i + rank, ragged=ragged,
                           with_ot=(t == 'itm'))
            b = to_device(b, device)
            b['img_feat'] = b['img_feat'].to(torch.bfloat16)          # fp16 features under amp O2 in the reference
            b['img_pos_feat'] = b['img_pos_feat'].to(torch.bfloat16)
            if 'feat_targets' in b:
                b['feat_targets'] = b['feat_targets'].to(torch.bfloat16)
            if self.merge_accum:
                from .data.merge import merge_batches
                b = merge_batches([b] * int(w['accum']))              # (the micro-batches of a step are the same resident batch here)
            self.batches[t] = b
        if self.reducer is None:
            # single process: the backward call returns without joining the weight-gradient stream; clip_grad_norm_ / step join
            # it (uniter_amd.optim.AdamW), so the embedding backward overlaps the deferred weight-gradient launch
            from . import ops as _ops
            self.model.uniter.encoder.grad_ready_hook = _ops.DeferWgradJoin()
        self.pool = [t for t, r in w.get('mix', ((tasks[0], 1),)) for _ in range(int(r))]
        self.rng = random.Random(seed)                                # same draw on every rank (data/loader.py:42-47)
        self.global_step = 0
        self.task_counts = {t: 0 for t in tasks}
        self.optimizer.zero_grad()
        self.optimizer.step()                                         # the reference's dummy first step (no-op: no grads)

    # ------------------------------------------------------------------------------------------------
    @property
    def examples_per_step(self):
        return self.w['batch'] * self.w['accum']

    @property
    def seq_len(self):
        return self.w['max_txt_len'] + self.w['num_bb']

    def flop_per_step(self):
        c = self.w['cfg']
        return 3.0 * self.w['accum'] * encoder_flops(self.w['batch'], self.seq_len, c['hidden_size'], c['intermediate_size'],
                                                     c['num_hidden_layers'])

    def _schedule_lr(self):
        from .optim import get_lr_sched
        self.global_step += 1
        lr = get_lr_sched(self.global_step, self.opts)
        if self.w['model'] == 'vqa':                                  # train_vqa.py:208-214
            for i, g in enumerate(self.optimizer.param_groups):
                g['lr'] = lr * self.w['lr_mul'] if i < 2 else lr
        else:
            for g in self.optimizer.param_groups:
                g['lr'] = lr

    def _loss(self, task, batch):
        kind = self.w['model']
        micro = batch.get('micro') if self.merge_accum else None      # merged micro-batches: per-micro-batch reductions, summed
        if micro is not None:
            from .data.merge import accumulated_itm_ot_loss, accumulated_loss
        if kind == 'nlvr2':
            loss = self.model(batch, compute_loss=True)
            return loss.mean() if micro is None else accumulated_loss(loss, micro)
        if kind == 'vqa':
            loss = self.model(batch, compute_loss=True)
            if micro is not None:
                return accumulated_loss(loss, micro, scale=batch['targets'].size(1))
            return loss.mean() * batch['targets'].size(1)             # train_vqa.py:188
        loss = self.model(batch, task=task, compute_loss=True)
        if task.startswith('itm'):                                    # pretrain.py:270-290
            itm_loss, ot_loss = loss
            if micro is not None:
                return accumulated_itm_ot_loss(itm_loss, ot_loss, batch['targets'], micro, self.w['itm_ot_lambda'])
            itm_loss = itm_loss.mean()
            if ot_loss is not None:
                ot_pos, ot_neg = ot_loss
                ot = (ot_pos.sum() - ot_neg.sum()) / (ot_pos.size(0) + ot_neg.size(0))
                return itm_loss + self.w['itm_ot_lambda'] * ot
            return itm_loss
        return loss.mean() if micro is None else accumulated_loss(loss, micro)

    def set_dp_mode(self, single_launch):
        """Data-parallel exchange mode of the following steps: True = ONE deferred weight-gradient launch whose gradient buckets go
        out behind flags (default), False = one backward call and one event-ordered collective per bucket (utils/distributed.py)."""
        from .utils import distributed as D
        if self.reducer is None or bool(self.reducer.single_launch) == bool(single_launch):
            return
        self.reducer = D.GradientReducer(self.arena, self.model.uniter.encoder, single_launch=bool(single_launch), **self._reducer_args)

    def dp_check_step(self, single_launch):
        """bench.py's start-up self-check (N > 1): forward + backward + gradient exchange of the first resident batch with dropout
        off in the given exchange mode; returns an order-sensitive 64-bit digest of the reduced bf16 gradient arena.  Parameters and
        optimizer state are not touched; the gradients are zero again on return and the runner keeps the mode."""
        from . import _lib
        from .utils.misc import set_dropout
        self.set_dp_mode(single_launch)
        task, batch = next(iter(self.batches.items()))
        set_dropout(self.model, 0.0)
        # (model/attention.py's MultiheadAttention — the NLVR2 paired-attention head — keeps its dropout as a float attribute that
        #  set_dropout does not reach: left on, two runs of ONE mode already differ and the check would always fall back)
        floats = [(m, m.dropout) for m in self.model.modules() if isinstance(getattr(m, 'dropout', None), float)]
        for m, _ in floats:
            m.dropout = 0.0
        try:
            self.reducer.begin()
            loss = self._loss(task, batch)
            loss.backward()
            self.reducer.finish(word_ids=None)
            _lib.join_wgrads()
            g = self.arena.grad.view(torch.int16).to(torch.int64)
            w = torch.arange(g.numel(), device=g.device, dtype=torch.int64).remainder_(8191).add_(1)
            digest = int((g * w).sum().item())
        finally:
            set_dropout(self.model, self.w['dropout'])
            for m, v in floats:
                m.dropout = v
            self.arena.grad.zero_()
        return digest

    def warm_up_tasks(self):
        """One untimed optimizer step per task of the mix, so that one-off work (tile selection for a task's head shapes,
        allocator growth) is not charged to whichever timed step happens to draw that task first."""
        saved = self.rng.getstate()
        for t in self.batches:
            self._forced_task = t
            self.train_step()
        self._forced_task = None
        self.rng.setstate(saved)
        for t in self.task_counts:
            self.task_counts[t] = 0

    def train_step(self):
        """One optimizer step: `accum` micro-batches of one task, summed gradients, allreduce, LR, clip, AdamW, zero_grad."""
        from .optim import clip_grad_norm_
        task = self.rng.choice(self.pool)                             # one task per optimizer step (data/loader.py:42-47)
        if getattr(self, '_forced_task', None) is not None:
            task = self._forced_task
        self.task_counts[task] += 1
        batch = self.batches[task]
        accum = 1 if self.merge_accum else self.w['accum']          # merged: one forward / backward over all micro-batches
        self._schedule_lr()
        loss = None
        seg = getattr(self, 'segment_events', None)                   # bench.py: GPU time of the forward / backward segments
        for micro in range(accum):
            last = micro == accum - 1
            if self.reducer is not None and last:
                self.reducer.begin()                                  # earlier micro-steps only accumulate locally
            if seg is not None:
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record()
            loss = self._loss(task, batch)
            if seg is not None:
                e1.record()
            loss.backward()
            if seg is not None:
                from . import _lib as _l
                _l.join_wgrads()                                      # the backward segment ends when every weight gradient is written
                e2.record()
                seg.append((e0, e1, e2))
        # (every task but MLM reaches the word-embedding table only through the input lookup: its gradient is exchanged as rows;
        #  the micro-batches of a step are the same resident batch here, so its ids are all the ids of the step)
        scale = self.reducer.finish(word_ids=None if task == 'mlm' else batch['input_ids']) if self.reducer is not None else 1.0
        clip_grad_norm_(self.optimizer, self.opts.grad_norm, grad_scale=scale)
        self.optimizer.step()
        self.optimizer.zero_grad()
        return loss
