"""Helper of tests/test_rccl_gpu.py, run as a subprocess on the GPU box with UNITER_DIST_FORCE=1 (a one-rank RCCL group):
two optimizer steps of a tiny NLVR2 model through GradientReducer with the REAL backward hook (the encoder backward hands
control back at bucket boundaries, the bucket's in-place allreduce runs on the reducer's side stream while the library's
internal wgrad stream is still busy) against the same two steps without any collective.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    from tests.common import IMG_DIM, TINY_CONFIG, load_golden
    from uniter_amd.model.nlvr2 import UniterForNlvr2PairedAttn
    from uniter_amd.optim import build_optimizer, clip_grad_norm_
    from uniter_amd.utils import distributed as D
    from uniter_amd.utils.arena import flatten_model
    from uniter_amd.utils.misc import Struct, set_dropout
    from uniter_amd.utils.synthetic import to_device
    assert os.environ.get("UNITER_DIST_FORCE") == "1"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    D.init("nccl")
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1 and D._on()
    calls = {"n": 0, "elems": 0}
    real_all_reduce = dist.all_reduce

    def counting_all_reduce(t, *a, **k):
        calls["n"] += 1
        calls["elems"] += t.numel()
        return real_all_reduce(t, *a, **k)
    dist.all_reduce = counting_all_reduce

    g = load_golden()
    batch = to_device(g.batch('nlvr2'), dev)
    opts = Struct(dict(optim='adamw', learning_rate=1e-3, betas=(0.9, 0.98), weight_decay=0.01))
    # UNITER_W1_WIDE=1: a configuration the deferred weight-gradient launch accepts (hidden / intermediate sizes multiples of 256,
    # a multiple of 64 tokens): the reducer then keeps the backward ONE call and waits for per-bucket flags (hipStreamWaitValue32)
    wide = os.environ.get("UNITER_W1_WIDE") == "1"
    lpb = int(os.environ.get("UNITER_W1_LAYERS_PER_BUCKET", "1"))
    if wide:
        import json as _json
        import tempfile
        from uniter_amd.utils.synthetic import make_batch
        cfg = dict(vocab_size=512, hidden_size=256, num_hidden_layers=4, num_attention_heads=4, intermediate_size=512,
                   hidden_act="gelu", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, max_position_embeddings=64,
                   type_vocab_size=2, initializer_range=0.02)
        cfg_path = os.path.join(tempfile.mkdtemp(), "wide.json")
        open(cfg_path, "w").write(_json.dumps(cfg))
        batch = to_device(make_batch('nlvr2', 8, max_txt_len=20, num_bb=12, img_dim=IMG_DIM, vocab_size=512, seed=5), dev)   # 8 x 32 tokens
    results = []
    for use_reducer in (False, True):
        if wide:
            torch.manual_seed(3)
            model = UniterForNlvr2PairedAttn.from_pretrained(cfg_path, {}, img_dim=IMG_DIM)
            model.init_type_embedding()
        else:
            w, nl = g.weights('pre'), g.weights('nlvr2')
            table3 = nl.pop('uniter.embeddings.token_type_embeddings.weight')
            model = UniterForNlvr2PairedAttn.from_pretrained(TINY_CONFIG, {**w, **nl}, img_dim=IMG_DIM)
            model.init_type_embedding()
            model.uniter.embeddings.token_type_embeddings.weight.data.copy_(table3)
        model.to(dev).bfloat16()
        set_dropout(model, 0.0)
        for m in model.modules():
            if hasattr(m, 'dropout') and isinstance(m.dropout, float):
                m.dropout = 0.0
        model.train()
        arena = flatten_model(model)
        D.broadcast_tensors([p.data for p in model.parameters()], 0)
        opt = build_optimizer(model, opts)
        # UNITER_W1_FUSE_ZERO=1: the training loop's optimizer mode (train.py::StepRunner) — zero_grad folded into the step and, with it,
        # the lazy zero_grad of round 6: from the second step on the encoder's backward OVERWRITES its parameter gradients (bucketed
        # deferred launch included).  The plain run keeps the eager zero_grad: the two must still end bit-identical.
        fuse = os.environ.get("UNITER_W1_FUSE_ZERO") == "1"
        opt.fuse_zero_grad = fuse and use_reducer
        reducer = D.GradientReducer(arena, model.uniter.encoder, layers_per_bucket=lpb,
                                    word_embeddings=model.uniter.embeddings.word_embeddings.weight) if use_reducer else None
        if not use_reducer:
            from uniter_amd import ops as _ops
            model.uniter.encoder.grad_ready_hook = _ops.DeferWgradJoin()      # the single-process training loop's hook
        before = calls["n"]
        buckets_seen = 0
        n_steps = 3 if fuse else 2
        for step in range(n_steps):
            if reducer is not None:
                reducer.begin()
            model(batch, compute_loss=True).mean().backward()
            # NLVR2 has no MLM head: the word-embedding gradient travels as rows
            scale = reducer.finish(word_ids=batch['input_ids'] if wide else None) if reducer is not None else 1.0
            if reducer is not None and reducer.single_launch:
                buckets_seen = max(buckets_seen, getattr(reducer, "last_flag_waits", 0))      # collectives enqueued behind a flag wait
            clip_grad_norm_(opt, 1.0, grad_scale=scale)
            opt.step()
            opt.zero_grad()
        torch.cuda.synchronize()
        results.append(({n: p.detach().clone() for n, p in model.named_parameters()}, calls["n"] - before, arena.numel))
    (plain, _, _), (reduced, n_calls, numel) = results
    same = all(torch.equal(plain[n], reduced[n]) for n in plain)
    n_layers = len(model.uniter.encoder.layer)
    print(json.dumps({"identical": bool(same), "allreduce_calls": n_calls, "encoder_layers": n_layers, "flag_waits": buckets_seen,
                      "single_launch": bool(reducer.single_launch),
                      "elements_reduced_per_step": calls["elems"] // n_steps if n_calls else 0, "arena_elements": numel, "steps": n_steps,
                      "backend": dist.get_backend()}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
