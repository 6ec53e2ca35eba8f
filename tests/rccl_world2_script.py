"""Helper of tests/test_rccl_gpu.py::test_two_rank_rccl_step_matches_single_process_mean: launched on a box with >= 2 GPUs as
`python -m torch.distributed.run --nproc-per-node 2` (one process per GPU, RCCL over xGMI).  Every rank runs two optimizer
steps of the tiny NLVR2 model on its OWN batch through GradientReducer (real backward hook, bucketed in-place bf16 sum
allreduce, 1/world folded into clip / AdamW); rank 0 then checks that (1) both ranks hold bit-identical parameters, and
(2) they equal — to bf16 gradient rounding — the parameters of a single process that averaged the two ranks' gradients
itself.  Prints one JSON line on rank 0."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def build(dev, g):
    from tests.common import IMG_DIM, TINY_CONFIG
    from uniter_amd.model.nlvr2 import UniterForNlvr2PairedAttn
    from uniter_amd.utils.arena import flatten_model
    from uniter_amd.utils.misc import set_dropout
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
    return model, flatten_model(model)


def main():
    from tests.common import load_golden
    from uniter_amd.optim import build_optimizer, clip_grad_norm_
    from uniter_amd.utils import distributed as D
    from uniter_amd.utils.misc import Struct
    from uniter_amd.utils.synthetic import to_device
    # UNITER_W2_BACKEND=gloo + UNITER_W2_ONE_GPU=1: both ranks share GPU 0 and reduce over gloo — the same code path (real
    # kernels, real backward hook, two ranks with different data) on the one-GPU test boxes, where RCCL cannot form a group
    backend = os.environ.get("UNITER_W2_BACKEND", "nccl")
    local = 0 if os.environ.get("UNITER_W2_ONE_GPU") == "1" else int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    D.init(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == 2 and dist.get_backend() == backend
    g = load_golden()
    base = g.batch('nlvr2')
    # rank r trains on the golden batch with its examples rolled by r pairs: different data per rank, same shapes
    batches = []
    for r in range(world):
        b = {k: (torch.roll(v, shifts=2 * r, dims=0) if torch.is_tensor(v) and v.dim() > 0 and v.size(0) == base['input_ids'].size(0) else v)
             for k, v in base.items()}
        if 'targets' in b and torch.is_tensor(b['targets']):
            b['targets'] = torch.roll(base['targets'], shifts=r, dims=0)
        batches.append(to_device(b, dev))
    opts = Struct(dict(optim='adamw', learning_rate=1e-3, betas=(0.9, 0.98), weight_decay=0.01))

    model, arena = build(dev, g)
    D.broadcast_tensors([p.data for p in model.parameters()], 0)
    opt = build_optimizer(model, opts)
    reducer = D.GradientReducer(arena, model.uniter.encoder, layers_per_bucket=1)
    for step in range(2):
        reducer.begin()
        model(batches[rank], compute_loss=True).mean().backward()
        scale = reducer.finish()
        clip_grad_norm_(opt, 1.0, grad_scale=scale)
        opt.step()
        opt.zero_grad()
    torch.cuda.synchronize()
    mine = arena.data.detach().clone()
    both = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    out = None
    if rank == 0:
        identical = bool(torch.equal(both[0], both[1]))
        # single-process reference: the same two steps with the two ranks' gradients averaged by hand (fp32 mean of the
        # bf16 gradients, rounded to bf16 once — the collective rounds the SUM once and the 1/world lands in fp32)
        ref_model, ref_arena = build(dev, g)
        ref_opt = build_optimizer(ref_model, opts)
        for step in range(2):
            acc = torch.zeros(ref_arena.numel, dtype=torch.float32, device=dev)
            for r in range(world):
                ref_opt.zero_grad()
                ref_model(batches[r], compute_loss=True).mean().backward()
                acc += ref_arena.grad.float()
            ref_arena.grad.copy_(acc.to(torch.bfloat16))         # the sum, as the allreduce leaves it
            clip_grad_norm_(ref_opt, 1.0, grad_scale=1.0 / world)
            ref_opt.step()
            ref_opt.zero_grad()
        torch.cuda.synchronize()
        diff = (both[0].float() - ref_arena.data.float()).abs()
        moved = (ref_arena.data.float() - arena.data.float()).abs().max()
        out = {"identical_across_ranks": identical, "max_abs_diff_vs_single_process": float(diff.max()),
               "mean_abs_diff_vs_single_process": float(diff.mean()), "arena_elements": int(arena.numel),
               "max_abs_param": float(ref_arena.data.float().abs().max()), "world": world, "backend": dist.get_backend()}
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
