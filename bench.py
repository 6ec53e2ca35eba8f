#!/usr/bin/env python3
"""Headline benchmark: UNITER-base training throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full optimizer step of the reference's NLVR2 fine-tuning loop (train_nlvr2.py:153-195) on a
synthetic batch already resident in HBM: UniterForNlvr2PairedAttn forward (32 sequences of 60 text + 36 region
tokens per GPU = 16 pairs), loss.mean(), backward, gradient allreduce (N > 1, RCCL, overlapped with backward),
LR schedule, global-norm clipping (2.0), fused AdamW, zero_grad.  Model = config/uniter-base.json shapes,
hyper-parameters = config/train-nlvr2-base-1gpu.json (lr 3e-5, betas (0.9, 0.98), wd 0.01, dropout 0.1,
accumulation 1), bf16 parameters/activations with fp32 master weights and fp32 Adam state (the apex-O2 analogue).
Prints ONE JSON line (rank 0).  `value` counts encoder sequences ("examples") per second over all GPUs.

Extra objects on the line:
  roofline      the dominant HIP kernel (largest summed in-situ duration per step) timed live with HIP events that the
                library records around every launch on the stream it is launched on, over extra optimizer steps right
                after the timed region; algorithmic FLOP / average launch duration vs the dense bf16 MFMA peak
                (2.5 PFLOP/s, MI355X_MICROARCH.md).  `traffic` = HBM bytes per launch from the committed PMC passes
                (profiles/*_pmc_traffic.json).  `step` adds the whole-step figure (encoder fwd+bwd algorithmic FLOP /
                step time).
  cpu_baseline  the CPU oracle (oracle/uniter_oracle.py, a port of the reference path) timed on this box's host
                cores on the same workload — a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

BASE_CFG = dict(vocab_size=28996, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
                max_position_embeddings=512, type_vocab_size=2, initializer_range=0.02)
TRAIN = dict(batch=32, max_txt_len=60, num_bb=36, learning_rate=3e-5, betas=(0.9, 0.98), weight_decay=0.01,
             grad_norm=2.0, warmup_steps=800, num_train_steps=8000, dropout=0.1, optim='adamw')
MFMA_PEAK_TFLOPS = 2500.0


def encoder_flops(B, L, H, I, n_layers):
    """Algorithmic FLOP of one encoder forward: n_layers * (24*T*H^2 + 4*T*L*H); I = 4H."""
    T = B * L
    return n_layers * (2.0 * T * H * (3 * H + H + 2 * I) + 4.0 * T * L * H)


def write_cfg(path):
    with open(path, "w") as f:
        json.dump(BASE_CFG, f)


def build_model(device, cfg_path, seed):
    from uniter_amd.model.nlvr2 import UniterForNlvr2PairedAttn
    from uniter_amd.utils.misc import set_dropout, set_random_seed
    set_random_seed(seed)
    model = UniterForNlvr2PairedAttn.from_pretrained(cfg_path, {}, img_dim=2048)      # no checkpoint: random init
    model.init_type_embedding()                                                       # use_img_type (train_nlvr2.py:117)
    model.to(device).bfloat16()
    set_dropout(model, TRAIN['dropout'])
    model.train()
    return model


def kernel_flop(rec):
    """Algorithmic FLOP of one launch of a timed kernel (None for the HBM-bound ones)."""
    k, M, N, K = rec["kind_id"], rec["M"], rec["N"], rec["K"]
    if k <= 5:
        return 2.0 * M * N * K                     # every GEMM flavour: M tokens x N out x K in (C-ABI argument order)
    if k == 13:
        return 2.0 * M * N                         # grouped wgrad: N = sum_i N_i*K_i weight elements (include/uniter_hip.h)
    if k == 6:
        return 4.0 * M * N * N * K * 64            # attention fwd: B=M, L=N, heads=K, d=64: QK^T + PV
    if k == 7:
        return 8.0 * M * N * N * K * 64            # attention bwd: dV, dP, dQ, dK (algorithmic 2x forward)
    return None


def timed_pass(train_step, steps):
    """Run `steps` more optimizer steps with the library's per-launch HIP events enabled (events are recorded on the
    stream each kernel is launched on — the main stream or the backward's wgrad side stream) and return one row per
    (kind, shape): launches per step, average launch duration in situ, achieved TFLOP/s."""
    from uniter_amd import _lib
    torch.cuda.synchronize()
    _lib.timing_begin()
    for _ in range(steps):
        train_step()
    recs = _lib.timing_end()
    rows = []
    for r in recs:
        us = r["total_us"] / max(r["calls"], 1)
        fl = kernel_flop(r)
        rows.append({"kernel": r["kind"], "kind_id": r["kind_id"], "shape": [r["M"], r["N"], r["K"]],
                     "launches_per_step": round(r["calls"] / steps, 2), "us": round(us, 2),
                     "us_per_step": round(r["total_us"] / steps, 1),
                     "tflops": None if fl is None else round(fl / us * 1e-6, 1)})
    rows.sort(key=lambda x: -x["us_per_step"])
    return rows


def pmc_traffic(kind_id, shape):
    """HBM bytes per launch of one encoder GEMM (flavour = epilogue id, shape = M, N, K of the C-ABI call) from the
    committed PMC passes (profiles/*_pmc_traffic.json, produced by scripts/profile_round.sh +
    scripts/summarize_profile.py; rocprofv3 cannot run inside this process).  None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None
    try:
        doc = json.load(open(files[-1]))
    except (OSError, ValueError):
        return None
    M, N, K = shape
    e = doc.get("by_shape", {}).get("%d:%d:%d:%d" % (kind_id, M, N, K))
    if e is None:
        return None
    return {"hbm_bytes": e["hbm_bytes"], "hbm_read_bytes": e["hbm_read_bytes"], "hbm_write_bytes": e["hbm_write_bytes"],
            "kernel": e["kernel"], "grid": e["grid"], "source": os.path.basename(files[-1])}


def usable_cores():
    """Host cores this process may actually use: affinity mask and cgroup CPU quota, not the raw host count."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def cpu_baseline(seed=77, budget_s=20.0):
    """The oracle (CPU port of the reference path) on the same workload: fwd + bwd + clip + AdamW, fp32.
    Runs in this process; main() calls it through a subprocess with a hard timeout."""
    from oracle import uniter_oracle as O
    from uniter_amd.model.nlvr2 import UniterForNlvr2PairedAttn
    from uniter_amd.utils.synthetic import make_batch
    cores = min(usable_cores(), 64)
    torch.set_num_threads(cores)
    cfg_path = os.path.join("/tmp", "uniter_base_cpu_%d.json" % os.getpid())
    write_cfg(cfg_path)
    torch.manual_seed(seed)
    ref_model = UniterForNlvr2PairedAttn.from_pretrained(cfg_path, {}, img_dim=2048)   # only a weight container
    ref_model.init_type_embedding()
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in ref_model.state_dict().items()}
    del ref_model
    os.remove(cfg_path)
    B = TRAIN['batch']
    batch = make_batch('nlvr2', B, TRAIN['max_txt_len'], TRAIN['num_bb'], seed=seed)
    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in sd.items()}

    def step(i):
        for v in sd.values():
            v.grad = None
        loss, _ = O.nlvr2_paired_attn_loss(sd, BASE_CFG, batch)
        loss.mean().backward()
        grads = [v.grad for v in sd.values() if v.grad is not None]
        _, coef = O.clip_coef(grads, TRAIN['grad_norm'])
        lr = O.get_lr_sched(i + 1, TRAIN['learning_rate'], TRAIN['warmup_steps'], TRAIN['num_train_steps'])
        with torch.no_grad():
            for k, v in sd.items():
                if v.grad is None:
                    continue
                m, s = state[k]
                if coef != 1.0:
                    v.grad.mul_(coef)                    # clip_grad_norm_ scales in place (pretrain.py:329-331)
                O.adamw_step_(v.data, v.grad, m, s, i + 1, lr, TRAIN['betas'], 1e-6,
                              0.0 if O.no_decay(k) else TRAIN['weight_decay'])

    t0 = time.time()
    step(0)                                   # warm-up
    warm = time.time() - t0
    times = []
    i = 1
    while i <= 3 and (sum(times) + warm) < budget_s:
        t0 = time.time()
        step(i)
        times.append(time.time() - t0)
        i += 1
    best = min(times) if times else warm
    return {"value": round(B / best, 2), "unit": "examples/s", "cores": cores, "kind": "port",
            "sample": "oracle/uniter_oracle.py (fp32 torch-CPU port of the reference NLVR2 paired-attn step: fwd+bwd+clip+AdamW) "
                      "on the same UNITER-base workload, B=%d x (60+36), 1 warm-up + %d timed step(s), best %.2f s/step, "
                      "%d torch threads" % (B, len(times), best, cores)}


def cpu_baseline_subprocess(timeout_s=150):
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], capture_output=True, text=True,
                           timeout=timeout_s, env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "examples/s", "cores": usable_cores(), "kind": "port",
                "sample": "cpu baseline subprocess failed: " + (r.stderr.strip().splitlines() or ["?"])[-1][:200]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "examples/s", "cores": usable_cores(), "kind": "port",
                "sample": "cpu baseline did not finish within %d s" % timeout_s}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--overlap", action="store_true",
                    help="run the optimizer update asynchronously under the next forward pass (AdamW.enable_overlap)")
    ap.add_argument("--graph", action="store_true",
                    help="capture the whole optimizer step into a hipGraph and replay it (N=1 only); measured slower than "
                         "eager issue on ROCm 7.2 (8.65 vs 8.27 ms), so it is opt-in")
    ap.add_argument("--ragged", action="store_true",
                    help="NOT the headline config: ragged synthetic batch (10-60 text tokens, 10-36 regions per sequence) "
                         "to measure padding-free execution (SURVEY.md section 8 f-3)")
    ap.add_argument("--pack", action="store_true", help="run the encoder on real tokens only (UniterModel.pack_padding)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline()), flush=True)
        return

    from uniter_amd.optim import build_optimizer, clip_grad_norm_, get_lr_sched
    from uniter_amd.utils import distributed as D
    from uniter_amd.utils.arena import flatten_model
    from uniter_amd.utils.misc import Struct
    from uniter_amd.utils.synthetic import make_batch, to_device

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                             "--master-addr 127.0.0.1 --master-port 29500 bench.py --gpus %d ..." % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the encoder path has no CPU fallback")
    local = D.local_rank()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    D.init("nccl")
    rank = D.rank()

    cfg_path = os.path.join("/tmp", "uniter_base_bench_%d.json" % os.getpid())
    write_cfg(cfg_path)
    opts = Struct(TRAIN)
    model = build_model(device, cfg_path, seed=77)                      # same seed -> same init on every rank
    arena = flatten_model(model)
    D.broadcast_tensors([p.data for p in model.parameters()], 0)        # train_nlvr2.py:118
    optimizer = build_optimizer(model, opts)
    overlap = bool(args.overlap and not args.graph)
    if overlap:
        # opt-in: the AdamW update runs on the optimizer stream in per-layer segments under the next step's forward.
        # Measured neutral on one MI355X (the forward pass is itself bound by the memory system the update saturates:
        # 5.51 vs 5.46 ms/step), so the default is the synchronous step with zero_grad folded into the update kernel.
        from uniter_amd.optim import overlap_boundaries
        optimizer.enable_overlap(overlap_boundaries(model))
    elif not args.graph:
        optimizer.fuse_zero_grad = True
    lpb = int(os.environ.get("UNITER_BENCH_LAYERS_PER_BUCKET", "4"))
    reducer = D.GradientReducer(arena, model.uniter.encoder, layers_per_bucket=lpb) if (world > 1 or D._on()) else None
    # each rank trains on its own shard (data/data.py:222): different synthetic batch per rank, resident in HBM
    batch = to_device(make_batch('nlvr2', TRAIN['batch'], TRAIN['max_txt_len'], TRAIN['num_bb'], seed=1000 + rank,
                                 ragged=args.ragged), device)
    model.uniter.pack_padding = bool(args.pack)
    real_tokens = int(batch['attn_masks'].sum().item())
    batch['img_feat'] = batch['img_feat'].to(torch.bfloat16)            # fp16 features under amp O2 in the reference
    batch['img_pos_feat'] = batch['img_pos_feat'].to(torch.bfloat16)

    optimizer.zero_grad()
    optimizer.step()                                                     # train_nlvr2.py:150-151 dummy step (no-op)
    global_step = 0

    def schedule_lr():                        # train_nlvr2.py:174-178 (host side, before the update)
        nonlocal global_step
        global_step += 1
        lr_this_step = get_lr_sched(global_step, opts)
        for group in optimizer.param_groups:
            group['lr'] = lr_this_step

    def device_step():                        # everything that runs on the GPU for one optimizer step
        if reducer is not None:
            reducer.begin()
        loss = model(batch, compute_loss=True)
        loss = loss.mean()
        loss.backward()
        scale = reducer.finish() if reducer is not None else 1.0
        clip_grad_norm_(optimizer, opts.grad_norm, grad_scale=scale)
        optimizer.step()
        optimizer.zero_grad()
        return loss

    # --graph (N == 1): the whole step is captured once into a hipGraph and replayed (one launch per step instead of
    # ~550).  Default is eager issue: the step is GPU-bound and graph replay measured slower on this ROCm.
    mode = "eager"
    train_step = None
    if world == 1 and args.graph:
        from uniter_amd.utils.graph import GraphedStep
        try:
            graphed = GraphedStep(device_step, optimizer, device, warmup=3, pre_step=schedule_lr).capture()
            train_step = graphed
            mode = "hipgraph"
        except Exception as e:                                  # pragma: no cover - depends on the runtime
            sys.stderr.write("hipGraph capture failed (%s: %s); falling back to eager steps\n" % (type(e).__name__, e))
            from uniter_amd import ops as _ops
            _ops.disable_graph_rng()
            optimizer._graph = None
    if train_step is None:
        def train_step():
            schedule_lr()
            return device_step()

    for _ in range(args.warmup):
        train_step()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = train_step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss.item())
    assert final_loss == final_loss, "loss is NaN"

    # In-situ per-launch timing for the roofline object: extra optimizer steps AFTER the timed region.  With several
    # ranks these steps contain the gradient collectives, so EVERY rank runs them (only rank 0 reports).
    kernels = None
    if not args.no_kernel_timing and mode == "eager":
        tsteps = max(1, min(args.steps, 5))
        kernels = timed_pass(train_step, tsteps)
        if world > 1:
            torch.distributed.barrier()

    if rank == 0:
        B, L = TRAIN['batch'], TRAIN['max_txt_len'] + TRAIN['num_bb']
        ms = elapsed / args.steps * 1e3
        value = B * world * args.steps / elapsed
        flop_step = 3.0 * encoder_flops(B, L, BASE_CFG['hidden_size'], BASE_CFG['intermediate_size'], BASE_CFG['num_hidden_layers'])
        step_tf = flop_step / (ms * 1e-3) * 1e-12
        roofline = None
        if kernels is not None:
            mfma = [k for k in kernels if k["tflops"] is not None]
            dom = max(mfma, key=lambda k: k["us_per_step"])
            M, N, K = dom["shape"]
            traffic = pmc_traffic(dom["kind_id"], dom["shape"])
            # algorithmic HBM bytes of the dominant kernel for reference (bf16 operands + output [+ aux read])
            roofline = {"bound": "mfma", "kernel": "%s M%d N%d K%d" % (dom["kernel"], M, N, K), "achieved": dom["tflops"],
                        "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(dom["tflops"] / MFMA_PEAK_TFLOPS, 4),
                        "traffic": None if traffic is None else traffic["hbm_bytes"],
                        "traffic_detail": traffic,
                        "avg_launch_us": dom["us"], "launches_per_step": dom["launches_per_step"],
                        "measured": "HIP events around every launch on its own stream during %d extra optimizer steps; in the "
                                    "backward pass this kernel shares the GPU with the wgrad side stream, so the in-situ "
                                    "duration is longer than the kernel alone (DESIGN.md section 5)" % tsteps,
                        "step": {"algorithmic_tflop_per_step": round(flop_step * 1e-12, 4), "achieved": round(step_tf, 1),
                                 "frac": round(step_tf / MFMA_PEAK_TFLOPS, 4),
                                 "note": "encoder fwd+bwd algorithmic FLOP (heads, embeddings, optimizer excluded) / whole step time"}}
        result = {
            "metric": "train examples/sec UNITER-base seq=60txt+36img bs32/GPU", "value": round(value, 1), "unit": "examples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "UNITER-base NLVR2 paired-attn finetune step (config/train-nlvr2-base-1gpu.json shapes): "
                                   "fwd+bwd+clip+fused AdamW, dropout 0.1, random-init weights",
                       "global_batch": B * world, "seq_len": L, "parallelism": "dp%d" % world, "launch": mode, "optimizer_overlap": overlap,
                       "ragged": bool(args.ragged), "pack_padding": bool(args.pack),
                       "real_token_fraction": round(real_tokens / float(B * batch['attn_masks'].size(1)), 3),
                       "examples": "encoder sequences (32/GPU = 16 NLVR2 pairs)"},
            "final_loss": round(final_loss, 4),
            "roofline": roofline,
        }
        if kernels is not None:
            result["kernels"] = kernels
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline_subprocess()
        else:
            result["cpu_baseline"] = None
        # RCCL prints its banner through C stdio, which sits in libc's buffer until exit when stdout is a pipe: flush
        # it now so that the JSON line is the LAST line this process writes
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except (OSError, AttributeError):
            pass
        print(json.dumps(result), flush=True)
    try:
        os.remove(cfg_path)
    except OSError:
        pass
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
