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

from uniter_amd.train import BASE as BASE_CFG, WORKLOADS, StepRunner, encoder_flops  # noqa: E402

TRAIN = WORKLOADS['c2']
MFMA_PEAK_TFLOPS = 2500.0


def write_cfg(path):
    with open(path, "w") as f:
        json.dump(BASE_CFG, f)


def build_model(device, cfg_path, seed):
    """UNITER-base NLVR2 paired-attention model, random init (kept for scripts/host_profile.py and friends)."""
    from uniter_amd.train import build_model as _build
    from uniter_amd.utils.misc import set_dropout
    model = _build('nlvr2', BASE_CFG, device, seed, cfg_path)
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


def segment_pass(runner, train_step, steps):
    """GPU time of the forward and backward segments of `steps` more optimizer steps: HIP events recorded on the compute
    stream around model(batch) and loss.backward() (uniter_amd/train.py records them when `segment_events` is a list).
    The backward segment ends after the library's weight-gradient side stream has been joined.  ms per optimizer step."""
    runner.segment_events = []
    torch.cuda.synchronize()
    for _ in range(steps):
        train_step()
    torch.cuda.synchronize()
    ev, runner.segment_events = runner.segment_events, None
    if not ev:
        return None
    fwd = sum(a.elapsed_time(b) for a, b, _ in ev) / steps
    bwd = sum(b.elapsed_time(c) for _, b, c in ev) / steps
    return fwd, bwd


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


def pmc_step_traffic():
    """Fabric-side bytes per optimizer step of every kernel of the forward + backward segments (everything but the optimizer's
    kernels), summed from the committed PMC passes (profiles/*_pmc_traffic.json: `by_kernel` rows, bytes per step).  None if
    the newest file has no such table."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None
    try:
        doc = json.load(open(files[-1]))
    except (OSError, ValueError):
        return None
    rows = doc.get("by_kernel")
    if not rows:
        return None
    tot = 0.0
    opt = 0.0
    for r in rows:
        b = float(r.get("hbm_bytes_per_step", 0.0))
        if "adamw" in r.get("kernel", "") or "gradsq" in r.get("kernel", ""):
            opt += b
        else:
            tot += b
    return {"fwd_bwd_bytes_per_step": int(tot), "optimizer_bytes_per_step": int(opt), "source": os.path.basename(files[-1]),
            "note": "FETCH_SIZE (x2: the gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE per kernel family, separate --pmc passes"}


def measured_step_traffic(config, timeout_s=150, extra_args=()):
    """Fabric-side bytes per optimizer step, MEASURED in this run: two `rocprofv3 --pmc` passes (FETCH_SIZE, then WRITE_SIZE — never
    in one pass, never with another trace domain; MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots") over a short child run of
    this very script (same workload, same kernels: 1 warm-up + 2 steps, no timing legs), summed per kernel and divided by the
    child's optimizer steps (= its adamw_kernel launches).  FETCH_SIZE is doubled (gfx950 tallies the 128-B requests of wide
    coalesced reads at 64 B), both counters are KiB.  Returns None when rocprofv3 is not on the box or a pass fails — the line then
    falls back to the committed passes (traffic_source says which)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    child = [sys.executable, os.path.abspath(__file__), "--config", config, "--steps", "2", "--warmup", "1", "--windows", "1",
             "--no-cpu-baseline", "--no-kernel-timing", "--no-traffic"] + list(extra_args)
    per = {}
    steps = None
    tmp = tempfile.mkdtemp(prefix="uniter_pmc_")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            env = dict(os.environ, TMPDIR="/tmp")
            try:
                r = subprocess.run([exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--"] + child,
                                   capture_output=True, text=True, timeout=timeout_s, env=env, cwd="/tmp")
            except subprocess.TimeoutExpired:
                return None
            files = glob.glob(os.path.join(out, "*", "*counter_collection.csv"))
            if r.returncode != 0 or not files:
                return None
            fam = {}
            n_adam = 0
            with open(files[0]) as f:
                for row in csv.DictReader(f):
                    if row["Counter_Name"] != counter:
                        continue
                    name = row["Kernel_Name"]
                    key = "optimizer" if ("adamw_kernel" in name or "gradsq_kernel" in name) else "fwd_bwd"
                    n_adam += 1 if "adamw_kernel" in name else 0
                    fam[key] = fam.get(key, 0.0) + float(row["Counter_Value"]) * 1024.0 * (2.0 if counter == "FETCH_SIZE" else 1.0)
            if n_adam == 0:
                return None
            # the child runs warm-up + timed steps (+ the dummy first optimizer step of the reference's loop, which launches
            # no AdamW kernel): every AdamW launch closes one step
            steps = n_adam
            per[counter] = {k: v / n_adam for k, v in fam.items()}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    rd, wr = per["FETCH_SIZE"], per["WRITE_SIZE"]
    return {"fwd_bwd_bytes_per_step": int(rd.get("fwd_bwd", 0.0) + wr.get("fwd_bwd", 0.0)),
            "fwd_bwd_read_bytes_per_step": int(rd.get("fwd_bwd", 0.0)), "fwd_bwd_write_bytes_per_step": int(wr.get("fwd_bwd", 0.0)),
            "optimizer_bytes_per_step": int(rd.get("optimizer", 0.0) + wr.get("optimizer", 0.0)),
            "steps_in_pass": steps, "source": "measured in this run", "child_argv": child[1:],
            "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over a 3-step child run of "
                    "this command; FETCH_SIZE x 2 (gfx950), KiB -> bytes; the counters sit on the L2's fabric side, so "
                    "Infinity-Cache hits are included"}


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


def _oracle_setup(seed):
    """Weights (as autograd leaves) and the batch of the headline workload on the CPU: the same seeded construction the
    GPU side uses, so both sides start from identical bf16-representable-or-not fp32 values (the GPU casts to bf16)."""
    from uniter_amd.model.nlvr2 import UniterForNlvr2PairedAttn
    from uniter_amd.utils.misc import set_random_seed
    from uniter_amd.utils.synthetic import make_batch
    cfg_path = os.path.join("/tmp", "uniter_base_cpu_%d.json" % os.getpid())
    write_cfg(cfg_path)
    set_random_seed(seed)
    ref_model = UniterForNlvr2PairedAttn.from_pretrained(cfg_path, {}, img_dim=2048)   # only a weight container
    ref_model.init_type_embedding()
    os.remove(cfg_path)
    # the GPU model holds bf16 weights: the oracle starts from exactly those values
    sd = {k: v.detach().to(torch.bfloat16).float().clone().requires_grad_(True) for k, v in ref_model.state_dict().items()}
    del ref_model
    batch = make_batch('nlvr2', TRAIN['batch'], TRAIN['max_txt_len'], TRAIN['num_bb'], seed=1000)
    return sd, batch


def _bf16_yardstick(sd, batch):
    """Gradients of the oracle run with its plain torch ops in bf16 on the GPU (what an unfused PyTorch-bf16 implementation
    of the same model reaches against the fp32 oracle) — the secondary yardstick of SURVEY.md section 8c.  Only in the
    parity leg of the cpu_baseline subprocess, after the timed region of the parent; None without a GPU."""
    if not torch.cuda.is_available():
        return None
    from oracle import uniter_oracle as O
    try:
        dev = torch.device("cuda", 0)
        sdb = {k: v.detach().to(dev, torch.bfloat16).requires_grad_(True) for k, v in sd.items()}
        b = {k: ((v.to(dev, torch.bfloat16) if v.is_floating_point() else v.to(dev)) if torch.is_tensor(v) else v) for k, v in batch.items()}
        loss, _ = O.nlvr2_paired_attn_loss(sdb, BASE_CFG, b)
        loss.float().mean().backward()
        return {k: v.grad.float().cpu() for k, v in sdb.items() if v.grad is not None}
    except Exception as e:                                   # pragma: no cover - depends on the box
        sys.stderr.write("bf16 yardstick unavailable (%s: %s)\n" % (type(e).__name__, e))
        return None


def cpu_baseline(seed=77, budget_s=20.0, parity_file=None):
    """The oracle (CPU port of the reference path) on the same workload: fwd + bwd + clip + AdamW, fp32.
    Runs in this process; main() calls it through a subprocess with a hard timeout.  With `parity_file` (loss and
    gradients of one dropout-free GPU step on the same weights and batch, written by main()) the same oracle forward /
    backward is also the parity check of the headline workload."""
    from oracle import uniter_oracle as O
    cores = min(usable_cores(), 64)
    torch.set_num_threads(cores)
    sd, batch = _oracle_setup(seed)
    B = TRAIN['batch']
    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in sd.items()}
    parity = None

    def fwd_bwd():
        for v in sd.values():
            v.grad = None
        loss, _ = O.nlvr2_paired_attn_loss(sd, BASE_CFG, batch)
        loss.mean().backward()
        return loss

    if parity_file and os.path.exists(parity_file):
        gpu = torch.load(parity_file)
        loss = fwd_bwd().detach()
        lrel = float((gpu['loss'].float() - loss).abs().max() / loss.abs().max())
        worst, worst_name, n_cmp, cos_min, zero_abs = 0.0, None, 0, 1.0, 0.0
        for k, g in gpu['grads'].items():
            ref = sd[k].grad
            if ref is None:
                continue
            g = g.float()
            den = float(ref.norm())
            # gradients that are mathematically zero (the key bias: softmax is invariant to a per-query shift of all
            # scores) are rounding noise on both sides — compare them absolutely against the layer's query-bias scale
            if den < 1e-6 * max(float(ref.numel()) ** 0.5, 1.0) or k.endswith('attention.self.key.bias'):
                zero_abs = max(zero_abs, float(g.abs().max()))
                continue
            rel = float((g - ref).norm()) / den
            cos = float((g * ref).sum() / (g.norm() * ref.norm() + 1e-30))
            n_cmp += 1
            cos_min = min(cos_min, cos)
            if rel > worst:
                worst, worst_name = rel, k
        yard = _bf16_yardstick(sd, batch)          # the oracle's own torch ops in bf16 on the GPU (tests' secondary yardstick)
        strict, worst_yard = 0, None
        if yard is not None:
            for k, g in gpu['grads'].items():
                ref = sd[k].grad
                if ref is None or k not in yard or k.endswith('attention.self.key.bias') or float(ref.norm()) < 1e-6 * max(float(ref.numel()) ** 0.5, 1.0):
                    continue
                rel = float((g.float() - ref).norm()) / float(ref.norm())
                limit = 5e-2 if k.startswith('uniter.') else 1e-1      # tests/test_gpu_parity.py GRAD_L2 / GRAD_L2_HEAD
                if rel > limit:
                    strict += 1
            if worst_name in yard:
                ref = sd[worst_name].grad
                worst_yard = float((yard[worst_name] - ref).norm()) / float(ref.norm())
        parity = {"loss_rel_err": round(lrel, 6), "grad_rel_l2_max": round(worst, 5), "grad_rel_l2_max_tensor": worst_name,
                  "grad_rel_l2_max_tensor_torch_bf16_yardstick": None if worst_yard is None else round(worst_yard, 5),
                  "gradients_over_absolute_bound": None if yard is None else strict,
                  "absolute_bounds": "rel-L2 <= 5e-2 (uniter.*) / 1e-1 (head); tensors over it are held to 2x the "
                                     "torch-bf16 yardstick by tests/test_gpu_parity.py::test_headline_nlvr2_base_step_vs_oracle",
                  "grad_cosine_min": round(cos_min, 6), "gradients_compared": n_cmp,
                  "zero_gradients_max_abs": float("%.3e" % zero_abs),
                  "what": "one dropout-free step of the headline workload (UNITER-base NLVR2 paired-attn, B=32, L=96, 12 layers) on "
                          "the GPU vs oracle/uniter_oracle.py in fp32 on the CPU, identical bf16-representable weights and batch"}

    def step(i):
        fwd_bwd()
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

    # what is timed: the reference's own modules when build() staged them (oracle/_ref, see oracle/make_ref.py) — else the port
    from oracle import ref_runner
    kind = "port"
    if ref_runner.available():
        try:
            cfg_path = os.path.join("/tmp", "uniter_base_ref_%d.json" % os.getpid())
            write_cfg(cfg_path)
            ref = ref_runner.ReferenceNlvr2Step(cfg_path, {k: v.detach().clone() for k, v in sd.items()}, TRAIN)
            os.remove(cfg_path)

            def step(i):                          # noqa: F811  (train_nlvr2.py:153-195 with the reference's model and AdamW)
                ref.step(batch, i + 1)
            kind = "reference"
        except Exception as e:                    # pragma: no cover - a staged tree that does not import: fall back, say so
            sys.stderr.write("staged reference unusable (%s: %s): timing the oracle port\n" % (type(e).__name__, e))
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
    what = ("the reference's own model/nlvr2.py UniterForNlvr2PairedAttn + optim/adamw.py AdamW (staged by build() into the git-ignored "
            "oracle/_ref/, apex FusedLayerNorm -> torch.nn.LayerNorm, fp32, dropout 0.1, train_nlvr2.py:153-195 step order)"
            if kind == "reference" else
            "oracle/uniter_oracle.py (fp32 torch-CPU port of the reference NLVR2 paired-attn step: fwd+bwd+clip+AdamW; no staged "
            "reference on this box, the oracle is pinned to the reference by tests/golden)")
    out = {"value": round(B / best, 2), "unit": "examples/s", "cores": cores, "kind": kind,
           "sample": "%s on the same UNITER-base workload, B=%d x (60+36), 1 warm-up + %d timed step(s), best %.2f s/step, "
                     "%d torch threads" % (what, B, len(times), best, cores)}
    if parity is not None:
        out["parity"] = parity
    return out


def cpu_baseline_subprocess(timeout_s=240, parity_file=None):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"]
    if parity_file:
        cmd += ["--parity-file", parity_file]
    try:
        env = dict(os.environ)
        if not parity_file:                          # the timing leg is CPU-only; the parity leg also runs the oracle's torch ops
            env.update(HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")   # in bf16 on the (now idle) GPU: the secondary yardstick
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "examples/s", "cores": usable_cores(), "kind": "port",
                "sample": "cpu baseline subprocess failed: " + (r.stderr.strip().splitlines() or ["?"])[-1][:200]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "examples/s", "cores": usable_cores(), "kind": "port",
                "sample": "cpu baseline did not finish within %d s" % timeout_s}


def gpu_parity_probe(runner, path):
    """One dropout-free forward + backward of the headline workload on the freshly initialised GPU model; loss and every
    gradient except the word-embedding table go to `path` for the oracle leg (cpu_baseline) to compare against."""
    from uniter_amd.utils.misc import set_dropout
    from uniter_amd.utils.synthetic import make_batch, to_device
    model = runner.model
    set_dropout(model, 0.0)
    batch = to_device(make_batch('nlvr2', TRAIN['batch'], TRAIN['max_txt_len'], TRAIN['num_bb'], seed=1000), runner.device)
    batch['img_feat'] = batch['img_feat'].to(torch.bfloat16)
    batch['img_pos_feat'] = batch['img_pos_feat'].to(torch.bfloat16)
    loss = model(batch, compute_loss=True)
    loss.mean().backward()
    from uniter_amd import _lib
    _lib.join_wgrads()                        # (the step runner lets backward return before the weight-gradient stream is joined)
    grads = {n: p.grad.detach().to('cpu') for n, p in model.named_parameters()
             if p.grad is not None and 'word_embeddings' not in n}
    torch.save({'loss': loss.detach().float().cpu(), 'grads': grads}, path)
    for p in model.parameters():
        if p.grad is not None:
            p.grad.zero_()
    set_dropout(model, runner.w['dropout'])
    torch.cuda.synchronize()

# ------------------------------------------------------------------------------------------------------------------------------
# N > 1: launching, rendezvous, start-up self-check of the data-parallel mode
# ------------------------------------------------------------------------------------------------------------------------------
def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(gpus):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: replace this process by
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same args>`
    (one rank per GPU, the reference's `horovodrun -np N` of pretrain.py:169-173 / utils/distributed.py).  Does not return."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL's peer mappings need it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    argv = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    sys.stderr.flush()
    os.execve(sys.executable, argv, env)


def init_group(backend):
    """Join the process group.  First incarnation: the env:// rendezvous of torch.distributed.run.  After a watchdog re-exec
    (UNITER_BENCH_INCARNATION > 0) the ranks meet again on the same TCP store under a fresh key prefix, so that nothing the first
    incarnation left in the store (its ncclUniqueId, its barrier counters) is read by the second."""
    import torch.distributed as dist
    inc = int(os.environ.get("UNITER_BENCH_INCARNATION", "0"))
    if inc == 0 or int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        from uniter_amd.utils import distributed as D
        D.init(backend)
        return
    import datetime
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    agent_store = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "False") == "True"
    store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"]), world,
                          is_master=(rank == 0 and not agent_store), timeout=datetime.timedelta(seconds=120))
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend, store=dist.PrefixStore("uniter_bench_inc%d" % inc, store), rank=rank, world_size=world)


def reexec_with_fallback(reason):
    """Watchdog action: the self-check of the single-launch data-parallel mode did not come back.  A stream that waits for a flag
    nobody raises cannot be recovered inside the process, so every rank (all of them sit behind the same stuck collective and time
    out together) replaces itself by a fresh bench.py with the per-bucket mode forced; torch.distributed.run keeps seeing the same
    PIDs.  Never returns."""
    env = dict(os.environ)
    env["UNITER_AMD_DP_SINGLE_LAUNCH"] = "0"
    env["UNITER_BENCH_DP_FALLBACK"] = reason
    env["UNITER_BENCH_INCARNATION"] = str(int(os.environ.get("UNITER_BENCH_INCARNATION", "0")) + 1)
    sys.stderr.write("bench.py rank %s: %s -> re-exec with UNITER_AMD_DP_SINGLE_LAUNCH=0\n" % (os.environ.get("RANK", "0"), reason))
    sys.stderr.flush()
    os.execve(sys.executable, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env)


def dp_self_check(run_mode, watchdog_s=30.0, steps=2):
    """Start-up self-check of the data-parallel gradient exchange (N > 1), before anything is timed.

    run_mode(single_launch: bool) -> int digest of the reduced gradient arena after one forward + backward + exchange of the same
    batch on the same (untouched) parameters, dropout off.  The default mode (ONE deferred weight-gradient launch whose buckets go out
    behind flags, `hipStreamWaitValue32` on the communication stream: uniter_amd/utils/distributed.py) must give, `steps` times, the
    digest of the conservative mode (one backward call and one event-ordered collective per bucket), and all ranks must agree.  A
    mismatch selects the conservative mode for the run; a check that does not return within `watchdog_s` seconds re-execs every
    rank with that mode forced.  Returns (mode, note): mode in {"single_launch_flags", "per_bucket"}."""
    import threading
    import torch.distributed as dist
    forced = os.environ.get("UNITER_BENCH_DP_FALLBACK")
    if os.environ.get("UNITER_AMD_DP_SINGLE_LAUNCH", "1") == "0":
        return "per_bucket", (forced or "UNITER_AMD_DP_SINGLE_LAUNCH=0 in the environment")
    done = threading.Event()

    def watchdog():
        if not done.wait(watchdog_s):
            reexec_with_fallback("self-check of the single-launch data-parallel mode did not return within %.0f s" % watchdog_s)
    t = threading.Thread(target=watchdog, daemon=True)
    t.start()
    try:
        ref = [run_mode(False) for _ in range(steps)]
        got = [run_mode(True) for _ in range(steps)]
        ok = int(got == ref)
        # every rank must have seen the same digests (the exchange leaves identical gradients everywhere) and the same verdict
        box = [None] * dist.get_world_size()
        dist.all_gather_object(box, (ok, ref, got))
    finally:
        done.set()
    agree = all(b[1] == box[0][1] for b in box)
    if all(b[0] for b in box) and agree:
        return "single_launch_flags", "digests of %d step(s) equal to the per-bucket mode on all %d ranks" % (steps, len(box))
    why = "single-launch digests differ from the per-bucket mode on rank(s) %s" % [i for i, b in enumerate(box) if not b[0]]
    if not agree:
        why += "; per-bucket digests differ between ranks"
    return "per_bucket", why


def dry_launch(args, world):
    """--dry-launch: everything of the N > 1 start-up that needs no GPU — rendezvous (gloo), the self-check protocol with its
    watchdog and re-exec, the barrier / MAX-over-ranks timing bracket — with a stand-in for the model.  Rank 0 prints one JSON line."""
    import torch.distributed as dist
    init_group("gloo")
    rank = dist.get_rank() if dist.is_initialized() else 0
    hang = os.environ.get("UNITER_BENCH_SELFTEST_HANG") == "1" and int(os.environ.get("UNITER_BENCH_INCARNATION", "0")) == 0
    diverge = os.environ.get("UNITER_BENCH_SELFTEST_DIVERGE") == "1"

    def run_mode(single):
        if single and hang and rank == world - 1:
            time.sleep(3600)                      # a rank whose flag never comes: the others block in the collective below
        g = torch.Generator().manual_seed(1234 + rank)
        x = torch.randint(-1000, 1000, (4096,), generator=g, dtype=torch.int64)
        if single and diverge and rank == 0:
            x[7] += 1
        if world > 1:
            dist.all_reduce(x)
        return int((x * torch.arange(1, 4097)).sum().item())
    mode, note = ("single_process", "world 1")
    if world > 1:
        mode, note = dp_self_check(run_mode, watchdog_s=float(os.environ.get("UNITER_BENCH_WATCHDOG_S", "30")))
        dist.barrier()
    t0 = time.perf_counter()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "backend": "gloo", "dp_mode": mode, "dp_mode_note": note,
                          "incarnation": int(os.environ.get("UNITER_BENCH_INCARNATION", "0"))}), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--windows", type=int, default=3,
                    help="timed windows of --steps optimizer steps each; the line reports the median window (min / max beside it)")
    ap.add_argument("--config", default="c2", choices=sorted(WORKLOADS),
                    help="c2 = the headline workload (BASELINE.json metric); c3 / c4 / c5 = the other north-star "
                         "configurations as one GPU's share of the job (uniter_amd/train.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the in-run PMC passes (rocprofv3 child runs, ~1.5 min) behind roofline.traffic; the committed passes are used")
    ap.add_argument("--overlap", action="store_true",
                    help="run the optimizer update asynchronously under the next forward pass (AdamW.enable_overlap)")
    ap.add_argument("--ragged", action="store_true",
                    help="NOT the headline config: ragged synthetic batch (10-60 text tokens, 10-36 regions per sequence) "
                         "to measure padding-free execution (SURVEY.md section 8 f-3)")
    ap.add_argument("--pack", action="store_true", help="run the encoder on real tokens only (UniterModel.pack_padding)")
    ap.add_argument("--merge-accum", action="store_true",
                    help="c3 / c4 / c5: run the micro-batches of an optimizer step as ONE batch (uniter_amd/data/merge.py: same "
                         "examples, loss and gradients as the accumulation loop); the line says so in config.micro_batches_merged")
    ap.add_argument("--dry-launch", action="store_true",
                    help="N > 1 start-up only, on the CPU with gloo: self-launch, rendezvous, data-parallel self-check, no model (tests)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--parity-file", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(parity_file=args.parity_file)), flush=True)
        return

    from uniter_amd.utils import distributed as D

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)                    # python bench.py --gpus N: becomes torch.distributed.run with N ranks
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if args.dry_launch:
        dry_launch(args, world)
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the encoder path has no CPU fallback")
    local = D.local_rank()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    init_group("nccl")
    rank = D.rank()

    cfg_path = os.path.join("/tmp", "uniter_bench_%d.json" % os.getpid())
    lpb = int(os.environ["UNITER_BENCH_LAYERS_PER_BUCKET"]) if "UNITER_BENCH_LAYERS_PER_BUCKET" in os.environ else None
    overlap = bool(args.overlap)
    runner = StepRunner(args.config, device, rank=rank, world=world, seed=77, ragged=args.ragged, pack=args.pack,
                        overlap=overlap, cfg_path=cfg_path, reducer_layers_per_bucket=lpb, merge_accum=args.merge_accum)
    dp_mode, dp_note = ("single_process", None)
    if runner.reducer is not None:
        dp_mode, dp_note = dp_self_check(runner.dp_check_step, watchdog_s=float(os.environ.get("UNITER_BENCH_WATCHDOG_S", "30")))
        runner.set_dp_mode(dp_mode == "single_launch_flags")
        if rank == 0:
            sys.stderr.write("data-parallel mode: %s (%s)\n" % (dp_mode, dp_note))
    first_batch = next(iter(runner.batches.values()))
    real_tokens = int(first_batch['attn_masks'].sum().item())

    # parity of the headline workload, checked in this very run: one dropout-free GPU step now, the oracle later in the
    # cpu_baseline subprocess (same weights: same seed; same batch: same seed)
    parity_file = None
    if rank == 0 and world == 1 and args.config == 'c2' and not args.no_cpu_baseline and not args.ragged:
        parity_file = os.path.join("/tmp", "uniter_bench_parity_%d.pt" % os.getpid())
        gpu_parity_probe(runner, parity_file)

    mode = "eager"              # (whole-step hipGraph replay was measured and removed in round 6: profiles/r06_hipgraph_vs_eager.txt)
    train_step = runner.train_step
    if len(runner.batches) > 1:
        runner.warm_up_tasks()                # every task of the mix once, untimed (on top of the W warm-up steps)
    for _ in range(args.warmup):
        train_step()
    # The timed region: `--windows` windows (default 3) of EXACTLY `--steps` optimizer steps each, every window bracketed by a
    # barrier + torch.cuda.synchronize() on both sides, its time the MAX over ranks.  `value` / `ms_per_step` are those of the
    # MEDIAN window (min and max beside them): one 90 ms window on boxes that differ by several per cent cannot resolve a
    # round-over-round change of 1-2 %.
    window_s = []
    for _ in range(max(1, args.windows)):
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = train_step()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            el = float(t.item())
        window_s.append(el)
    elapsed = sorted(window_s)[len(window_s) // 2]
    final_loss = float(loss.item())
    assert final_loss == final_loss, "loss is NaN"

    # In-situ per-launch timing for the roofline object: extra optimizer steps AFTER the timed region.  With several
    # ranks these steps contain the gradient collectives, so EVERY rank runs them (only rank 0 reports).
    kernels = None
    segments = None
    if not args.no_kernel_timing:
        tsteps = max(1, min(args.steps, 5))
        segments = segment_pass(runner, train_step, max(tsteps, min(args.steps, 10)))
        kernels = timed_pass(train_step, tsteps)
        if world > 1:
            torch.distributed.barrier()

    if rank == 0:
        w = runner.w
        B, L = runner.examples_per_step, runner.seq_len
        ms = elapsed / args.steps * 1e3
        value = B * world * args.steps / elapsed
        flop_step = runner.flop_per_step()
        step_tf = flop_step / (ms * 1e-3) * 1e-12
        # roofline: the north star states its target on the ENCODER's forward + backward (>= 40 % of the dense bf16 MFMA peak), so
        # that is what `achieved` / `frac` are: algorithmic encoder FLOP per step (n_layers x (24 T H^2 +
        # 4 T L H), x3 for fwd + bwd) over the GPU time of model(batch) + loss.backward(), measured in this run with HIP events on
        # the compute stream.  The launch with the most MFMA time per step keeps its own figure under `dominant_kernel`.
        roofline = {"bound": "mfma", "kernel": "encoder forward + backward (all kernels of model(batch) and loss.backward(); embeddings and task head included in the time, excluded from the FLOP)",
                    "achieved": round(step_tf, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(step_tf / MFMA_PEAK_TFLOPS, 4),
                    "traffic": None, "subject": "whole step (no segment events in this mode)",
                    "step": {"algorithmic_tflop_per_step": round(flop_step * 1e-12, 4), "achieved": round(step_tf, 1),
                             "frac": round(step_tf / MFMA_PEAK_TFLOPS, 4),
                             "note": "encoder fwd+bwd algorithmic FLOP (heads, embeddings, optimizer excluded) / whole step time (median window)"}}
        if segments is not None:
            fwd_ms, bwd_ms = segments
            fb_tf = flop_step / ((fwd_ms + bwd_ms) * 1e-3) * 1e-12
            roofline.update({"achieved": round(fb_tf, 1), "frac": round(fb_tf / MFMA_PEAK_TFLOPS, 4), "subject": "encoder_fwd_bwd"})
            roofline["encoder_fwd_bwd"] = {"fwd_ms": round(fwd_ms, 3), "bwd_ms": round(bwd_ms, 3), "achieved": round(fb_tf, 1),
                                           "frac": round(fb_tf / MFMA_PEAK_TFLOPS, 4), "unit": "TFLOP/s",
                                           "measured": "HIP events on the compute stream around model(batch) and loss.backward() "
                                                       "over extra optimizer steps after the timed region (per optimizer step)"}
        if kernels is not None:
            mfma = [k for k in kernels if k["tflops"] is not None]
            dom = max(mfma, key=lambda k: k["us_per_step"])
            M, N, K = dom["shape"]
            traffic = pmc_traffic(dom["kind_id"], dom["shape"])
            label = "%s M%d N%d K%d" % (dom["kernel"], M, N, K)
            note = "in situ (the kernel's own stream)"
            if dom["kind_id"] == 13 and K > 4:
                # the deferred launch: every weight / bias / LayerNorm parameter gradient of one backward call (K problems
                # = 4 per layer) over M tokens, N = weight elements of all its problems
                label = "deferred weight gradients of %d layers in one launch (gemm8_multi_kernel): %d tokens x %d weight elements" % (K // 4, M, N)
                note = ("the launch runs on the library's weight-gradient stream after the backward chain of the call; the "
                        "embedding backward of the main stream overlaps its tail")
            roofline["dominant_kernel"] = {"kernel": label, "achieved": dom["tflops"], "unit": "TFLOP/s",
                                           "frac": round(dom["tflops"] / MFMA_PEAK_TFLOPS, 4),
                                           "avg_launch_us": dom["us"], "launches_per_step": dom["launches_per_step"],
                                           "traffic": None if traffic is None else traffic["hbm_bytes"], "traffic_detail": traffic,
                                           "measured": "HIP events around every launch on its own stream during %d extra optimizer steps; %s" % (tsteps, note)}
            step_traffic = None
            if world == 1 and not args.no_traffic:
                # (after the timed region and the timing legs, with this process's GPU work finished: the child has the GPU to itself)
                torch.cuda.synchronize()
                try:
                    step_traffic = measured_step_traffic(args.config, extra_args=[f for f, on in (
                        ("--merge-accum", args.merge_accum), ("--overlap", args.overlap), ("--ragged", args.ragged), ("--pack", args.pack)) if on])
                except Exception as e:                          # profiling must never cost the line
                    sys.stderr.write("in-run PMC traffic pass failed (%s: %s); using the committed passes\n" % (type(e).__name__, e))
                    step_traffic = None
            if step_traffic is None:
                step_traffic = pmc_step_traffic()
            if step_traffic is not None:
                roofline["traffic"] = step_traffic["fwd_bwd_bytes_per_step"]
                roofline["traffic_detail"] = step_traffic
        metric = "train examples/sec UNITER-base seq=60txt+36img bs32/GPU"
        if args.config != 'c2':
            metric = "train examples/sec (%s: %s seq=%dtxt+%dimg bs%dx%d/GPU)" % (
                args.config, "UNITER-large" if w['cfg']['hidden_size'] == 1024 else "UNITER-base", w['max_txt_len'], w['num_bb'],
                w['batch'], w['accum'])
        result = {
            "metric": metric, "value": round(value, 1), "unit": "examples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "%s: %s" % (args.config, w['desc']),
                       "global_batch": B * world, "micro_batch": w['batch'], "grad_accumulation": w['accum'], "seq_len": L,
                       "micro_batches_merged": bool(runner.merge_accum),
                       "parallelism": "dp%d" % world, "dp_mode": dp_mode, "dp_mode_note": dp_note, "launch": mode, "optimizer_overlap": overlap,
                       "ragged": bool(args.ragged), "pack_padding": bool(args.pack),
                       "real_token_fraction": round(real_tokens / float(first_batch['attn_masks'].numel()), 3),
                       "examples": "encoder sequences per optimizer step" + (" (32/GPU = 16 NLVR2 pairs)" if args.config == 'c2' else ""),
                       "task_draws": runner.task_counts},
            "final_loss": round(final_loss, 4),
            "roofline": roofline,
            # `roofline.achieved` / `avg_launch_us` are measured in THIS run (HIP events); `roofline.traffic` is not: it is the
            # per-launch PMC figure of the committed rocprofv3 passes (profiles/*_pmc_traffic.json, separate --pmc runs)
            "traffic_source": None if roofline.get("traffic") is None else (
                "measured in this run (rocprofv3 --pmc child passes)" if roofline["traffic_detail"].get("source") == "measured in this run"
                else "committed (%s; not measured in this run)" % roofline["traffic_detail"].get("source", "profiles/*_pmc_traffic.json")),
            "timed_windows": {"n": len(window_s), "steps_each": args.steps, "reported": "median",
                              "ms_per_step": [round(x / args.steps * 1e3, 3) for x in window_s],
                              "ms_per_step_min": round(min(window_s) / args.steps * 1e3, 3),
                              "ms_per_step_max": round(max(window_s) / args.steps * 1e3, 3)},
        }
        if kernels is not None:
            result["kernels"] = kernels
        if world == 1 and not args.no_cpu_baseline and args.config == 'c2':
            cb = cpu_baseline_subprocess(parity_file=parity_file)
            if isinstance(cb, dict) and "parity" in cb:
                result["parity"] = cb.pop("parity")
            result["cpu_baseline"] = cb
        else:
            result["cpu_baseline"] = None
        if parity_file and os.path.exists(parity_file):
            os.remove(parity_file)
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
