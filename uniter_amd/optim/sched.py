"""Learning-rate schedules (reference optim/sched.py:10-46): multipliers as pure functions of the step."""
from math import ceil


def noam_schedule(step, warmup_step=4000):
    """Transformer ("Noam") schedule: linear warm-up, then inverse square-root decay."""
    if step <= warmup_step:
        return step / warmup_step
    return (warmup_step ** 0.5) * (step ** -0.5)


def warmup_linear(step, warmup_step, tot_step):
    """BERT schedule: linear warm-up to 1 over `warmup_step`, linear decay to 0 at `tot_step`."""
    if step < warmup_step:
        return step / warmup_step
    remaining = (tot_step - step) / (tot_step - warmup_step)
    return max(0, remaining)


def vqa_schedule(step, warmup_interval, decay_interval, decay_start, decay_rate):
    """MCAN-style VQA schedule: 1/4, 2/4, 3/4 plateaus, 1, then geometric decay every `decay_interval`."""
    for quarter in (1, 2, 3):
        if step < quarter * warmup_interval:
            return quarter / 4
    if step >= decay_start:
        return decay_rate ** ceil((step - decay_start) / decay_interval)
    return 1


def get_lr_sched(global_step, opts):
    """opts.learning_rate x warmup_linear, floored at 1e-8 once the decay reaches zero (optim/sched.py:40-46)."""
    lr_this_step = opts.learning_rate * warmup_linear(global_step, opts.warmup_steps, opts.num_train_steps)
    return lr_this_step if lr_this_step > 0 else 1e-8
