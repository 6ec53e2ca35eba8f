"""Drop-in counterparts of the reference's ``optim`` package (optim/__init__.py:1-7)."""
from .adamw import AdamW, clip_grad_norm_, overlap_boundaries  # noqa: F401
from .misc import build_optimizer, build_vqa_optimizer  # noqa: F401
from .sched import get_lr_sched, noam_schedule, vqa_schedule, warmup_linear  # noqa: F401
