"""uniter_amd — MI355X (gfx950) native UNITER encoder training path.

Drop-in for the reference's ``model`` / ``optim`` / ``utils.distributed`` Python surface
(ChenRocks/UNITER); the arithmetic runs in hand-written HIP kernels behind the C ABI of
``include/uniter_hip.h`` (``uniter_amd/csrc``).
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"
