"""Stage ranges for rocprofv3's marker trace ("roctx ranges per stage").

The reference times epochs and iterations with `time.time()` (train_epoch.py:27-31,95-104) and has no profiler hooks.
Here every stage of NlosPose.forward (models/NlosPose.py:63-76), the losses, the backward of each stage and the optimizer
step can be bracketed by a named range that `rocprofv3 --kernel-trace --marker-trace --stats` reports next to the kernels:

    HP_ROCTX=1 rocprofv3 --kernel-trace --marker-trace --stats -d out -- python3 bench.py --steps 3

Off by default: `stage()` is then a no-op context manager and `mark_backward()` returns its argument untouched (no
autograd node is added).  Ranges are host-side brackets around the ENQUEUE of a stage's kernels; the kernels themselves
are in the kernel trace, correlated by the profiler.
"""
from __future__ import annotations

import os

import torch

from . import _lib

_live = None          # None: not decided yet (HP_ROCTX is read on first use); True / False afterwards
_bwd_open = 0         # id of the backward-stage range that is open, 0 if none
_bwd_cb_queued = False


def enable(on: bool = True) -> bool:
    """Switch ranges on / off; True if they are live (False when the box has no marker library)."""
    global _live
    _live = bool(_lib.lib().hp_range_enable(1 if on else 0))
    return _live


def live() -> bool:
    global _live
    if _live is None:
        _live = enable(True) if os.environ.get("HP_ROCTX", "0") not in ("", "0") else False
    return _live


class stage:
    """`with stage("lct"):` -- a named range on the calling thread; free when ranges are off."""
    __slots__ = ("name", "on")

    def __init__(self, name: str):
        self.name = name
        self.on = False

    def __enter__(self):
        if live():
            self.on = _lib.lib().hp_range_push(self.name.encode()) > 0
        return self

    def __exit__(self, *exc):
        if self.on:
            _lib.lib().hp_range_pop()
        return False


def _bwd_close():
    global _bwd_open, _bwd_cb_queued
    if _bwd_open:
        _lib.lib().hp_range_stop(_bwd_open)
    _bwd_open = 0
    _bwd_cb_queued = False


class _BackwardMark(torch.autograd.Function):
    """Identity on a stage's OUTPUT: its backward runs right before that stage's backward nodes, closes the range of the
    stage behind it and opens `bwd:<name>`.  The last range of a pass is closed by an end-of-backward callback (start / stop
    ranges, not push / pop: autograd runs nodes on its device thread, the callback on the thread that called backward)."""

    @staticmethod
    def forward(ctx, x, name):
        ctx.name = name
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        global _bwd_open, _bwd_cb_queued
        L = _lib.lib()
        if _bwd_open:
            L.hp_range_stop(_bwd_open)
        _bwd_open = int(L.hp_range_start(("bwd:" + ctx.name).encode()))
        if not _bwd_cb_queued:
            _bwd_cb_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(_bwd_close)
        return g, None


def mark_backward(x, name: str):
    """Tag the output of stage `name` so that its backward shows up as `bwd:<name>`; identity when ranges are off."""
    if live() and torch.is_tensor(x) and x.requires_grad and torch.is_grad_enabled():
        return _BackwardMark.apply(x, name)
    return x
