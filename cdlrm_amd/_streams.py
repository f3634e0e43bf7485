"""HIP stream / event handles for the engine.  On a HIP device these are torch.cuda streams and events; on the
host (the gloo multi-rank orchestration tests, where the kernels are replaced by a test double) they are no-ops."""
from __future__ import annotations

import contextlib

import torch


class _NullStream:
    def wait_stream(self, other):
        pass

    def wait_event(self, ev):
        pass

    def synchronize(self):
        pass


class _NullEvent:
    def record(self, stream=None):
        pass

    def synchronize(self):
        pass


def is_hip(device) -> bool:
    return torch.device(device).type == "cuda"


def new_stream(device, priority=None):
    """A side stream (default priority: the trainer's own queue is created at high priority, see main_no_ddp.Run)."""
    if not is_hip(device):
        return _NullStream()
    return torch.cuda.Stream(device=device) if priority is None else torch.cuda.Stream(device=device, priority=priority)


def current_stream(device):
    return torch.cuda.current_stream(device) if is_hip(device) else _NullStream()


def new_event(device, timing=False):
    return torch.cuda.Event(enable_timing=timing) if is_hip(device) else _NullEvent()


def on_stream(stream):
    return torch.cuda.stream(stream) if isinstance(stream, torch.cuda.Stream) else contextlib.nullcontext()


def pinned(t: torch.Tensor, device) -> torch.Tensor:
    return t.pin_memory() if is_hip(device) else t
