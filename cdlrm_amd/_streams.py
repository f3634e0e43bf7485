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


_BACKGROUND = {}        # device index -> the process's background stream (created once, never destroyed)


def background_stream(device):
    """The look-ahead plan's stream: the LEAST urgent priority the device offers (torch's own stream pool only hands out
    the default and the high level), created through the library and wrapped as a torch stream.  Where the plan's scans
    and the training step compete for CUs the step goes first.  One per device and process, kept for the life of the
    process: the caching allocator may hold events on it (record_stream) long after a pipeline is gone."""
    if not is_hip(device):
        return _NullStream()
    from . import _lib
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _BACKGROUND.get(idx)
    if st is None:
        with torch.cuda.device(idx):
            h = _lib.raw().cdlrm_stream_create(1 << 20)
        if not h:
            raise _lib.CdlrmError(-22, _lib.raw().cdlrm_last_error().decode("utf-8", "replace"))
        st = torch.cuda.ExternalStream(int(h), device=torch.device("cuda", idx))
        _BACKGROUND[idx] = st
    return st


_LOW = {}               # device index -> the process's second least-urgent stream (created once, never destroyed)


def low_priority_stream(device):
    """Another stream of the least urgent priority, distinct from background_stream's (the engine's chunk sort: work whose
    result is needed many steps later must not queue behind a plan's DMA copies, nor in front of anything a step waits for).
    One per device and process, like background_stream's."""
    if not is_hip(device):
        return _NullStream()
    from . import _lib
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _LOW.get(idx)
    if st is None:
        with torch.cuda.device(idx):
            h = _lib.raw().cdlrm_stream_create(1 << 20)
        if not h:
            raise _lib.CdlrmError(-22, _lib.raw().cdlrm_last_error().decode("utf-8", "replace"))
        st = torch.cuda.ExternalStream(int(h), device=torch.device("cuda", idx))
        _LOW[idx] = st
    return st


def current_stream(device):
    return torch.cuda.current_stream(device) if is_hip(device) else _NullStream()


def new_event(device, timing=False):
    return torch.cuda.Event(enable_timing=timing) if is_hip(device) else _NullEvent()


def on_stream(stream):
    return torch.cuda.stream(stream) if isinstance(stream, torch.cuda.Stream) else contextlib.nullcontext()


def pinned(t: torch.Tensor, device) -> torch.Tensor:
    return t.pin_memory() if is_hip(device) else t
