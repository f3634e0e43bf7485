"""Thin functional layer over the C ABI (include/cdlrm_hip.h) on torch tensors.

Torch is plumbing here (device memory, streams); every computation is a libcdlrm_hip.so call.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import Geometry, Plan, check, ptr, stream_ptr
from ._lib import Victims as _VictimsStruct


def _require_cuda(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError("cdlrm_amd: %s must live on the MI355X (got %s); there is no CPU path" % (name, t.device))


class CacheCtx:
    """Opaque library context of one cache group (geometry + bound cache state)."""

    def __init__(self, table_rows: Sequence[int], cache_sets: Sequence[int], dim: int, num_ways: int,
                 aux_rows: int, device: torch.device, aux_phases: int = 1):
        self.T = len(table_rows)
        self.D, self.ways, self.aux = int(dim), int(num_ways), int(aux_rows)
        self.aux_phases = max(1, int(aux_phases))
        self.table_rows = [int(x) for x in table_rows]
        self.cache_sets = [int(x) for x in cache_sets]
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("cdlrm_amd: a HIP device is required (no CPU fallback)")
        _lib.require_gpu("cdlrm_amd.ops.CacheCtx")
        _lib.warn_hw_queues()
        self.rows = [self.ways * p + self.aux * self.aux_phases for p in self.cache_sets]
        self.row_base, self.tag_base, self.set_base = [0], [0], [0]
        for k in range(self.T):
            self.row_base.append(self.row_base[-1] + self.rows[k])
            self.tag_base.append(self.tag_base[-1] + self.cache_sets[k] * self.ways)
            self.set_base.append(self.set_base[-1] + self.cache_sets[k])
        self.total_rows, self.total_tags, self.total_sets = self.row_base[-1], self.tag_base[-1], self.set_base[-1]
        self.bm_words = [((n + 63) // 64 + 1023) // 1024 * 1024 for n in self.table_rows]
        self.total_bm_words = sum(self.bm_words)
        tr = (C.c_int64 * self.T)(*self.table_rows)
        cs = (C.c_int64 * self.T)(*self.cache_sets)
        geo = Geometry(self.T, self.D, self.ways, self.aux, tr, cs, self.device.index or 0, self.aux_phases)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(_lib.lib().cdlrm_ctx_create(C.byref(geo), C.byref(h)))
        self.handle = h
        self._host_ptrs = None
        self._keep = None

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                # straight to the library: the garbage collector may run this while a step is being recorded
                _lib.raw().cdlrm_ctx_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def bind_cache(self, tags: torch.Tensor, weight: torch.Tensor):
        _require_cuda(tags, "tags"); _require_cuda(weight, "weight")
        assert tags.dtype == torch.int64 and tags.numel() == self.total_tags and tags.is_contiguous()
        assert weight.dtype == torch.float32 and tuple(weight.shape) == (self.total_rows, self.D) and weight.is_contiguous()
        check(_lib.lib().cdlrm_ctx_bind_cache(self.handle, tags.data_ptr(), weight.data_ptr()))
        self._keep = (tags, weight)

    def bind_host_tables(self, ptrs: Sequence[int]):
        ptrs = [int(p) for p in ptrs]
        if ptrs == self._host_ptrs:
            return
        arr = (C.c_void_p * self.T)(*ptrs)
        check(_lib.lib().cdlrm_ctx_bind_host_tables(self.handle, arr))
        self._host_ptrs = ptrs

    def bind_victims(self, victims: Optional["Victims"]):
        """Serve the aux-row fill of the per-iteration probe from the window's HBM-resident victim rows (None:
        from the host tables again)."""
        check(_lib.lib().cdlrm_ctx_bind_victims(self.handle, C.byref(victims.c) if victims is not None else None))
        self._victims = victims

    def check(self, stream=None):
        """Raise if a kernel flagged out-of-range input (synchronises the stream)."""
        check(_lib.lib().cdlrm_ctx_check_sync(self.handle, stream_ptr(stream)))


# ---- per-iteration path --------------------------------------------------------------------------

def embbag_probe(ctx: CacheCtx, idx: torch.Tensor, stream=None, aux_phase: int = 0, out=None):
    """idx int64 [T, n] on device -> (slots int32 [T, n], miss_pos int32 [T, n], miss_count int32 [T]).
    aux_phase: which of the ctx's aux regions receives the misses' rows (double-buffered aux, see engine.py)."""
    _require_cuda(idx, "idx")
    assert idx.dtype == torch.int64 and idx.dim() == 2 and idx.shape[0] == ctx.T and idx.stride(1) == 1
    n = idx.shape[1]
    if out is not None:
        slots, miss_pos, miss_count = out
        assert slots.shape == (ctx.T, n) and slots.dtype == torch.int32 and slots.is_contiguous()
        assert miss_pos.shape == (ctx.T, n) and miss_pos.dtype == torch.int32 and miss_pos.is_contiguous()
        assert miss_count.numel() == ctx.T and miss_count.dtype == torch.int32
    else:
        slots = torch.empty((ctx.T, n), dtype=torch.int32, device=idx.device)
        miss_pos = torch.empty((ctx.T, n), dtype=torch.int32, device=idx.device)
        miss_count = torch.empty((ctx.T,), dtype=torch.int32, device=idx.device)
    check(_lib.lib().cdlrm_embbag_probe(ctx.handle, idx.data_ptr(), n, idx.stride(0) if n else 0, slots.data_ptr(),
                                        miss_pos.data_ptr(), miss_count.data_ptr(), int(aux_phase),
                                        stream_ptr(stream)))
    return slots, miss_pos, miss_count


def window_resolve(ctx: CacheCtx, idx: torch.Tensor, seg_len: int, wslots: torch.Tensor, wsrc: torch.Tensor, stream=None,
                   batch_len: int = 0):
    """Resolve every lookup of idx [T, n] (consecutive global batches of batch_len lookups, each cut into rank slices of
    seg_len; batch_len 0 = one run of seg_len-long segments) against the CURRENT tags and the bound victim list, once:
    wslots / wsrc int32 [T, n] contiguous (see include/cdlrm_hip.h)."""
    _require_cuda(idx, "idx")
    assert idx.dtype == torch.int64 and idx.dim() == 2 and idx.shape[0] == ctx.T and idx.stride(1) == 1
    n = idx.shape[1]
    for w in (wslots, wsrc):
        assert w.dtype == torch.int32 and tuple(w.shape) == (ctx.T, n) and w.is_contiguous()
    check(_lib.lib().cdlrm_window_resolve(ctx.handle, idx.data_ptr(), n, idx.stride(0), int(batch_len), int(seg_len), wslots.data_ptr(),
                                          wsrc.data_ptr(), stream_ptr(stream)))


def embbag_take(ctx: CacheCtx, idx: torch.Tensor, wslots: torch.Tensor, wsrc: torch.Tensor, slots_out: torch.Tensor,
                aux_phase: int = 0, stream=None):
    """One batch out of a resolved window: idx / wslots / wsrc are [T, n] column VIEWS of the window's tensors."""
    n = idx.shape[1]
    assert idx.dtype == torch.int64 and idx.stride(1) == 1 and wslots.stride(1) == 1 and wsrc.stride(1) == 1
    assert wslots.shape == (ctx.T, n) and wsrc.shape == (ctx.T, n) and wslots.stride(0) == wsrc.stride(0)
    assert slots_out.dtype == torch.int32 and slots_out.shape == (ctx.T, n) and slots_out.is_contiguous()
    check(_lib.lib().cdlrm_embbag_take(ctx.handle, idx.data_ptr(), n, idx.stride(0), wslots.data_ptr(), wsrc.data_ptr(),
                                       wslots.stride(0), slots_out.data_ptr(), int(aux_phase), stream_ptr(stream)))


def victim_writeback_work(ctx: CacheCtx, n: int) -> torch.Tensor:
    nbytes = int(_lib.lib().cdlrm_victim_writeback_work_bytes(ctx.T, n))
    return torch.empty((nbytes + 255) // 256 * 256, dtype=torch.uint8, device=ctx.device)


def victim_writeback(ctx: CacheCtx, idx: torch.Tensor, slots: torch.Tensor, wsrc: Optional[torch.Tensor], aux_phase: int,
                     work: torch.Tensor, stream=None):
    """--evict-victim-cache: the batch's trained aux rows (its misses) back to their host rows and to their copies among the
    window's resident victim rows; behind the step's embedding update, in front of the next batch's take."""
    n = idx.shape[1]
    assert idx.dtype == torch.int64 and idx.stride(1) == 1 and slots.dtype == torch.int32 and slots.shape == (ctx.T, n) \
        and slots.is_contiguous()
    check(_lib.lib().cdlrm_victim_writeback(ctx.handle, idx.data_ptr(), n, idx.stride(0), slots.data_ptr(),
                                            0 if wsrc is None else wsrc.data_ptr(), 0 if wsrc is None else wsrc.stride(0),
                                            int(aux_phase), work.data_ptr(), stream_ptr(stream)))


class TimingEvent:
    """A HIP timing event of the library's own (its handle exists from creation on, so a launch tape can hold it)."""

    def __init__(self):
        h = _lib.raw().cdlrm_event_create(1)
        if not h:
            raise _lib.CdlrmError(-22, _lib.raw().cdlrm_last_error().decode("utf-8", "replace"))
        self.handle = int(h)

    def elapsed_us(self, stop: "TimingEvent") -> float:
        """Microseconds from this event to `stop` (waits for `stop`)."""
        us = C.c_float(0.0)
        check(_lib.raw().cdlrm_event_elapsed_us(self.handle, stop.handle, C.byref(us)))
        return float(us.value)

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                _lib.raw().cdlrm_event_destroy(h)
            except Exception:
                pass


def event_attach_next(event: "torch.cuda.Event", stream):
    """`event` completes with the next linear_bwd / interact_bwd launched on `stream` (attached to the launch: no marker
    packet on the queue).  The event must have been recorded once before (torch creates the HIP event lazily)."""
    h = event.cuda_event
    assert h, "record the event once before attaching it"
    check(_lib.lib().cdlrm_event_attach_next(int(h), stream.cuda_stream))


def event_record(event: "torch.cuda.Event", stream):
    """event.record(stream) as a library call (the handle is an argument a launch tape can keep in a cell)."""
    h = event.cuda_event
    assert h, "record the event once through torch first (it creates the HIP event lazily)"
    check(_lib.lib().cdlrm_event_record(int(h), stream.cuda_stream))


def time_next_gather(ctx: CacheCtx, start: TimingEvent, stop: TimingEvent):
    """The next embbag_fwd on this context leaves its own start / stop timestamps in the two events (attached to the
    launch: nothing is added to the queue)."""
    check(_lib.lib().cdlrm_ctx_time_next_gather(ctx.handle, start.handle, stop.handle))


def embbag_fwd(ctx: CacheCtx, slots: torch.Tensor, offsets: Optional[torch.Tensor], out: torch.Tensor,
               ld_bag: int, ld_table: int, n_bags: Optional[int] = None, stream=None):
    n = slots.shape[1]
    nb = n if offsets is None else offsets.shape[1]
    if n_bags is not None:
        nb = n_bags
    check(_lib.lib().cdlrm_embbag_fwd(ctx.handle, slots.data_ptr(), ptr(offsets), n, nb,
                                      0 if offsets is None else offsets.stride(0), out.data_ptr(), ld_bag, ld_table,
                                      stream_ptr(stream)))


def embbag_bwd_work(ctx: CacheCtx, n: int, device) -> torch.Tensor:
    """(zero-filled: embbag_bwd_apply_sorted uses the buffer's scratch without a prepare in front)"""
    nbytes = int(_lib.lib().cdlrm_embbag_bwd_work_bytes(ctx.T, n, ctx.D))
    work = torch.zeros((nbytes + 255) // 256 * 256, dtype=torch.uint8, device=device)
    # the fill runs on the CURRENT stream, the buffer's first user (a prepare) on a side stream of the caller's: it has to have
    # landed before anything else touches the buffer (once per buffer)
    torch.cuda.current_stream(device).synchronize()
    return work


def embbag_bwd_sgd(ctx: CacheCtx, slots: torch.Tensor, offsets: Optional[torch.Tensor], grad: torch.Tensor,
                   ld_bag: int, ld_table: int, lr: float, work: torch.Tensor, touched: Optional[torch.Tensor] = None,
                   stream=None):
    n = slots.shape[1]
    nb = n if offsets is None else offsets.shape[1]
    check(_lib.lib().cdlrm_embbag_bwd_sgd(ctx.handle, slots.data_ptr(), ptr(offsets), n, nb,
                                          0 if offsets is None else offsets.stride(0), grad.data_ptr(), ld_bag,
                                          ld_table, float(lr), work.data_ptr(), ptr(touched), stream_ptr(stream)))


def embbag_bwd_prepare(ctx: CacheCtx, slots: torch.Tensor, work: torch.Tensor, stream=None):
    check(_lib.lib().cdlrm_embbag_bwd_prepare(ctx.handle, slots.data_ptr(), slots.shape[1], work.data_ptr(),
                                              stream_ptr(stream)))


def embbag_bwd_apply(ctx: CacheCtx, n: int, offsets: Optional[torch.Tensor], grad: torch.Tensor, ld_bag: int,
                     ld_table: int, lr: float, work: torch.Tensor, touched: Optional[torch.Tensor] = None, stream=None):
    nb = n if offsets is None else offsets.shape[1]
    check(_lib.lib().cdlrm_embbag_bwd_apply(ctx.handle, ptr(offsets), n, nb, 0 if offsets is None else offsets.stride(0),
                                            grad.data_ptr(), ld_bag, ld_table, float(lr), work.data_ptr(), ptr(touched),
                                            stream_ptr(stream)))


def embbag_bwd_apply_rest(ctx: CacheCtx, n: int, offsets: Optional[torch.Tensor], grad: torch.Tensor, ld_bag: int,
                          ld_table: int, lr: float, work: torch.Tensor, touched: Optional[torch.Tensor] = None, stream=None):
    """embbag_bwd_apply behind gather_interact_bwd_sgd: the once-only slots are done, the runs of >= 2 lookups are left."""
    nb = n if offsets is None else offsets.shape[1]
    check(_lib.lib().cdlrm_embbag_bwd_apply_rest(ctx.handle, ptr(offsets), n, nb, 0 if offsets is None else offsets.stride(0),
                                                 grad.data_ptr(), ld_bag, ld_table, float(lr), work.data_ptr(), ptr(touched),
                                                 stream_ptr(stream)))


def embbag_bwd_once_flags(ctx: CacheCtx, work: torch.Tensor, n: int) -> int:
    """Address of the once-only flags (uint8 [T, n]) a prepare left in `work`."""
    import ctypes as C
    out = C.c_void_p()
    check(_lib.lib().cdlrm_embbag_bwd_once_flags(ctx.handle, work.data_ptr(), n, C.byref(out)))
    return int(out.value)


def embbag_bwd_sorted(ctx: CacheCtx, nb: int, n: int, device) -> torch.Tensor:
    """Caller-owned buffer for the sorted slot lists of a look-ahead chunk of nb batches of n lookups per table."""
    nbytes = int(_lib.lib().cdlrm_embbag_bwd_sorted_bytes(ctx.T, nb, n))
    # zero-filled: lists nobody has sorted yet are still lists of valid (slot 0, position 0) keys
    buf = torch.zeros((nbytes + 255) // 256 * 256, dtype=torch.uint8, device=device)
    torch.cuda.current_stream(device).synchronize()
    return buf


def embbag_bwd_prepare_window(ctx: CacheCtx, wslots: torch.Tensor, batch_len: int, nb: int, n: int, sorted_buf: torch.Tensor,
                              stream=None, j0: int = 0, count: Optional[int] = None):
    """Sort the slot ids of batches [j0, j0 + count) of a chunk of nb (default: all of them), every table, at once: wslots int32
    [T, >= (nb - 1) * batch_len + n] (the resolver's chunk, or a view starting at this rank's first column)."""
    assert wslots.dtype == torch.int32 and wslots.stride(1) == 1
    check(_lib.lib().cdlrm_embbag_bwd_prepare_window(ctx.handle, wslots.data_ptr(), wslots.stride(0), batch_len, nb, n, int(j0),
                                                     int(nb - j0 if count is None else count), sorted_buf.data_ptr(),
                                                     stream_ptr(stream)))


def embbag_bwd_sorted_views(ctx: CacheCtx, sorted_buf: torch.Tensor, nb: int, n: int, j: int):
    """(keys, meta, once) addresses of batch j inside a sorted chunk; table t's lists lie t * nb * n elements further."""
    import ctypes as C
    k, m, o = C.c_void_p(), C.c_void_p(), C.c_void_p()
    check(_lib.lib().cdlrm_embbag_bwd_sorted_views(ctx.handle, sorted_buf.data_ptr(), nb, n, j, C.byref(k), C.byref(m), C.byref(o)))
    return int(k.value), int(m.value), int(o.value)


def embbag_bwd_apply_sorted(ctx: CacheCtx, n: int, grad: torch.Tensor, ld_bag: int, ld_table: int, lr: float, work: torch.Tensor,
                            keys: int, meta: int, tstride: int, aux_phase: int, rest: bool,
                            touched: Optional[torch.Tensor] = None, stream=None):
    """embbag_bwd_apply / _apply_rest over a sorted chunk's lists (addresses from embbag_bwd_sorted_views)."""
    check(_lib.lib().cdlrm_embbag_bwd_apply_sorted(ctx.handle, n, grad.data_ptr(), ld_bag, ld_table, float(lr), work.data_ptr(),
                                                   keys, meta, tstride, int(aux_phase), int(bool(rest)), ptr(touched),
                                                   stream_ptr(stream)))


# ---- look-ahead window plan -------------------------------------------------------------------------

class WindowPlan:
    """Caller-owned buffers of one insert plan (cdlrm_plan in the header) + the call sequence."""

    def __init__(self, ctx: CacheCtx, max_window: int, cap_uniq: Optional[int] = None, cap_win: Optional[int] = None):
        dev = ctx.device
        self.ctx = ctx
        per_table = [min(int(max_window), n) for n in ctx.table_rows]
        self.cap_uniq = int(cap_uniq) if cap_uniq is not None else max(16, sum(per_table))
        slots_cap = [min(u, ctx.ways * p) for u, p in zip(per_table, ctx.cache_sets)]
        self.cap_win = int(cap_win) if cap_win is not None else max(16, sum(slots_cap))
        i64, i32, u8 = torch.int64, torch.int32, torch.uint8
        z = lambda n, dt: torch.zeros(max(int(n), 1), dtype=dt, device=dev)
        e = lambda n, dt: torch.empty(max(int(n), 1), dtype=dt, device=dev)
        T = ctx.T
        self.bitmap = z(ctx.total_bm_words, i64)
        self.uniq = e(self.cap_uniq, i64)
        self.uniq_off = z(T + 1, i64)
        self.prot = z(ctx.total_sets, i64)
        self.hit = e((self.cap_uniq + 15) // 16 * 16, u8)
        self.kept = e(self.cap_uniq, i32)
        self.kept_off = z(T + 1, i64)
        self.way = e((self.cap_uniq + 15) // 16 * 16, u8)
        self.flags = z((self.cap_uniq + 15) // 16 * 16, u8)
        self.winner = torch.full((ctx.total_rows,), -1, dtype=i32, device=dev)
        self.win_claim = e(self.cap_win, i32)
        self.win_idx = e(self.cap_win, i64)
        self.win_row = e(self.cap_win, i64)
        self.win_tag = e(self.cap_win, i64)
        self.win_off = z(T + 1, i64)
        self.stage = torch.empty((self.cap_win, ctx.D), dtype=torch.float32, device=dev)
        self.ev_tag = torch.full((self.cap_win,), -1, dtype=i64, device=dev)
        self.c = Plan(self.bitmap.data_ptr(), self.uniq.data_ptr(), self.uniq_off.data_ptr(), self.cap_uniq,
                      self.prot.data_ptr(), self.hit.data_ptr(), self.kept.data_ptr(), self.kept_off.data_ptr(),
                      self.way.data_ptr(), self.flags.data_ptr(), self.winner.data_ptr(), self.win_claim.data_ptr(),
                      self.win_idx.data_ptr(), self.win_row.data_ptr(), self.win_tag.data_ptr(),
                      self.win_off.data_ptr(), self.cap_win, self.stage.data_ptr(), self.ev_tag.data_ptr())

    # K1
    def unique(self, idx: torch.Tensor, stream=None):
        _require_cuda(idx, "window indices")
        assert idx.dtype == torch.int64 and idx.dim() == 2 and idx.shape[0] == self.ctx.T and idx.stride(1) == 1
        check(_lib.lib().cdlrm_window_unique(self.ctx.handle, C.byref(self.c), idx.data_ptr(), idx.shape[1],
                                             idx.stride(0), stream_ptr(stream)))

    def unique_add(self, idx: torch.Tensor, stream=None):
        """Streamed K1: fold one chunk [T, n] of the window into the bitmap (any number of chunks, then unique_finish)."""
        _require_cuda(idx, "window indices")
        assert idx.dtype == torch.int64 and idx.dim() == 2 and idx.shape[0] == self.ctx.T and idx.stride(1) == 1
        check(_lib.lib().cdlrm_window_unique_add(self.ctx.handle, C.byref(self.c), idx.data_ptr(), idx.shape[1],
                                                 idx.stride(0), stream_ptr(stream)))

    def unique_finish(self, stream=None):
        check(_lib.lib().cdlrm_window_unique_finish(self.ctx.handle, C.byref(self.c), stream_ptr(stream)))

    def set_unique(self, uniqs: Sequence[torch.Tensor]):
        """CacheEmbeddings drop-in entry: the caller already has the sorted unique lists."""
        off = [0]
        for u in uniqs:
            off.append(off[-1] + int(u.numel()))
        if off[-1] > self.cap_uniq:
            raise RuntimeError("plan capacity %d < %d unique indices" % (self.cap_uniq, off[-1]))
        if off[-1]:
            self.uniq[:off[-1]] = torch.cat([u.reshape(-1).to(self.uniq.device, torch.int64) for u in uniqs])
        self.uniq_off.copy_(torch.tensor(off, dtype=torch.int64))

    # K2
    def probe(self, stream=None):
        check(_lib.lib().cdlrm_plan_probe(self.ctx.handle, C.byref(self.c), stream_ptr(stream)))

    def offsets(self, stream=None):
        """(uniq_off, kept_off, win_off) as python lists; synchronises the stream."""
        T = self.ctx.T
        bufs = [(C.c_int64 * (T + 1))() for _ in range(3)]
        check(_lib.lib().cdlrm_plan_offsets_sync(self.ctx.handle, C.byref(self.c), bufs[0], bufs[1], bufs[2],
                                                 stream_ptr(stream)))
        return [list(b) for b in bufs]

    # K3
    def assign(self, q: Optional[torch.Tensor] = None, seed: int = 0, stream=None):
        if q is not None:
            _require_cuda(q, "q")
            assert q.dtype == torch.float32 and q.is_contiguous()
        check(_lib.lib().cdlrm_plan_assign(self.ctx.handle, C.byref(self.c), ptr(q), int(seed) & (2 ** 64 - 1),
                                           stream_ptr(stream)))

    # K5a
    def fetch(self, src_ptrs: Sequence[int], by_position: bool, stream=None):
        arr = (C.c_void_p * self.ctx.T)(*[int(p) for p in src_ptrs])
        check(_lib.lib().cdlrm_plan_fetch(self.ctx.handle, C.byref(self.c), arr, 1 if by_position else 0,
                                          stream_ptr(stream)))

    def victims(self, victims: "Victims", stream=None, list_only: bool = False):
        """After assign(): list the window's indices that stay outside the cache and fetch their host rows
        (list_only: the caller moves the rows itself -- host_gather_rows + one DMA copy)."""
        fn = _lib.lib().cdlrm_plan_victims_list if list_only else _lib.lib().cdlrm_plan_victims
        check(fn(self.ctx.handle, C.byref(self.c), C.byref(victims.c), stream_ptr(stream)))

    # K4 + K5b
    def commit(self, stream=None):
        check(_lib.lib().cdlrm_plan_commit(self.ctx.handle, C.byref(self.c), stream_ptr(stream)))

    # K14
    def writeback(self, dst_ptrs: Sequence[int], average: bool, stream=None):
        arr = (C.c_void_p * self.ctx.T)(*[int(p) for p in dst_ptrs])
        check(_lib.lib().cdlrm_plan_writeback(self.ctx.handle, C.byref(self.c), arr, 1 if average else 0,
                                              stream_ptr(stream)))


class Victims:
    """Caller-owned buffers of one window's victim rows (cdlrm_victims in the header)."""

    def __init__(self, ctx: CacheCtx, cap: int):
        dev = ctx.device
        self.cap = max(1, int(cap))
        self.pos = torch.empty(self.cap, dtype=torch.int32, device=dev)
        self.idx = torch.zeros(self.cap, dtype=torch.int64, device=dev)
        self.off = torch.zeros(ctx.T + 2, dtype=torch.int64, device=dev)     # [T + 1] offsets + the un-capped list length
        self.rows = torch.empty((self.cap, ctx.D), dtype=torch.float32, device=dev)
        self.c = _VictimsStruct(self.pos.data_ptr(), self.idx.data_ptr(), self.off.data_ptr(), self.rows.data_ptr(),
                                self.cap)


def host_gather_rows(table_ptrs: Sequence[int], idx: torch.Tensor, off: Sequence[int], dim: int, dst: torch.Tensor,
                     threads: int = 16):
    """HOST-side gather (CPU threads, releases the GIL): dst[j] = table_{t(j)}[idx[j]] for j < off[-1].  idx: CPU int64,
    dst: CPU fp32 [>= off[-1], dim] (pinned for the DMA that follows)."""
    T = len(table_ptrs)
    assert idx.device.type == "cpu" and idx.dtype == torch.int64 and idx.is_contiguous()
    assert dst.device.type == "cpu" and dst.dtype == torch.float32 and dst.is_contiguous() and dst.shape[1] == dim
    assert len(off) == T + 1 and off[-1] <= idx.numel() and off[-1] <= dst.shape[0]
    tp = (C.c_void_p * T)(*[int(p) for p in table_ptrs])
    oa = (C.c_int64 * (T + 1))(*[int(o) for o in off])
    check(_lib.lib().cdlrm_host_gather_rows(tp, idx.data_ptr(), oa, T, int(dim), dst.data_ptr(), int(threads)))


def gather_rows(src_ptr: int, index: torch.Tensor, dim: int, stream=None) -> torch.Tensor:
    _require_cuda(index, "index")
    out = torch.empty((index.numel(), dim), dtype=torch.float32, device=index.device)
    check(_lib.lib().cdlrm_gather_rows(int(src_ptr), index.data_ptr(), index.numel(), dim, out.data_ptr(),
                                       stream_ptr(stream)))
    return out


# ---- table aggregation -------------------------------------------------------------------------------

def agg_compact(ctx: CacheCtx, touched: torch.Tensor, rows_out: torch.Tensor, count_out: torch.Tensor, stream=None):
    check(_lib.lib().cdlrm_agg_compact(ctx.handle, touched.data_ptr(), touched.numel(), rows_out.data_ptr(),
                                       rows_out.numel(), count_out.data_ptr(), stream_ptr(stream)))


def agg_gather(ctx: CacheCtx, rows: torch.Tensor, count: torch.Tensor, scale: float, buf: torch.Tensor, cap: int,
               stream=None, first: int = 0):
    """buf[i] = weight[rows[i]] / scale for i < min(count - first, cap): `rows` points at entry `first` of the list
    whose length is the device word `count` (chunked merges pass slices)."""
    check(_lib.lib().cdlrm_agg_gather(ctx.handle, rows.data_ptr(), count.data_ptr(), float(scale), buf.data_ptr(),
                                      int(cap), int(first), stream_ptr(stream)))


def agg_scatter(ctx: CacheCtx, rows: torch.Tensor, count: torch.Tensor, buf: torch.Tensor, cap: int, stream=None,
                first: int = 0):
    check(_lib.lib().cdlrm_agg_scatter(ctx.handle, rows.data_ptr(), count.data_ptr(), buf.data_ptr(), int(cap),
                                       int(first), stream_ptr(stream)))


def agg_mark_tier(ctx: CacheCtx, slots: torch.Tensor, value: int, tier: torch.Tensor, stream=None):
    """tier[row] = value for every cache slot the [T, n] view of resolved slot ids names (cdlrm_agg_mark_tier)."""
    assert slots.dtype == torch.int32 and slots.dim() == 2 and slots.shape[0] == ctx.T and slots.stride(1) == 1
    assert tier.dtype == torch.uint8 and tier.numel() == ctx.total_rows
    check(_lib.lib().cdlrm_agg_mark_tier(ctx.handle, slots.data_ptr(), slots.shape[1], slots.stride(0), int(value),
                                         tier.data_ptr(), stream_ptr(stream)))


def agg_split(ctx: CacheCtx, rows: torch.Tensor, count: int, tier: torch.Tensor, n_classes: int, rows_out: torch.Tensor,
              class_off: torch.Tensor, stream=None):
    """Stable counting sort of the merge's row list by tier byte (cdlrm_agg_split); class_off: int64 [n_classes + 1] on the device."""
    assert rows.dtype == torch.int64 and rows_out.dtype == torch.int64 and rows_out.numel() >= count
    assert class_off.dtype == torch.int64 and class_off.numel() >= n_classes + 1
    check(_lib.lib().cdlrm_agg_split(ctx.handle, rows.data_ptr(), int(count), tier.data_ptr(), int(n_classes), rows_out.data_ptr(),
                                     class_off.data_ptr(), stream_ptr(stream)))


# ---- dense model -------------------------------------------------------------------------------------

def interact_fwd(feat: torch.Tensor, itself: bool, R: torch.Tensor, stream=None):
    B, F, D = feat.shape
    check(_lib.lib().cdlrm_interact_fwd(feat.data_ptr(), B, F, D, int(bool(itself)), R.data_ptr(), R.stride(0),
                                        stream_ptr(stream)))


def interact_bwd(feat: torch.Tensor, dR: torch.Tensor, itself: bool, dfeat: torch.Tensor, stream=None, x_act: int = 0):
    """x_act: activation that produced feature 0 (the bottom MLP's output); its gradient row then leaves as the
    pre-activation gradient."""
    B, F, D = feat.shape
    check(_lib.lib().cdlrm_interact_bwd(feat.data_ptr(), dR.data_ptr(), dR.stride(0), B, F, D, int(bool(itself)),
                                        int(x_act), dfeat.data_ptr(), stream_ptr(stream)))


def gather_interact_supported(ctx: CacheCtx) -> bool:
    """Shapes the fused gather + interaction kernels take (D in 32 / 64 / 128 / 256, 16 < T + 1 <= 32)."""
    return bool(_lib.raw().cdlrm_gather_interact_supported(ctx.handle))      # (a query: never on a step's tape)


def gather_interact_fwd(ctx: CacheCtx, slots: torch.Tensor, x: torch.Tensor, itself: bool, R: torch.Tensor, stream=None):
    """Cached EmbeddingBag forward (Criteo layout) + dot interaction in one launch: x [B, D] (any row pitch) is feature 0,
    features 1 .. T are the cache rows `slots` [T, n] names.  Bit-identical to embbag_fwd + interact_fwd."""
    B = x.shape[0]
    assert slots.dtype == torch.int32 and slots.stride(1) == 1 and x.stride(1) == 1
    check(_lib.lib().cdlrm_gather_interact_fwd(ctx.handle, slots.data_ptr(), slots.stride(0), x.data_ptr(), x.stride(0), B,
                                               int(bool(itself)), R.data_ptr(), R.stride(0), stream_ptr(stream)))


def gather_interact_bwd(ctx: CacheCtx, slots: torch.Tensor, x: torch.Tensor, dR: torch.Tensor, itself: bool,
                        dfeat: torch.Tensor, stream=None, x_act: int = 0):
    """interact_bwd with the rows read again from the cache (before the batch's embedding update rewrites them)."""
    B = x.shape[0]
    assert slots.dtype == torch.int32 and slots.stride(1) == 1 and x.stride(1) == 1 and dfeat.is_contiguous()
    check(_lib.lib().cdlrm_gather_interact_bwd(ctx.handle, slots.data_ptr(), slots.stride(0), x.data_ptr(), x.stride(0),
                                               dR.data_ptr(), dR.stride(0), B, int(bool(itself)), int(x_act),
                                               dfeat.data_ptr(), stream_ptr(stream)))


def gather_interact_bwd_sgd(ctx: CacheCtx, slots: torch.Tensor, x: torch.Tensor, dR: torch.Tensor, itself: bool,
                            dfeat: torch.Tensor, once: int, ld_once: int, lr: float, stream=None, x_act: int = 0):
    """gather_interact_bwd + the SGD step of the slots the batch reads once (their gradient rows are not written).  once: address
    of the sort's flags (embbag_bwd_once_flags, ld_once = n; embbag_bwd_sorted_views, ld_once = nb * n).  embbag_bwd_apply_rest /
    embbag_bwd_apply_sorted(rest=True) does the other slots."""
    B = x.shape[0]
    assert slots.dtype == torch.int32 and slots.stride(1) == 1 and x.stride(1) == 1 and dfeat.is_contiguous()
    check(_lib.lib().cdlrm_gather_interact_bwd_sgd(ctx.handle, slots.data_ptr(), slots.stride(0), x.data_ptr(), x.stride(0),
                                                   dR.data_ptr(), dR.stride(0), B, int(bool(itself)), int(x_act),
                                                   dfeat.data_ptr(), int(once), int(ld_once), float(lr), stream_ptr(stream)))


ACT = {"none": 0, "relu": 1, "sigmoid": 2}


GEMM_ALONE = 0x100        # include/cdlrm_hip.h: CDLRM_GEMM_ALONE


def linear_fwd(X: torch.Tensor, W: torch.Tensor, b: Optional[torch.Tensor], Y: torch.Tensor, act: int, stream=None,
               alone: bool = False):
    """alone: no other GEMM runs beside this launch (scheduling hint, CDLRM_GEMM_ALONE)."""
    M, K = X.shape
    N = W.shape[0]
    assert W.shape[1] == K and W.is_contiguous() and X.stride(1) == 1 and Y.stride(1) == 1
    check(_lib.lib().cdlrm_linear_fwd(X.data_ptr(), X.stride(0), W.data_ptr(), ptr(b), Y.data_ptr(), Y.stride(0), M, N,
                                      K, act | (GEMM_ALONE if alone else 0), stream_ptr(stream)))


def linear_bwd_work(M: int, N: int, K: int, device) -> torch.Tensor:
    nbytes = int(_lib.lib().cdlrm_linear_bwd_work_bytes(M, N, K))
    return torch.empty((nbytes + 255) // 256 * 256, dtype=torch.uint8, device=device)


def linear_bwd(X, W, Y, dY, dX, dW, db, act: int, work: torch.Tensor, stream=None, x_act: int = 0, alone: bool = False):
    """act: this layer's activation, applied backward to dY in place (0: dY already is the pre-activation
    gradient).  x_act: the activation that produced X; dX then leaves as the layer below's pre-activation gradient."""
    M, K = X.shape
    N = W.shape[0]
    check(_lib.lib().cdlrm_linear_bwd(X.data_ptr(), X.stride(0), W.data_ptr(), ptr(Y), 0 if Y is None else Y.stride(0),
                                      dY.data_ptr(), dY.stride(0), ptr(dX), 0 if dX is None else dX.stride(0),
                                      ptr(dW), ptr(db), M, N, K, act | (GEMM_ALONE if alone else 0), int(x_act), work.data_ptr(),
                                      stream_ptr(stream)))


def mlp_wgrad_work(M: int, Ns: Sequence[int], Ks: Sequence[int], device) -> torch.Tensor:
    """Scratch for mlp_wgrad over layers with output widths Ns and input widths Ks at batch M."""
    n = len(Ns)
    NA = C.c_int32 * n
    nbytes = int(_lib.lib().cdlrm_mlp_wgrad_work_bytes(n, int(M), NA(*[int(x) for x in Ns]), NA(*[int(x) for x in Ks])))
    return torch.empty((nbytes + 255) // 256 * 256, dtype=torch.uint8, device=device)


class WgradPlan:
    """Host-side argument block of cdlrm_mlp_wgrad for a fixed set of layers and buffers (built once per batch shape;
    `set_x` re-points one layer's input, e.g. the dense features of the current batch)."""

    def __init__(self, Xs, dZs, dWs, dbs, work: torch.Tensor):
        import ctypes as C
        n = len(Xs)
        assert len(dZs) == n and len(dWs) == n and len(dbs) == n
        self.n = n
        self.M = int(dZs[0].shape[0])
        self._keep = (list(Xs), list(dZs), list(dWs), list(dbs), work)
        PA, IA, NA = C.c_void_p * n, C.c_int64 * n, C.c_int32 * n
        self.X = PA(*[x.data_ptr() for x in Xs])
        self.ld_x = IA(*[x.stride(0) for x in Xs])
        self.dZ = PA(*[d.data_ptr() for d in dZs])
        self.ld_dz = IA(*[d.stride(0) for d in dZs])
        self.dW = PA(*[w.data_ptr() for w in dWs])
        self.db = PA(*[ptr(b) for b in dbs])
        self.N = NA(*[int(w.shape[0]) for w in dWs])
        self.K = NA(*[int(w.shape[1]) for w in dWs])
        for x, d, w in zip(Xs, dZs, dWs):
            assert x.shape[0] == self.M and d.shape[0] == self.M and x.stride(1) == 1 and d.stride(1) == 1
            assert w.is_contiguous() and d.shape[1] == w.shape[0] and x.shape[1] <= w.shape[1] <= x.stride(0)
        need = int(_lib.lib().cdlrm_mlp_wgrad_work_bytes(n, self.M, self.N, self.K))
        assert work.numel() * work.element_size() >= need and work.data_ptr() % 256 == 0, "work too small: ops.mlp_wgrad_work"
        self.work = work

    def set_x(self, i: int, x: torch.Tensor):
        assert x.shape[0] == self.M and x.stride(1) == 1
        self.X[i] = x.data_ptr()
        self.ld_x[i] = x.stride(0)
        self._keep[0][i] = x

    def set_params(self, Ws, bs):
        """The layers' parameters (W[i] [N, K] contiguous, b[i] [N] or None), for mlp_wgrad(..., lr=): the SGD step then
        happens in the weight-gradient launches."""
        import ctypes as C
        assert len(Ws) == self.n and len(bs) == self.n
        for i, w in enumerate(Ws):
            assert w.is_contiguous() and tuple(w.shape) == (int(self.N[i]), int(self.K[i]))
        PA = C.c_void_p * self.n
        self.P_w = PA(*[w.data_ptr() for w in Ws])
        self.P_b = PA(*[ptr(b) for b in bs])
        self._params = (list(Ws), list(bs))


def mlp_wgrad(plan: WgradPlan, stream=None, lr: Optional[float] = None):
    """dW[i] = dZ[i]^T X[i], db[i] = column sums of dZ[i] for every layer of the plan (one grouped launch at small M).
    lr given (and plan.set_params called): W[i] -= lr * dW[i], b[i] -= lr * db[i] in the same launches."""
    if lr is None:
        check(_lib.lib().cdlrm_mlp_wgrad(plan.n, plan.X, plan.ld_x, plan.dZ, plan.ld_dz, plan.dW, plan.db, plan.M, plan.N,
                                         plan.K, plan.work.data_ptr(), stream_ptr(stream)))
    else:
        check(_lib.lib().cdlrm_mlp_wgrad_sgd(plan.n, plan.X, plan.ld_x, plan.dZ, plan.ld_dz, plan.dW, plan.db, plan.P_w,
                                             plan.P_b, float(lr), plan.M, plan.N, plan.K, plan.work.data_ptr(),
                                             stream_ptr(stream)))


def bce_fwd_bwd(Z: torch.Tensor, target: torch.Tensor, loss_buf: torch.Tensor, dZ: Optional[torch.Tensor], stream=None,
                sigmoid_bwd: bool = False):
    assert loss_buf.numel() >= 65
    check(_lib.lib().cdlrm_bce_fwd_bwd(Z.data_ptr(), target.data_ptr(), Z.numel(), loss_buf.data_ptr(), ptr(dZ),
                                       1 if sigmoid_bwd else 0, stream_ptr(stream)))


LOSS = {"bce": 0, "mse": 1, "wbce": 2}


def loss_fwd_bwd(Z: torch.Tensor, target: torch.Tensor, loss_buf: torch.Tensor, dZ: Optional[torch.Tensor], *,
                 kind: int = 0, weights=(1.0, 1.0), threshold: float = 0.0, Zc: Optional[torch.Tensor] = None,
                 sigmoid_bwd: bool = False, stream=None):
    """loss_fn_wrap (main_no_ddp.py:212-221) with every arm + the --loss-threshold clamp (model_no_ddp.py:311-314)."""
    check(_lib.lib().cdlrm_loss_fwd_bwd(Z.data_ptr(), target.data_ptr(), Z.numel(), int(kind), float(weights[0]),
                                        float(weights[1]), float(threshold), loss_buf.data_ptr(), ptr(dZ), ptr(Zc),
                                        1 if sigmoid_bwd else 0, stream_ptr(stream)))


def head_scratch(device) -> torch.Tensor:
    return torch.zeros(int(_lib.lib().cdlrm_head_scratch_floats()), dtype=torch.float32, device=device)


def head_fwd_bwd(Y: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], target: torch.Tensor, Z: torch.Tensor,
                 dZ: torch.Tensor, dY: Optional[torch.Tensor], loss_buf: torch.Tensor, scratch: torch.Tensor, *,
                 x_act: int = 0, kind: int = 0, weights=(1.0, 1.0), threshold: float = 0.0,
                 Zc: Optional[torch.Tensor] = None, finish: bool = True, stream=None):
    """Last top layer (out_features 1, sigmoid) + loss + the layer's input gradient in one launch.  finish=False leaves
    the loss as partial sums in `scratch`: head_finish() on a stream ordered behind this call completes loss_buf."""
    B, K = Y.shape
    assert Y.stride(1) == 1 and w.numel() >= K and Z.numel() == B and dZ.numel() == B and target.numel() == B
    assert dY is None or (dY.shape == Y.shape and dY.stride(1) == 1)
    check(_lib.lib().cdlrm_head_fwd_bwd(Y.data_ptr(), Y.stride(0), w.data_ptr(), ptr(bias), target.data_ptr(), B, K,
                                        int(kind), float(weights[0]), float(weights[1]), float(threshold), int(x_act),
                                        Z.data_ptr(), ptr(Zc), dZ.data_ptr(), ptr(dY), 0 if dY is None else dY.stride(0),
                                        loss_buf.data_ptr(), scratch.data_ptr(), 1 if finish else 0, stream_ptr(stream)))


def head_finish(scratch: torch.Tensor, B: int, loss_buf: torch.Tensor, stream=None, acc: Optional[torch.Tensor] = None):
    """acc: float64 [2] on the device, += [correct predictions, loss * B] of this batch."""
    assert acc is None or (acc.dtype == torch.float64 and acc.numel() >= 2 and acc.is_contiguous())
    check(_lib.lib().cdlrm_head_finish(scratch.data_ptr(), int(B), loss_buf.data_ptr(), ptr(acc), stream_ptr(stream)))


def act_bwd(dX: torch.Tensor, X: torch.Tensor, act: int, stream=None):
    """dX *= act'(X) in place over a 2-D block (views with row pitches allowed)."""
    assert dX.shape == X.shape and dX.dim() == 2 and dX.stride(1) == 1 and X.stride(1) == 1
    check(_lib.lib().cdlrm_act_bwd(dX.data_ptr(), dX.stride(0), X.data_ptr(), X.stride(0), X.shape[0], X.shape[1],
                                   int(act), stream_ptr(stream)))


def sgd_step2(param: torch.Tensor, grad: torch.Tensor, off0: int, n0: int, off1: int, n1: int, lr: float, stream=None):
    assert param.is_contiguous() and grad.is_contiguous() and param.numel() == grad.numel()
    assert 0 <= off0 and off0 + n0 <= param.numel() and 0 <= off1 and off1 + n1 <= param.numel()
    check(_lib.lib().cdlrm_sgd_step2(param.data_ptr(), grad.data_ptr(), int(off0), int(n0), int(off1), int(n1), float(lr),
                                     stream_ptr(stream)))


def scale_div(x: torch.Tensor, divisor: float, stream=None):
    assert x.is_contiguous()
    check(_lib.lib().cdlrm_scale_div(x.data_ptr(), x.numel(), float(divisor), stream_ptr(stream)))


def scatter_rows(dst_ptr: int, index: torch.Tensor, rows: torch.Tensor, average: bool, stream=None,
                 distinct: bool = True):
    """dst[index[i]] = rows[i] (or the average with the old row).  distinct=False: the list may repeat an index (the
    reference's eviction lists do): the averaging arm then blends every entry against the OLD destination row first
    (cdlrm_blend_rows) and scatters the blended rows -- repeats carry identical rows, so the result is defined."""
    _require_cuda(index, "index"); _require_cuda(rows, "rows")
    assert rows.is_contiguous() and index.is_contiguous() and index.dtype == torch.int64
    if average and not distinct:
        out = torch.empty_like(rows)
        check(_lib.lib().cdlrm_blend_rows(int(dst_ptr), index.data_ptr(), rows.data_ptr(), index.numel(), rows.shape[1],
                                          out.data_ptr(), stream_ptr(stream)))
        rows, average = out, False
    check(_lib.lib().cdlrm_scatter_rows(int(dst_ptr), index.data_ptr(), rows.data_ptr(), index.numel(), rows.shape[1],
                                        1 if average else 0, stream_ptr(stream)))


def mark_rows(ctx: CacheCtx, slots: torch.Tensor, touched: torch.Tensor, stream=None):
    assert slots.dtype == torch.int32 and slots.is_contiguous() and slots.shape[0] == ctx.T
    check(_lib.lib().cdlrm_mark_rows(ctx.handle, slots.data_ptr(), slots.shape[1], touched.data_ptr(),
                                     stream_ptr(stream)))


def sgd_step(param: torch.Tensor, grad: torch.Tensor, lr: float, stream=None):
    assert param.is_contiguous() and grad.is_contiguous() and param.numel() == grad.numel()
    check(_lib.lib().cdlrm_sgd_step(param.data_ptr(), grad.data_ptr(), param.numel(), float(lr), stream_ptr(stream)))


def roc_auc(scores: torch.Tensor, targets: torch.Tensor) -> float:
    """Area under the ROC curve of `scores` against binary `targets`, on the device the scores live on: the Mann-Whitney rank
    sum with average ranks for tied scores (= sklearn.metrics.roc_auc_score), float64 accumulation.  The rank-0 test loop's
    second figure beside the accuracy (the reference parses --mlperf-auc-threshold, main_no_ddp.py:119-120,
    and never computes the AUC).  NaN when only one class is present."""
    s = scores.reshape(-1).to(torch.float32)
    t = targets.reshape(-1).to(s.device) > 0.5
    n = s.numel()
    n_pos = int(t.sum())
    n_neg = n - n_pos
    if n_pos == 0 or n_neg == 0:
        return float("nan")
    order = torch.argsort(s, stable=True)
    ss = s[order]
    # average rank of every run of equal scores: (first + last) / 2 of its 1-based positions
    new = torch.ones(n, dtype=torch.bool, device=s.device)
    new[1:] = ss[1:] != ss[:-1]
    run_id = torch.cumsum(new.to(torch.int64), 0) - 1
    first = torch.nonzero(new).reshape(-1)
    last = torch.cat([first[1:], torch.tensor([n], device=s.device)]) - 1
    avg_rank = ((first + last).to(torch.float64) * 0.5 + 1.0)[run_id]
    rank_sum_pos = avg_rank[t[order]].sum()
    u = rank_sum_pos - n_pos * (n_pos + 1) / 2.0
    return float(u / (float(n_pos) * float(n_neg)))
