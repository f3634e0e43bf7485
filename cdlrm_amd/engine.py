"""The fused per-GPU training engine behind `main_no_ddp.Run` and bench.py.

`TrainEngine.step()` is the reference's loop body (main_no_ddp.py:401-423) as one fixed sequence of
libcdlrm_hip.so launches on preallocated buffers -- no autograd graph, no per-table Python loop:

    bottom MLP -> gather (slots probed one step ahead) -> interaction -> top MLP -> BCE -> top dgrad chain
          -> interaction bwd -> fused embedding bwd + sparse SGD || bottom dgrad chain || weight gradients
          -> grad all-reduce (RCCL) -> dense SGD -> every table_agg_freq steps: touched-row merge across ranks

Five HIP streams: main (the chain above), side (embedding backward; the slot sort of batches without a look-ahead
window), pref (the NEXT batch's take / tag probe and aux-row fill into the other aux region; at long batches also the top
MLP's weight gradients), the window plan's, and a least-priority one on which `WindowResolver` sorts the slot ids of a
look-ahead chunk slice by slice, batches ahead of their steps (the embedding backward's sort, off every queue a step waits
for; its once-only flags let the interaction backward update the slots a batch reads once).  The launch sequence of a step is
recorded once per control path and replayed (`_step_taped`).

`WindowPipeline` is the look-ahead side: it plans window w+1 (unique scan, tag probe, way choice, winners-
only pinned-host -> HBM row fetch) on a side HIP stream while window w trains, and commits it at the
boundary (main_no_ddp.py:393-399, 148-209; cache_manager.py:66-115).
"""
from __future__ import annotations

import math
import os
from typing import List, Optional

import torch
import torch.distributed as dist

from . import _lib
from . import _streams as S
from . import ops
from ._lib import record as rec
from .model_no_ddp import DLRM_Net, Embedding_Table_Cache_Group, Embedding_Table_Group, _linears


def square_bags(lS_o, lS_i, device=None, multiple: int = 256):
    """Ragged multi-hot bags (the reference's random front end: T index lists of different lengths + T offset lists,
    dlrm_data_pytorch.py:763-805) -> the rectangular layout the kernels take: (offsets int64 [T, n_bags + 1],
    indices int64 [T, n]).  Tables are squared off with ONE extra bag: the padding lookups of table k repeat its first
    index of the batch (no new row is touched, no new window index appears) and are pooled into bag n_bags, a scratch row
    behind the batch whose gradient is zero -- adding lr * 0 leaves the cache rows bit-identical.  n is rounded up to
    `multiple` so that a stream of batches reuses a few buffer shapes.  Rectangular inputs pass through unchanged."""
    if isinstance(lS_i, torch.Tensor) and lS_i.dim() == 2:
        off = lS_o if isinstance(lS_o, torch.Tensor) or lS_o is None else torch.stack(list(lS_o))
        return (None if off is None else off.to(device) if device is not None else off,
                lS_i.to(device) if device is not None else lS_i)
    idx_l = [torch.as_tensor(x).reshape(-1).to(torch.int64) for x in lS_i]
    off_l = [torch.as_tensor(x).reshape(-1).to(torch.int64) for x in lS_o]
    T, nb = len(idx_l), int(off_l[0].numel())
    lens = [int(x.numel()) for x in idx_l]
    assert min(lens) >= 1 and all(int(o.numel()) == nb for o in off_l), "every table needs >= 1 lookup and n_bags offsets"
    n = (max(lens) + multiple - 1) // multiple * multiple
    idx = torch.empty(T, n, dtype=torch.int64)
    off = torch.empty(T, nb + 1, dtype=torch.int64)
    for k in range(T):
        idx[k, :lens[k]] = idx_l[k]
        idx[k, lens[k]:] = idx_l[k][0]
        off[k, :nb] = off_l[k]
        off[k, nb] = lens[k]
    if device is not None:
        idx, off = idx.to(device), off.to(device)
    return off, idx


def pad_window(lists, device=None):
    """A look-ahead window of ragged per-table index lists -> int64 [T, n]: the plan only needs each table's SET of
    indices, so short tables are padded with their own first index."""
    lists = [torch.as_tensor(x).reshape(-1).to(torch.int64) for x in lists]
    n = max(int(x.numel()) for x in lists)
    out = torch.empty(len(lists), n, dtype=torch.int64)
    for k, x in enumerate(lists):
        out[k, :x.numel()] = x
        out[k, x.numel():] = x[0]
    return out.to(device) if device is not None else out


class WindowPipeline:
    def __init__(self, cache_group: Embedding_Table_Cache_Group, host_tables: Embedding_Table_Group, max_window: int,
                 *, parity_rng: bool = False, seed: int = 0, average_on_writeback: bool = False, rank: int = 0,
                 world_size: int = 1, cap_uniq: Optional[int] = None, cap_win: Optional[int] = None,
                 victim_rows: Optional[int] = None, host_gather: bool = False, gather_threads: int = 16,
                 shard_fetch: Optional[bool] = None, process_group=None, write_back: bool = True,
                 force_collectives: bool = False):
        """victim_rows: capacity (rows) of each of the two HBM buffers that hold the host rows of a window's
        non-cached indices (None: every unique index of a window, at most 8 GiB per buffer; 0: off -- every miss
        reads the host table over PCIe as the reference does).
        host_gather: move the plan's bulk rows (winners, victims) the way the reference's Prefetcher does -- CPU threads
        gather them from the host tables into pinned staging, ONE DMA copy brings them to HBM -- from a background
        thread, instead of GPU waves reading host memory (which slows every kernel beside them ~2x while they run;
        a DMA copy costs the training ~2 %).  Not with parity_rng (that path draws its random numbers in-line).
        shard_fetch (host_gather plans at world_size > 1; default on): every rank's plan is the same list of rows, so
        each rank fetches only its 1/world slice from the host tables (CPU gather + DMA over ITS PCIe link) and the
        slices are exchanged over xGMI with one all-gather per list at the window boundary (commit(), main thread,
        same point of the step sequence on every rank).  Host-DRAM and PCIe traffic per window drop from
        world x (winners + victims) to 1 x; the reference instead broadcasts whole cache tables from rank 0
        (main_no_ddp.py:318-319).
        write_back=False: evicted rows are dropped instead of written to the host tables (repeatability tests that must
        leave the host tables as they found them; never in training).
        force_collectives: take the multi-rank code paths (sharded fetch + in-place all-gather, barrier) at world_size 1
        too -- a 1-rank communicator executes every collective the N-rank run issues (tests/test_rccl_one_rank.py)."""
        self.cg, self.host = cache_group, host_tables
        self.write_back = bool(write_back)
        self.ctx = cache_group.ctx
        self.plan = ops.WindowPlan(self.ctx, max_window, cap_uniq, cap_win)
        # Each of the two victim buffers starts at <= 8 GiB and GROWS when a window's plan lists more non-cached indices than it
        # holds (host-gather plans: the count is on the host before any row moves), up to a fifth of the device's memory per
        # buffer: a Zipf window at c3 leaves 9-10 M victims (5 GB), a UNIFORM one 76 M (39 GB) -- cut at 8 GiB, four of five
        # misses of every step read the host table over PCIe inside the step (0.90 instead of 0.70 ms/step).  288 GB of HBM are
        # there to be used; what does not fit even then is still served from the host table, as the reference serves all of it.
        self.victim_limit = 0
        if victim_rows is None:
            victim_rows = min(self.plan.cap_uniq, (8 << 30) // (4 * self.ctx.D))
            if S.is_hip(cache_group.weight.device):
                total = torch.cuda.get_device_properties(cache_group.weight.device).total_memory
                self.victim_limit = min(self.plan.cap_uniq, int(total * 0.2) // (4 * self.ctx.D))
        self.victims = [ops.Victims(self.ctx, victim_rows) for _ in range(2)] if victim_rows > 0 else None
        self._vnext = 0
        self.dev = cache_group.weight.device
        # the plan's stream: least urgent priority -- the window scan shares the GPU with the training step, which goes first
        self.side = S.background_stream(self.dev)
        self.parity_rng, self.seed, self.avg = parity_rng, int(seed), average_on_writeback
        self.rank, self.world = rank, world_size
        self.multi = self.world > 1 or bool(force_collectives)
        self.host_ptrs = host_tables.device_pointers()
        self.ctx.bind_host_tables(self.host_ptrs)
        # every rank holds a private copy of the host tables (hostmem.make_host_tables "replicas"): every rank writes its evictions
        # back -- identical rows on every rank at a commit (sync_touched_to_rank0 precedes it), so the copies stay identical
        self.replicated_host = bool(getattr(host_tables, "_replicated", False))
        self.window_no = 0
        self._commits = 0
        self.planned = None          # event: plan of the next window is ready
        self.written_back = None     # event: evictions of the last commit are in the host tables
        self.last_offsets = None
        self.host_gather = bool(host_gather) and not parity_rng and S.is_hip(self.dev)
        self.gather_threads = int(gather_threads)
        if shard_fetch is None:
            shard_fetch = True
        self.shard = bool(shard_fetch) and self.host_gather and self.multi
        self.pg = process_group
        self._exchange = []          # (buffer, chunk rows) all-gathers commit() owes for the plan in flight
        self._worker = None          # background thread of a host-gather plan
        self._worker_err = None
        self._pin = {}               # pinned host staging, grown on demand
        # where the last host-gather plan's time went (ms, host clock of the plan thread; `dma` from events at commit()):
        # the GPU half (window scan, tag probe, way choice, victim list), list copies to the host, first-touch allocation of
        # pinned staging (first window only, or when a list outgrows it), the CPU threads' row gather, the DMA copies
        self._breakdown = None
        self._bd = None

    @property
    def breakdown(self):
        """Itemised cost of the last committed host-gather plan (None before the first).  Reading it waits for that plan's
        DMA copies (instrumentation: commit() itself never blocks the host for it)."""
        return self.resolve_breakdown(self._breakdown)

    def breakdown_deferred(self):
        """The same record WITHOUT waiting for anything: hand it to resolve_breakdown() later (bench.py: after the timed region)."""
        return self._breakdown

    @staticmethod
    def resolve_breakdown(bd):
        if bd is not None and "_dma_events" in bd:
            evs = bd.pop("_dma_events")
            if evs and isinstance(evs[0][0], torch.cuda.Event):
                evs[-1][1].synchronize()
                bd["dma"] = float(sum(a.elapsed_time(b) for a, b in evs))
        return bd

    def _pinned(self, key, shape, dtype):
        t = self._pin.get(key)
        if t is None or t.shape[0] < shape[0]:
            import time as _time
            t0 = _time.perf_counter()
            rows = int(shape[0] * 1.25) + 1024
            t = torch.empty((rows,) + tuple(shape[1:]), dtype=dtype, pin_memory=True)
            self._pin[key] = t
            if self._bd is not None:
                self._bd["pinned_alloc"] += (_time.perf_counter() - t0) * 1e3
        return t

    def reserve_staging(self, win_rows: int, victim_rows: int):
        """Allocate the pinned host staging of a host-gather plan up front (rows of the winners' / victims' lists a window
        may produce): pinning GBs of host memory costs ~1 s the first time a plan needs it; a trainer that knows its window
        size pays it at set-up, not inside its first window."""
        D = self.ctx.D
        for key, rows in (("win", win_rows), ("vic", victim_rows)):
            if rows > 0:
                self._pinned(key + "_idx", (int(rows),), torch.int64)
                self._pinned(key + "_rows", (int(rows), D), torch.float32)

    def _shard_range(self, n, cap_rows):
        """This rank's slice of a list of n rows that travels as `world` equal chunks: (chunk, lo, hi), or None when the
        list is fetched whole (sharding off, empty list, or the padded length would not fit the buffer)."""
        if not self.shard or n == 0:
            return None
        chunk = -(-n // self.world)
        if chunk * self.world > cap_rows:
            return None
        lo = min(self.rank * chunk, n)
        return chunk, lo, min(lo + chunk, n)

    def _fetch_list(self, key, idx_dev, off, n, dst_dev, D):
        """Rows of one plan list (n entries, `off` = per-table offsets into it) -> dst_dev[:n]: CPU-thread gather into
        pinned staging + one DMA copy, of the whole list or of this rank's slice (the exchange is queued for commit())."""
        import time as _time
        side, bd = self.side, self._bd
        sh = self._shard_range(n, dst_dev.shape[0])
        chunk, lo, hi = sh if sh is not None else (0, 0, n)
        m = hi - lo
        alloc0 = bd["pinned_alloc"] if bd is not None else 0.0
        t0 = _time.perf_counter()
        idx_h = self._pinned(key + "_idx", (max(m, 1),), torch.int64)
        idx_h[:m].copy_(idx_dev[lo:hi], non_blocking=True)
        side.synchronize()
        t1 = _time.perf_counter()
        if bd is not None:          # (a first-touch allocation of the index staging is itemised as pinned_alloc, not here)
            bd["lists_to_host"] -= bd["pinned_alloc"] - alloc0
        if m > 0:
            rows_h = self._pinned(key + "_rows", (m, D), torch.float32)
            t2 = _time.perf_counter()
            off_r = [min(max(int(o), lo), hi) - lo for o in off]
            ops.host_gather_rows(self.host_ptrs, idx_h, off_r, D, rows_h, self.gather_threads)
            t3 = _time.perf_counter()
            ev0, ev1 = S.new_event(self.dev, timing=True), S.new_event(self.dev, timing=True)
            ev0.record(side)
            dst_dev[lo:hi].copy_(rows_h[:m], non_blocking=True)
            ev1.record(side)
            if bd is not None:
                bd["cpu_gather"] += (t3 - t2) * 1e3
                bd["_dma_events"].append((ev0, ev1))
                bd["rows"][key] = m
        if bd is not None:
            bd["lists_to_host"] += (t1 - t0) * 1e3
        if sh is not None:
            self._exchange.append((dst_dev, chunk))

    def _plan_host_gather(self, window_idx, lists_ready, t_launch):
        """Background half of a host-gather plan: wait for the winner / victim lists, copy them down, gather the rows on
        the CPU, issue the two DMA copies and the `planned` event on the plan stream."""
        try:
            import time as _time
            torch.cuda.set_device(self.dev)
            plan, side, T, D = self.plan, self.side, self.ctx.T, self.ctx.D
            vic = self.victims[self._vnext] if self.victims is not None else None
            bd = self._bd = dict(gpu_scan_probe_assign=0.0, lists_to_host=0.0, pinned_alloc=0.0, cpu_gather=0.0, dma=None,
                                 rows={}, threads=self.gather_threads, sharded=self.shard, _dma_events=[])
            lists_ready.synchronize()
            bd["gpu_scan_probe_assign"] = (_time.perf_counter() - t_launch) * 1e3
            t0 = _time.perf_counter()
            _, _, wo = plan.offsets(stream=side)
            W = wo[T]
            bd["lists_to_host"] += (_time.perf_counter() - t0) * 1e3
            with torch.cuda.stream(side):
                self._fetch_list("win", plan.win_idx, wo, W, plan.stage, D)
                if vic is not None:
                    t0 = _time.perf_counter()
                    voff = vic.off.cpu().tolist()
                    bd["lists_to_host"] += (_time.perf_counter() - t0) * 1e3
                    need = voff[T + 1]               # what the window has; voff[T] = what the buffer holds of it
                    if need > vic.cap and self.victim_limit > vic.cap:
                        # more victims than this buffer holds: a larger one (this is the buffer of the window being planned:
                        # nothing reads it yet), and the list again
                        t0 = _time.perf_counter()
                        side.synchronize()
                        old_cap, vic = vic.cap, None        # (no reference of ours keeps the old buffer alive)
                        grown = self._grow_victims(old_cap, min(self.victim_limit, int(need * 1.1) + 4096), self._commits == 0)
                        vic = self.victims[self._vnext]
                        if grown:
                            plan.victims(vic, stream=side, list_only=True)
                            voff = vic.off.cpu().tolist()
                        bd["victims_regrown_ms"] = (_time.perf_counter() - t0) * 1e3
                    bd["victims_in_window"], bd["victim_capacity"] = int(voff[T + 1]), int(vic.cap)
                    V = min(voff[T], vic.cap)
                    voff = [min(o, V) for o in voff[:T + 1]]
                    self._fetch_list("vic", vic.idx, voff, V, vic.rows, D)
                bd["host_total"] = (_time.perf_counter() - t_launch) * 1e3
                if isinstance(window_idx, torch.Tensor) and window_idx.is_cuda:
                    window_idx.record_stream(side)
                self.planned = S.new_event(self.dev)
                self.planned.record(side)
        except BaseException as e:          # surfaced by commit()
            self._worker_err = e

    def _grow_victims(self, old_cap: int, want: int, both: bool) -> bool:
        """Replace the victim buffer of the window being planned (both: the other one too -- the very first plan, when neither is
        bound: sized now, beside a plan nobody trains next to, instead of inside the second window; a 40 GB allocation holds the
        device for ~1 s) by one of `want` rows -- or of what the device has FREE right now, less 4 GiB of headroom for the
        trainer's own later allocations: this runs in the plan's background thread in the middle of training, where an
        out-of-memory error would surface windows after set-up.  True: the buffers are new objects (the list has to be made
        again); False: the old ones stay -- the misses that do not fit read the host table, as the reference serves all of them."""
        n_buf = 2 if both else 1
        row = 4 * self.ctx.D + 12                     # row + index + position per entry (ops.Victims)
        if self.multi:
            # Several ranks: the capacity has to come out THE SAME on every rank -- with the sharded fetch V = min(victims,
            # capacity) sets the chunk sizes (and the number) of commit()'s all-gathers, and the bound victim set has to be one
            # set.  So nothing rank-local enters: `want` derives from the window's victim count (the replicated plan: equal
            # everywhere) and victim_limit (a fifth of the device's TOTAL memory); this rank's free memory does not, and there is
            # no smaller-size fallback -- a rank that cannot allocate what the others allocate stops the job here, loudly,
            # instead of training on a different victim set.
            cap = int(want)
            if cap <= old_cap:
                return False
            slots = [self._vnext, self._vnext ^ 1][:n_buf]
            for i in slots:
                self.victims[i] = None
            try:
                for i in slots:
                    self.victims[i] = ops.Victims(self.ctx, cap)
            except torch.OutOfMemoryError as e:
                raise RuntimeError("rank %d: victim buffers of %d rows (the capacity every rank allocates) do not fit this "
                                   "device: lower --lookahead or pass victim_rows" % (self.rank, cap)) from e
            return True
        free = torch.cuda.mem_get_info(self.dev)[0] if S.is_hip(self.dev) else (1 << 62)
        free += n_buf * old_cap * row                 # what releasing the old buffer(s) gives back
        cap = int(min(want, (free - (4 << 30)) // (n_buf * row)))
        if cap <= old_cap:
            return False
        slots = [self._vnext, self._vnext ^ 1][:n_buf]
        for c in (cap, old_cap):
            try:
                for i in slots:
                    self.victims[i] = None
                for i in slots:
                    self.victims[i] = ops.Victims(self.ctx, c)
                return True
            except torch.OutOfMemoryError:          # (the allocator's view and the driver's can differ): the old size again
                for i in slots:
                    self.victims[i] = None
                torch.cuda.empty_cache()
        raise RuntimeError("victim buffers: not even the previous capacity could be allocated again")

    def _unique(self, window_idx, side):
        """K1 on the plan stream: one [T, n] tensor, or a window streamed as chunks (a callable returning an iterator
        of [T, n_c] tensors, produced on the plan stream: a window that does not fit HBM in one piece)."""
        plan = self.plan
        if callable(window_idx):
            for chunk in window_idx():
                plan.unique_add(chunk, stream=side)
                del chunk
            plan.unique_finish(stream=side)
        else:
            plan.unique(window_idx, stream=side)

    def plan_window(self, window_idx, q_source=None):
        """Launch the plan of one window on the side stream.  window_idx: [T, n] int64 on the device, or a callable
        that yields the window's chunks (called under the plan stream; see _unique)."""
        plan, side = self.plan, self.side
        side.wait_stream(S.current_stream(self.dev))          # window_idx may have been produced there
        if self.host_gather:
            import threading
            import time as _time
            assert self._worker is None, "the previous plan was never committed"
            t_launch = _time.perf_counter()
            with S.on_stream(side):
                self._unique(window_idx, side)
                plan.probe(stream=side)
                plan.assign(None, seed=self.seed * 1000003 + self.window_no, stream=side)
                if self.victims is not None:
                    plan.victims(self.victims[self._vnext], stream=side, list_only=True)
                lists_ready = S.new_event(self.dev)
                lists_ready.record(side)
            self.planned = None
            self._worker_err = None
            self._exchange = []
            self._worker = threading.Thread(target=self._plan_host_gather, args=(window_idx, lists_ready, t_launch),
                                            daemon=True)
            self._worker.start()
            self.window_no += 1
            return
        with S.on_stream(side):
            self._unique(window_idx, side)
            plan.probe(stream=side)
            if self.parity_rng:
                # Categorical.sample() draws from the torch CPU generator, table by table (main_no_ddp.py:184-185)
                uo, ko, _ = plan.offsets(stream=side)
                T, ways = self.ctx.T, self.ctx.ways
                src = q_source if q_source is not None else (lambda M, w: torch.empty(M, w).exponential_(1) if M else torch.empty(0, w))
                qs = [src(ko[k + 1] - ko[k], ways) for k in range(T)]
                q = torch.cat(qs).contiguous().to(self.dev, non_blocking=False) if ko[T] else None
                self.last_offsets = (uo, ko)
                plan.assign(q if q is not None else torch.empty(1, ways, device=self.dev), stream=side)
            else:
                plan.assign(None, seed=self.seed * 1000003 + self.window_no, stream=side)
            plan.fetch(self.host_ptrs, False, stream=side)
            if self.victims is not None:
                plan.victims(self.victims[self._vnext], stream=side)
            if isinstance(window_idx, torch.Tensor) and window_idx.is_cuda:
                window_idx.record_stream(side)
            self.planned = S.new_event(self.dev)
            self.planned.record(side)
        self.window_no += 1

    def commit(self):
        """At the window boundary: swap the fetched rows in, write the tags, write the evicted rows back."""
        main = S.current_stream(self.dev)
        if self._worker is not None:            # host-gather plan: the background half has to have issued its copies
            self._worker.join()
            self._worker = None
            if self._worker_err is not None:
                raise self._worker_err
        assert self.planned is not None, "plan_window() first"
        main.wait_event(self.planned)
        if self._bd is not None:
            # (the DMA copies' own duration is read from their timing events when `breakdown` is QUERIED, not here: a commit
            #  keeps the host free to issue the next steps -- on a late plan the wait above is the GPU's, not the host's)
            self._breakdown, self._bd = self._bd, None
        # sharded fetch: every rank holds its slice of each list in place; one in-place all-gather per list (RCCL over
        # xGMI) completes them.  Issued here, on the main thread and stream, at the same point of the step sequence on
        # every rank, so it orders with the per-step gradient exchanges on the same communicator.
        for buf, chunk in self._exchange:
            full = buf[:chunk * self.world]
            dist.all_gather_into_tensor(full, full[self.rank * chunk:(self.rank + 1) * chunk], group=self.pg)
        self._exchange = []
        self.plan.commit(stream=main)
        self._commits += 1
        if self.victims is not None:
            # from here on the per-iteration probe serves misses from this window's resident victim rows (the buffer
            # of the previous window is free for the next plan: every probe that read it is ordered before this point)
            self.ctx.bind_victims(self.victims[self._vnext])
            self._vnext ^= 1
        done = S.new_event(self.dev)
        done.record(main)
        with S.on_stream(self.side):
            self.side.wait_event(done)
            # evictions come from rank 0's copy only (main_no_ddp.py:208, 312-315); with private host-table copies every rank
            # applies them to its own
            if (self.rank == 0 or self.replicated_host) and self.write_back:
                self.plan.writeback(self.host_ptrs, self.avg, stream=self.side)
            self.written_back = S.new_event(self.dev)
            self.written_back.record(self.side)
        self.planned = None

    def close(self):
        """Wait for a plan that is still in flight (its background thread runs C++ threads of its own; a process that exits
        under them aborts in teardown).  Call before the process ends; the plan's result is dropped."""
        w, self._worker = self._worker, None
        if w is not None:
            w.join()
        if S.is_hip(self.dev):
            self.side.synchronize()

    def plan_in_flight(self) -> bool:
        """Is the plan of the next window running right now (its kernels / DMA copies / CPU gather)?  bench.py flags the
        roofline-kernel samples taken beside one."""
        if self._worker is not None and self._worker.is_alive():
            return True
        return self.planned is not None and S.is_hip(self.dev) and not self.planned.query()

    def wait_writeback(self):
        """Host-side: the evicted rows are in the host tables (all ranks may read them afterwards)."""
        if self.written_back is not None:
            self.written_back.synchronize()
        if self.multi:
            dist.barrier(group=self.pg)

    def eviction_data(self):
        """(idx, rows) per table, as the reference queues them (main_no_ddp.py:199); synchronises."""
        _, _, wo = self.plan.offsets()
        out = []
        for k in range(self.ctx.T):
            tag = self.plan.ev_tag[wo[k]:wo[k + 1]]
            valid = tag != -1
            out.append((tag[valid], self.plan.stage[wo[k]:wo[k + 1]][valid]))
        return out


class WindowResolver:
    """The look-ahead window's lookups resolved ONCE (tags only change at a refill, main_no_ddp.py:393-399): tag match,
    ordered miss numbering per batch and the place of every miss in the window's victim rows (cdlrm_window_resolve), in
    chunks of `chunk` batches issued on the engine's prefetch stream ahead of the training position.  `batch(j)` hands the
    engine the views of batch j; what is left per iteration is cdlrm_embbag_take (copy the slot ids, copy the miss rows).
    Create it right after the window's commit(); call ensure() after every step.

    Chunk results live in a fixed RING of three buffers owned by the engine (chunk c in slot c % 3), not in per-chunk
    allocations: the resolve writes them on the prefetch stream while takes read them on the side stream, streams the
    caching allocator knows nothing about -- a block freed on the host while queued takes still read it would be handed
    straight to the next chunk's resolve.  A slot is recycled behind an event recorded on the side stream when the recycling
    resolve is issued, and chunk c is only issued once batch() has been asked for the SECOND batch of chunk c - 2 (the step
    of that chunk's first batch is then on the queues): every take of chunk c - 3, the slot's previous holder, has been
    ISSUED by then, so the event covers them all however far the host runs ahead of the GPU."""

    RING = 3

    def __init__(self, engine: "TrainEngine", window_idx: torch.Tensor, global_batch: int, *, chunk: int = 16):
        self.eng, self.ctx = engine, engine.ctx
        self.idx = window_idx
        self.B = int(global_batch)
        self.lbs = -(-self.B // engine.world)
        self.rank = engine.rank
        # this rank's slice of a global batch (main_no_ddp.py:388-391): the last rank's is shorter when world does not divide B
        self.col0 = min(self.rank * self.lbs, self.B)
        self.width = max(0, min(self.lbs, self.B - self.col0))
        n = int(window_idx.shape[1])
        assert n % self.B == 0, "a window is a whole number of batches"
        self.nb = n // self.B
        self.CH = max(1, int(chunk))
        self.chunks = {}            # chunk number -> (wslots, wsrc, ready event)
        self.done = 0               # chunks issued
        self.maxj = -1              # highest batch handed out by batch()
        key = ("wres", self.CH * self.B)
        if key not in engine._bufs:
            T = self.ctx.T
            engine._bufs[key] = [tuple(torch.empty(T * self.CH * self.B, dtype=torch.int32, device=window_idx.device)
                                       for _ in range(2)) for _ in range(self.RING)]
        self._ring = engine._bufs[key]
        if ("wres_ev",) not in engine._bufs:        # one persistent event per ring slot (a launch tape keeps its handle in a cell)
            evs = [S.new_event(engine.dev) for _ in range(self.RING)]
            if S.is_hip(engine.dev):
                for e in evs:
                    e.record(S.current_stream(engine.dev))
            engine._bufs[("wres_ev",)] = evs
        self._ring_ev = engine._bufs[("wres_ev",)]
        # The embedding backward's slot sort, per CHUNK instead of per step (round 6): a window's slot ids are final once its
        # chunk is resolved, so the lists of SL batches x T tables are sorted by ONE set of launches (four launches for four
        # batches instead of four per batch, 416 instead of 104 sorting workgroups each), on the side stream behind a step's
        # embedding update, batches ahead of their use.  A step whose batch has its lists sorted issues no sort, and its
        # interaction backward finds the once-only flags there (TrainEngine.fuse_once) without waiting for anything: the
        # slice's event is waited for by the stream that runs the batch's take, which the step's gather waits for anyway.
        self.sort_chunks = bool(getattr(engine, "sort_chunks", False)) and S.is_hip(engine.dev) and self.width > 0
        # (short batches: the slice's four runtime calls are host time the issuing thread does not have every other step --
        #  per-rank 1024: 0.1852 ms with slices of 2 against 0.1824 with the per-step sort on the launch tape -- and its kernels
        #  are small: as many batches as make ~16 k lookups per table)
        sl = int(getattr(engine, "sort_slice", 2))
        self.SL = max(1, min(max(sl, (sl * 8192) // max(self.width, 1)), self.CH))
        self.sorted_upto = 0        # batches [0, sorted_upto) of the window have had their slice's turn
        self._sorted_at = {}        # first batch of a slice whose sort was issued -> batches in it (engine.sort_chunks at that time)
        engine._sev_waited = engine._sev_waited_cur = None   # (the slice events are shared by every resolver of the engine)
        if self.sort_chunks:
            skey = ("wsorted", self.CH, self.width)
            if skey not in engine._bufs:
                engine._bufs[skey] = [ops.embbag_bwd_sorted(self.ctx, self.CH, self.width, window_idx.device)
                                      for _ in range(self.RING)]
            self._sorted_ring = engine._bufs[skey]
            nsl = -(-self.CH // self.SL)
            ekey = ("wsorted_ev", self.CH, self.SL)
            if ekey not in engine._bufs:
                engine._bufs[ekey] = [[S.new_event(engine.dev) for _ in range(nsl)] for _ in range(self.RING)]
            self._sorted_ev = engine._bufs[ekey]
        self._armed = None          # chunk handed to the engine, to be placed by its next step
        if getattr(engine, "_pending_resolve", None) is not None:
            # a chunk of the window this one replaces that no step has placed: its ring slot belongs to this window now
            engine._pending_resolve, engine.mark_next = None, False
        self._started = False       # the first chunks of a window are issued at once (no step of the window is in flight)
        self.ensure(self.CH + 2)
        self._started = True

    def _long_batch(self) -> bool:
        """Long local batches on a HIP device with an engine that can place a resolve inside its step (TrainEngine.
        place_resolve_min): a chunk's resolve issued at once lands on whatever the training queue runs then -- one interaction
        forward in sixteen took 60 us instead of 23 -- so the step places it behind its interaction forward."""
        eng = self.eng
        return (hasattr(eng, "_issue_resolve") and S.is_hip(eng.dev)
                and (self.width >= getattr(eng, "place_resolve_min", 1 << 62) or not eng._side_gather(self.width)))

    def _prepare(self, c: int):
        ctx = self.ctx
        self.chunks.pop(c - self.RING, None)
        last = getattr(self.eng, "_last_sort_ev", None)
        if last is not None and c < self.RING:
            # The slice sorts read the ring slots' slot ids on a least-priority stream of their own.  Inside a window the slot's
            # previous holder is chunk c - 3, whose slices were all waited for by the takes of its batches, long ago; the first
            # chunks of a window recycle slots of the PREVIOUS window, which may have been left early -- a sort whose batches
            # were never trained has no take behind it: this resolve (prefetch stream, issued after this call) is ordered behind
            # every sort issued so far
            self.eng.pref.wait_event(last)
        b0, b1 = c * self.CH, min(self.nb, (c + 1) * self.CH)
        cols = self.idx[:, b0 * self.B:b1 * self.B]
        w = int(cols.shape[1])
        ring_ws, ring_src = self._ring[c % self.RING]
        ws, wsrc = ring_ws[:ctx.T * w].view(ctx.T, w), ring_src[:ctx.T * w].view(ctx.T, w)
        ev = self._ring_ev[c % self.RING]
        self.chunks[c] = (ws, wsrc, ev)
        return dict(cols=cols, lbs=self.lbs, ws=ws, wsrc=wsrc, ev=ev, B=self.B, first=(c == 0))

    def ensure(self, upto_batch: int, urgent: bool = False):
        """Issue the resolve of every chunk that holds a batch < upto_batch (prefetch stream; no host wait).  Called after every
        step.  At long batches a chunk that falls due is handed to the engine, whose NEXT step issues it behind its interaction
        forward (TrainEngine._issue_resolve); urgent (batch() needs the chunk now), the first chunks of a window, short batches
        and engines without that hook: at once."""
        eng = self.eng
        if self._armed is not None:
            if eng._pending_resolve is None:
                self._armed = None                      # the step issued since has placed it
            elif urgent and self._armed * self.CH < int(upto_batch):
                self._armed = None
                eng.flush_pending_resolve()             # needed before a step could place it
        nchunks = -(-self.nb // self.CH)
        want = min(nchunks, -(-min(int(upto_batch), self.nb) // self.CH))
        while self.done < want and self._armed is None:
            c = self.done
            if c >= self.RING and self.maxj < (c - 2) * self.CH + 1:
                break               # the slot's previous chunk may still have takes to issue: a later ensure() / batch() issues it
            self.done += 1
            pr = self._prepare(c)
            if not urgent and c > 0 and self._started and self._long_batch() and eng._pending_resolve is None:
                eng._pending_resolve, eng.mark_next = pr, True
                self._armed = c
                break               # one chunk per step
            if hasattr(eng, "_issue_resolve"):
                eng._issue_resolve(pr, lambda fn, *a: fn(*a), S.current_stream(eng.dev), placed=False)
            else:                   # (host-logic tests: a bare engine stand-in)
                ops.window_resolve(self.ctx, pr["cols"], self.lbs, pr["ws"], pr["wsrc"], stream=eng.pref, batch_len=self.B)
        if not urgent:
            # (callers pass the batch just trained + CH + 2) the slice that holds the batch after next: the next step issues that
            # batch's take, and the take's stream waits for the slice
            self.ensure_sorted(int(upto_batch) - self.CH - 2 + 3)

    def _chunk_nb(self, c: int) -> int:
        return min(self.nb, (c + 1) * self.CH) - c * self.CH

    def ensure_sorted(self, upto_batch: int):
        """Issue the slot sort of every slice that holds a batch < upto_batch (side stream, behind whatever the last step put
        there; no host wait).  A slice of a chunk whose resolve has not been ISSUED yet stays for a later call."""
        if not self.sort_chunks:
            return
        eng = self.eng
        upto = min(int(upto_batch), self.nb)
        while self.sorted_upto < upto:
            b0 = self.sorted_upto
            c = b0 // self.CH
            if c >= self.done or c not in self.chunks:
                break
            if self._armed is not None and c >= self._armed and getattr(eng, "_pending_resolve", None) is not None:
                break               # handed to the engine, not yet issued
            ws, _, ev = self.chunks[c]
            nbc = self._chunk_nb(c)
            j0 = b0 - c * self.CH
            cnt = min(self.SL, nbc - j0)
            if getattr(eng, "sort_chunks", True):       # (switched per slice by tools/ab_step.py)
                st = eng.sort_stream(self.width)
                st.wait_event(ev)                       # the chunk's resolve (prefetch stream)
                # the ring slot's previous lists were read by embedding updates three chunks back; what the wait really picks is
                # WHERE in the step the sort runs: behind the last step's embedding update = in that step's tail
                after = getattr(eng, "sort_after", "emb_done")
                if after == "interacted":
                    st.wait_event(eng._events["interacted"])
                elif eng._emb_done is not None:
                    st.wait_event(eng._emb_done)
                if getattr(eng, "sort_delay", 0):       # (tools/race_check.py: a late sort shows a consumer that is not ordered behind it)
                    with torch.cuda.stream(st):
                        torch.cuda._sleep(int(eng.sort_delay))
                ops.embbag_bwd_prepare_window(self.ctx, ws[:, self.col0:], self.B, nbc, self.width,
                                              self._sorted_ring[c % self.RING], stream=st, j0=j0, count=cnt)
                self._sorted_ev[c % self.RING][j0 // self.SL].record(st)
                eng._last_sort_ev = self._sorted_ev[c % self.RING][j0 // self.SL]
                self._sorted_at[b0] = cnt
            self.sorted_upto = b0 + cnt
            self._sorted_at.pop(b0 - 3 * self.CH, None)

    def sorted_views(self, j: int):
        """(keys, meta, once addresses of batch j's sorted lists, elements between two tables' lists, the slice's event) -- or
        None: not sorted (yet)."""
        if not self.sort_chunks or j >= self.sorted_upto:
            return None
        c = j // self.CH
        nbc = self._chunk_nb(c)
        jl = j - c * self.CH
        if (j - jl % self.SL) not in self._sorted_at:
            return None
        k, m, o = ops.embbag_bwd_sorted_views(self.ctx, self._sorted_ring[c % self.RING], nbc, self.width, jl)
        return k, m, o, nbc * self.width, self._sorted_ev[c % self.RING][jl // self.SL]

    def batch(self, j: int):
        """(wslots view, wsrc view, ready event) of this rank's lookups of batch j of the window."""
        c = j // self.CH
        self.maxj = max(self.maxj, j)
        self.ensure(j + 1, urgent=True)
        ws, wsrc, ev = self.chunks[c]
        col = (j - c * self.CH) * self.B + self.col0
        if c >= 2 and (c - 2) in self.chunks:       # chunks far behind the training position are not needed any more
            del self.chunks[c - 2]
        # (4th entry: where this batch sits in which window -- the engine's row merge looks AHEAD from there, lookahead())
        return ws[:, col:col + self.width], wsrc[:, col:col + self.width], ev, (self, j)

    def lookahead(self, j: int, kmax: int):
        """[(k, slot ids of GLOBAL batch j + k -- every rank's lookups --, ready event)] for k = 1, 2, ... while the batch
        belongs to this window and its chunk's resolve has been issued (at most kmax): the rows the next steps will use."""
        out = []
        eng = self.eng
        for k in range(1, int(kmax) + 1):
            b = j + k
            if b >= self.nb:
                break
            c = b // self.CH
            if c >= self.done or c not in self.chunks:
                break
            if self._armed is not None and c >= self._armed and getattr(eng, "_pending_resolve", None) is not None:
                break               # handed to the engine, not yet issued
            ws, _, ev = self.chunks[c]
            col = (b - c * self.CH) * self.B
            out.append((k, ws[:, col:col + self.B], ev))
        return out


class TrainEngine:
    def __init__(self, cache_group: Embedding_Table_Cache_Group, dlrm: DLRM_Net, host_tables: Embedding_Table_Group,
                 *, lr: float, lr_embeds: float, world_size: int = 1, rank: int = 0, table_agg_freq: int = 1,
                 table_agg_op: str = "mean", process_group=None, loss: str = "bce", loss_weights=(1.0, 1.0),
                 defer_top_update: bool = False, force_collectives: bool = False):
        """loss / loss_weights: --loss-function / --loss-weights (main_no_ddp.py:364-372); the --loss-threshold clamp
        is read from `dlrm.loss_threshold`.
        defer_top_update: the top MLP's weight gradients, their all-reduce and their SGD update leave the critical
        path -- they run on a side stream beside the interaction backward, the bottom MLP's backward and the HEAD of the
        next step (gather, bottom MLP forward), which only waits for them in front of its interaction.  Same values,
        different schedule; readers of the top MLP's weights outside step()/evaluate() call finish() first.
        force_collectives: run the multi-rank control flow (gradient all-reduce, touched-row flags and merge, exchange
        stream) at world_size 1 as well: over a 1-rank RCCL communicator every collective of the N-rank step executes and is
        the identity, so the result must equal the one-rank fast path bit for bit (tests/test_rccl_one_rank.py)."""
        self.cg, self.dlrm, self.host = cache_group, dlrm, host_tables
        self.ctx = cache_group.ctx
        self.dev = cache_group.weight.device
        self.lr, self.lr_embeds = float(lr), float(lr_embeds)
        self.world, self.rank = int(world_size), int(rank)
        self.multi = self.world > 1 or bool(force_collectives)      # the multi-rank control flow is on
        self.agg_freq, self.agg_op = int(table_agg_freq), table_agg_op
        self.pg = process_group
        self.T, self.D = self.ctx.T, self.ctx.D
        self.F = self.T + 1
        self.itself = bool(dlrm.arch_interaction_itself)
        assert dlrm.arch_interaction_op in ("dot", "cat"), "--arch-interaction-op: dot or cat (model_no_ddp.py:272-304)"
        self.cat = dlrm.arch_interaction_op == "cat"
        self.loss_kind = ops.LOSS[loss]
        self.loss_weights = (float(loss_weights[0]), float(loss_weights[1]))
        thr = float(getattr(dlrm, "loss_threshold", 0.0) or 0.0)
        self.loss_threshold = thr if 0.0 < thr < 1.0 else 0.0
        self.defer_top = bool(defer_top_update)
        self.bot = dlrm._acts(dlrm.bot_l, dlrm.sigmoid_bot)
        self.top = dlrm._acts(dlrm.top_l, dlrm.sigmoid_top)
        self._flatten_params()
        self.ctx.bind_host_tables(host_tables.device_pointers())
        self._bufs = {}
        self.iter = 0
        self.comm = S.new_stream(self.dev) if self.multi else None
        self.side = S.new_stream(self.dev)
        self.pref = S.new_stream(self.dev)          # next batch's tag probe + aux-row fill
        self.agg_rows = None
        # rows per chunk of the touched-row merge (32 MB at D = 128: large enough for xGMI bandwidth, small enough that
        # the gather of chunk i+1 hides under the reduction of chunk i)
        self.agg_chunk_rows = 1 << 16
        # The merge of step j is due, row by row, before the first later step that USES the row (on any rank) -- the look-ahead
        # window knows those steps --, so only the rows the very next batch needs are exchanged at step j; the others follow in
        # deadline order over the next steps, `merge_budget_rows` per step beyond what the next step needs (MergePump below).
        self.lazy_merge = True
        # per step, beyond what the next step needs: max(this, the rank's lookups per step) rows -- 32 MB at c3's per-rank batch,
        # 109 MB at c5's: about one step's length of xGMI time at ring rates, so that the next step's gradient exchange (same
        # communicator stream) finds the stream free; a c3 merge drains in ~29 steps, a c5 merge in ~41 (look-ahead: 34-65)
        self.merge_budget_rows = 1 << 16
        self.merge_budget_auto = True           # (tests switch the per-step-lookups floor off to keep rows on their way)
        self._pump = None
        self._pref = None
        self._phase = 0                             # aux region of the batch being trained
        self._emb_done = None
        ne = lambda: S.new_event(self.dev)
        self._events = dict(probed={k: ne() for k in range(4)}, probed_inline=ne(), gathered=ne(), interacted=ne(),
                            emb_done=ne(), wgrad_done=ne(), top_dz=ne(), top_updated=ne(), fwd_mark=ne(), res_slot=ne(),
                            tier_marked=ne(), bot_dz=ne(), bot_wg=ne())
        if S.is_hip(self.dev):
            # torch creates the HIP event at the first record: give every engine event its handle now (a launch tape stores
            # handles; a wait recorded before the event's first real record would otherwise push that tape back to Python)
            cur = S.current_stream(self.dev)
            for e in list(self._events["probed"].values()) + [v for k, v in self._events.items() if k != "probed"]:
                e.record(cur)
        self._head_scratch = ops.head_scratch(self.dev)
        # running print statistics [correct predictions, loss * mbs] summed over steps in float64 on the device
        # (main_no_ddp.py:427-433 keeps them on the host and synchronises twice per step for it): the head's finish launch adds
        # to them; readers call finish() first and zero them when they start a new interval
        self.stat_acc = torch.zeros(2, dtype=torch.float64, device=self.dev)
        self.loss_sync = True                   # see step()
        # output head in one launch (last layer + loss + its input gradient) when the last top layer is 1-wide + sigmoid
        l_last, a_last = self.top[-1]
        self.fused_head = l_last.out_features == 1 and a_last == 2
        # top-MLP weight gradients at long local batches: the SAME stream as the prefetch (their work never overlaps in
        # time: probe/fill right after the gather, weight gradients late in the backward) -- four streams in all
        # (main, side, pref, the window plan's), one per default hardware queue
        self.wst = self.pref
        # measured (c3 shapes): the split loses ~3 % at B <= 2048 (two more launches on a latency-bound step), gains
        # 8 % at 4096 (0.580 -> 0.534 ms) and at 8192 (0.853 -> 0.782 ms)
        self.split_wgrad_min = 2049
        # Local batches below this take the TWO-AUX-REGION schedule (the next batch's take on the prefetch stream at the head of
        # the step, into the other aux region; its slot sort at the head of the next step; stand-alone gather, where there is one,
        # on the side stream beside the bottom MLP's forward), from it on the CHAINED take (one aux region; take and sort of the
        # next batch behind this batch's embedding update, the gather / fused interaction forward alone on the training queue).
        # Decided by samples/s on the fused build (round 5, tools/ab_step.py, one box, six rounds each, three sessions): at 8192
        # the two-region schedule is 0.8-1.3 % faster (0.5588 / 0.5591 / 0.5593 against 0.5662 / 0.5636 / 0.5651 ms; the driver's
        # 20 steps 0.5737 against 0.5797, a whole window 0.5865 against 0.5906) although the fused kernel -- the HBM-roofline
        # kernel -- then runs beside the next batch's take and this batch's sort: 28.3 us in the step against 23.9 alone (0.57
        # against 0.68 of 8 TB/s; bench.py reports both, roofline.frac and roofline.alone).  At 65536 the chained take stays
        # (3.718 against 3.741 ms), 4096 is a tie, 2048 / 1024 keep two regions (0.2422 / 0.2433, 0.1798 / 0.1845 the other way).
        self.gather_alone_min = 16384
        # The top MLP's forward and its input-gradient chain run with no other GEMM beside them in every schedule of the step
        # (the weight gradients start behind the chain, top_wgrad_after): these launches carry CDLRM_GEMM_ALONE and, where their
        # 128x128 tiles fill the chip (local batch 8192 x 512-wide layers: 256 tiles on 256 CUs; c5), take the one-workgroup-per-CU
        # kernel (csrc/gemm_wide.h).  The bottom MLP's backward runs beside the weight gradients and never does (round 6: with
        # the hint on every GEMM the c3 step took 0.5790 ms against 0.5580 without, with it on these 0.5562).
        self.wide_gemm = True
        # ... and the bottom MLP's forward (it runs at the end of the previous step, beside the next batch's take and slot sort:
        # small kernels that fit beside a wide workgroup): its 512 -> 256 layer on 64x128 tiles, c3 0.5495 against 0.5522 ms
        self.wide_gemm_bottom = True
        # the SGD step of the slots a batch reads ONCE rides in the fused interaction backward (cdlrm_gather_interact_bwd_sgd):
        # those lookups' gradient rows are never written or read back, the sorted path is left with the repeated slots.
        # Bit-identical (a single addend has no order); costs the training queue one wait for the slot sort
        self.fuse_once = True
        # the slot sort of the embedding backward per look-ahead chunk slice instead of per step (WindowResolver.ensure_sorted);
        # steps of batches without sorted lists (no resolver, multi-hot bags) sort their own
        self.sort_chunks = True
        # batches per slice.  tools/ab_step.py, one box, four rounds each, c3, against the per-step sort (0.547-0.552 ms): slices of
        # 1 / 2 / 4 / 8 / 16 batches -1.1 / -1.5 .. -2.2 / -1.1 / -0.8 / +0.4 %: what pays is the sort OFF the queues a step waits for
        # and the folded once-only update it allows, not the batching -- a long slice is a long visitor in one step's tail
        # (on the prefetch stream, see sort_on: 1 / 2 / 4 batches -2.7 / -3.3 / -2.7 %, 2 batches behind the interaction backward
        #  instead of the embedding update -3.1 %)
        self.sort_slice = 2
        self.sort_after = "emb_done"
        # The stream of the slice sorts: "pref" -- issued behind a step, they follow that step's top-MLP weight gradients (and its
        # `top_updated` record) on the prefetch stream and run in the step's tail; the next take on that stream is behind them in
        # order, the one after needs them.  "side": behind the embedding update; "own": a least-priority stream of their own.
        # tools/ab_step.py, one box: c3 0.5364 / 0.5397 / 0.5380 ms, per-rank 1024 0.1804 / 0.1815 / 0.1835, c5 3.651 / 3.615 /
        # 3.575.  "own" wins at c5 but is fragile: in the CLI's process (tools/run_cli_c3.sh, same kernels, same schedule) the
        # step took 1.15 ms instead of 0.54 with it -- whatever hardware queue the runtime gives a SIXTH stream there, the
        # training queue's kernels ended up behind the sort's waits -- and 0.61 again under rocprofv3; no extra stream, no such
        # dependence on the runtime's queue mapping.  (The slice lengths above were measured with "own".)
        self.sort_on = "pref"
        self._cur_sorted = self._next_sorted = None
        self._sev_waited = self._sev_waited_cur = None
        self.slice_wait = True      # (False: tools/race_check.py --negative -- nobody waits for the slices: the check must notice)
        self._last_sort_ev = None
        # --evict-victim-cache (main_no_ddp.py:96, parsed and unused by the reference): behind every step's embedding update
        # the trained aux rows of the batch's misses go back to their host rows and to their copies among the window's victim
        # rows (ops.victim_writeback).  One rank only; the step then runs un-pipelined (no take of the next batch ahead of
        # this batch's write-back, no launch tape) -- an optional mode, not the timed path.  Run() also plans the next window
        # at the boundary instead of in the background: the plan's row gather has to see the window's last write-back.
        self.evict_victim = False
        self._vwb_work = {}
        self._gslot = None
        self._res = self._next_res = None
        self._tapes = {}
        self.tape_fallbacks = []                # why a recorded step could not become a native tape (should stay empty)
        # schedule knobs (attributes, not environment switches: tests/test_engine_parity.py runs each of them both ways)
        self.use_tape = S.is_hip(self.dev)      # replay recorded launch sequences (_step_taped)
        self.native_tape = True                 # ... from the C side (csrc/tape.hip) instead of from Python
        # ... in lanes: the training queue's calls by this thread, each side queue's by a helper thread of the library.  At a
        # per-rank batch of 1024 the thread that issues a step's ~45 runtime calls, not the GPU, sets the step time
        self.tape_lanes = 3
        # ... below this local batch only: from it on the GPU needs several times the host's issue time per step and a
        # single issuing thread keeps the program order of the tape (A / B at c3: 0.6331 one lane, 0.6396 three)
        self.tape_lanes_below = 4096
        # cross-stream events of the step complete WITH the kernel they follow (attached to its launch) instead of being
        # recorded behind it: no marker packet, no bubble on the training queue
        self.attach_events = True
        self.fold_top_wait = True               # short batches: the wait for the deferred top-MLP update rides on the side stream
        self.chain_take = True                  # long batches: the next batch's take rides behind the embedding update
        self.fuse_sgd = True                    # one rank: the dense SGD rides in the weight gradients' reduction pass
        # one rank, long batches: where the top MLP's weight gradients start -- "top_dz" (behind the top MLP's input-gradient
        # chain, beside the interaction backward and the bottom MLP's backward), "interacted", "bot_dz" (behind the bottom MLP's
        # input-gradient chain) or "bot_wg" (behind the bottom MLP's weight gradients: they then run into the next step's
        # bottom MLP, gather and interaction forward, which leave the MFMA idle)
        self.top_wgrad_after = "top_dz"
        # (Round 5, measured and removed: the weight gradients of the top MLP's LAST TWO layers started behind the first GEMM of
        #  the input-gradient chain -- their dZ are final behind the head kernel, that GEMM is the last reader of their weights --
        #  to use the MFMA capacity the chain leaves: c3 0.5679 against 0.5646 ms, c5 3.788 against 3.719, per-rank 4096 a tie.
        #  Like every placement beside that chain since round 1, it costs the chain more than it takes off the second half.)
        # Criteo layout + dot interaction: the gather IS the interaction's operand load (cdlrm_gather_interact_fwd / _bwd) -- the
        # [B, T, D] block between cached EmbeddingBag and interact_features is neither written nor read back (c3: 109 MB + 113 MB
        # per step), the backward reads the rows again from the cache, in front of the batch's embedding update.  Bit-identical
        # to the two operators (False: gather + interaction as two launches; multi-hot bags and "cat" always take those).
        self.fuse_gather = True
        # fused gather, a knob measured and left OFF (round 5): the sort of the batch's slot ids (the embedding backward's prepare;
        # side stream) started BEHIND the interaction forward -- an event attached to that launch -- instead of at the head of
        # the step (two aux regions) or at the tail of the previous one (chained take), so that it runs under the top MLP's
        # forward.  tools/ab_step.py, same box, six rounds: c3 0.5646 against 0.5632 ms (chained take), 0.5660 against 0.5625 (two
        # aux regions), per-rank 4096 0.3337 / 0.3306, 1024 0.1787 / 0.1776 -- slower everywhere: the first half of the step is
        # the training queue's GEMM chain alone, and the sort takes its CUs.  (A third placement -- the next batch's sort behind
        # this batch's embedding update in the step's tail, two aux regions -- measured 0.5669 against 0.5643, 0.1844 / 0.1810 at
        # 1024, and was removed.)
        self.sort_after_fwd = False
        # (Also measured and removed: the sort in FRONT of the side stream's wait for the previous top-MLP update, two aux
        #  regions: c3 0.5625 against 0.5608, 2048 0.2392 / 0.2378, 1024 0.1801 / 0.1787, only 4096 gained, 0.3304 / 0.3328.)
        # WindowResolver hands the NEXT step a look-ahead chunk to resolve (`_pending_resolve`, taken when `mark_next` is set): the
        # step issues it on the prefetch stream right behind its interaction forward -- in front of its own weight gradients on that
        # stream --, so the resolve (random 128-B tag reads) runs beside the top MLP's GEMMs, which leave HBM idle, instead of
        # behind the weight gradients at the end of the step, where it lands on the next step's gather (the roofline kernel)
        # local batches from this size on PLACE the resolve (either take schedule); below it a chunk is small and is issued at once
        self.place_resolve_min = 8192
        self.mark_next = False
        self._mark_this = False
        self._fwd_marked = False
        self._pending_resolve = None

    # all Linear weights in one flat buffer (one all-reduce, main_no_ddp.py:234-247), biases behind them
    def _flatten_params(self):
        """All Linear weights in one flat buffer, biases behind them.  The first top-MLP layer reads the interaction
        output R, whose width D + npairs (479 at c3) is not a multiple of 4: its weight is stored with the row pitch
        rounded up to 4 (zero pad column, R gets a matching zero column) so that layer runs on the 16-byte-load GEMM
        path; `layer.weight` stays a [out, in] view of that storage."""
        lin = _linears(self.dlrm.bot_l) + _linears(self.dlrm.top_l)
        top0 = _linears(self.dlrm.top_l)[0]
        kp = {l: ((l.in_features + 3) // 4 * 4 if l is top0 else l.in_features) for l in lin}
        nw = sum(l.out_features * kp[l] for l in lin)
        nb = sum(l.bias.numel() for l in lin)
        self.param_flat = torch.zeros(nw + nb, dtype=torch.float32, device=self.dev)
        self.grad_flat = torch.zeros(nw + nb, dtype=torch.float32, device=self.dev)
        self.n_weight = nw
        nbot = len(_linears(self.dlrm.bot_l))
        nw_bot = sum(l.out_features * kp[l] for l in lin[:nbot])
        nb_bot = sum(l.bias.numel() for l in lin[:nbot])
        # (weights offset, count, biases offset, count) of the bottom / top MLP inside the flat buffers
        self.rng_bot = (0, nw_bot, nw, nb_bot)
        self.rng_top = (nw_bot, nw - nw_bot, nw + nb_bot, nb - nb_bot)
        # the weight-gradient ranges the two exchanges at world > 1 carry (top MLP's, bottom MLP's)
        self._grad_views = (self.grad_flat[nw_bot:nw], self.grad_flat[0:nw_bot])
        off_w, off_b = 0, nw
        self.W, self.gW, self.gb = {}, {}, {}
        for l in lin:
            n = l.out_features * kp[l]
            Wp = self.param_flat[off_w:off_w + n].view(l.out_features, kp[l])
            Wp[:, :l.in_features].copy_(l.weight.data)
            self.W[l] = Wp                                          # what the kernels see: [out, kp] contiguous
            l.weight.data = Wp[:, :l.in_features]
            self.gW[l] = self.grad_flat[off_w:off_w + n].view(l.out_features, kp[l])
            l.weight.grad = self.gW[l][:, :l.in_features]
            off_w += n
            m = l.bias.numel()
            self.param_flat[off_b:off_b + m].copy_(l.bias.data)
            l.bias.data = self.param_flat[off_b:off_b + m]
            self.gb[l] = self.grad_flat[off_b:off_b + m]
            l.bias.grad = self.gb[l]
            off_b += m
        self.r_width = kp[top0]

    def _issue_resolve(self, pr, rec, main, placed: bool):
        """One look-ahead chunk's resolve on the prefetch stream (WindowResolver prepares `pr`).  placed: from inside a step,
        behind its interaction forward (an event recorded on the training queue: one marker packet per chunk, i.e. per 16 steps)."""
        pst, ev = self.pref, self._events
        if placed:
            if not self._fwd_marked:            # (else: fwd_mark completes with the interaction forward's launch)
                rec(ev["fwd_mark"].record, main)
            rec(pst.wait_event, ev["fwd_mark"])
        elif pr["first"]:
            pst.wait_stream(main)               # the commit (tags, victims) is on the main stream
        # the ring slot's previous chunk (this window's c - 3, or a chunk of the previous window): its takes are all issued, on
        # the side stream (chained / in-line takes) or on this very stream (two-phase takes: in order)
        rec(ev["res_slot"].record, self.side)
        rec(pst.wait_event, ev["res_slot"])
        if self.multi:
            # ... and the row merge's deadline pass (main stream, _pump_start) has read the slot ids it wanted from it
            rec(pst.wait_event, ev["tier_marked"])
        ops.window_resolve(self.ctx, pr["cols"], pr["lbs"], pr["ws"], pr["wsrc"], stream=pst, batch_len=pr["B"])
        if S.is_hip(self.dev):
            ops.event_record(pr["ev"], pst)       # (a library call: on a launch tape the slot's event handle is a cell)
        else:
            pr["ev"].record(pst)

    def flush_pending_resolve(self):
        """Issue a handed-over chunk at once (its batches are needed before a step could place it)."""
        pr, self._pending_resolve = self._pending_resolve, None
        self.mark_next = False
        if pr is not None:
            self._issue_resolve(pr, lambda fn, *a: fn(*a), S.current_stream(self.dev), placed=False)

    def sort_stream(self, local_batch: int = 0):
        """Where a look-ahead chunk's slot lists are sorted (WindowResolver.ensure_sorted): `sort_on`.  (The side stream under the
        chained take, long batches, instead of the prefetch stream: c5 3.645 against 3.614 ms -- no gain, one rule for all.)"""
        if self.sort_on == "side":
            return self.side
        if self.sort_on == "pref":
            return self.pref
        return S.low_priority_stream(self.dev)

    def _fused_gather(self, lS_o) -> bool:
        """This step's gather rides in the interaction kernels (fuse_gather)."""
        return bool(self.fuse_gather and lS_o is None and not self.cat and S.is_hip(self.dev)
                    and ops.gather_interact_supported(self.ctx))

    def _side_gather(self, B: int) -> bool:
        """Short local batches run the gather on the side stream (see _fwd_bwd)."""
        return B < self.gather_alone_min and not (self.defer_top and self.cat)

    def _chain(self, B: int, next_idx, lS_o) -> bool:
        """Long batches on the window-resident probe: the next batch's take rides behind this step's embedding update on
        the side stream (see _fwd_bwd)."""
        return (self.chain_take and not self._side_gather(B) and next_idx is not None and self._next_res is not None
                and lS_o is None)

    def _reduce_avg(self) -> bool:
        """grad /= W followed by all-reduce(SUM) (main_no_ddp.py:239-244) as ONE all-reduce(AVG): on RCCL, for a
        power-of-two world size -- dividing by 2, 4, 8 is exact in fp32 and commutes with every rounding of the sum, so the
        result is bit-identical while one kernel launch per exchange disappears.  Other sizes / backends (gloo in the
        tests) keep the reference's two steps."""
        r = getattr(self, "_avg_ok", None)
        if r is None:
            W = self.world
            r = (W & (W - 1)) == 0 and dist.get_backend(self.pg) == "nccl"
            self._avg_ok = r
        return r

    def _buffers(self, B):
        if B in self._bufs:
            return self._bufs[B]
        dev, D, F = self.dev, self.D, self.F
        f32 = torch.float32
        npairs = F * (F + 1) // 2 if self.itself else F * (F - 1) // 2
        b = dict()
        # one scratch row behind the batch in the feature blocks: the extra bag of squared-off ragged inputs
        # (square_bags) pools into it; its gradient row stays zero
        if self.cat:
            # "cat" interaction (model_no_ddp.py:297-299): R = cat([x] + ly) IS the feature block the bottom MLP and
            # the gather write into -- no interaction kernel, no copy
            assert self.r_width >= F * D
            b["R"] = torch.zeros(B + 1, self.r_width, dtype=f32, device=dev)[:B]
            b["dR"] = torch.zeros(B + 1, self.r_width, dtype=f32, device=dev)[:B]
            b["feat"] = b["R"].as_strided((B, F, D), (self.r_width, D, 1))
            b["dfeat"] = b["dR"].as_strided((B, F, D), (self.r_width, D, 1))
        else:
            b["feat"] = torch.zeros(B + 1, F, D, dtype=f32, device=dev)[:B]
            b["dfeat"] = torch.zeros(B + 1, F, D, dtype=f32, device=dev)[:B]
            assert self.r_width >= D + npairs
            b["R"] = torch.zeros(B, self.r_width, dtype=f32, device=dev)      # pad column (if any) stays zero
            b["dR"] = torch.zeros(B, self.r_width, dtype=f32, device=dev)
        b["Zc"] = torch.empty(B, 1, dtype=f32, device=dev) if self.loss_threshold > 0.0 else None
        # activations / gradients of the hidden layers
        b["bot_y"] = [torch.empty(B, l.out_features, dtype=f32, device=dev) for l, _ in self.bot[:-1]]
        b["bot_dy"] = [torch.empty(B, l.out_features, dtype=f32, device=dev) for l, _ in self.bot[:-1]]
        b["top_y"] = [torch.empty(B, l.out_features, dtype=f32, device=dev) for l, _ in self.top]
        b["top_dy"] = [torch.empty(B, l.out_features, dtype=f32, device=dev) for l, _ in self.top]
        b["loss"] = torch.zeros(65, dtype=f32, device=dev)
        work = 0
        for l, _ in self.bot + self.top:
            work = max(work, ops.linear_bwd_work(B, l.out_features, self.W[l].shape[1], dev).numel())
        work = max(work, ops.mlp_wgrad_work(B, [l.out_features for l, _ in self.bot + self.top],
                                            [self.W[l].shape[1] for l, _ in self.bot + self.top], dev).numel())
        b["lin_work"] = torch.empty(work, dtype=torch.uint8, device=dev)
        # weight/bias gradients of all layers are taken at the end of the backward, from the pre-activation
        # gradients the dgrad chain leaves in these buffers (one grouped launch at small batches)
        layers = [l for l, _ in self.bot + self.top]
        x0 = torch.empty(B, self.W[layers[0]].shape[1], dtype=f32, device=dev)   # stand-in: re-pointed at X every step
        xs = [x0] + b["bot_y"] + [b["R"]] + b["top_y"][:-1]
        dzs = b["bot_dy"] + [b["dfeat"][:, 0, :]] + b["top_dy"]
        b["wgrad"] = ops.WgradPlan(xs, dzs, [self.gW[l] for l in layers], [self.gb[l] for l in layers], b["lin_work"])
        # long local batches: the top MLP's weight gradients run on their own stream beside the interaction backward
        # and the bottom MLP's backward -- (bottom plan, top plan), each with its own scratch
        b["wgrad_split"] = None
        if self.defer_top or (S.is_hip(dev) and B >= self.split_wgrad_min):
            nb = len(self.bot)
            gw = [self.gW[l] for l in layers]
            gb = [self.gb[l] for l in layers]
            wk = lambda ls: ops.mlp_wgrad_work(B, [l.out_features for l in ls], [self.W[l].shape[1] for l in ls], dev)
            b["wgrad_split"] = (ops.WgradPlan(xs[:nb], dzs[:nb], gw[:nb], gb[:nb], wk(layers[:nb])),
                                ops.WgradPlan(xs[nb:], dzs[nb:], gw[nb:], gb[nb:], wk(layers[nb:])))
            # one rank: nothing sits between a sub-network's weight gradients and its SGD step, so the step rides in the
            # gradients' reduction pass (cdlrm_mlp_wgrad_sgd) -- one launch less per sub-network and step
            b["wgrad_split"][0].set_params([self.W[l] for l in layers[:nb]], [l.bias.data for l in layers[:nb]])
            b["wgrad_split"][1].set_params([self.W[l] for l in layers[nb:]], [l.bias.data for l in layers[nb:]])
        self._bufs[B] = b
        return b

    def _probe_bufs(self, n, which):
        """(slots, miss_pos, miss_count) of one pipeline stage, allocated once: they are written and read on side
        streams, where the caching allocator's per-stream reuse rules would not protect per-call temporaries."""
        key = ("probe", n, which)
        if key not in self._bufs:
            i32 = torch.int32
            self._bufs[key] = (torch.empty(self.T, n, dtype=i32, device=self.dev),
                               torch.empty(self.T, n, dtype=i32, device=self.dev),
                               torch.empty(self.T, dtype=i32, device=self.dev))
        return self._bufs[key]

    def _emb_work(self, n):
        key = ("emb", n)
        if key not in self._bufs:
            self._bufs[key] = ops.embbag_bwd_work(self.ctx, n, self.dev)
        return self._bufs[key]

    # ----------------------------------------------------------------------------------------------
    def sync_touched_to_rank0(self):
        """The reference refills rank 0's cache and then broadcasts EVERY cache table from rank 0
        (main_no_ddp.py:318-319), which also overwrites whatever the other ranks changed since the last
        table-agg merge.  Same result without moving 10.9 GB: only the rows some rank touched since that merge can
        differ, so broadcast exactly those (union of the touched flags), leaving the flags set for the next merge."""
        if not self.multi:
            return
        ctx = self.ctx
        self._pump_finish()         # the refill that follows reads and evicts cache rows: the last merge lands first
        self._agg_alloc()
        self._agg_flags.copy_(self.cg.touched)
        dist.all_reduce(self._agg_flags, op=dist.ReduceOp.MAX, group=self.pg)
        U = self._agg_list(self._agg_flags)
        if U == 0:
            return
        buf = self._agg_buffer(U)
        ops.agg_gather(ctx, self.agg_rows, self.agg_count, 1.0, buf, U)
        dist.broadcast(buf, src=0, group=self.pg)
        ops.agg_scatter(ctx, self.agg_rows, self.agg_count, buf, U)

    def _agg_alloc(self):
        if self.agg_rows is None:
            self.agg_rows = torch.empty(self.ctx.total_rows, dtype=torch.int64, device=self.dev)
            self.agg_count = torch.zeros(1, dtype=torch.int64, device=self.dev)
            self.agg_count_host = S.pinned(torch.zeros(1, dtype=torch.int64), self.dev)
            self._agg_flags = torch.zeros(self.ctx.total_rows, dtype=torch.uint8, device=self.dev)
            self._agg_counted = S.new_event(self.dev)
            self._agg_buf = None

    def _agg_list(self, flags) -> int:
        """flags -> sorted row list (device) and its length on the host.  The host waits for THIS copy only (an event
        behind the compaction), not for the stream: kernels queued behind it keep the GPU busy meanwhile."""
        ops.agg_compact(self.ctx, flags, self.agg_rows, self.agg_count)
        self.agg_count_host.copy_(self.agg_count, non_blocking=True)
        self._agg_counted.record(S.current_stream(self.dev))
        self._agg_counted.synchronize()
        return int(self.agg_count_host[0])

    def _agg_buffer(self, U: int) -> torch.Tensor:
        """[>= U, D] exchange buffer, kept between merges (grown by doubling: no allocation in a steady-state merge)."""
        if self._agg_buf is None or self._agg_buf.shape[0] < U:
            cap = 1 << max(12, (U - 1).bit_length())
            self._agg_buf = torch.empty(min(cap, self.ctx.total_rows), self.D, dtype=torch.float32, device=self.dev)
        return self._agg_buf[:U]

    def step(self, X: torch.Tensor, lS_i: torch.Tensor, T: torch.Tensor, lS_o: Optional[torch.Tensor] = None,
             j: Optional[int] = None, gather_events: Optional[list] = None, next_idx: Optional[torch.Tensor] = None,
             res=None, next_res=None, loss_sync: bool = True):
        """One training iteration on this rank's slice (next_idx: the NEXT batch's indices, if it belongs to the same
        window: its tag probe and aux fill are then issued behind this step's embedding backward).  X [B, m_den] fp32, lS_i [T, n] int64, T [B, 1] fp32, all
        on the device; lS_o None = Criteo layout (one lookup per bag), else int64 [T, B] offsets -- or [T, B + 1] from
        square_bags() for ragged multi-hot tables; j = batch number inside the epoch (table-agg schedule).
        res / next_res: WindowResolver.batch(j) / .batch(j + 1) -- the window-resident probe: this batch's (the next
        batch's) slot ids and miss sources were resolved once for the whole window, the per-step tag probe shrinks to
        cdlrm_embbag_take.
        loss_sync=False: the loss buffer is completed OFF the training queue -- the head kernel leaves per-workgroup partial
        sums, and the one-workgroup launch that adds them up (and accumulates `stat_acc`) runs on the side stream in front of
        the embedding backward instead of holding the training queue for ~6 us; the returned buffer and `stat_acc` are then
        valid after finish() (bench.py, Run), not right behind step() on the current stream.
        Returns the device loss buffer (element 0 = the loss, 1 = correct predictions, 2 = loss * batch)."""
        B, n = X.shape[0], lS_i.shape[1]
        if self.evict_victim:
            assert not self.multi, "--evict-victim-cache is defined for one rank (every rank would write its own misses' rows)"
            next_idx = next_res = None          # the next batch's take follows this batch's write-back
        self.loss_sync = bool(loss_sync) or not self.fused_head
        self._mark_this = bool(self.mark_next and self._pending_resolve is not None and lS_o is None)
        self.mark_next = False
        self._res, self._next_res = res, (next_res if next_idx is not None else None)
        if lS_o is not None:
            assert lS_o.shape[1] in (B, B + 1) and (not self.multi or lS_o.shape[1] == B)
        if j is None:
            j = self.iter
        # (the row merge at the end of a table-agg step leaves aux rows alone -- the backward never flags them --, so the
        #  next batch's probe / aux fill may run ahead across it like across any other step)
        sgd_done = False
        # bench.py: HIP timing events of the roofline kernel -- a pre-created (start, stop) pair of ops.TimingEvent, or a list
        # that receives a new pair.  They are attached to the gather's launch (cdlrm_ctx_time_next_gather); on a tape the
        # pair's handles are cells.
        if gather_events is None:
            self._gslot = None
        elif isinstance(gather_events, tuple):
            self._gslot = gather_events
        else:
            self._gslot = (ops.TimingEvent(), ops.TimingEvent())
            gather_events.append(self._gslot)
        if self._pump is not None:
            # the deadlines of a pending merge were computed for the batches FOLLOWING the merge step in its window: a step that
            # is not the expected next batch of that window (another resolver, a jump) gets the whole merge first
            p, pos = self._pump, (res[3] if (res is not None and len(res) > 3) else None)
            if pos is None or pos[0] is not p["rs"] or pos[1] != p["j0"] + (self.iter - p["step0"]):
                self._pump_finish()
        if self._pump is not None:
            self._pump_wait()       # rows of an earlier merge that THIS step uses have landed
        self._cur_sorted = self._next_sorted = None
        if lS_o is None and not self.evict_victim:
            if res is not None and len(res) > 3:
                self._cur_sorted = res[3][0].sorted_views(res[3][1])
            if self._next_res is not None and len(next_res) > 3:
                self._next_sorted = next_res[3][0].sorted_views(next_res[3][1])
                if self._next_sorted is not None and self._next_sorted[4] is not self._sev_waited and self.slice_wait:
                    # the next batch's take (two aux regions: prefetch stream; chained / single region: side stream) waits for
                    # its slice's sort: this step's successor -- gather, interaction backward (the once-only flags), embedding
                    # update -- is ordered behind that take
                    if self.ctx.aux_phases >= 2 and not self._chain(B, next_idx, lS_o):
                        self.pref.wait_event(self._next_sorted[4])
                    else:
                        # (not on both: with two aux regions the `gathered` record THIS step's fused forward waits for sits on the
                        #  side stream -- a wait for a sort that is still running would hold the training queue, as in round 6's
                        #  first placement of the sort)
                        self.side.wait_event(self._next_sorted[4])
                    self._sev_waited = self._next_sorted[4]
            pf = self._pref
            if (self._cur_sorted is not None and pf is not None and pf["ptr"] == lS_i.data_ptr()
                    and pf["shape"] == tuple(lS_i.shape) and not pf.get("sorted_ok")):
                # this batch's take was issued by the previous step, BEFORE its slice's sort (a chunk resolved late: short chunks):
                # nothing orders this step behind that sort -- it sorts its own slots
                self._cur_sorted = None
            if self._cur_sorted is not None and self._cur_sorted[4] is not self._sev_waited_cur and self.slice_wait:
                # this batch's take, if no earlier step issued it, runs in line on the side stream; its gather follows that stream
                self.side.wait_event(self._cur_sorted[4])
                self._sev_waited_cur = self._cur_sorted[4]
        if res is not None:
            # an in-line take (no prefetched result for this batch) runs on the side stream: behind the chunk's resolve
            self.side.wait_event(res[2])
        if self._next_res is not None and next_res[2] is not (res[2] if res is not None else None):
            # the next batch's take may run on the side stream too (single aux region / chained take): behind ITS chunk
            self.side.wait_event(next_res[2])
        if self.use_tape and lS_o is None and not self.evict_victim:
            sgd_done = self._step_taped(X, lS_i, T, next_idx)
        else:
            sgd_done = self._fwd_bwd(X, lS_i, T, lS_o, gather_events, next_idx)
        if self.evict_victim:
            # on the side stream, behind the embedding update _fwd_bwd issued there and in front of the next step's take
            if n not in self._vwb_work:
                self._vwb_work[n] = ops.victim_writeback_work(self.ctx, n)
            ops.victim_writeback(self.ctx, lS_i, self._last_slots, res[1] if (res is not None and lS_o is None) else None,
                                 self._last_phase, self._vwb_work[n], stream=self.side)
        if self._mark_this:
            self._pending_resolve = None        # issued by this step
        # ---- dense gradient exchange + SGD ----
        if self.multi and self.defer_top:
            # two exchanges, issued in the same order on every rank: the top MLP's (its weight gradients were launched
            # on the side stream right after the top dgrad chain) runs beside the rest of this step's backward and the
            # head of the next step; the bottom MLP's is the only one on the critical path
            wst, W = self.wst, float(self.world)
            gt, gb = self._grad_views
            avg = self._reduce_avg()
            with S.on_stream(wst):
                if avg:
                    dist.all_reduce(gt, op=dist.ReduceOp.AVG, group=self.pg)
                else:
                    ops.scale_div(gt, W, stream=wst)             # layer.weight.grad /= world (:239); biases untouched
                    dist.all_reduce(gt, op=dist.ReduceOp.SUM, group=self.pg)
                ops.sgd_step2(self.param_flat, self.grad_flat, *self.rng_top, self.lr, stream=wst)
                self._events["top_updated"].record(wst)
            if avg:
                dist.all_reduce(gb, op=dist.ReduceOp.AVG, group=self.pg)
            else:
                ops.scale_div(gb, W)
                dist.all_reduce(gb, op=dist.ReduceOp.SUM, group=self.pg)
            if next_idx is None or not (self._side_gather(B) and self.ctx.aux_phases >= 2):
                S.current_stream(self.dev).wait_event(self._events["emb_done"])
            ops.sgd_step2(self.param_flat, self.grad_flat, *self.rng_bot, self.lr)
            sgd_done = True
        elif self.multi:
            gw = self.grad_flat[:self.n_weight]
            if self._reduce_avg():
                dist.all_reduce(gw, op=dist.ReduceOp.AVG, group=self.pg)
            else:
                ops.scale_div(gw, float(self.world))             # layer.weight.grad /= world (:239); biases untouched
                dist.all_reduce(gw, op=dist.ReduceOp.SUM, group=self.pg)
            # the join _fwd_bwd left out: the embedding backward / sparse SGD on the side stream ran beside the
            # all-reduce (the reference overlaps optimizer_embeds.step() with it the same way, :412-414)
            S.current_stream(self.dev).wait_event(self._events["emb_done"])
        if not sgd_done:
            ops.sgd_step(self.param_flat, self.grad_flat, self.lr)
        # ---- periodic cache-row merge (main_no_ddp.py:417-423) ----
        if j is None:
            j = self.iter
        if self.multi and j > 0 and j % self.agg_freq == 0:
            # (one rank averages with itself, W[u] = W[u] / 1: nothing to do, and no flags are kept at world == 1)
            if self._emb_done is not None:      # the merge reads (and the flag reset races with) this step's row updates
                S.current_stream(self.dev).wait_event(self._emb_done)
            self.table_aggregate()
            # the next step's gather may run on the side stream: it has to see the merged rows
            self.side.wait_stream(S.current_stream(self.dev))
        elif self._pump is not None:
            self._pump_advance()    # the rows the NEXT step needs, and this step's share of the rest
        self.iter += 1
        return self._buffers(B)["loss"]

    def _fwd_bwd(self, X, lS_i, T, lS_o, gather_events, next_idx=None):
        """Forward + backward of one iteration up to: all dense gradients in grad_flat, cache rows updated."""
        ctx, cg = self.ctx, self.cg
        B = X.shape[0]
        n = lS_i.shape[1]
        buf = self._buffers(B)
        feat, dfeat, R, dR = buf["feat"], buf["dfeat"], buf["R"], buf["dR"]
        F, D = self.F, self.D
        # ---- forward ----
        # tag probe + aux-row fill (PCIe) on the side stream, under the bottom MLP -- or already done: the previous
        # step issues them for this batch behind its embedding backward (software-pipelined across iterations)
        # Stream/event calls go through `rec` so that a recorded step (see _step_taped) replays them too; the events
        # are persistent engine objects (a wait always sees the latest record, whichever tape issued it).
        main = S.current_stream(self.dev)
        side = self.side
        ev = self._events
        two_phase = ctx.aux_phases >= 2
        # Long batches with the window-resident probe: the next batch's take costs ~10 us stand-alone, so it no longer needs
        # a stream and an aux region of its own (that pipeline hid 250 us of PCIe reads).  It follows this step's embedding
        # update in order on the side stream, and ONE event recorded there -- after the take and, on one rank, after the
        # deferred top-MLP update -- is all the next step's gather waits for: the main queue carries one wait per step
        # instead of three (probe, top update, embedding update) plus a record, each a 6-8 us bubble (measured).
        chain = self._chain(B, next_idx, lS_o)
        if chain:
            two_phase = False
        if self.defer_top and self.cat:
            # the previous step's top weight gradients read R = the feature block this step's first kernels overwrite
            rec(main.wait_event, ev["top_updated"])
        pref, self._pref = self._pref, None
        top_waited = prepared = False
        if pref is not None and pref["ptr"] == lS_i.data_ptr() and pref["shape"] == tuple(lS_i.shape):
            slots, miss_pos, miss_count, probed = pref["res"]
            self._phase = pref["phase"]
            top_waited = bool(pref.get("chained_top"))
            prepared = bool(pref.get("prepared"))
        else:
            rec(side.wait_stream, main)
            if self._res is not None and lS_o is None:
                slots, miss_pos, miss_count = self._probe_bufs(n, self._phase)
                ops.embbag_take(ctx, lS_i, self._res[0], self._res[1], slots, aux_phase=self._phase, stream=side)
            else:
                slots, miss_pos, miss_count = ops.embbag_probe(ctx, lS_i, stream=side, aux_phase=self._phase,
                                                               out=self._probe_bufs(n, self._phase))
            probed = ev["probed_inline"]
            rec(probed.record, side)
        n_bags = B if lS_o is None else lS_o.shape[1]
        self._last_slots, self._last_phase = slots, self._phase          # (ops.victim_writeback reads them behind the step)

        fused = self._fused_gather(lS_o)

        def gather(st):
            if fused:       # the rows are the interaction kernel's operand loads: nothing to launch here
                return
            if self._gslot is not None:         # bench.py: the roofline kernel's own start / stop timestamps
                ops.time_next_gather(ctx, self._gslot[0], self._gslot[1])
            ops.embbag_fwd(ctx, slots, lS_o, feat[:, 1:, :], feat.stride(0), D, n_bags=n_bags, stream=st)

        # Short local batches: the gather (6 us at 1024) goes to the SIDE stream, beside the bottom MLP's forward -- it
        # needs the probe result and the previous step's embedding update, which ran on that very stream (in order: no
        # event), not the bottom MLP.  Long batches: it is the HBM-roofline kernel and runs alone on the main stream.
        side_gather = self._side_gather(B)
        if side_gather:
            rec(side.wait_event, probed)
            if self.fold_top_wait and self.defer_top and not self.cat and not top_waited:
                # the previous step's deferred top-MLP update is waited for HERE, on the side stream in front of the gather:
                # the one wait the training queue has in front of the interaction (`gathered`) then covers it too -- one
                # barrier packet less on that queue.  (The update has landed long before the embedding update the gather
                # follows in order.)
                rec(side.wait_event, ev["top_updated"])
                top_waited = True
            gather(side)
            rec(ev["gathered"].record, side)
        cur = X
        bot_acts = [X]
        for i, (l, act) in enumerate(self.bot):
            y = feat[:, 0, :] if i == len(self.bot) - 1 else buf["bot_y"][i]
            ops.linear_fwd(cur, self.W[l], l.bias.data, y, act, alone=self.wide_gemm_bottom)
            bot_acts.append(y)
            cur = y
        if side_gather:
            rec(main.wait_event, ev["gathered"])
        else:
            # the gather runs alone on the main stream (it is the HBM-roofline kernel: nothing competes for bandwidth)
            rec(main.wait_event, probed)
            gather(main)
        if next_idx is not None and two_phase:
            # Software pipelining across iterations: the NEXT batch's tag probe and aux-row fill (~0.25 ms of PCIe
            # reads at c3) start right behind this batch's gather, on their own stream, into the OTHER aux region
            # (this batch still reads and updates its own aux rows).  The other region was last used by the previous
            # batch: its embedding update must have landed (emb_done).
            pst = self.pref
            if B >= self.gather_alone_min:
                # long batches: the gather is the HBM-roofline kernel and runs alone; the probe starts behind it
                ev_g = ev["gathered"]
                rec(ev_g.record, main)
                rec(pst.wait_event, ev_g)
            # short batches: every event recorded on the main queue costs a ~6 us bubble there (measured), more than the
            # probe could take from a 6 us gather -- it starts as soon as the other aux region is free
            if self._emb_done is not None:
                rec(pst.wait_event, self._emb_done)
            ph = 1 - self._phase
            if self._next_res is not None:
                res = self._probe_bufs(n, ph)
                ops.embbag_take(ctx, next_idx, self._next_res[0], self._next_res[1], res[0], aux_phase=ph, stream=pst)
            else:
                res = ops.embbag_probe(ctx, next_idx, stream=pst, aux_phase=ph, out=self._probe_bufs(n, ph))
            evp = ev["probed"][ph]
            rec(evp.record, pst)
            self._pref = dict(ptr=next_idx.data_ptr(), shape=tuple(next_idx.shape), phase=ph,
                              res=(res[0], res[1], res[2], evp), sorted_ok=self._next_sorted is not None)
        # the backward's sort of the slot ids needs nothing but the probe result: run it on the side, under the MLPs
        emb_work = self._emb_work(n)
        if not side_gather:
            rec(side.wait_event, probed)
        attach = self.attach_events and S.is_hip(self.dev)
        # the sort behind the interaction forward (sort_after_fwd): issued below, behind that launch
        defer_sort = bool(self.sort_after_fwd and fused and attach and not prepared)
        # sv: this batch's slot lists were sorted with its look-ahead chunk (WindowResolver.ensure_sorted): no sort here, and the
        # once-only slots can be updated by the interaction backward (it reads the sort's flags)
        # (the flags of a batch's OWN sort would cost the training queue a wait for the side stream in front of the interaction
        #  backward: measured 21 us, three times what the folding saves)
        sv = self._cur_sorted if (lS_o is None and n == B) else None
        once = bool(self.fuse_once and fused and sv is not None)
        if sv is not None:
            defer_sort = False
        elif not prepared and not defer_sort:     # (prepared: a chained take sorted this batch's slots right behind itself)
            ops.embbag_bwd_prepare(ctx, slots, emb_work, stream=side)
        if self.defer_top and not self.cat and not top_waited:
            # the previous step's top-MLP update (weight gradients read R / top_y / top_dy, then all-reduce and SGD on
            # the side stream) has to have landed before this step overwrites those buffers and reads the weights
            # (top_waited: the event this step's gather waited for was recorded behind that update)
            rec(main.wait_event, ev["top_updated"])
        self._fwd_marked = False
        if fused:
            if self._gslot is not None:         # bench.py: the kernel that does the gather, timed by its own launch
                ops.time_next_gather(ctx, self._gslot[0], self._gslot[1])
            if attach and (defer_sort or self._mark_this):
                # what the side queues start behind the interaction forward waits for an event that completes WITH this launch
                ops.event_attach_next(ev["fwd_mark"], main)
                self._fwd_marked = True
            ops.gather_interact_fwd(ctx, slots, feat[:, 0, :], self.itself, R)
            if defer_sort:
                rec(side.wait_event, ev["fwd_mark"])
                ops.embbag_bwd_prepare(ctx, slots, emb_work, stream=side)
        elif not self.cat:
            ops.interact_fwd(feat, self.itself, R)
        if self._mark_this:
            self._issue_resolve(self._pending_resolve, rec, main, placed=True)
        cur = R
        top_acts = [R]
        fused_head = self.fused_head
        for i, (l, act) in enumerate(self.top):
            y = buf["top_y"][i]
            if not (fused_head and i == len(self.top) - 1):
                ops.linear_fwd(cur, self.W[l], l.bias.data, y, act, alone=self.wide_gemm)
            top_acts.append(y)
            cur = y
        Z = cur
        # ---- backward ----
        # No stand-alone activation-backward pass: every gradient buffer holds the PRE-activation gradient of its
        # layer.  The loss kernel applies the last sigmoid's derivative, each dgrad GEMM applies the derivative of
        # the activation that produced its input (x_act) in its epilogue, the interaction backward does the same
        # for the bottom MLP's output, and the bias gradients are column sums taken inside the wgrad GEMMs.
        last_act = self.top[-1][1]
        split = buf["wgrad_split"]
        attach = self.attach_events and S.is_hip(self.dev)
        nb_, wst = len(self.bot), self.wst

        n_top = len(self.top)
        if fused_head:
            # last layer + loss + the layer's input gradient in one launch
            l = self.top[-1][0]
            dX = dR if n_top == 1 else buf["top_dy"][-2]
            ops.head_fwd_bwd(top_acts[-2][:, :l.in_features], self.W[l], l.bias.data, T, Z, buf["top_dy"][-1],
                             dX[:, :l.in_features], buf["loss"], self._head_scratch,
                             x_act=(self.top[-2][1] if n_top > 1 else 0), kind=self.loss_kind,
                             weights=self.loss_weights, threshold=self.loss_threshold, Zc=buf["Zc"], finish=False)
            if self.loss_sync:      # partial sums -> loss, running statistics: here, or on the side stream below
                ops.head_finish(self._head_scratch, B, buf["loss"], acc=self.stat_acc)
            dY = dX
        else:
            ops.loss_fwd_bwd(Z, T, buf["loss"], buf["top_dy"][-1], kind=self.loss_kind, weights=self.loss_weights,
                             threshold=self.loss_threshold, Zc=buf["Zc"], sigmoid_bwd=(last_act == 2))
            rec(self.stat_acc.add_, buf["loss"][1:3])
            dY = buf["top_dy"][-1]
        # where the top MLP's weight gradients start (one rank: a knob; several ranks: behind the input-gradient chain, their
        # exchange follows them)
        late = "top_dz"
        if split is not None and self.defer_top and not self.multi and not self.cat:
            late = self.top_wgrad_after
            if late == "bot_dz" and len(self.bot) < 2:
                late = "interacted"
        for i in reversed(range(n_top - 1 if fused_head else n_top)):
            l, act = self.top[i]
            if i == len(self.top) - 1 and act == 2:
                act = 0                                      # already applied by the loss kernel
            elif i < len(self.top) - 1:
                act = 0                                      # applied by the dgrad epilogue of layer i+1
            dX = dR if i == 0 else buf["top_dy"][i - 1]
            if i == 0 and attach and split is not None and late == "top_dz":
                # `top_dz` (every top-layer dZ is final: the weight gradients may start) completes WITH the chain's last GEMM
                # -- attached to its launch instead of recorded behind it: a record is a marker packet of its own and left
                # a 6-8 us bubble on the training queue
                ops.event_attach_next(ev["top_dz"], main)
            ops.linear_bwd(top_acts[i], self.W[l], top_acts[i + 1], dY, dX, None, None, act,
                           buf["lin_work"], x_act=(self.top[i - 1][1] if i > 0 else 0), alone=self.wide_gemm)
            dY = dX
        def top_wgrad(after):
            # Every top-layer dZ is final: the top MLP's weight gradients run on their own stream, beside the bottom MLP's
            # backward.  (Launching each layer's weight gradient as soon as ITS dZ exists -- beside the dgrad chain itself
            # -- measured slower: 0.810 vs 0.782 ms at B=8192; the chain is the critical path and loses CUs to them.)
            rec(wst.wait_event, after)
            fused = self.defer_top and not self.multi and self.fuse_sgd
            ops.mlp_wgrad(split[1], stream=wst, lr=self.lr if fused else None)
            if not self.defer_top:
                rec(ev["wgrad_done"].record, wst)
            elif not self.multi:
                if not fused:
                    ops.sgd_step2(self.param_flat, self.grad_flat, *self.rng_top, self.lr, stream=wst)
                rec(ev["top_updated"].record, wst)

        # (One event on the main queue for both side streams -- recorded behind the interaction backward -- instead of one in
        #  front of it for the weight gradients and one behind it for the embedding backward measured slower at c3, 0.698 vs
        #  0.675 ms: the saved bubble is worth less than the 57 us the weight gradients start later.  The interaction backward
        #  split by rows -- the dense feature's row as its own launch, the rest on the side queue -- measured slower too, 0.718
        #  vs 0.663 ms.  Both schedules were removed in round 3.)
        if split is not None and late == "top_dz":
            if not (attach and n_top - (1 if fused_head else 0) > 0):
                rec(ev["top_dz"].record, main)
            top_wgrad(ev["top_dz"])
        if self.cat:
            # dR is the gradient of the feature block itself; only the bottom MLP's output needs its activation's
            # derivative (the dot path applies it in the interaction backward's epilogue)
            ops.act_bwd(dfeat[:, 0, :], feat[:, 0, :], self.bot[-1][1])
        else:
            if attach:
                ops.event_attach_next(ev["interacted"], main)      # completes with the interaction backward's launch
            if once:
                ops.gather_interact_bwd_sgd(ctx, slots, feat[:, 0, :], dR, self.itself, dfeat, sv[2], sv[3], self.lr_embeds,
                                            x_act=self.bot[-1][1])
            elif fused:
                ops.gather_interact_bwd(ctx, slots, feat[:, 0, :], dR, self.itself, dfeat, x_act=self.bot[-1][1])
            else:
                ops.interact_bwd(feat, dR, self.itself, dfeat, x_act=self.bot[-1][1])
        # embedding backward + sparse SGD on a side stream, overlapped with the bottom-MLP backward and the
        # gradient all-reduce (the reference overlaps optimizer_embeds.step() with the all-reduce, :412-414)
        if not attach or self.cat:
            rec(ev["interacted"].record, main)
        rec(side.wait_event, ev["interacted"])
        if late == "interacted":
            top_wgrad(ev["interacted"])
        if fused_head and not self.loss_sync:
            # the head's partial sums -> loss buffer + running statistics, off the training queue.  The next head kernel
            # overwrites the partials only behind the next gather, which is ordered behind this stream's embedding update
            ops.head_finish(self._head_scratch, B, buf["loss"], stream=side, acc=self.stat_acc)
        if sv is not None:
            ops.embbag_bwd_apply_sorted(ctx, n, dfeat[:, 1:, :], dfeat.stride(0), D, self.lr_embeds, emb_work, sv[0], sv[1], sv[3],
                                        self._phase, once, cg.touched if self.multi else None, stream=side)
        else:
            ops.embbag_bwd_apply(ctx, n, lS_o, dfeat[:, 1:, :], dfeat.stride(0), D, self.lr_embeds, emb_work,
                                 cg.touched if self.multi else None, stream=side)
        emb_done = ev["emb_done"]
        rec(emb_done.record, side)
        self._emb_done = emb_done
        if next_idx is not None and not two_phase:
            # single aux region: the next batch's fill can only follow this batch's embedding update
            which = 2 + (self.iter & 1)
            if self._next_res is not None:
                res = self._probe_bufs(n, which)
                ops.embbag_take(ctx, next_idx, self._next_res[0], self._next_res[1], res[0], aux_phase=0, stream=side)
            else:
                res = ops.embbag_probe(ctx, next_idx, stream=side, out=self._probe_bufs(n, which))
            chain_sort = chain and not (self.sort_after_fwd and fused and attach) and self._next_sorted is None
            if chain_sort:
                # ... and the sort of the next batch's slot ids for ITS backward: here it ends well before the step does; issued
                # at the head of the next step it shared HBM with that step's gather (the roofline kernel).  (sort_after_fwd: the
                # next step issues it behind its interaction forward instead)
                ops.embbag_bwd_prepare(ctx, res[0], emb_work, stream=side)
            # (weight gradients that start later are waited for by the next step's training queue, in front of its
            #  interaction forward: they may run beside the next gather)
            chained_top = chain and self.defer_top and not self.multi and split is not None and late == "top_dz"
            if chained_top:
                rec(side.wait_event, ev["top_updated"])     # recorded above, behind this step's top-MLP SGD
            evp = ev["probed"][which]
            rec(evp.record, side)
            self._pref = dict(ptr=next_idx.data_ptr(), shape=tuple(next_idx.shape), phase=0,
                              res=(res[0], res[1], res[2], evp), chained_top=chained_top, prepared=chain_sort,
                              sorted_ok=self._next_sorted is not None)
        dY = dfeat[:, 0, :]
        for i in reversed(range(1, len(self.bot))):         # layer 0 has no input gradient
            l, act = self.bot[i]
            dX = buf["bot_dy"][i - 1]
            if i == 1 and late == "bot_dz" and attach:
                ops.event_attach_next(ev["bot_dz"], main)
            ops.linear_bwd(bot_acts[i], self.W[l], bot_acts[i + 1], dY, dX, None, None, 0,
                           buf["lin_work"], x_act=self.bot[i - 1][1])
            dY = dX
        if late == "bot_dz":
            if not attach:
                rec(ev["bot_dz"].record, main)
            top_wgrad(ev["bot_dz"])
        sgd_included = False
        if split is not None:
            split[0].set_x(0, X)
            fused = self.defer_top and not self.multi and self.fuse_sgd
            ops.mlp_wgrad(split[0], lr=self.lr if fused else None)
            if not self.defer_top:
                rec(main.wait_event, ev["wgrad_done"])
            elif not self.multi:
                if not fused:
                    ops.sgd_step2(self.param_flat, self.grad_flat, *self.rng_bot, self.lr)
                sgd_included = True
            if late == "bot_wg":
                rec(ev["bot_wg"].record, main)
                top_wgrad(ev["bot_wg"])
        else:
            plan = buf["wgrad"]
            plan.set_x(0, X)
            ops.mlp_wgrad(plan)
        if self.multi:
            return False # step() joins AFTER it has issued the gradient all-reduce: the exchange overlaps the embedding update
        if next_idx is None:
            rec(main.wait_stream, side)      # full join
        elif chain:
            pass                             # the next gather waits for the take's event, recorded behind emb_done
        elif not (side_gather and two_phase):
            rec(main.wait_event, emb_done)   # cache rows are updated; the prefetched probe keeps running
        # else: the next step's gather runs on the side stream, in order behind this embedding update, and its probe waits
        # for emb_done on its own stream -- nothing on the main stream reads the cache rows before the next full join
        # (window boundary, row merge, evaluate(), finish()), so the main queue is spared one more wait (6-8 us bubble)
        return sgd_included

    # ----------------------------------------------------------------------------------------------
    def evaluate(self, X: torch.Tensor, lS_i: torch.Tensor, lS_o: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Forward only (the test loop of main_no_ddp.py:479-494: `cache_group(lS_o, lS_i, emb_tables, rank)` then
        `dlrm(X, lookups)` under no_grad): tag probe with the aux-miss path (test indices outside the cache read their
        host rows -- into the aux region of the batch trained last, whose rows are dead), cached gather, bottom MLP,
        interaction, top MLP.  Returns Z [B, 1] (a buffer reused by the next call)."""
        ctx = self.ctx
        B, n = X.shape[0], lS_i.shape[1]
        assert n <= ctx.aux, "test batch larger than the aux table (test_mini_batch_size <= aux_table_size)"
        if self._pump is not None:
            # Rows of the last merge are still on their way, and landing them means COLLECTIVES.  The test loop runs on rank 0
            # only (main_no_ddp.py:478-494): a rank that issued the remaining exchanges from here would issue them alone, in
            # other pieces and at another place of the communicator's order than its peers (a hang, or other ranks' gradients
            # reduced into rows).  Every rank calls drain_merge() at the same step first -- Run does, in front of its rank-0
            # test block.
            raise RuntimeError("evaluate(): a row merge is still draining; every rank has to call drain_merge() first")
        self.finish()
        pend = self._pref
        if pend is not None and pend["phase"] == self._phase:
            # the next training batch's take already filled THIS aux region (single-region / chained take: the long-batch
            # default): wait for it, let the test batch overwrite the region, and drop the prefetched result -- the next
            # step() then takes (and sorts) its batch again in line
            S.current_stream(self.dev).wait_event(pend["res"][3])
            self._pref = None
        buf = self._buffers(B)
        feat, R = buf["feat"], buf["R"]
        F, D = self.F, self.D
        slots, _, _ = ops.embbag_probe(ctx, lS_i, aux_phase=self._phase, out=self._probe_bufs(n, "eval"))
        cur = X
        for i, (l, act) in enumerate(self.bot):
            y = feat[:, 0, :] if i == len(self.bot) - 1 else buf["bot_y"][i]
            ops.linear_fwd(cur, self.W[l], l.bias.data, y, act)
            cur = y
        if self._fused_gather(lS_o):
            ops.gather_interact_fwd(ctx, slots, feat[:, 0, :], self.itself, R)
        else:
            ops.embbag_fwd(ctx, slots, lS_o, feat[:, 1:, :], feat.stride(0), D, n_bags=(B if lS_o is None else lS_o.shape[1]))
            if not self.cat:
                ops.interact_fwd(feat, self.itself, R)
        cur = R
        for i, (l, act) in enumerate(self.top):
            y = buf["top_y"][i]
            ops.linear_fwd(cur, self.W[l], l.bias.data, y, act, alone=self.wide_gemm)
            cur = y
        if self.loss_threshold > 0.0:       # DLRM_Net.forward returns the clamped prediction (model_no_ddp.py:311-314)
            cur = torch.clamp(cur, min=self.loss_threshold, max=1.0 - self.loss_threshold)
        return cur

    def drain_merge(self):
        """Land every row of a merge that is still travelling in deadline order (MergePump).  COLLECTIVE: call it on every rank
        at the same step, or on none -- in front of anything only some ranks do next (the rank-0 test loop, a checkpoint)."""
        self._pump_finish()

    def finish(self):
        """Order the current stream behind a deferred top-MLP update (defer_top_update): call before reading the top
        MLP's weights, gradients or activation buffers outside step() / evaluate().  Collective while a row merge is
        draining (drain_merge): every rank calls it at the same step, as Run does at print boundaries and at the end."""
        if self.defer_top:
            S.current_stream(self.dev).wait_event(self._events["top_updated"])
        if self._emb_done is not None:          # a pipelined short-batch step leaves the embedding update un-joined
            S.current_stream(self.dev).wait_event(self._emb_done)
        self._pump_finish()                     # rows of the last merge that are still on their way

    def prediction(self, B: int) -> torch.Tensor:
        """Z of the last step at batch size B as DLRM_Net.forward returns it (clamped under --loss-threshold)."""
        buf = self._buffers(B)
        return buf["Zc"] if buf["Zc"] is not None else buf["top_y"][-1]

    def _step_taped(self, X, lS_i, T, next_idx):
        """The same launch sequence as _fwd_bwd, replayed from a recording.  At small local batches the ~30 launches
        of a step take less GPU time than the Python around them (argument marshalling, stream lookups): the first
        step of each control path (batch shape, aux phase, prefetched or in-line probe, next batch handed over or
        not) runs _fwd_bwd under `_lib.start_recording`, later ones re-issue the recorded (function, arguments)
        list.  Pointers that change from step to step (X, T, the index tensors) are shared ctypes cells patched before
        each replay; everything else the step touches is a preallocated buffer with a fixed address."""
        import ctypes as C
        B, n = X.shape[0], lS_i.shape[1]
        main = S.current_stream(self.dev)
        pref = self._pref
        hit = pref is not None and pref["ptr"] == lS_i.data_ptr() and pref["shape"] == tuple(lS_i.shape)
        phase = pref["phase"] if hit else self._phase
        nxt = next_idx is not None
        key = (B, n, main.cuda_stream, hit, phase, nxt, self._emb_done is not None, X.stride(0), lS_i.stride(0),
               next_idx.stride(0) if nxt else 0,
               (self.iter & 1) if (self.ctx.aux_phases < 2 or self._chain(B, next_idx, None)) else 0,
               bool(hit and pref.get("chained_top")), bool(hit and pref.get("prepared")),
               self._gslot is not None, self.loss_sync,
               (int(self._pending_resolve["cols"].shape[1]), self._pending_resolve["cols"].stride(0)) if self._mark_this else None,
               self.tape_lanes, self.tape_lanes_below, self.attach_events, self.fold_top_wait, self.top_wgrad_after,
               self.fuse_gather, self.sort_after_fwd, self.wide_gemm, self.wide_gemm_bottom, self.fuse_once,
               None if self._cur_sorted is None else self._cur_sorted[3], self._next_sorted is not None,
               self._res[0].stride(0) if (self._res is not None and not hit) else 0,
               self._next_res[0].stride(0) if self._next_res is not None else 0)
        tape = self._tapes.get(key)
        if tape is None:
            calls = []
            _lib.start_recording(calls)
            try:
                included = self._fwd_bwd(X, lS_i, T, None, None, next_idx)
                if not self.multi and not included:   # no gradient exchange in between: the dense SGD rides on the tape too
                    ops.sgd_step(self.param_flat, self.grad_flat, self.lr)
            finally:
                _lib.stop_recording()
            # pointer arguments equal to one of the per-step tensors become shared cells
            cells = {"X": C.c_void_p(X.data_ptr()), "T": C.c_void_p(T.data_ptr()), "idx": C.c_void_p(lS_i.data_ptr())}
            if nxt:
                cells["next"] = C.c_void_p(next_idx.data_ptr())
            if self._res is not None and not hit:
                cells["ws"] = C.c_void_p(self._res[0].data_ptr())
                cells["wsrc"] = C.c_void_p(self._res[1].data_ptr())
            if self._next_res is not None:
                cells["nws"] = C.c_void_p(self._next_res[0].data_ptr())
                cells["nwsrc"] = C.c_void_p(self._next_res[1].data_ptr())
            if self._cur_sorted is not None:
                sv = self._cur_sorted
                cells["skeys"], cells["smeta"], cells["sonce"] = C.c_void_p(sv[0]), C.c_void_p(sv[1]), C.c_void_p(sv[2])
            if self._gslot is not None:
                cells["g0"], cells["g1"] = C.c_void_p(self._gslot[0].handle), C.c_void_p(self._gslot[1].handle)
            if self._mark_this:             # the placed resolve's chunk: index columns, ring slot, the slot's event
                pr = self._pending_resolve
                cells["rcols"], cells["rws"] = C.c_void_p(pr["cols"].data_ptr()), C.c_void_p(pr["ws"].data_ptr())
                cells["rwsrc"], cells["rev"] = C.c_void_p(pr["wsrc"].data_ptr()), C.c_void_p(int(pr["ev"].cuda_event))
            by_value = {c.value: c for c in cells.values()}
            if len(by_value) != len(cells):
                return not self.multi      # aliased inputs: stay on the untaped path
            prog = []
            for fn, args in calls:
                if getattr(fn, "argtypes", None) is not None:
                    args = tuple(by_value.get(a, a) if isinstance(a, int) and not isinstance(a, bool) else a for a in args)
                    prog.append((fn, args, True))
                else:
                    prog.append((fn, args, False))
            post = self._pref
            native = None
            if self.native_tape and _lib.native_tape_ok():
                # the same calls as a C-side tape: one library call per step instead of ~45 interpreted ones (0.22 ms of
                # host time per step, more than the GPU needs at a per-rank batch of 1024)
                try:
                    lanes = self.tape_lanes if B < self.tape_lanes_below else 1
                    native = _lib.NativeTape(prog, cells, main_stream=main.cuda_stream if lanes > 1 else None, max_lanes=lanes)
                except _lib.TapeUnsupported as e:
                    native = None
                    self.tape_fallbacks.append(str(e))      # this control path replays from Python (bench.py reports it)
            self._tapes[key] = dict(prog=prog, cells=cells, phase=self._phase, native=native,
                                    pref=None if post is None else (post["phase"], post["res"], post.get("chained_top", False),
                                                                    post.get("prepared", False)))
            return not self.multi
        cells = tape["cells"]
        cells["X"].value = X.data_ptr()
        cells["T"].value = T.data_ptr()
        cells["idx"].value = lS_i.data_ptr()
        if nxt:
            cells["next"].value = next_idx.data_ptr()
        if "ws" in cells:
            cells["ws"].value = self._res[0].data_ptr()
            cells["wsrc"].value = self._res[1].data_ptr()
        if "nws" in cells:
            cells["nws"].value = self._next_res[0].data_ptr()
            cells["nwsrc"].value = self._next_res[1].data_ptr()
        if "skeys" in cells:
            sv = self._cur_sorted
            cells["skeys"].value, cells["smeta"].value, cells["sonce"].value = sv[0], sv[1], sv[2]
        if "g0" in cells:
            cells["g0"].value, cells["g1"].value = self._gslot[0].handle, self._gslot[1].handle
        if "rcols" in cells:
            pr = self._pending_resolve
            cells["rcols"].value, cells["rws"].value = pr["cols"].data_ptr(), pr["ws"].data_ptr()
            cells["rwsrc"].value, cells["rev"].value = pr["wsrc"].data_ptr(), int(pr["ev"].cuda_event)
        bufs = self._buffers(B)
        (bufs["wgrad_split"][0] if bufs["wgrad_split"] is not None else bufs["wgrad"]).set_x(0, X)
        if tape["native"] is not None:
            rc = tape["native"].replay()
            if rc:
                _lib.check(rc)
        else:
            for fn, args, is_lib in tape["prog"]:
                rc = fn(*args)
                if is_lib and rc:
                    _lib.check(rc)
        # the state _fwd_bwd leaves behind
        self._phase = tape["phase"]
        self._emb_done = self._events["emb_done"]
        self._pref = None
        if tape["pref"] is not None:
            self._pref = dict(ptr=next_idx.data_ptr(), shape=tuple(next_idx.shape), phase=tape["pref"][0], res=tape["pref"][1],
                              chained_top=tape["pref"][2], prepared=tape["pref"][3], sorted_ok=self._next_sorted is not None)
        return not self.multi

    # ---- the touched-row merge in deadline order ----------------------------------------------------------------------
    # broadcast_and_aggregate (main_no_ddp.py:250-292) replaces every row any rank touched since the last merge by its mean
    # (max, sum) over the ranks, at step j, before step j + 1 runs.  What step j + 1 can observe of that is the rows IT uses;
    # a row nobody uses before step j + d may be merged any time before step j + d -- it does not change in between on any
    # rank.  The window's resolved slot ids (WindowResolver.lookahead: every rank's lookups of the next batches) say which
    # rows the next K steps use: the merge's row list is sorted by the first batch that needs a row (classes: batch 1, 2, 3-4,
    # 5-8, 9-16, 17-32, 33-64, later = "cold", due at batch K + 1) and exchanged in that order on the exchange stream -- gather, all-
    # reduce, scatter per chunk -- a few chunks per step, while the training steps run; a step waits for the chunk that holds
    # the last row it needs.  At c3 (100 steps x 8192 lookups x 26 tables, Zipf 1.05: 1.86 M rows, 0.95 GB) the next batch
    # needs 2.3 % of the merge's rows, the next 16 batches 13 %.  Same values as the merge in one piece: every rank builds the
    # same list from the same flags and the same window, and a row's gather reads what the one-piece merge would have read.
    MERGE_CLASS_FIRST = (1, 2, 3, 5, 9, 17, 33)     # first batch (after the merge step) of each deadline class; then: cold
    MERGE_LOOKAHEAD = 64                            # batches ahead the classes reach (the resolver decides how many are known)

    def _pump_start(self, U, buf, scale, rop) -> bool:
        pos = self._res[3] if (self._res is not None and len(self._res) > 3) else None
        if pos is None:
            return False
        rs, j = pos
        la = rs.lookahead(j, self.MERGE_LOOKAHEAD)
        if not la:
            return False            # nothing known about the next batch (end of the window): the merge in one piece
        ctx, main = self.ctx, S.current_stream(self.dev)
        first = self.MERGE_CLASS_FIRST
        C = len(first) + 1
        K = la[-1][0]
        if ("agg_tier",) not in self._bufs:
            self._bufs[("agg_tier",)] = (torch.empty(ctx.total_rows, dtype=torch.uint8, device=self.dev),
                                         torch.empty(ctx.total_rows, dtype=torch.int64, device=self.dev),
                                         torch.zeros(C + 1, dtype=torch.int64, device=self.dev),
                                         S.pinned(torch.zeros(C + 1, dtype=torch.int64), self.dev))
        tier, rows_sorted, off_dev, off_host = self._bufs[("agg_tier",)]
        tier.fill_(C - 1)
        seen = set()
        for k, ws, ev in la:
            if id(ev) not in seen:
                main.wait_event(ev)
                seen.add(id(ev))
        for c in reversed(range(C - 1)):        # latest deadline first: the earliest class that names a row wins
            hi = first[c + 1] if c + 1 < len(first) else self.MERGE_LOOKAHEAD + 1
            for k, ws, ev in la:
                if first[c] <= k < hi:
                    ops.agg_mark_tier(ctx, ws, c, tier)
        ops.agg_split(ctx, self.agg_rows, U, tier, C, rows_sorted, off_dev)
        self._events["tier_marked"].record(main)
        off_host.copy_(off_dev, non_blocking=True)
        self._agg_counted.record(main)
        self._agg_counted.synchronize()
        off = [int(x) for x in off_host.tolist()]
        ready = S.new_event(self.dev)
        ready.record(main)
        # (the budget cuts the exchange into pieces: it has to be the same number on every rank -- the resolver's lbs =
        #  ceil(B / world), not this rank's slice width, which is shorter on the last rank when world does not divide B)
        budget = int(self.merge_budget_rows)
        if self.merge_budget_auto:
            budget = max(budget, int(rs.lbs) * self.T)
        self._pump = dict(rows=rows_sorted, U=U, buf=buf, scale=scale, rop=rop, off=off, K=K, step0=self.iter, issued=0,
                          waited=0, chunks=[], ready=ready, budget=budget, rs=rs, j0=j)
        self._pump_advance()
        return True

    def _pump_need(self, d: int) -> int:
        """Rows of the pending merge that have to have landed before the d-th step after it."""
        p = self._pump
        if d > p["K"]:
            return p["U"]
        need = 0
        for c, f in enumerate(self.MERGE_CLASS_FIRST):
            if f <= d:
                need = p["off"][c + 1]
        return need

    def _pump_issue(self, lo: int, hi: int):
        p, ctx, comm = self._pump, self.ctx, self.comm
        with S.on_stream(comm):
            if not p["chunks"]:
                comm.wait_event(p["ready"])
            ops.agg_gather(ctx, p["rows"][lo:], self.agg_count, p["scale"], p["buf"][lo:hi], hi - lo, first=lo)
            dist.all_reduce(p["buf"][lo:hi], op=p["rop"], group=self.pg)
            ops.agg_scatter(ctx, p["rows"][lo:], self.agg_count, p["buf"][lo:hi], hi - lo, first=lo)
            ev = S.new_event(self.dev)
            ev.record(comm)
        p["chunks"].append((hi, ev))
        p["issued"] = hi

    def _pump_advance(self, everything: bool = False):
        """Issue the chunks the NEXT step needs, plus `merge_budget_rows` of the rest (everything: all of it)."""
        p = self._pump
        if p is None:
            return
        d_next = self.iter + 1 - p["step0"]
        target = p["U"] if everything else max(self._pump_need(d_next), min(p["U"], p["issued"] + p["budget"]))
        ch = max(1, int(self.agg_chunk_rows))
        while p["issued"] < target:
            self._pump_issue(p["issued"], min(target, p["issued"] + ch))

    def _pump_wait(self, everything: bool = False):
        """The training (and the side) stream wait for the chunk that holds the last row this step needs."""
        p = self._pump
        need = p["U"] if everything else self._pump_need(self.iter - p["step0"])
        if need > p["waited"]:
            assert p["issued"] >= need, "merge rows this step needs were never issued"
            for hi, ev in p["chunks"]:
                if hi >= need:
                    S.current_stream(self.dev).wait_event(ev)
                    self.side.wait_event(ev)
                    p["waited"] = hi
                    break
        if p["waited"] >= p["U"]:
            self._pump = None

    def _pump_finish(self):
        if self._pump is not None:
            self._pump_advance(everything=True)
            self._pump_wait(everything=True)
            self._pump = None

    def table_aggregate(self):
        """broadcast_and_aggregate (main_no_ddp.py:250-292): average the rows any rank touched since the last
        merge.  Slot ids are global, so the union of touched rows is an all-reduce(MAX) of the flag bytes (the reference
        all-gathers the slot lists and takes torch.unique per table); the rows then travel as ONE compacted [U, D]
        buffer, in chunks: while chunk i is reduced over xGMI on the exchange stream, chunk i+1 is gathered and chunk
        i-1 scattered back on the main stream.  No allocation in steady state; the host waits only for the row count."""
        cg, ctx = self.cg, self.ctx
        touched = cg.touched
        if not self.multi:
            touched.zero_()
            return
        main = S.current_stream(self.dev)
        self._pump_finish()         # (a merge period shorter than the look-ahead: the previous merge's rows land first)
        dist.all_reduce(touched, op=dist.ReduceOp.MAX, group=self.pg)
        self._agg_alloc()
        U = self._agg_list(touched)                   # consumes (clears) the flags
        if U == 0:
            return
        buf = self._agg_buffer(U)
        if self.agg_op == "mean":
            scale, rop = float(self.world), dist.ReduceOp.SUM
        elif self.agg_op == "sum":
            scale, rop = 1.0, dist.ReduceOp.SUM
        elif self.agg_op == "max":
            scale, rop = 1.0, dist.ReduceOp.MAX
        else:
            raise ValueError(self.agg_op)
        if self.lazy_merge and self.comm is not None and self._pump_start(U, buf, scale, rop):
            return
        ch = self.agg_chunk_rows
        nch = (U + ch - 1) // ch
        if nch == 1 or self.comm is None or not S.is_hip(self.dev):
            ops.agg_gather(ctx, self.agg_rows, self.agg_count, scale, buf, U)
            dist.all_reduce(buf, op=rop, group=self.pg)
            ops.agg_scatter(ctx, self.agg_rows, self.agg_count, buf, U)
            return
        comm = self.comm
        gathered = [S.new_event(self.dev) for _ in range(nch)]
        reduced = [S.new_event(self.dev) for _ in range(nch)]

        def gather(i):
            lo, hi = i * ch, min(U, (i + 1) * ch)
            ops.agg_gather(ctx, self.agg_rows[lo:], self.agg_count, scale, buf[lo:hi], hi - lo, first=lo)
            gathered[i].record(main)

        def reduce(i):
            lo, hi = i * ch, min(U, (i + 1) * ch)
            with S.on_stream(comm):
                comm.wait_event(gathered[i])
                dist.all_reduce(buf[lo:hi], op=rop, group=self.pg)
                reduced[i].record(comm)

        def scatter(i):
            lo, hi = i * ch, min(U, (i + 1) * ch)
            main.wait_event(reduced[i])
            ops.agg_scatter(ctx, self.agg_rows[lo:], self.agg_count, buf[lo:hi], hi - lo, first=lo)

        gather(0)
        for i in range(nch):
            reduce(i)
            if i + 1 < nch:
                gather(i + 1)
            if i >= 1:
                scatter(i - 1)
        scatter(nch - 1)
