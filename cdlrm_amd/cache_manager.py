"""Host-side mirror of the reference's cache_manager.py: the look-ahead Prefetcher.

Reference (cache_manager.py:8-115): an mp.Process that walks a second copy of the data loader, groups
`lookahead` batches into a window, sorts/dedups the window's indices per table on CPU pool workers, gathers those
rows from the host tables and pushes (rows, uniq, map) into `batch_fifo`; a second process applies eviction
write-backs to the host tables.

Here the scan, the row gather and the write-back are HIP kernels (cdlrm_window_unique / cdlrm_gather_rows /
cdlrm_scatter_rows) and the producer is a thread with its own HIP stream in the trainer process, so device
buffers are handed over without IPC.  Names, argument order, window grouping and FIFO protocol are the
reference's.
"""
from __future__ import annotations

import math
import os
import queue
import threading
from typing import List, Optional

import torch

from . import ops


def window_groups(num_batches: int, lookahead: int, cache_workers: int) -> List[List[int]]:
    """Batch ids (per epoch) of every FIFO entry, in order -- the grouping rule of Prefetcher.run
    (cache_manager.py:85-110): flush when j > 0 and collected % (lookahead*cache_workers) == 0 or at the last
    batch; the batch that triggers a flush starts the next group, except the epoch's last batch, which joins the
    group being flushed; a flushed group is cut into slices of `lookahead` batches."""
    limit = lookahead * cache_workers
    out: List[List[int]] = []
    cur: List[int] = []
    collected = 0
    for j in range(num_batches):
        last = j == num_batches - 1
        if (j > 0 and collected % limit == 0) or last:
            if last:
                cur.append(j)
            for p in range(math.ceil(len(cur) / lookahead)):
                out.append(cur[p * lookahead:(p + 1) * lookahead])
            cur = [j]
            collected = 1
        else:
            cur.append(j)
            collected += 1
    return out


class UniqueIndexMap:
    """Stands for the dense inverse map `map[idx] = position in uniq` the reference materialises as an int64
    [max+1, 1] tensor (cache_manager.py:36-40, up to 320 MB per table and window).  uniq is sorted, so the
    position is a binary search; indexing returns the same [k, 1] int64 tensor the reference's map would."""

    def __init__(self, uniq: torch.Tensor):
        self.uniq = uniq

    @property
    def shape(self):
        n = int(self.uniq[-1]) + 1 if self.uniq.numel() else 0
        return torch.Size([n, 1])

    def __getitem__(self, idx: torch.Tensor) -> torch.Tensor:
        idx = idx.to(self.uniq.device)
        pos = torch.searchsorted(self.uniq, idx)
        pos = torch.where((pos < self.uniq.numel()) & (self.uniq[pos.clamp(max=max(self.uniq.numel() - 1, 0))] == idx),
                          pos, torch.full_like(pos, -1))
        return pos.view(-1, 1)


class _Scanner:
    """Per-host-table-group scan context: geometry + plan buffers for cdlrm_window_unique."""

    def __init__(self, table_rows: List[int], dim: int, device: torch.device, max_window: int):
        self.ctx = ops.CacheCtx(table_rows, [1] * len(table_rows), dim, 1, 0, device)
        self.max_window = 0
        self.plan: Optional[ops.WindowPlan] = None
        self._reserve(max_window)

    def _reserve(self, n: int):
        if n > self.max_window:
            self.plan = ops.WindowPlan(self.ctx, n, cap_win=16)
            self.max_window = n


_scanners = {}
_scanner_lock = threading.Lock()


def _scanner_for(emb_tables, device: torch.device, n: int) -> _Scanner:
    key = (id(emb_tables), str(device), threading.get_ident())
    with _scanner_lock:
        sc = _scanners.get(key)
        if sc is None:
            rows = [int(E.weight.shape[0]) for E in emb_tables.emb_l]
            sc = _Scanner(rows, int(emb_tables.emb_l[0].weight.shape[1]), device, n)
            _scanners[key] = sc
    sc._reserve(n)
    return sc


class Prefetcher:
    def __init__(self, args, emb_tables_cpu, batch_fifo, eviction_fifo, finish_event, cache_ld):
        # Shared variables (cache_manager.py:12-18)
        self.args = args
        self.emb_tables_cpu = emb_tables_cpu
        self.batch_fifo = batch_fifo
        self.eviction_fifo = eviction_fifo
        self.finish_event = finish_event
        self.cache_ld = cache_ld
        self._thread: Optional[threading.Thread] = None
        self._evict_thread: Optional[threading.Thread] = None
        self.device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None

    @staticmethod
    def pin_pool(p, core):
        """cache_manager.py:21-25 (taskset via os.system there)."""
        try:
            os.sched_setaffinity(0, {core + 3 + p})
        except OSError:
            pass
        return 1

    @staticmethod
    def process_batch_slice(slice, emb_tables_cpu):
        """cache_manager.py:28-46 -> (rows_per_table, uniq_per_table, map_per_table), all on the device.
        uniq_k = sorted unique of slice[k] (bitmap + popcount scan on the GPU); rows_k = W_host[k][uniq_k] gathered
        by the GPU from the pinned table; map_k answers map_k[idx] -> position in uniq_k."""
        if not torch.cuda.is_available():
            raise RuntimeError("cdlrm_amd: a HIP device is required (no CPU fallback)")
        device = torch.device("cuda", torch.cuda.current_device())
        if isinstance(slice, (list, tuple)):
            slice = torch.stack([torch.as_tensor(s).reshape(-1) for s in slice])
        idx = slice.to(device=device, dtype=torch.int64)
        if idx.stride(-1) != 1:
            idx = idx.contiguous()
        sc = _scanner_for(emb_tables_cpu, device, idx.shape[1])
        sc.plan.unique(idx)
        uo, _, _ = sc.plan.offsets()
        sc.ctx.check()
        lists_of_unique_indices = [sc.plan.uniq[uo[k]:uo[k + 1]].clone() for k in range(len(uo) - 1)]
        unique_indices_maps = [UniqueIndexMap(u) for u in lists_of_unique_indices]
        cached_entries_per_table = emb_tables_cpu.fetch_unique_idx_slices(lists_of_unique_indices)
        return cached_entries_per_table, lists_of_unique_indices, unique_indices_maps

    @staticmethod
    def eviction_manager(emb_tables, eviction_fifo, average_on_writeback, core, timeout):
        """cache_manager.py:49-64: apply queued evictions to the host tables until the queue stays empty for
        `timeout` seconds.  The scatter into the pinned tables is a HIP kernel (rows travel over PCIe)."""
        # The reference pins its eviction PROCESS to `core` (taskset, cache_manager.py:52); here the manager is a thread of the
        # trainer process -- or a direct call --, so the calling thread is pinned for the duration of the loop and gets its mask
        # back afterwards: a caller left on one core would hand that mask to every thread it creates later (an 8-thread torch
        # CPU region spinning on one core runs ~100x slower: seen in the test suite).
        prev = None
        try:
            prev = os.sched_getaffinity(0)
            os.sched_setaffinity(0, {core})
        except OSError:
            prev = None
        ptrs = emb_tables.device_pointers()
        try:
            while True:
                eviction_data = eviction_fifo.get(timeout=timeout) if timeout > 0 else eviction_fifo.get()
                for k, table_eviction_data in enumerate(eviction_data):
                    idxs, embeddings = table_eviction_data[0], table_eviction_data[1]
                    if idxs.numel() == 0:
                        continue
                    dev = embeddings.device if embeddings.is_cuda else torch.device("cuda", torch.cuda.current_device())
                    ops.scatter_rows(ptrs[k], idxs.to(dev, torch.int64).contiguous(),
                                     embeddings.to(dev, torch.float32).contiguous(), average_on_writeback,
                                     distinct=False)
                torch.cuda.synchronize()
                # queue.Queue / Manager().Queue: lets a caller wait until the host tables hold the rows
                # (`eviction_fifo.join()`); the reference offers no such point and reads whatever has landed
                done = getattr(eviction_fifo, "task_done", None)
                if done is not None:
                    done()
        except queue.Empty:
            print('Eviction queue empty longer than expected. Exiting eviction manager...')
        finally:
            if prev is not None:
                try:
                    os.sched_setaffinity(0, prev)
                except OSError:
                    pass

    # -- the producer --------------------------------------------------------------------------------
    def window_slices(self):
        """Yield the [T, n] index tensor of every FIFO entry, epoch by epoch, in the reference's order."""
        args = self.args
        for epoch in range(args.nepochs):
            groups = window_groups(len(self.cache_ld), args.lookahead, args.cache_workers)
            gi, buf = 0, []
            for j, (_, _, sparse_idxs, _) in enumerate(self.cache_ld):
                buf.append(torch.as_tensor(sparse_idxs) if not isinstance(sparse_idxs, (list, tuple))
                           else torch.stack([torch.as_tensor(s).reshape(-1) for s in sparse_idxs]))
                # the groups partition the batch ids in order: emit every group whose last batch has arrived
                while gi < len(groups) and groups[gi][-1] <= j:
                    g = groups[gi]
                    first = j - len(buf) + 1
                    yield torch.cat([buf[b - first] for b in g], dim=1)
                    gi += 1
                    nxt = groups[gi][0] if gi < len(groups) else j + 1
                    buf = buf[nxt - first:]

    def run(self):
        """cache_manager.py:66-115: one (rows, uniq, map) triple per window into batch_fifo, then wait for the
        finish event.  Runs on its own HIP stream so the scan overlaps the trainer's kernels."""
        if self.device is not None:
            torch.cuda.set_device(self.device)
        self._evict_thread = threading.Thread(
            target=Prefetcher.eviction_manager,
            args=(self.emb_tables_cpu, self.eviction_fifo, self.args.average_on_writeback,
                  self.args.main_start_core + 2, self.args.eviction_fifo_timeout), daemon=True)
        self._evict_thread.start()
        stream = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(stream):
            for sl in self.window_slices():
                a = Prefetcher.process_batch_slice(sl, self.emb_tables_cpu)
                stream.synchronize()
                self.batch_fifo.put((a[0], a[1], a[2]))
        self._evict_thread.join()
        self.finish_event.wait()

    # mp.Process-like surface (main_no_ddp.py:637, 646)
    def start(self):
        self._thread = threading.Thread(target=self.run, daemon=True)
        self._thread.start()

    def join(self, timeout=None):
        if self._thread is not None:
            self._thread.join(timeout)
