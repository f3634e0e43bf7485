"""Host master-table memory for the GPU path: pinned (single process) or shared + registered (one
process per GPU, all mapping the same /dev/shm file, main_no_ddp.py:621-622 `emb_tables.share_memory()`)."""
from __future__ import annotations

import os
from typing import List, Sequence

import numpy as np
import torch

from .model_no_ddp import Embedding_Table_Group


def cpu_share() -> int:
    """CPUs this process may actually use: the cgroup's quota (cpu.max / cfs_quota_us) where there is one, else the affinity
    mask.  `os.cpu_count()` reports the machine (256 on the GPU box whose containers get 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2: "<quota> <period>" or "max <period>"
            q, p = f.read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, p = int(f.read()), int(g.read())
                if q > 0 and p > 0:
                    n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return max(1, n)


def shared_table_dir() -> str:
    """Where the one shared mapping of the host tables lives at world > 1 (a tmpfs: /dev/shm unless CDLRM_SHM_DIR says otherwise)."""
    return os.environ.get("CDLRM_SHM_DIR", "/dev/shm")


def shared_table_dir_free():
    """Bytes free there (None: cannot tell)."""
    try:
        st = os.statvfs(shared_table_dir())
        return int(st.f_bavail) * int(st.f_frsize)
    except OSError:
        return None


def memory_limit():
    """This job's host-memory limit in bytes (cgroup v2 memory.max, else cgroup v1 memory.limit_in_bytes; None: none / unknown)."""
    for path in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            txt = open(path).read().strip()
        except OSError:
            continue
        if txt.isdigit() and int(txt) < (1 << 60):
            return int(txt)
    return None


REPLICA_LIMIT_FRACTION = 0.85       # W private copies + W x pinned staging may take this much of the job's memory limit


def host_tables_mode(table_bytes: int, world: int, *, staging_bytes: int = 0, shm_free="ask", limit="ask"):
    """How `world` ranks of one node get the host tables -- ONE decision for make_host_tables (the run) and bench.py --plan-only
    (the plan), so the two cannot disagree: returns (mode, error) with mode "private" (one rank), "shared" (one tmpfs mapping,
    rank 0 writes evictions back) or "replicas" (a private pinned copy per rank, every rank writes back), error a message when the
    chosen mode cannot hold (else None).  CDLRM_HOST_TABLES = shared | replicas | auto (default) forces or leaves the choice."""
    if world <= 1:
        return "private", None
    forced = os.environ.get("CDLRM_HOST_TABLES", "auto")
    if forced not in ("shared", "replicas", "auto"):
        raise ValueError("CDLRM_HOST_TABLES: shared, replicas or auto")
    if shm_free == "ask":
        shm_free = shared_table_dir_free()
    if limit == "ask":
        limit = memory_limit()
    fits_shm = shm_free is None or shm_free >= table_bytes + (1 << 30)
    mode = forced if forced != "auto" else ("shared" if fits_shm else "replicas")
    err = None
    if mode == "shared" and not fits_shm:
        err = ("the shared host tables (%.0f GB) do not fit %s (%.0f GB free): CDLRM_SHM_DIR=<larger tmpfs>, "
               "CDLRM_HOST_TABLES=replicas, or cap the tables (--max-ind-range)" % (table_bytes / 1e9, shared_table_dir(), shm_free / 1e9))
    if mode == "replicas" and limit and world * (table_bytes + staging_bytes) > REPLICA_LIMIT_FRACTION * limit:
        err = ("%d private copies of the host tables (%.0f GB each, + %.1f GB of pinned staging per rank) exceed %d %% of this job's "
               "memory limit (%.0f GB) and %s holds %.0f GB: CDLRM_SHM_DIR=<larger tmpfs> or cap the tables (--max-ind-range)"
               % (world, table_bytes / 1e9, staging_bytes / 1e9, int(REPLICA_LIMIT_FRACTION * 100), limit / 1e9, shared_table_dir(),
                  (shm_free or 0) / 1e9))
    return mode, err


def fill_uniform_from_device(dst: torch.Tensor, n_rows: int, device, seed: int, chunk_rows: int = 1 << 21):
    """dst[n, m] ~ U(-sqrt(1/n), sqrt(1/n)) (the reference's init distribution, model_no_ddp.py:70-73), drawn on
    the GPU and copied down in chunks: fast enough for the 96 GB Terabyte-shape tables."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    bound = float(np.sqrt(1.0 / n_rows))
    m = dst.shape[1]
    for r0 in range(0, n_rows, chunk_rows):
        r1 = min(n_rows, r0 + chunk_rows)
        t = torch.rand(r1 - r0, m, generator=g, device=device, dtype=torch.float32)
        t.mul_(2 * bound).sub_(bound)
        dst[r0:r1].copy_(t, non_blocking=False)


def make_host_tables(ln_emb: Sequence[int], m_spa: int, *, device, seed: int = 123, rank: int = 0, world: int = 1,
                     shm_name: str = "cdlrm_host_tables", barrier=None) -> Embedding_Table_Group:
    """Synthetic host tables for bench / Run on synthetic data."""
    ln = [int(n) for n in ln_emb]
    eg = Embedding_Table_Group(m_spa, np.array(ln), init="empty_meta")

    def private_copy():
        for k, n in enumerate(ln):
            w = torch.empty(n, m_spa, dtype=torch.float32, pin_memory=True)
            fill_uniform_from_device(w, n, device, seed * 1009 + k)
            eg.emb_l[k].weight.data = w
        eg._pinned = True
        return eg

    if world == 1:
        return private_copy()
    total = sum(ln) * m_spa
    # Two ways to give W ranks the host tables (CDLRM_HOST_TABLES = shared | replicas | auto):
    #   shared   -- ONE tmpfs mapping registered by every rank (the reference's emb_tables.share_memory()); rank 0 writes evictions;
    #   replicas -- every rank pins its OWN copy, filled from the same seed (identical bits), and applies every eviction write-back
    #               to it: the evicted rows are the same on all ranks (the insert is replicated and sync_touched_to_rank0 has just
    #               made every touched row rank 0's), so the copies stay identical.  W x the memory, no tmpfs, NUMA-local reads.
    # auto: shared when the tmpfs has room for the tables, replicas otherwise.  Rank 0 decides, everybody follows.
    import torch.distributed as dist
    mode = ["auto"]
    if rank == 0:           # rank 0 decides (host_tables_mode: the same function bench.py --plan-only prints), everybody follows
        m, err = host_tables_mode(total * 4, world, staging_bytes=int(os.environ.get("CDLRM_PLAN_STAGING_BYTES", "0")))
        mode[0] = ("error: " + err) if err else m
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast_object_list(mode, src=0)
    if mode[0].startswith("error"):
        raise RuntimeError(mode[0])
    if mode[0] == "replicas":
        private_copy()
        eg._replicated = True
        barrier()
        return eg
    # The backing file: created by rank 0 under a name nobody can predict, exclusively (O_EXCL: never an existing file),
    # without following a symlink, readable by the owner only; the other ranks learn the name from rank 0.  (A fixed
    # world-writable path could be pre-created or symlinked by another local user.)
    import secrets
    name = [None]
    if rank == 0:
        name[0] = "%s_%s" % (shm_name, secrets.token_hex(8))
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast_object_list(name, src=0)
    elif world > 1:
        raise RuntimeError("make_host_tables(world > 1) needs an initialised process group to share the table file's name")
    shm_dir = shared_table_dir()
    path = os.path.join(shm_dir, name[0])
    if rank == 0:
        # a tmpfs smaller than the tables does not fail at ftruncate: it kills the process with SIGBUS at the first page it
        # cannot back, in the middle of the fill.  Say it here instead.
        free = shared_table_dir_free()
        if free is not None and free < total * 4 + (1 << 30):
            raise RuntimeError("host tables of %.1f GB do not fit %s (%.1f GB free): point CDLRM_SHM_DIR at a larger tmpfs, or cap "
                               "the tables (--max-ind-range)" % (total * 4 / 1e9, shm_dir, free / 1e9))
        fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_RDWR | getattr(os, "O_NOFOLLOW", 0), 0o600)
        try:
            os.ftruncate(fd, total * 4)
        finally:
            os.close(fd)
    barrier()
    flat = torch.from_file(path, shared=True, size=total, dtype=torch.float32)
    off = 0
    for k, n in enumerate(ln):
        eg.emb_l[k].weight.data = flat[off:off + n * m_spa].view(n, m_spa)
        off += n * m_spa
    eg._flat = flat
    # one registration for the whole mapping
    from . import _lib
    import ctypes as C
    alias = C.c_void_p()
    _lib.check(_lib.lib().cdlrm_host_register(flat.data_ptr(), total * 4, C.byref(alias)))
    if alias.value != flat.data_ptr():
        raise RuntimeError("registered host memory has a different device alias; unsupported")
    eg._registered.append(flat.data_ptr())
    eg._pinned = True
    if rank == 0:
        for k, n in enumerate(ln):
            fill_uniform_from_device(eg.emb_l[k].weight.data, n, device, seed * 1009 + k)
    barrier()
    if rank == 0:
        os.unlink(path)          # the mappings keep the memory alive; nothing is left behind in /dev/shm
    return eg
