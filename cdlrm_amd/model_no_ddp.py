"""Host-side mirror of the reference's model_no_ddp.py (same class / method names, argument order and
return structure) over libcdlrm_hip.so.

    Embedding_Table_Group        host master tables (pinned, GPU-visible)      model_no_ddp.py:21-98
    Embedding_Table_Cache_Group  per-GPU N-way set-associative row cache       model_no_ddp.py:101-212
    DLRM_Net                     bottom/top MLP + pairwise-dot interaction     model_no_ddp.py:215-316
    isPrime                                                                    model_no_ddp.py:319-331

What differs, by design: tags live in HBM next to the rows (one flat int64 buffer, one flat
fp32 row buffer for all tables), the per-iteration probe/gather/backward/SGD are single multi-table
launches, and `ly` comes back as views of the [B, T+1, D] interaction operand.
"""
from __future__ import annotations

import sys
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import ops


def isPrime(n):
    """Same quirky trial division as the reference (model_no_ddp.py:319-331): starts at 3, tests
    i*i < n, so 1 and 2 are "not prime" while 4, 6, 8, 9, 25 ... pass.  It fixes the cache geometry."""
    if n == 1 or n == 2:
        return False
    i = 3
    while i * i < n:
        if n % i == 0:
            return False
        i += 1
    return True


# --------------------------------------------------------------------------------------------------
# host master tables
# --------------------------------------------------------------------------------------------------


class _HostTable(nn.Module):
    """emb_l[k] of the host group: `.weight` fp32 [n, m] (the reference keeps an nn.EmbeddingBag here,
    model_no_ddp.py:67-74); calling it is EmbeddingBag(mode="sum") on the host rows."""

    def __init__(self, n, m, weight):
        super().__init__()
        self.num_embeddings, self.embedding_dim = n, m
        self.weight = nn.Parameter(weight, requires_grad=False)

    def forward(self, input, offsets=None):
        return torch.nn.functional.embedding_bag(input, self.weight, offsets, mode="sum")


class Embedding_Table_Group(nn.Module):
    """Full embedding tables in host memory (model_no_ddp.py:21-98).  `emb_l[k].weight` is the fp32
    [n_k, m] master table, initialised U(-sqrt(1/n), sqrt(1/n)) from the numpy global RNG exactly as the
    reference does (:70-73).  `pin()` moves the tables into pinned memory so the HIP kernels read and
    write rows over PCIe without staging copies."""

    def __init__(self, m_spa=None, ln_emb=None, qr_flag=False, qr_operation="mult", qr_collisions=0,
                 qr_threshold=200, md_flag=False, md_threshold=200, init="numpy"):
        super().__init__()
        self._pinned = False
        self._registered: List[int] = []
        if (m_spa is not None) and (ln_emb is not None):
            self.qr_flag = qr_flag
            if self.qr_flag:
                self.qr_collisions, self.qr_operation, self.qr_threshold = qr_collisions, qr_operation, qr_threshold
            self.md_flag = md_flag
            if self.md_flag:
                self.md_threshold = md_threshold
            self.m_spa = m_spa
            self.emb_l = self.create_emb(m_spa, np.asarray(ln_emb), init)

    def create_emb(self, m, ln, init="numpy"):
        emb_l = nn.ModuleList()
        for i in range(0, ln.size):
            n = int(ln[i])
            if self.qr_flag and n > self.qr_threshold:
                # model_no_ddp.py:52-56: tables above the threshold become quotient-remainder pairs (HIP operator,
                # cdlrm_amd/tricks/qr_embedding_bag.py); no numpy draw is consumed for them, as in the reference
                from .tricks.qr_embedding_bag import QREmbeddingBag
                emb_l.append(QREmbeddingBag(n, m, self.qr_collisions, operation=self.qr_operation, mode="sum",
                                            sparse=True))
                continue
            if self.md_flag and n > self.md_threshold:
                # model_no_ddp.py:57-66: m is the per-table width list (md_solver); PrEmbeddingBag of width m[i] projected
                # to max(m), its table drawn from the numpy generator like the plain ones.  Stand-alone HIP operator
                # (cdlrm_amd/tricks/md_embedding_bag.py): a cache row has ONE width, so such a table cannot feed the
                # cache -- no `.weight`, as in the reference.  Dead at the reference's CLI (main_no_ddp.py:612-621 builds
                # the group without md_flag).
                from .tricks.md_embedding_bag import PrEmbeddingBag
                _m, base = int(m[i]), int(max(m))
                EE = PrEmbeddingBag(n, _m, base)
                W = np.random.uniform(low=-np.sqrt(1 / n), high=np.sqrt(1 / n), size=(n, _m)).astype(np.float32)
                EE.embs.weight.data = torch.tensor(W, requires_grad=False)
                emb_l.append(EE)
                continue
            if not isinstance(m, (int, np.integer)):
                # the reference reaches nn.EmbeddingBag(n, <list>) here and raises the same TypeError
                raise TypeError("a plain table needs one integer width, got %r (table %d is not above md_threshold)"
                                % (m, i))
            if init == "numpy":
                W = np.random.uniform(low=-np.sqrt(1 / n), high=np.sqrt(1 / n), size=(n, m)).astype(np.float32)
                Wt = torch.from_numpy(W)
            elif init == "empty":
                Wt = torch.empty(n, m, dtype=torch.float32)
            elif init == "empty_meta":      # storage is attached by the caller (hostmem.make_host_tables)
                Wt = torch.empty(0, m, dtype=torch.float32)
            else:
                raise ValueError(init)
            emb_l.append(_HostTable(n, m, Wt))
        return emb_l

    # -- GPU visibility ---------------------------------------------------------------------------
    def pin(self):
        """Pinned (page-locked, GPU-mapped) copies of the tables; idempotent."""
        if not self._pinned:
            for E in self.emb_l:
                if not hasattr(E, "weight"):
                    continue
                w = E.weight.data
                if not w.is_pinned():
                    p = torch.empty(w.shape, dtype=w.dtype, pin_memory=True)
                    p.copy_(w)
                    E.weight.data = p
            self._pinned = True
        return self

    def register_shared(self):
        """For tables in shared memory (share_memory() / a /dev/shm file mapped by several ranks):
        page-lock them in THIS process and make them GPU-visible (hipHostRegister)."""
        from . import _lib
        import ctypes as C
        for E in self.emb_l:
            w = E.weight.data
            if w.data_ptr() in self._registered or w.is_pinned():
                continue
            alias = C.c_void_p()
            _lib.check(_lib.lib().cdlrm_host_register(w.data_ptr(), w.numel() * 4, C.byref(alias)))
            if alias.value != w.data_ptr():
                raise RuntimeError("registered host memory has a different device alias; unsupported")
            self._registered.append(w.data_ptr())
        self._pinned = True
        return self

    def _plain(self, E, k):
        if not hasattr(E, "weight"):
            # the reference fails the same way: fetch_unique_idx_slices reads E.weight (model_no_ddp.py:84), which a
            # QREmbeddingBag (weight_q / weight_r) or PrEmbeddingBag (embs.weight) does not have -- such tables cannot feed
            # the cache
            raise AttributeError("'%s' object has no attribute 'weight' (table %d is a quotient-remainder / mixed-"
                                 "dimension table; the cached path needs plain host tables)" % (type(E).__name__, k))
        return E

    def device_pointers(self) -> List[int]:
        if not self._pinned:
            self.pin()
        return [int(self._plain(E, k).weight.data.data_ptr()) for k, E in enumerate(self.emb_l)]

    def fetch_unique_idx_slices(self, lists_of_unique_indices):
        """rows[k] = W_host[k][uniq_k] (model_no_ddp.py:80-87), gathered by the GPU straight from the
        pinned tables; returns device tensors."""
        out = []
        ptrs = self.device_pointers()
        for k, uniq in enumerate(lists_of_unique_indices):
            if not uniq.is_cuda:
                uniq = uniq.cuda()
            out.append(ops.gather_rows(ptrs[k], uniq.to(torch.int64).contiguous(), self.emb_l[k].weight.shape[1]))
        return out

    def forward(self, lS_o, lS_i):
        # host-side EmbeddingBag over the master tables (model_no_ddp.py:89-98); not on the cached path
        ly = []
        for k, sparse_index_group_batch in enumerate(lS_i):
            ly.append(self.emb_l[k](sparse_index_group_batch, lS_o[k]))
        return ly


# --------------------------------------------------------------------------------------------------
# the cache
# --------------------------------------------------------------------------------------------------


class _CacheTable:
    """emb_l[k] of the cache group: exposes `.weight` (a view of the flat row buffer) like the
    nn.EmbeddingBag the reference keeps per table (model_no_ddp.py:138)."""

    def __init__(self, group, k):
        self._g, self._k = group, k

    @property
    def weight(self):
        g, k = self._g, self._k       # the reference's [ways*P + aux, m] table (a second aux region may follow it)
        return g.weight.data[g.row_base[k]:g.row_base[k] + g.num_ways * g.cache_sizes[k] + g.aux_table_size]


class LookupList(list):
    """`ly`: a list of T [n_bags, D] tensors (views) that also carries the packed [n_bags, T+1, D]
    interaction operand they live in, so DLRM_Net can skip the torch.cat of model_no_ddp.py:276."""
    packed: Optional[torch.Tensor] = None


class _VictimEntries:
    """victim_cache_entries (model_no_ddp.py:187): (aux_storage_idxs, missing_sparse_idxs) per table,
    materialised on access (it needs the miss counts on the host)."""

    def __init__(self, n):
        self._n = n
        self._last = None

    def _set(self, group, idx, miss_pos, miss_count):
        self._last = (group, idx, miss_pos, miss_count)

    def __len__(self):
        return self._n

    def __getitem__(self, k):
        if self._last is None:
            return None
        g, idx, miss_pos, miss_count = self._last
        m = int(miss_count[k])
        pos = miss_pos[k, :m].long()
        aux = g.cache_sizes[k] * g.num_ways + torch.arange(m, device=pos.device)
        return aux, idx[k][pos]


class _CacheLookupFn(torch.autograd.Function):
    """Autograd glue of the drop-in surface: forward = probe + fused gather (HIP), backward stashes the
    dense gradient of the pooled rows for `CacheSGD.step()` (fused backward + sparse SGD, HIP)."""

    @staticmethod
    def forward(ctx, anchor, group, idx, offsets):
        packed, slots = group._lookup(idx, offsets)
        ctx.group = group
        ctx.slots, ctx.offsets = slots, offsets
        ctx.mark_non_differentiable(slots)
        return packed, slots

    @staticmethod
    def backward(ctx, grad_packed, _):
        ctx.group._pending.append((ctx.slots, ctx.offsets, grad_packed.contiguous()))
        return None, None, None, None


class Embedding_Table_Cache_Group(nn.Module):
    """Per-GPU N-way set-associative cache of embedding rows (model_no_ddp.py:101-212).

    State (reference names kept): `occupancy_tables[k]` int64 [P_k, ways] (-1 = empty),
    `emb_l[k].weight` fp32 [ways*P_k + aux, m]; slot = P_k*way + set, aux slots from P_k*ways.
    Here both are views of two flat device buffers (`tags`, `weight`).
    """

    def __init__(self, m_spa, ln_emb, max_cache_size, aux_table_size, num_ways, cache_init="normal", aux_phases=2,
                 device=None):
        """aux_phases = 2 (default) appends a second aux region to every table so that the fused engine can fill the
        NEXT batch's miss rows while the current batch still trains on its own (engine.py); callers that never pass
        aux_phase see exactly the reference's tables.
        Build-only keywords (not in the reference's signature, model_no_ddp.py:102): cache_init ("normal" = the
        reference's nn.EmbeddingBag default init drawn from the torch CPU generator; "zeros" / "empty" for synthetic
        runs), aux_phases, device (allocate the "zeros" / "empty" buffers directly in HBM instead of moving 10-34 GB
        through host memory with .to(device))."""
        super().__init__()
        self.aux_phases = max(1, int(aux_phases))
        self.ln_emb = np.asarray(ln_emb)
        self.num_ways = int(num_ways)
        self.m_spa = int(m_spa)
        self.aux_table_size = int(aux_table_size)
        self.max_cache_size = self.find_next_prime(max_cache_size)
        self.cache_sizes = [int(n) if int(n) < self.max_cache_size else self.max_cache_size for n in self.ln_emb]
        rows = [self.num_ways * p + self.aux_table_size * self.aux_phases for p in self.cache_sizes]
        self.row_base = [0]
        self.tag_base = [0]
        for k, r in enumerate(rows):
            self.row_base.append(self.row_base[-1] + r)
            self.tag_base.append(self.tag_base[-1] + self.cache_sizes[k] * self.num_ways)
        on_dev = device if (device is not None and cache_init in ("zeros", "empty")) else None
        w = torch.empty(self.row_base[-1], self.m_spa, dtype=torch.float32, device=on_dev)
        if cache_init == "normal":
            # nn.EmbeddingBag default init, table by table, from the torch CPU generator (:138)
            # (only the reference's rows draw from the generator -- the insert's Exp(1) draws come from the same
            #  stream later; the second aux region is scratch)
            w.zero_()
            for k in range(len(rows)):
                ref_rows = self.num_ways * self.cache_sizes[k] + self.aux_table_size
                w[self.row_base[k]:self.row_base[k] + ref_rows].normal_()
        elif cache_init == "zeros":
            w.zero_()
        elif cache_init != "empty":
            raise ValueError(cache_init)
        self.weight = nn.Parameter(w, requires_grad=False)
        self.register_buffer("tags", torch.full((self.tag_base[-1],), -1, dtype=torch.int64, device=on_dev))
        self.emb_l = [_CacheTable(self, k) for k in range(len(rows))]
        self.victim_cache_entries = _VictimEntries(len(rows))
        self._ctx: Optional[ops.CacheCtx] = None
        self._pending = []
        self._anchor = None
        self._bwd_work = {}
        self._touched = None
        self._arange_ok = {}

    # -- reference helpers ------------------------------------------------------------------------
    def find_next_prime(self, max_cache_size):
        for i in range(max_cache_size, 2 * max_cache_size):
            if isPrime(i):
                return i

    def compute_set_indices(self, table_idx, lookup_idxs):
        return torch.remainder(lookup_idxs, self.cache_sizes[table_idx])

    @property
    def occupancy_tables(self):
        return [self.tags[self.tag_base[k]:self.tag_base[k + 1]].view(self.cache_sizes[k], self.num_ways)
                for k in range(len(self.cache_sizes))]

    # -- device plumbing --------------------------------------------------------------------------
    def _apply(self, fn, *a, **kw):
        r = super()._apply(fn, *a, **kw)
        self._ctx = None          # buffers moved: rebind lazily
        self._anchor = None
        return r

    @property
    def ctx(self) -> ops.CacheCtx:
        if self._ctx is None:
            # ops.CacheCtx refuses any non-HIP device: the module must be moved to the MI355X first (.to(rank))
            self._ctx = ops.CacheCtx([int(n) for n in self.ln_emb], self.cache_sizes, self.m_spa, self.num_ways,
                                     self.aux_table_size, self.weight.device, aux_phases=self.aux_phases)
            self._ctx.bind_cache(self.tags, self.weight.data)
        return self._ctx

    def flush_to_host(self, emb_tables: "Embedding_Table_Group", average: bool = False):
        """End-of-run cache flush (the reference only ever writes EVICTED rows back,
        cache_manager.py:57-62): every valid tag's cache row goes to its host row, `W_host[k][tag] = cache row`
        (or the average with average=True, as --average-on-writeback does for evictions).  Aux rows are transient
        copies and are not written.  Synchronises."""
        ptrs = emb_tables.device_pointers()
        for k in range(len(self.cache_sizes)):
            P = self.cache_sizes[k]
            tags = self.occupancy_tables[k].t().reshape(-1)                  # way-major: element w*P + s = slot id
            valid = (tags != -1).nonzero(as_tuple=False).flatten()
            if valid.numel() == 0:
                continue
            rows = self.weight.data[self.row_base[k]:self.row_base[k] + self.num_ways * P].index_select(0, valid)
            ops.scatter_rows(ptrs[k], tags.index_select(0, valid).contiguous(), rows.contiguous(), average)
        torch.cuda.synchronize(self.weight.device)

    @property
    def touched(self) -> torch.Tensor:
        """uint8 flag per cache row, set by the fused backward: the rows the table-agg merge exchanges."""
        if self._touched is None or self._touched.device != self.weight.device:
            self._touched = torch.zeros(self.row_base[-1], dtype=torch.uint8, device=self.weight.device)
        return self._touched

    def _is_arange(self, lS_o, n):
        """Criteo layout check (offsets == arange, data_loader_terabyte.py:83-87)."""
        if lS_o is None:
            return True
        if isinstance(lS_o, (list, tuple)):
            lS_o = torch.stack([torch.as_tensor(o) for o in lS_o])
        if lS_o.shape[-1] != n:
            return False
        key = (lS_o.data_ptr(), lS_o._version, tuple(lS_o.shape), str(lS_o.device))
        if key not in self._arange_ok:
            if len(self._arange_ok) > 64:
                self._arange_ok.clear()
            ar = torch.arange(n, device=lS_o.device, dtype=lS_o.dtype)
            self._arange_ok[key] = bool((lS_o == ar).all())
        return self._arange_ok[key]

    def _lookup(self, idx, offsets):
        ctx = self.ctx
        T, D = len(self.cache_sizes), self.m_spa
        n = idx.shape[1]
        nb = n if offsets is None else offsets.shape[1]
        slots, miss_pos, miss_count = ops.embbag_probe(ctx, idx)
        packed = torch.empty((nb, T + 1, D), dtype=torch.float32, device=idx.device)
        ops.embbag_fwd(ctx, slots, offsets, packed[:, 1:, :], (T + 1) * D, D)
        self.victim_cache_entries._set(self, idx, miss_pos, miss_count)
        return packed, slots

    def forward(self, lS_o, lS_i, emb_tables, rank):
        """model_no_ddp.py:149-212 -> (ly, cache_group_idxs).  lS_i: [T, n] int64 (CPU or device) or a list
        of T equally long 1-D tensors; lS_o likewise (None = Criteo layout)."""
        T = len(self.emb_l)
        if (lS_o is not None and T != len(lS_o)) or (T != len(lS_i)):
            sys.exit("ERROR: corrupted model input detected in parallel_forward call")
        dev = self.weight.device
        if isinstance(lS_i, (list, tuple)):
            if len({int(x.numel()) for x in lS_i}) != 1:
                raise NotImplementedError("tables with different lookup counts per batch (ragged multi-hot) "
                                          "are not supported by the fused path yet")
            lS_i = torch.stack([torch.as_tensor(x).reshape(-1) for x in lS_i])
        idx = lS_i.to(device=dev, dtype=torch.int64, non_blocking=True)
        if idx.stride(-1) != 1:
            idx = idx.contiguous()
        n = idx.shape[1]
        if self._is_arange(lS_o, n):
            offsets = None
        else:
            if isinstance(lS_o, (list, tuple)):
                lS_o = torch.stack([torch.as_tensor(o) for o in lS_o])
            offsets = lS_o.to(device=dev, dtype=torch.int64).contiguous()
        self.ctx.bind_host_tables(emb_tables.device_pointers())
        if torch.is_grad_enabled():
            if self._anchor is None or self._anchor.device != dev:
                self._anchor = torch.zeros(1, device=dev, requires_grad=True)
            packed, slots = _CacheLookupFn.apply(self._anchor, self, idx, offsets)
        else:
            packed, slots = self._lookup(idx, offsets)
        ly = LookupList(packed[:, k + 1, :] for k in range(T))
        ly.packed = packed
        cache_group_idxs = [slots[k] for k in range(T)]
        if len(self.emb_l) != len(ly):
            sys.exit("ERROR: corrupted intermediate result in parallel_forward call")
        return ly, cache_group_idxs

    # -- fused backward + sparse SGD (optimizer_embeds.step(), main_no_ddp.py:376, 413) ------------
    def apply_pending_sgd(self, lr: float):
        ctx = self.ctx
        T, D = len(self.cache_sizes), self.m_spa
        for slots, offsets, grad_packed in self._pending:
            n = slots.shape[1]
            if n not in self._bwd_work:
                self._bwd_work[n] = ops.embbag_bwd_work(ctx, n, slots.device)
            ops.embbag_bwd_sgd(ctx, slots, offsets, grad_packed[:, 1:, :], (T + 1) * D, D, lr, self._bwd_work[n],
                               self.touched)
        self._pending = []


class CacheSGD:
    """Stands where the reference has `optim.SGD(cache_group.parameters(), lr=lr_embeds)`
    (main_no_ddp.py:376): zero_grad() / step() with the same meaning, but the step is the fused
    HIP backward + sparse row update."""

    def __init__(self, cache_group: Embedding_Table_Cache_Group, lr: float):
        self.group, self.lr = cache_group, float(lr)
        self.param_groups = [{"lr": self.lr, "params": [cache_group.weight]}]

    def zero_grad(self, set_to_none: bool = True):
        self.group._pending = []

    def step(self):
        self.group.apply_pending_sgd(self.param_groups[0]["lr"])


# --------------------------------------------------------------------------------------------------
# dense model
# --------------------------------------------------------------------------------------------------


def _linears(seq) -> List[nn.Linear]:
    return [l for l in seq if isinstance(l, nn.Linear)]


def _dense_weight(l: nn.Linear) -> torch.Tensor:
    """[out, in] contiguous weight for the kernels.  After a TrainEngine has taken the module over, layer.weight can be
    a strided view of the engine's padded flat storage; the autograd surface then works on a packed copy."""
    w = l.weight.data
    return w if w.is_contiguous() else w.contiguous()


class _DlrmDenseFn(torch.autograd.Function):
    """Whole dense forward/backward on the HIP kernels: bottom MLP -> (writes feature 0 of the packed
    operand) -> dot interaction -> top MLP.  One autograd node instead of ~20."""

    @staticmethod
    def forward(ctx, net, X, packed, *params):
        st = net._dense_forward(X, packed)
        ctx.net, ctx.st = net, st
        return st["Z"]

    @staticmethod
    def backward(ctx, dZ):
        net, st = ctx.net, ctx.st
        dpacked, grads = net._dense_backward(st, dZ.contiguous())
        return (None, None, dpacked) + tuple(grads)


class DLRM_Net(nn.Module):
    """model_no_ddp.py:215-316.  `bot_l` / `top_l` are nn.Sequential of nn.Linear + ReLU/Sigmoid holding
    the parameters (numpy-seeded init identical to the reference, :255-261); the arithmetic runs on the
    FP32-MFMA kernels."""

    def __init__(self, ln_bot=None, ln_top=None, arch_interaction_op=None, arch_interaction_itself=False,
                 sync_dense_params=True, sigmoid_bot=-1, sigmoid_top=-1, loss_threshold=0.0):
        super().__init__()
        if (ln_bot is not None) and (ln_top is not None) and (arch_interaction_op is not None):
            self.output_d = 0
            self.parallel_model_batch_size = -1
            self.parallel_model_is_not_prepared = True
            self.arch_interaction_op = arch_interaction_op
            self.arch_interaction_itself = arch_interaction_itself
            self.sync_dense_params = sync_dense_params
            self.loss_threshold = loss_threshold
            self.cpu = torch.device("cpu")
            self.sigmoid_bot, self.sigmoid_top = sigmoid_bot, sigmoid_top
            self.bot_l = self.create_mlp(np.asarray(ln_bot), sigmoid_bot)
            self.top_l = self.create_mlp(np.asarray(ln_top), sigmoid_top)
            self._work = {}

    def create_mlp(self, ln, sigmoid_layer):
        layers = nn.ModuleList()
        for i in range(0, ln.size - 1):
            n = ln[i]
            m = ln[i + 1]
            LL = nn.Linear(int(n), int(m), bias=True)
            mean = 0.0
            std_dev = np.sqrt(2 / (m + n))
            W = np.random.normal(mean, std_dev, size=(m, n)).astype(np.float32)
            std_dev = np.sqrt(1 / m)
            bt = np.random.normal(mean, std_dev, size=m).astype(np.float32)
            LL.weight.data = torch.tensor(W, requires_grad=True)
            LL.bias.data = torch.tensor(bt, requires_grad=True)
            layers.append(LL)
            layers.append(nn.Sigmoid() if i == sigmoid_layer else nn.ReLU())
        return torch.nn.Sequential(*layers)

    # -- helpers ------------------------------------------------------------------------------------
    def _acts(self, seq, sigmoid_layer):
        ls = _linears(seq)
        return [(l, ops.ACT["sigmoid"] if i == sigmoid_layer else ops.ACT["relu"]) for i, l in enumerate(ls)]

    def _pack(self, x_or_none, ly, B, D, device):
        packed = getattr(ly, "packed", None)
        if packed is not None:
            return packed
        z = torch.zeros(B, D, device=device) if x_or_none is None else x_or_none
        return torch.stack([z] + list(ly), dim=1).contiguous()

    def _mlp_fwd(self, x, layers, out_last=None):
        acts = [x]
        cur = x
        for i, (l, act) in enumerate(layers):
            if out_last is not None and i == len(layers) - 1:
                y = out_last
            else:
                y = torch.empty(cur.shape[0], l.out_features, dtype=torch.float32, device=cur.device)
            ops.linear_fwd(cur, _dense_weight(l), l.bias.data, y, act)
            acts.append(y)
            cur = y
        return acts

    def _mlp_bwd(self, acts, layers, dY, need_dx):
        grads = [None] * (2 * len(layers))
        B = acts[0].shape[0]
        for i in reversed(range(len(layers))):
            l, act = layers[i]
            N, K = l.out_features, l.in_features
            key = (B, N, K)
            if key not in self._work:
                self._work[key] = ops.linear_bwd_work(B, N, K, dY.device)
            dW = torch.empty(l.out_features, l.in_features, dtype=torch.float32, device=dY.device)
            db = torch.empty_like(l.bias.data)
            dX = torch.empty(B, K, dtype=torch.float32, device=dY.device) if (i > 0 or need_dx) else None
            ops.linear_bwd(acts[i], _dense_weight(l), acts[i + 1], dY, dX, dW, db, act, self._work[key])
            grads[2 * i], grads[2 * i + 1] = dW, db
            dY = dX
        return dY, grads

    def _dense_forward(self, X, packed):
        B, F, D = packed.shape
        itself = bool(self.arch_interaction_itself)
        bot = self._acts(self.bot_l, self.sigmoid_bot)
        top = self._acts(self.top_l, self.sigmoid_top)
        X = X.contiguous()
        bacts = self._mlp_fwd(X, bot, out_last=packed[:, 0, :])
        if self.arch_interaction_op == "dot":
            npairs = F * (F + 1) // 2 if itself else F * (F - 1) // 2
            R = torch.empty(B, D + npairs, dtype=torch.float32, device=X.device)
            ops.interact_fwd(packed, itself, R)
        elif self.arch_interaction_op == "cat":
            R = packed.view(B, F * D)
        else:
            sys.exit("ERROR: --arch-interaction-op=" + self.arch_interaction_op + " is not supported")
        tacts = self._mlp_fwd(R, top)
        return dict(bacts=bacts, tacts=tacts, packed=packed, R=R, Z=tacts[-1], bot=bot, top=top)

    def _dense_backward(self, st, dZ):
        packed = st["packed"]
        dR, gtop = self._mlp_bwd(st["tacts"], st["top"], dZ, True)
        if self.arch_interaction_op == "dot":
            dpacked = torch.empty_like(packed)
            ops.interact_bwd(packed, dR, bool(self.arch_interaction_itself), dpacked)
        else:
            dpacked = dR.view(packed.shape).clone()
        # dY of the bottom MLP's last layer is feature 0 of dpacked (strided rows)
        _, gbot = self._mlp_bwd(st["bacts"], st["bot"], dpacked[:, 0, :], False)
        return dpacked, gbot + gtop

    # -- reference API ------------------------------------------------------------------------------
    def interact_features(self, x, ly):
        """model_no_ddp.py:272-304 (forward only helper; the training path uses forward())."""
        B, D = x.shape
        if self.arch_interaction_op == "dot":
            packed = torch.stack([x] + list(ly), dim=1).contiguous()
            F = packed.shape[1]
            itself = bool(self.arch_interaction_itself)
            npairs = F * (F + 1) // 2 if itself else F * (F - 1) // 2
            R = torch.empty(B, D + npairs, dtype=torch.float32, device=x.device)
            ops.interact_fwd(packed, itself, R)
            return R
        elif self.arch_interaction_op == "cat":
            return torch.cat([x] + list(ly), dim=1)
        sys.exit("ERROR: --arch-interaction-op=" + self.arch_interaction_op + " is not supported")

    def _params(self):
        ps = []
        for l in _linears(self.bot_l) + _linears(self.top_l):
            ps += [l.weight, l.bias]
        return ps

    def forward(self, dense_x, ly):
        B = dense_x.shape[0]
        D = _linears(self.bot_l)[-1].out_features
        packed = self._pack(None, ly, B, D, dense_x.device)
        if torch.is_grad_enabled():
            p = _DlrmDenseFn.apply(self, dense_x, packed, *self._params())
        else:
            p = self._dense_forward(dense_x, packed)["Z"]
        if 0.0 < self.loss_threshold < 1.0:
            z = torch.clamp(p, min=self.loss_threshold, max=(1.0 - self.loss_threshold))
        else:
            z = p
        return z


class HipBCELoss(nn.Module):
    """torch.nn.BCELoss(reduction="mean") on the HIP kernel (main_no_ddp.py:368)."""

    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, Z, T):
            buf = torch.empty(65, dtype=torch.float32, device=Z.device)
            dZ = torch.empty_like(Z)
            ops.bce_fwd_bwd(Z.contiguous(), T.contiguous(), buf, dZ)
            ctx.save_for_backward(dZ)
            return buf[0]

        @staticmethod
        def backward(ctx, g):
            (dZ,) = ctx.saved_tensors
            return dZ * g, None

    def forward(self, Z, T):
        return HipBCELoss._Fn.apply(Z, T)
