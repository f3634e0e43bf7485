"""CPU oracle for the cDLRM look-ahead-cache data-parallel training path.

THIS FILE IS TEST INFRASTRUCTURE.  It is a from-scratch CPU restatement (torch-CPU / numpy, single
thread semantics) of the reference algorithm for the hot path named by BASELINE.json:north_star.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it, and only
as the checker / the timed CPU baseline -- never as the product path.  The product path
(`cdlrm_amd/`) never imports this module and fails loudly when the HIP library is missing.

Parity pin: every function below is checked in `tests/test_oracle_golden.py` against golden vectors
captured from the *imported reference* (`tools/make_golden.py`, run in the build container where
`/root/reference` exists; vectors committed under `tests/golden/`).  Citations are file:line into the
reference tree (lkp411/cDLRM).

Conventions
-----------
* `occ`     : int64 [P_k, ways] tag ("occupancy") table of one embedding table, -1 = empty.
* `weight`  : fp32 [ways*P_k + aux, D] cache rows of one table; slot = P_k*way + set (way-major).
* duplicate (set, way) claims inside one window: the claimant that comes LAST in ascending-index
  order wins both the tag and the row (the reference's single-thread `index_put_` behaviour,
  main_no_ddp.py:204-206).  This oracle implements that rule explicitly, it does not rely on the
  thread count of the torch build that runs it.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

# --------------------------------------------------------------------------------------------------
# a-5  cache geometry  (model_no_ddp.py:319-331 isPrime, :122-125 find_next_prime, :130-147)
# --------------------------------------------------------------------------------------------------


def is_prime_ref(n: int) -> bool:
    """model_no_ddp.py:319-331 -- trial division that starts at 3 and stops at i*i < n.

    Quirks kept on purpose (they decide the cache geometry): 1 and 2 -> False; every n whose
    smallest odd factor f has f*f >= n passes (4, 6, 8, 9, 10, 14, 25, ...).
    """
    if n == 1 or n == 2:
        return False
    i = 3
    while i * i < n:
        if n % i == 0:
            return False
        i += 1
    return True


def find_next_prime(max_cache_size: int) -> Optional[int]:
    """model_no_ddp.py:122-125 -- first i in [c, 2c) with is_prime_ref(i) (None if the range is empty)."""
    for i in range(max_cache_size, 2 * max_cache_size):
        if is_prime_ref(i):
            return i
    return None


def cache_geometry(ln_emb: Sequence[int], max_cache_size: int, num_ways: int, aux_table_size: int):
    """model_no_ddp.py:113,130-142 -- (P, cache_sizes[k]=min(n_k,P), rows_k = ways*P_k + aux)."""
    P = find_next_prime(max_cache_size)
    cache_sizes = [int(n) if int(n) < P else P for n in ln_emb]
    rows = [num_ways * p + aux_table_size for p in cache_sizes]
    return P, cache_sizes, rows


def new_occupancy_tables(cache_sizes: Sequence[int], num_ways: int) -> List[torch.Tensor]:
    """model_no_ddp.py:144-147."""
    return [torch.full((int(p), num_ways), -1, dtype=torch.int64) for p in cache_sizes]


# --------------------------------------------------------------------------------------------------
# a-1  window grouping  (cache_manager.py:85-110)
# --------------------------------------------------------------------------------------------------


def window_groups(num_batches: int, lookahead: int, cache_workers: int) -> List[List[int]]:
    """Batch ids (per epoch) that `Prefetcher.run` concatenates before each flush.

    cache_manager.py:85-110: flush when j>0 and collected % (L*cache_workers) == 0, or at the last
    batch; the batch that triggers a flush starts the next group (:106-107) except the very last
    batch of the epoch, which is appended to the group being flushed (:92-93).  Each flushed group is
    cut into slices of L*B examples (:75, :96-100) -> one (rows, uniq, map) triple per slice.
    Returns the list of per-slice batch-id lists, in FIFO order.
    """
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


# --------------------------------------------------------------------------------------------------
# a-2  lookahead unique scan + host row gather  (cache_manager.py:28-46, model_no_ddp.py:80-87)
# --------------------------------------------------------------------------------------------------


def process_batch_slice(slice_idx: Sequence[torch.Tensor], host_tables: Sequence[torch.Tensor],
                        build_maps: bool = True):
    """Per table k: sorted unique of the window's indices, dense inverse map, host rows.

    cache_manager.py:32 `torch.unique` (sorted ascending, int64); :36-40 map = -1*ones(max+1,1),
    map[uniq] = arange(U); :44 -> model_no_ddp.py:84 rows = W_host[k][uniq].
    """
    rows, uniqs, maps = [], [], []
    for k in range(len(host_tables)):
        idx = torch.as_tensor(slice_idx[k]).reshape(-1).to(torch.int64)
        u = torch.from_numpy(np.unique(idx.numpy()))
        uniqs.append(u)
        if build_maps:
            m = torch.full((int(u.max()) + 1, 1), -1, dtype=torch.int64)
            m[u] = torch.arange(u.shape[0]).view(-1, 1)
            maps.append(m)
        else:
            maps.append(None)
        rows.append(host_tables[k][u])
    return rows, uniqs, maps


# --------------------------------------------------------------------------------------------------
# a-3  set-associative insert / evict for one window  (main_no_ddp.py:148-209)
# --------------------------------------------------------------------------------------------------


def draw_q(M: int, ways: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """The Exp(1) draw `Categorical.sample()` consumes (torch.multinomial fast path):
    one float32 per (claimant, way), row-major, from the torch CPU generator."""
    q = torch.empty(M, ways, dtype=torch.float32)
    if M > 0:
        q.exponential_(1, generator=generator)
    return q


def choose_ways(avail_rows: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
    """main_no_ddp.py:183-185: Categorical(avail.float()).sample() == argmax_w(p_w / q_w),
    p = avail / sum(avail) in float32, first maximal index on ties."""
    a = avail_rows.to(torch.float32)
    p = a / a.sum(-1, keepdim=True)
    return torch.argmax(p / q, dim=-1)


def cache_embeddings_table(uniq: torch.Tensor, rows: torch.Tensor, occ: torch.Tensor,
                           weight: torch.Tensor, P: int,
                           q_source: Callable[[int, int], torch.Tensor]):
    """One table of `CacheEmbeddings` (main_no_ddp.py:151-206).  Mutates `occ` and `weight`.

    Returns dict(way, kept_idx, kept_set, evict_idx, evict_rows, q).
    `q_source(M, ways)` supplies the Exp(1) draw (parity mode: the torch CPU generator).
    """
    ways = occ.shape[1]
    uniq = uniq.to(torch.int64)
    set_idx = torch.remainder(uniq, P)                                   # :155
    eq = occ[set_idx] == uniq.view(-1, 1)                                # :160
    hit = eq.any(dim=1)
    hit_pos = hit.nonzero(as_tuple=False).flatten()                      # :161
    miss_pos = (~hit).nonzero(as_tuple=False).flatten()                  # :162
    hit_sets = set_idx[hit_pos]                                          # :164
    hit_ways = eq.nonzero(as_tuple=True)[1]                              # :165
    nec_idx = uniq[miss_pos]                                             # :167
    nec_set = set_idx[miss_pos]                                          # :168
    avail = torch.ones(occ.shape, dtype=torch.bool)                      # :171
    avail[hit_sets, hit_ways] = False                                    # :172
    full = ~avail.any(dim=1)                                             # :173  (per set)
    keep = ~full[nec_set]                                                # :176-177
    nec_idx = nec_idx[keep]                                              # :179
    nec_set = nec_set[keep]                                              # :180
    pos_in_uniq = miss_pos[keep]                                         # == map[nec_idx] (:205)
    M = int(nec_idx.shape[0])
    q = q_source(M, ways)
    if M > 0:
        way = choose_ways(avail[nec_set], q)                             # :183-185
    else:
        way = torch.zeros(0, dtype=torch.int64)
    # eviction list, read from the PRE-update state, one entry per claimant of an occupied slot
    old = occ[nec_set, way]                                              # :190
    ev = (old != -1).nonzero(as_tuple=False).flatten()
    ev_slots = P * way[ev] + nec_set[ev]                                 # :194
    evict_idx = old[ev].clone()                                          # :196
    evict_rows = weight[ev_slots].clone()                                # :197
    # commit, last claimant in ascending-index order wins a contested (set, way)
    slots = P * way + nec_set                                            # :203
    if M > 0:
        s_np = slots.numpy()
        _, first_in_rev = np.unique(s_np[::-1], return_index=True)
        winners = torch.from_numpy(np.sort(M - 1 - first_in_rev))
        occ[nec_set[winners], way[winners]] = nec_idx[winners]           # :204
        weight[slots[winners]] = rows[pos_in_uniq[winners]]              # :205-206
    return dict(way=way, kept_idx=nec_idx, kept_set=nec_set, evict_idx=evict_idx,
                evict_rows=evict_rows, q=q, slots=slots)


def cache_embeddings(rows_per_table, uniqs, occ_tables, weights, cache_sizes,
                     q_source: Optional[Callable[[int, int], torch.Tensor]] = None):
    """`CacheEmbeddings` over all tables, in table order (the RNG stream is shared, :151).

    Returns (eviction_data, details) where eviction_data is the list of (idx, rows) tuples the
    reference puts on `eviction_fifo` (main_no_ddp.py:199, 208-209).
    """
    if q_source is None:
        q_source = lambda M, w: draw_q(M, w)
    eviction_data, details = [], []
    for k in range(len(rows_per_table)):
        d = cache_embeddings_table(uniqs[k], rows_per_table[k], occ_tables[k], weights[k],
                                   int(cache_sizes[k]), q_source)
        eviction_data.append((d["evict_idx"], d["evict_rows"]))
        details.append(d)
    return eviction_data, details


def dedup_evictions(evict_idx: torch.Tensor, evict_rows: torch.Tensor):
    """Eviction entries repeat when several claimants pick one occupied slot (SURVEY App. A window 2);
    every repeat carries the same (tag, row), so the write-back (a-13) only depends on the set of
    distinct tags.  Returns the entries sorted by tag with repeats removed."""
    if evict_idx.numel() == 0:
        return evict_idx, evict_rows
    u, first = np.unique(evict_idx.numpy(), return_index=True)
    return torch.from_numpy(u), evict_rows[torch.from_numpy(first)]


# --------------------------------------------------------------------------------------------------
# a-13 eviction write-back to the host master tables  (cache_manager.py:57-62)
# --------------------------------------------------------------------------------------------------


def eviction_writeback(host_tables, eviction_data, average_on_writeback: bool = False):
    """cache_manager.py:61-62: W[idx] = emb, or (W[idx]+emb)/2 with --average-on-writeback.
    Repeated idx entries carry identical rows, so the result is well defined."""
    for k, (idx, emb) in enumerate(eviction_data):
        if idx.numel() == 0:
            continue
        di, de = dedup_evictions(idx, emb)
        if average_on_writeback:
            host_tables[k][di] = (host_tables[k][di] + de) / 2
        else:
            host_tables[k][di] = de


# --------------------------------------------------------------------------------------------------
# a-6  per-iteration tag probe + cached EmbeddingBag forward  (model_no_ddp.py:149-212)
# --------------------------------------------------------------------------------------------------


def probe_table(occ: torch.Tensor, idx: torch.Tensor, P: int):
    """model_no_ddp.py:166-187 for one table: slot per lookup, misses -> aux slots in position order.

    Returns (slots int64 [n], miss_pos int64 [m], miss_idx int64 [m])."""
    ways = occ.shape[1]
    idx = idx.to(torch.int64)
    set_idx = torch.remainder(idx, P)
    eq = occ[set_idx] == idx.view(-1, 1)
    hit = eq.any(dim=1)
    hit_pos = hit.nonzero(as_tuple=False).flatten()
    miss_pos = (~hit).nonzero(as_tuple=False).flatten()
    hit_ways = eq[hit_pos].nonzero(as_tuple=True)[1]
    slots = torch.empty(idx.shape, dtype=torch.int64)
    slots[hit_pos] = P * hit_ways + set_idx[hit_pos]                     # :174
    slots[miss_pos] = P * ways + torch.arange(miss_pos.shape[0])         # :177
    return slots, miss_pos, idx[miss_pos]


def cache_forward(occ_tables, weights, cache_sizes, lS_o, lS_i, host_tables):
    """`Embedding_Table_Cache_Group.forward` (model_no_ddp.py:149-212).  Mutates the aux rows of
    `weights` (:179).  Returns (ly list of fp32 [n_bags, D], cache_group_idxs list of int32 [n])."""
    ly, cg = [], []
    for k in range(len(weights)):
        P = int(cache_sizes[k])
        slots, miss_pos, miss_idx = probe_table(occ_tables[k], lS_i[k], P)
        if miss_pos.numel() > 0:
            weights[k][slots[miss_pos]] = host_tables[k][miss_idx]       # :179
        V = torch.nn.functional.embedding_bag(slots, weights[k], lS_o[k].to(torch.int64),
                                              mode="sum")               # :200-202
        ly.append(V)
        cg.append(slots.to(torch.int32))                                 # :204
    return ly, cg


# --------------------------------------------------------------------------------------------------
# a-7  EmbeddingBag backward + sparse SGD on cache rows  (main_no_ddp.py:376, 409, 413)
# --------------------------------------------------------------------------------------------------


def bag_ids(offsets: torch.Tensor, n_idx: int) -> torch.Tensor:
    """bag id of every lookup given EmbeddingBag offsets."""
    offsets = offsets.to(torch.int64)
    b = torch.zeros(n_idx, dtype=torch.int64)
    if offsets.numel() > 1:
        starts = offsets[1:]
        starts = starts[starts < n_idx]
        b.index_add_(0, starts, torch.ones_like(starts))
    return torch.cumsum(b, 0)


def embbag_bwd_sgd(weight: torch.Tensor, slots: torch.Tensor, offsets: torch.Tensor,
                   grad_out: torch.Tensor, lr: float):
    """W[slot] -= lr * sum_{lookups i with slot_i == slot} dL/dV[bag(i)].

    torch: embedding_bag backward (sparse) -> COO grad; SGD.step coalesces it (duplicates summed in
    position order) and does `param.add_(grad, alpha=-lr)`."""
    slots = slots.to(torch.int64)
    b = bag_ids(offsets, slots.shape[0])
    g = grad_out[b]
    u, inv = torch.unique(slots, return_inverse=True)
    acc = torch.zeros(u.shape[0], weight.shape[1], dtype=weight.dtype)
    acc.index_add_(0, inv, g)
    weight[u] = weight[u] + (-lr) * acc


# --------------------------------------------------------------------------------------------------
# a-8 / a-9 / a-10  dense model  (model_no_ddp.py:215-316, main_no_ddp.py:212-221)
# --------------------------------------------------------------------------------------------------


def init_mlp(ln: Sequence[int]):
    """model_no_ddp.py:244-262: per layer W ~ N(0, sqrt(2/(m+n))) [m,n], b ~ N(0, sqrt(1/m)) [m],
    drawn from the *numpy* global RNG in that order, float32.  The reference constructs `nn.Linear(n, m)` first
    (:252), whose default init draws from the torch CPU generator before the weights are replaced: those draws are
    reproduced (and discarded) here because the insert's Exp(1) draws come from the same stream later."""
    ws, bs = [], []
    for i in range(len(ln) - 1):
        n, m = int(ln[i]), int(ln[i + 1])
        torch.nn.Linear(n, m, bias=True)
        W = np.random.normal(0.0, np.sqrt(2 / (m + n)), size=(m, n)).astype(np.float32)
        b = np.random.normal(0.0, np.sqrt(1 / m), size=m).astype(np.float32)
        ws.append(torch.tensor(W))
        bs.append(torch.tensor(b))
    return ws, bs


def init_host_tables(ln_emb: Sequence[int], m_spa: int):
    """model_no_ddp.py:70-73: U(-sqrt(1/n), sqrt(1/n)) [n, m] float32 from the numpy global RNG."""
    out = []
    for n in ln_emb:
        n = int(n)
        W = np.random.uniform(low=-np.sqrt(1 / n), high=np.sqrt(1 / n), size=(n, m_spa)).astype(np.float32)
        out.append(torch.tensor(W))
    return out


def mlp_forward(x, ws, bs, sigmoid_layer: int = -1):
    """model_no_ddp.py:244-270: Linear + ReLU, Sigmoid at `sigmoid_layer`."""
    for i, (W, b) in enumerate(zip(ws, bs)):
        x = torch.nn.functional.linear(x, W, b)
        x = torch.sigmoid(x) if i == sigmoid_layer else torch.relu(x)
    return x


def interaction_pairs(nf: int, itself: bool = False):
    """model_no_ddp.py:288-290."""
    offset = 1 if itself else 0
    li = [i for i in range(nf) for j in range(i + offset)]
    lj = [j for i in range(nf) for j in range(i + offset)]
    return li, lj


def interact_features(x, ly, op: str = "dot", itself: bool = False):
    """model_no_ddp.py:272-304."""
    if op == "dot":
        B, d = x.shape
        T = torch.cat([x] + list(ly), dim=1).view((B, -1, d))
        Z = torch.bmm(T, torch.transpose(T, 1, 2))
        li, lj = interaction_pairs(Z.shape[1], itself)
        Zflat = Z[:, torch.tensor(li, dtype=torch.long), torch.tensor(lj, dtype=torch.long)]
        return torch.cat([x, Zflat], dim=1)
    if op == "cat":
        return torch.cat([x] + list(ly), dim=1)
    raise ValueError("unsupported interaction op " + op)


def dlrm_forward(X, ly, bot, top, op="dot", itself=False, loss_threshold=0.0):
    """model_no_ddp.py:306-316; sigmoid on the last top layer (main_no_ddp.py:358)."""
    x = mlp_forward(X, bot[0], bot[1], -1)
    z = interact_features(x, ly, op, itself)
    p = mlp_forward(z, top[0], top[1], len(top[0]) - 1)
    if 0.0 < loss_threshold < 1.0:
        p = torch.clamp(p, min=loss_threshold, max=1.0 - loss_threshold)
    return p


def loss_fn(Z, T, kind="bce", loss_ws=None):
    """main_no_ddp.py:212-221, 364-372."""
    if kind == "bce":
        return torch.nn.functional.binary_cross_entropy(Z, T, reduction="mean")
    if kind == "mse":
        return torch.nn.functional.mse_loss(Z, T, reduction="mean")
    if kind == "wbce":
        w = loss_ws[T.data.view(-1).long()].view_as(T)
        return (w * torch.nn.functional.binary_cross_entropy(Z, T, reduction="none")).mean()
    raise ValueError(kind)


# --------------------------------------------------------------------------------------------------
# a-12 table aggregation  (main_no_ddp.py:250-292)
# --------------------------------------------------------------------------------------------------


def table_aggregate(weights_per_rank, touched_per_rank, reduce_op="mean"):
    """`broadcast_and_aggregate` for one table, W ranks emulated in-process.

    weights_per_rank: list of W fp32 [rows, D] (mutated); touched_per_rank: list of W int32 slot-id
    tensors (each rank's cache_group_idxs window for this table).  :268-292."""
    W = len(weights_per_rank)
    u = torch.unique(torch.cat([t.reshape(-1) for t in touched_per_rank]), sorted=True).long()
    if reduce_op == "mean":
        acc = weights_per_rank[0][u] / W
        for r in range(1, W):
            acc = acc + weights_per_rank[r][u] / W
    elif reduce_op == "sum":
        acc = weights_per_rank[0][u].clone()
        for r in range(1, W):
            acc = acc + weights_per_rank[r][u]
    elif reduce_op == "max":
        acc = weights_per_rank[0][u].clone()
        for r in range(1, W):
            acc = torch.maximum(acc, weights_per_rank[r][u])
    else:
        raise ValueError(reduce_op)
    for r in range(W):
        weights_per_rank[r][u] = acc
    return u


# --------------------------------------------------------------------------------------------------
# a-15 QR embedding bag operator  (tricks/qr_embedding_bag.py:156-174)
# --------------------------------------------------------------------------------------------------


def qr_embedding_bag(idx, offsets, weight_q, weight_r, num_collisions: int, operation="mult"):
    """q = (idx / c).long() is a float32 true division then truncation (wrong above 2**24, kept),
    r = idx % c; two sum-pooled bags combined by mult/add/concat."""
    iq = (idx / num_collisions).long()
    ir = torch.remainder(idx, num_collisions).long()
    eq = torch.nn.functional.embedding_bag(iq, weight_q, offsets, mode="sum")
    er = torch.nn.functional.embedding_bag(ir, weight_r, offsets, mode="sum")
    if operation == "concat":
        return torch.cat((eq, er), dim=1)
    if operation == "add":
        return eq + er
    if operation == "mult":
        return eq * er
    raise ValueError(operation)


# --------------------------------------------------------------------------------------------------
# a-16 mixed-dimension trick  (tricks/md_embedding_bag.py:20-78)
# --------------------------------------------------------------------------------------------------


def md_solver(n, alpha, d0=None, B=None, round_dim=True, k=None):
    """md_embedding_bag.py:20-57 in numpy float32: sizes sorted ascending (result stays in that order), divided by the
    query counts, d = lambda * n^-alpha with lambda = d0 * n_min^alpha or B / sum n^(1-alpha); clamp at 1, entry 0 = d0
    when given; round half-to-even to long; optional round(log2) to a power of two (returned as float32)."""
    n = np.asarray(n, dtype=np.int64)
    order = np.argsort(n, kind="stable")
    nf = n[order].astype(np.float32)
    kk = np.ones(len(n), dtype=np.float32) if k is None else np.asarray(k, dtype=np.float32)[order]
    nf = (nf / kk).astype(np.float32)
    a = np.float32(alpha)
    if d0 is not None:
        lamb = np.float32(d0) * np.power(nf[0], a, dtype=np.float32)
    elif B is not None:
        lamb = np.float32(B) / np.sum(np.power(nf, np.float32(1) - a, dtype=np.float32), dtype=np.float32)
    else:
        raise ValueError("Must specify either d0 or B")
    d = (np.float32(lamb) * np.power(nf, -a, dtype=np.float32)).astype(np.float32)
    d = np.maximum(d, np.float32(1))
    if d0 is not None:
        d[0] = d0
    d = np.rint(d).astype(np.int64)
    if round_dim:
        return np.power(np.float32(2), np.rint(np.log2(d.astype(np.float32)))).astype(np.float32)
    return d


def pr_embedding_bag(idx, offsets, weight, proj=None):
    """md_embedding_bag.py:60-78: sum-pooled bag of width embedding_dim, then x @ proj^T (no bias) when given."""
    e = torch.nn.functional.embedding_bag(idx, weight, offsets, mode="sum")
    return e if proj is None else torch.nn.functional.linear(e, proj)


# --------------------------------------------------------------------------------------------------
# a-14 the whole training loop, W ranks emulated in-process  (main_no_ddp.py:324-502)
# --------------------------------------------------------------------------------------------------


class OracleTrainer:
    """Replays `Run`'s op order (main_no_ddp.py:386-425) around the restated steps above, for
    `world_size` emulated ranks that share ONE tag state (main_no_ddp.py:295-306) and hold one
    cache/MLP replica each.  Deterministic schedule: prefetch distance 0 (the window's host rows are
    gathered right before the insert) and eviction write-back applied synchronously, rank 0's copy
    only (main_no_ddp.py:208, 312-315)."""

    def __init__(self, ln_emb, m_spa, ln_bot, ln_top, *, cache_size, num_ways, mini_batch_size,
                 world_size=1, lr=0.1, lr_embeds=0.3, lookahead=2, table_agg_freq=1,
                 table_agg_op="mean", loss="bce", itself=False, op="dot", seed=123,
                 average_on_writeback=False, host_tables=None, cache_init="normal", loss_weights=None,
                 loss_threshold=0.0, evict_victim_cache=False):
        self.ln_emb = [int(n) for n in ln_emb]
        self.W = world_size
        self.lr, self.lr_embeds = lr, lr_embeds
        self.L, self.agg_freq, self.agg_op = lookahead, table_agg_freq, table_agg_op
        self.loss_kind, self.itself, self.op = loss, itself, op
        # --loss-weights as main_no_ddp.py:370 builds them (float64), --loss-threshold (model_no_ddp.py:311-314)
        self.loss_ws = None if loss_weights is None else torch.tensor([float(w) for w in loss_weights], dtype=torch.float64)
        self.loss_threshold = float(loss_threshold)
        self.avg_wb = average_on_writeback
        # --evict-victim-cache (main_no_ddp.py:96) is parsed and victim_cache_entries (model_no_ddp.py:187) recorded by the
        # reference, neither is ever used.  The build's definition (world size 1): behind the embedding SGD of a step,
        # emb_tables[k].weight[missing_sparse_idxs] = cache[k].weight[aux_storage_idxs] -- assigned in position order, so of
        # several misses of one index the last one's aux row stays (parity unpinned beyond this restatement).
        self.evict_victim = bool(evict_victim_cache)
        assert not (self.evict_victim and world_size != 1)
        self.ways = num_ways
        self.B = mini_batch_size
        self.lbs = math.ceil(mini_batch_size / world_size)
        # main process (main_no_ddp.py:509-512, 621)
        np.random.seed(seed)
        torch.manual_seed(seed)
        self.host = host_tables if host_tables is not None else init_host_tables(self.ln_emb, m_spa)
        self.P, self.cache_sizes, rows = cache_geometry(self.ln_emb, cache_size, num_ways, mini_batch_size)
        self.occ = new_occupancy_tables(self.cache_sizes, num_ways)
        self.weights, self.bot, self.top = [], [], []
        for r in range(self.W):
            # every trainer process re-seeds (main_no_ddp.py:335-337), builds the cache group
            # (nn.EmbeddingBag default N(0,1) init, model_no_ddp.py:138) and then the MLPs
            np.random.seed(seed)
            torch.manual_seed(seed)
            if cache_init == "normal":
                self.weights.append([torch.randn(n_rows, m_spa) for n_rows in rows])
            else:
                self.weights.append([torch.zeros(n_rows, m_spa) for n_rows in rows])
            self.bot.append(init_mlp(ln_bot))
            self.top.append(init_mlp(ln_top))
        self.touched = [[] for _ in range(self.W)]
        self.losses: List[float] = []

    # one refill (main_no_ddp.py:309-321): rank-0 insert, every rank ends with rank 0's cache
    def refill(self, window_idx, q_source=None):
        rows, uniqs, _ = process_batch_slice(window_idx, self.host, build_maps=False)
        ev, details = cache_embeddings(rows, uniqs, self.occ, self.weights[0], self.cache_sizes, q_source)
        eviction_writeback(self.host, ev, self.avg_wb)
        for r in range(1, self.W):
            for k in range(len(self.ln_emb)):
                self.weights[r][k].copy_(self.weights[0][k])            # broadcast, :318-319
        return ev, details

    def evaluate(self, X, lS_o, lS_i):
        """The test loop body of main_no_ddp.py:484-487 on rank 0: forward through the cache group (aux rows of the
        misses are overwritten with the host rows, as in training) and the MLPs, no gradient.  Returns Z [B, 1]."""
        with torch.no_grad():
            ly, _ = cache_forward(self.occ, self.weights[0], self.cache_sizes, [lS_o[k] for k in range(len(self.ln_emb))],
                                  [lS_i[k] for k in range(len(self.ln_emb))], self.host)
            return dlrm_forward(X, ly, self.bot[0], self.top[0], self.op, self.itself, self.loss_threshold)

    def step(self, j, X, lS_o, lS_i, T):
        """One iteration of main_no_ddp.py:387-423 for all emulated ranks.  Returns per-rank losses."""
        Wn, lbs = self.W, self.lbs
        rank_loss, grads_w = [], []
        params = []
        for r in range(Wn):
            Xr = X[r * lbs:(r + 1) * lbs]
            if Wn == 1:         # whole batch: multi-hot / ragged bags pass through as they are
                Ir = [lS_i[k] for k in range(len(self.ln_emb))]
                Or = [lS_o[k] for k in range(len(self.ln_emb))]
            else:               # the rank slice of the Criteo layout (one lookup per sample)
                Ir = [lS_i[k][r * lbs:(r + 1) * lbs] for k in range(len(self.ln_emb))]
                # lS_o[:, :local_batch_size] (:390).  When world does not divide the batch the reference's last rank gets
                # lbs offsets for fewer lookups -- an empty trailing bag, [lbs, D] pooled rows beside a shorter X -- and
                # raises in interact_features' cat (model_no_ddp.py:276).  Defined here (and in the engine) as the natural
                # extension: the last rank trains on its short slice, one bag per lookup it has.
                Or = [lS_o[k][:Ir[k].numel()] for k in range(len(self.ln_emb))]
            Tr = T[r * lbs:(r + 1) * lbs]
            ly, cg = cache_forward(self.occ, self.weights[r], self.cache_sizes, Or, Ir, self.host)
            ly = [v.detach().requires_grad_(True) for v in ly]
            bw = [w.detach().requires_grad_(True) for w in self.bot[r][0]]
            bb = [b.detach().requires_grad_(True) for b in self.bot[r][1]]
            tw = [w.detach().requires_grad_(True) for w in self.top[r][0]]
            tb = [b.detach().requires_grad_(True) for b in self.top[r][1]]
            Z = dlrm_forward(Xr, ly, (bw, bb), (tw, tb), self.op, self.itself, self.loss_threshold)
            E = loss_fn(Z, Tr, self.loss_kind, self.loss_ws)
            E.backward()
            rank_loss.append(float(E.detach()))
            params.append((bw, bb, tw, tb))
            # embedding SGD (optimizer_embeds.step, :413)
            for k in range(len(self.ln_emb)):
                embbag_bwd_sgd(self.weights[r][k], cg[k].long(), Or[k], ly[k].grad, self.lr_embeds)
            self.touched[r].append(list(cg))      # per-table lists (ragged for multi-hot bags)
            if self.evict_victim:
                for k in range(len(self.ln_emb)):
                    first_aux = int(self.cache_sizes[k]) * self.ways
                    sl = cg[k].long()
                    pos = torch.nonzero(sl >= first_aux).flatten()
                    for p in pos.tolist():          # position order: the last occurrence of an index wins
                        self.host[k][int(Ir[k][p])] = self.weights[r][k][int(sl[p])]
        # aggregate_gradients (:234-247): weight grads averaged, bias grads NOT reduced
        for li in range(len(params[0][0])):
            g = sum(params[r][0][li].grad / Wn for r in range(Wn))
            for r in range(Wn):
                params[r][0][li].grad = g.clone()
        for li in range(len(params[0][2])):
            g = sum(params[r][2][li].grad / Wn for r in range(Wn))
            for r in range(Wn):
                params[r][2][li].grad = g.clone()
        for r in range(Wn):
            bw, bb, tw, tb = params[r]
            self.bot[r] = ([(w - self.lr * w.grad).detach() for w in bw],
                           [(b - self.lr * b.grad).detach() for b in bb])
            self.top[r] = ([(w - self.lr * w.grad).detach() for w in tw],
                           [(b - self.lr * b.grad).detach() for b in tb])
        # table aggregation (:417-423)
        if j > 0 and j % self.agg_freq == 0:
            if Wn > 1 or True:
                for k in range(len(self.ln_emb)):
                    touched = [torch.cat([t[k] for t in self.touched[r]]) for r in range(Wn)]
                    table_aggregate([self.weights[r][k] for r in range(Wn)], touched, self.agg_op)
            self.touched = [[] for _ in range(Wn)]
        self.losses.append(rank_loss)
        return rank_loss
