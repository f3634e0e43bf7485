"""The reference-named free functions and classes of the drop-in surface, EXECUTED with the reference's signatures
against the reference's own outputs (tests/golden, captured by tools/make_golden.py from the imported reference):

    Prefetcher(args, emb, batch_fifo, eviction_fifo, event, cache_ld).start()      cache_manager.py:8-115
      -> load_caches_and_broadcast(cache_group, batch_fifo, eviction_fifo, rank)   main_no_ddp.py:309-321
      -> cache_group(lS_o, lS_i, emb_tables, rank) -> dlrm(X, ly) -> loss -> backward
      -> aggregate_gradients(dlrm) / optimizer_embeds.step() / wait_wrap / optimizer_mlps.step()   :234-247, 412-415
      -> broadcast_and_aggregate(cache_group, idxs, rank, reduce_op)               :250-292
    CacheEmbeddings(rows, uniqs, maps, cache_group, eviction_fifo, rank)           :148-209
    Prefetcher.eviction_manager(emb, fifo, average_on_writeback, core, timeout)    cache_manager.py:49-64
    Embedding_Table_Group(qr_flag=True)                                            model_no_ddp.py:52-56
"""
import os
import queue
import sys
import threading
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.asarray(a))


class GatedLoader:
    """A cache loader (the second copy of the data loader the Prefetcher walks, dlrm_data_pytorch.py:465-483) whose
    pace the test controls: the batch that makes the Prefetcher flush window g is held back until the trainer has
    inserted window g-1 and its evictions are in the host tables -- prefetch distance 0, the schedule the golden runs
    were captured with (the reference itself runs up to batch_fifo_size windows ahead and reads whatever has landed)."""

    def __init__(self, batches, L):
        self.batches, self.L = batches, L
        n = len(batches)
        self.n_groups = (n + L - 1) // L
        self.gates = [threading.Event() for _ in range(self.n_groups)]
        self.gates[0].set()
        # The reference flushes group g when the first batch of group g+1 arrives (or the epoch's last batch,
        # cache_manager.py:91-93, 106-107); this package's Prefetcher emits it as soon as the group's own last batch has
        # arrived.  Holding back the group's LAST batch is early enough for both.
        self.trigger = {min(g * L + L - 1, n - 1): g for g in range(self.n_groups)}

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        for j, b in enumerate(self.batches):
            g = self.trigger.get(j)
            if g is not None:
                assert self.gates[g].wait(timeout=120), "gate %d never opened" % g
            yield b


def _ref_batches(g):
    """(X, lS_o, lS_i, T) stream of tools/make_golden.py:ref_train."""
    from test_engine_parity import make_batches
    T = len(g["ln_emb"])
    B = int(g["B"])
    lS_o = torch.arange(B, dtype=torch.int64).repeat(T, 1)
    return [(X, lS_o, lS_i, Tt) for X, lS_i, Tt in make_batches(g)]


def test_reference_loop_prefetcher_fifo_refill_train(golden):
    """The reference's loop body run UNCHANGED on this package's objects (INTEGRATION.md path A): a Prefetcher thread
    fills batch_fifo, load_caches_and_broadcast refills from it, evictions travel through eviction_fifo to the
    Prefetcher's eviction manager, the step is cache_group -> dlrm -> BCELoss -> backward -> both optimizers."""
    from cdlrm_amd.cache_manager import Prefetcher
    from cdlrm_amd.main_no_ddp import load_caches_and_broadcast, wait_wrap
    from cdlrm_amd.model_no_ddp import CacheSGD, DLRM_Net, Embedding_Table_Cache_Group, Embedding_Table_Group, HipBCELoss
    g = golden("train_small")
    ln_emb = np.array([int(x) for x in g["ln_emb"]])
    m_spa, seed, B, L = int(g["m_spa"]), int(g["seed"]), int(g["B"]), int(g["L"])
    nf = len(ln_emb) + 1
    ln_top = np.array([m_spa + nf * (nf - 1) // 2] + [int(x) for x in g["top"]])
    torch.cuda.set_device(0)
    np.random.seed(seed)
    torch.manual_seed(seed)
    emb_tables = Embedding_Table_Group(m_spa, ln_emb).pin()
    np.random.seed(seed)
    torch.manual_seed(seed)
    cache_group = Embedding_Table_Cache_Group(m_spa, ln_emb, int(g["cache_size"]), B, int(g["ways"])).to(DEV)
    dlrm = DLRM_Net(np.array(g["ln_bot"]), ln_top, "dot", False, True, -1, ln_top.size - 2, 0.0).to(DEV)
    loss_fn = HipBCELoss()
    optimizer_mlps = torch.optim.SGD(dlrm.parameters(), lr=float(g["lr"]))
    optimizer_embeds = CacheSGD(cache_group, lr=float(g["lr_emb"]))
    batches = _ref_batches(g)
    cache_ld = GatedLoader(batches, L)
    args = SimpleNamespace(lookahead=L, cache_workers=1, nepochs=1, mini_batch_size=B, average_on_writeback=False,
                           main_start_core=0, eviction_fifo_timeout=10)
    batch_fifo, eviction_fifo, finish_event = queue.Queue(maxsize=8), queue.Queue(maxsize=8), threading.Event()
    cm = Prefetcher(args, emb_tables, batch_fifo, eviction_fifo, finish_event, cache_ld)
    cm.start()
    rank = 0
    losses = []
    for j, (X, lS_o, lS_i, Tt) in enumerate(batches):
        if j % L == 0:
            torch.manual_seed(5000 + j)                  # the Exp(1) stream the reference consumed for this refill
            reqs = load_caches_and_broadcast(cache_group, batch_fifo, eviction_fifo, rank)
            wait_wrap(reqs)
            eviction_fifo.join()                         # the eviction manager has applied this refill's write-back
            gq = j // L + 1
            if gq < cache_ld.n_groups:
                cache_ld.gates[gq].set()                 # the Prefetcher may now gather the next window's host rows
        lookups, cache_group_idxs = cache_group(lS_o, lS_i, emb_tables, rank)
        Z = dlrm(X.to(DEV), lookups)
        E = loss_fn(Z, Tt.to(DEV))
        optimizer_mlps.zero_grad()
        optimizer_embeds.zero_grad()
        E.backward()
        optimizer_embeds.step()
        optimizer_mlps.step()
        losses.append(float(E.detach()))
        assert len(cache_group_idxs) == len(ln_emb) and cache_group_idxs[0].dtype == torch.int32
    finish_event.set()
    cm.join(timeout=15)
    cache_group.ctx.check()
    np.testing.assert_allclose(np.array(losses), g["losses"], rtol=1e-5)
    for k in range(len(ln_emb)):
        assert torch.equal(cache_group.occupancy_tables[k].cpu(), t(g[f"occ_{k}"])), k
        w = cache_group.emb_l[k].weight[: int(g["ways"]) * cache_group.cache_sizes[k]].double().sum().item()
        np.testing.assert_allclose(w, float(g[f"weight_sum_{k}"]), rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(emb_tables.emb_l[k].weight.data.double().sum().item(), float(g[f"host_sum_{k}"]),
                                   rtol=1e-6, atol=1e-5)
    assert batch_fifo.empty()


@pytest.mark.parametrize("name", ["cache_windows_small", "cache_windows_uniform"])
def test_cache_embeddings_signature_on_reference_windows(golden, name):
    """CacheEmbeddings(rows, uniqs, maps, cache_group, eviction_fifo, rank) fed with the reference's own
    process_batch_slice outputs, window after window: tags bit-exact, every cache row, the eviction list it queues
    (as a set: one entry per evicted tag, rows exact), and the host tables after eviction_manager applied it."""
    from cdlrm_amd.cache_manager import Prefetcher, UniqueIndexMap
    from cdlrm_amd.main_no_ddp import CacheEmbeddings
    from cdlrm_amd.model_no_ddp import Embedding_Table_Cache_Group, Embedding_Table_Group
    g = golden(name)
    ln_emb = np.array([int(x) for x in g["ln_emb"]])
    T, m_spa, ways, B = len(ln_emb), int(g["m_spa"]), int(g["ways"]), int(g["B"])
    np.random.seed(0)
    host = Embedding_Table_Group(m_spa, ln_emb)
    for k in range(T):
        host.emb_l[k].weight.data = t(g[f"host0_{k}"]).clone()
    host.pin()
    cg = Embedding_Table_Cache_Group(m_spa, ln_emb, int(g["cache_size"]), B, ways, cache_init="zeros", aux_phases=1)
    for k in range(T):
        cg.emb_l[k].weight.copy_(t(g[f"weight0_{k}"]))
    cg = cg.to(DEV)
    assert cg.cache_sizes == [int(x) for x in g["cache_sizes"]]
    for w in range(int(g["nwin"])):
        uniqs = [t(g[f"w{w}_uniq_{k}"]) for k in range(T)]                     # CPU tensors, as the reference hands them
        rows = [t(g[f"w{w}_rows_{k}"]) for k in range(T)]
        maps = [UniqueIndexMap(u) for u in uniqs]
        fifo = queue.Queue()
        torch.manual_seed(int(g[f"w{w}_qseed"]))
        CacheEmbeddings(rows, uniqs, maps, cg, fifo, 0)
        ev = fifo.get_nowait()
        assert len(ev) == T
        for k in range(T):
            assert torch.equal(cg.occupancy_tables[k].cpu(), t(g[f"w{w}_occ_{k}"])), (w, k)
            nrow = ways * cg.cache_sizes[k]
            assert torch.equal(cg.emb_l[k].weight[:nrow].cpu(), t(g[f"w{w}_weight_{k}"])[:nrow]), (w, k)
            # the reference lists one entry per CLAIMANT of an occupied slot (repeats carry the same tag and row)
            want_idx, want_rows = t(g[f"w{w}_ev_idx_{k}"]), t(g[f"w{w}_ev_rows_{k}"])
            got_idx, got_rows = ev[k][0].cpu(), ev[k][1].cpu()
            assert got_idx.numel() == torch.unique(got_idx).numel()
            assert set(got_idx.tolist()) == set(want_idx.tolist()), (w, k)
            lut = {int(i): r for i, r in zip(want_idx.tolist(), want_rows)}
            for i, r in zip(got_idx.tolist(), got_rows):
                assert torch.equal(r, lut[i]), (w, k, i)
        # write-back through the drop-in eviction manager (returns when the queue stays empty for `timeout` s)
        evq = queue.Queue()
        evq.put(ev)
        Prefetcher.eviction_manager(host, evq, False, min(os.sched_getaffinity(0)), 1)
        for k in range(T):
            assert torch.equal(host.emb_l[k].weight.data, t(g[f"w{w}_host_{k}"])), (w, k)
        # forward probe through the reference-named forward()
        lS_i = t(g[f"w{w}_fwd_lS_i"])
        lS_o = torch.arange(B, dtype=torch.int64).repeat(T, 1)
        with torch.no_grad():
            ly, cgi = cg(lS_o, lS_i, host, 0)
        for k in range(T):
            assert torch.equal(cgi[k].cpu(), t(g[f"w{w}_fwd_idx_{k}"])), (w, k)
            assert torch.equal(ly[k].cpu(), t(g[f"w{w}_fwd_ly_{k}"])), (w, k)
    cg.ctx.check()


@pytest.mark.parametrize("avg", [0, 1])
def test_eviction_manager_writeback_arms(golden, avg):
    """Prefetcher.eviction_manager with --average-on-writeback off / on against the reference's
    (cache_manager.py:57-62), on an eviction list that repeats entries (as the reference's lists do)."""
    from cdlrm_amd.cache_manager import Prefetcher
    from cdlrm_amd.model_no_ddp import Embedding_Table_Group
    g = golden("writeback_avg%d" % avg)
    np.random.seed(0)
    host = Embedding_Table_Group(4, np.array([40, 9]))
    for k in range(2):
        host.emb_l[k].weight.data = t(g[f"before_{k}"]).clone()
    host.pin()
    evq = queue.Queue()
    evq.put([(t(g[f"idx_{k}"]), t(g[f"emb_{k}"])) for k in range(2)])
    mask = os.sched_getaffinity(0)
    Prefetcher.eviction_manager(host, evq, bool(avg), min(mask), 1)
    # the manager pins its thread to `core` while it runs (cache_manager.py:52 pins its process) and hands the mask back: a
    # caller left on one core would pass that on to every thread it starts afterwards
    assert os.sched_getaffinity(0) == mask
    for k in range(2):
        assert torch.equal(host.emb_l[k].weight.data, t(g[f"after_{k}"])), k


def test_average_on_writeback_through_the_window_plan(golden):
    """--average-on-writeback through the fused path (cdlrm_plan_writeback's averaging arm): host tables after three
    windows against the oracle's trainer with average_on_writeback=True."""
    from cdlrm_amd.engine import TrainEngine, WindowPipeline
    from cdlrm_amd.model_no_ddp import DLRM_Net, Embedding_Table_Cache_Group, Embedding_Table_Group
    from oracle import cdlrm_oracle as O
    ln_emb, m_spa, B, L, ways, cache_size, seed = [3000, 50, 7, 1200], 16, 48, 3, 4, 40, 19
    ln_bot = np.array([13, 32, m_spa])
    nf = len(ln_emb) + 1
    ln_top = np.array([m_spa + nf * (nf - 1) // 2, 24, 1])
    rng = np.random.RandomState(4)
    batches = []
    for j in range(9):
        X = torch.from_numpy(rng.rand(B, 13).astype(np.float32))
        idx = torch.stack([torch.from_numpy((rng.zipf(1.2, size=B).astype(np.int64) * 2654435761 % n)) for n in ln_emb])
        Tt = torch.from_numpy(np.round(rng.rand(B, 1)).astype(np.float32))
        batches.append((X, idx, Tt))
    torch.set_num_threads(1)
    otr = O.OracleTrainer(ln_emb, m_spa, ln_bot, ln_top, cache_size=cache_size, num_ways=ways, mini_batch_size=B,
                          lr=0.1, lr_embeds=0.3, lookahead=L, table_agg_freq=10 ** 9, seed=seed,
                          average_on_writeback=True)
    lS_o = torch.arange(B).repeat(len(ln_emb), 1)
    for j, (X, idx, Tt) in enumerate(batches):
        if j % L == 0:
            torch.manual_seed(300 + j)
            otr.refill(torch.cat([b[1] for b in batches[j:j + L]], dim=1))
        otr.step(j, X, lS_o, idx, Tt)
    np.random.seed(seed)
    torch.manual_seed(seed)
    host = Embedding_Table_Group(m_spa, np.array(ln_emb)).pin()
    np.random.seed(seed)
    torch.manual_seed(seed)
    cg = Embedding_Table_Cache_Group(m_spa, np.array(ln_emb), cache_size, B, ways).to(DEV)
    dl = DLRM_Net(ln_bot, ln_top, "dot", False, True, -1, ln_top.size - 2, 0.0).to(DEV)
    eng = TrainEngine(cg, dl, host, lr=0.1, lr_embeds=0.3)
    pipe = WindowPipeline(cg, host, L * B, parity_rng=True, average_on_writeback=True)
    losses = []
    for j, (X, idx, Tt) in enumerate(batches):
        if j % L == 0:
            torch.manual_seed(300 + j)
            pipe.plan_window(torch.cat([b[1] for b in batches[j:j + L]], dim=1).to(DEV))
            pipe.commit()
            pipe.wait_writeback()
        losses.append(float(eng.step(X.to(DEV), idx.to(DEV), Tt.to(DEV), j=j)[0]))
    cg.ctx.check()
    np.testing.assert_allclose(np.array(losses), np.array([l[0] for l in otr.losses]), rtol=1e-5)
    np.random.seed(seed)
    fresh = O.init_host_tables(ln_emb, m_spa)
    changed = 0
    for k in range(len(ln_emb)):
        assert torch.equal(cg.occupancy_tables[k].cpu(), otr.occ[k]), k
        np.testing.assert_allclose(host.emb_l[k].weight.data.numpy(), otr.host[k].numpy(), rtol=2e-5, atol=1e-7)
        changed += int((host.emb_l[k].weight.data != fresh[k]).any(dim=1).sum())
    assert changed > 10, "the fixture must write rows back"


def test_qr_flag_builds_quotient_remainder_host_tables():
    """Embedding_Table_Group(qr_flag=True) (model_no_ddp.py:52-56): tables above qr_threshold become QREmbeddingBag
    pairs, the others stay plain and draw from the numpy stream exactly as without the flag; forward() runs the HIP
    operator; feeding the cache from a QR table fails as it does in the reference (no `.weight`)."""
    from cdlrm_amd.model_no_ddp import Embedding_Table_Group
    from cdlrm_amd.tricks.qr_embedding_bag import QREmbeddingBag
    from oracle import cdlrm_oracle as O
    ln_emb, m = np.array([1000, 50, 5000]), 8
    np.random.seed(5)
    torch.manual_seed(5)
    eg = Embedding_Table_Group(m, ln_emb, qr_flag=True, qr_operation="mult", qr_collisions=4, qr_threshold=200)
    assert isinstance(eg.emb_l[0], QREmbeddingBag) and isinstance(eg.emb_l[2], QREmbeddingBag)
    assert not isinstance(eg.emb_l[1], QREmbeddingBag)
    assert tuple(eg.emb_l[0].weight_q.shape) == (250, m) and tuple(eg.emb_l[0].weight_r.shape) == (4, m)
    np.random.seed(5)
    want = np.random.uniform(low=-np.sqrt(1 / 50), high=np.sqrt(1 / 50), size=(50, m)).astype(np.float32)
    assert np.array_equal(eg.emb_l[1].weight.data.numpy(), want)        # the only table that consumes numpy draws
    eg = eg.to(DEV)
    rng = np.random.RandomState(1)
    lS_i = [torch.from_numpy(rng.randint(0, n, 12)) for n in ln_emb]
    lS_o = [torch.tensor([0, 3, 4, 9]) for _ in ln_emb]
    ly = eg([o.to(DEV) for o in lS_o], [i.to(DEV) for i in lS_i])
    for k in (0, 2):
        ref = O.qr_embedding_bag(lS_i[k], lS_o[k], eg.emb_l[k].weight_q.detach().cpu(), eg.emb_l[k].weight_r.detach().cpu(), 4, "mult")
        np.testing.assert_allclose(ly[k].detach().cpu().numpy(), ref.numpy(), rtol=1e-6, atol=1e-7)
    ref1 = torch.nn.functional.embedding_bag(lS_i[1], eg.emb_l[1].weight.data.cpu(), lS_o[1], mode="sum")
    np.testing.assert_allclose(ly[1].cpu().numpy(), ref1.numpy(), rtol=1e-6, atol=1e-7)
    with pytest.raises(AttributeError):
        eg.fetch_unique_idx_slices([torch.tensor([1, 2]) for _ in ln_emb])


def test_md_flag_builds_mixed_dimension_host_tables():
    """Embedding_Table_Group(md_flag=True) (model_no_ddp.py:57-66) with md_solver's per-table widths: tables above
    md_threshold become PrEmbeddingBag(n, m[i], max(m)) drawn from the numpy stream; forward() runs the HIP operator and
    every table comes out max(m) wide; a table at or below the threshold fails as in the reference (one width needed)."""
    from cdlrm_amd.model_no_ddp import Embedding_Table_Group
    from cdlrm_amd.tricks.md_embedding_bag import PrEmbeddingBag, md_solver
    from oracle import cdlrm_oracle as O
    ln_emb = np.array([300, 90000, 2500])
    m = md_solver(torch.tensor(ln_emb), 0.3, d0=16, round_dim=True).long().tolist()     # sorted-table order, as :612-618
    assert m == [int(x) for x in O.md_solver(ln_emb, 0.3, d0=16)] and m[0] == 16 and m[-1] < 16
    np.random.seed(9)
    torch.manual_seed(9)
    eg = Embedding_Table_Group(m, ln_emb, md_flag=True, md_threshold=200)
    assert all(isinstance(E, PrEmbeddingBag) for E in eg.emb_l)
    np.random.seed(9)
    for k, n in enumerate(ln_emb):
        want = np.random.uniform(low=-np.sqrt(1 / n), high=np.sqrt(1 / n), size=(n, m[k])).astype(np.float32)
        assert np.array_equal(eg.emb_l[k].embs.weight.data.numpy(), want)
    eg = eg.to(DEV)
    rng = np.random.RandomState(2)
    lS_i = [torch.from_numpy(rng.randint(0, n, 12)) for n in ln_emb]
    lS_o = [torch.tensor([0, 3, 4, 9]) for _ in ln_emb]
    ly = eg([o.to(DEV) for o in lS_o], [i.to(DEV) for i in lS_i])
    for k, E in enumerate(eg.emb_l):
        P = E.proj.weight.detach().cpu() if m[k] < max(m) else None
        ref = O.pr_embedding_bag(lS_i[k], lS_o[k], E.embs.weight.detach().cpu(), P)
        assert tuple(ly[k].shape) == (4, max(m))
        np.testing.assert_allclose(ly[k].detach().cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)
    with pytest.raises(AttributeError):
        eg.fetch_unique_idx_slices([torch.tensor([1, 2]) for _ in ln_emb])
    with pytest.raises(TypeError):
        Embedding_Table_Group(m, ln_emb, md_flag=True, md_threshold=1000)


# ---- two trainer processes: aggregate_gradients + broadcast_and_aggregate + load_caches_and_broadcast ----------------

def _w2_worker(rid, port, name, host_shared, ret):
    import faulthandler
    faulthandler.dump_traceback_later(200, exit=True)
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import torch.distributed as dist
        from test_distributed_gloo import _batches
        from cdlrm_amd.cache_manager import Prefetcher
        from cdlrm_amd.main_no_ddp import (aggregate_gradients, broadcast_and_aggregate, load_caches_and_broadcast,
                                           share_occupancy_tables, wait_wrap)
        import cdlrm_amd.model_no_ddp as M
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rid, world_size=2)
        torch.cuda.set_device(0)
        dev = "cuda:0"
        g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        ln_emb = np.array([int(x) for x in g["ln_emb"]])
        m_spa, seed, B, L = int(g["m_spa"]), int(g["seed"]), int(g["B"]), int(g["L"])
        agg_freq, agg_op = int(g["agg_freq"]), str(g["agg_op"])
        T = len(ln_emb)
        lbs = B // 2
        nf = T + 1
        ln_top = np.array([m_spa + nf * (nf - 1) // 2] + [int(x) for x in g["top"]])
        emb_tables = M.Embedding_Table_Group(m_spa, ln_emb, init="empty_meta")
        for k in range(T):
            emb_tables.emb_l[k].weight.data = host_shared[k]
        emb_tables.register_shared()
        np.random.seed(seed)
        torch.manual_seed(seed)
        cache_group = M.Embedding_Table_Cache_Group(m_spa, ln_emb, int(g["cache_size"]), B, int(g["ways"])).to(dev)
        dlrm = M.DLRM_Net(np.array(g["ln_bot"]), ln_top, "dot", False, True, -1, ln_top.size - 2, 0.0).to(dev)
        share_occupancy_tables(cache_group, None, rid)
        loss_fn = M.HipBCELoss()
        optimizer_mlps = torch.optim.SGD(dlrm.parameters(), lr=float(g["lr"]))
        optimizer_embeds = M.CacheSGD(cache_group, lr=float(g["lr_emb"]))
        batches = _batches(g)
        lS_o = torch.arange(lbs, dtype=torch.int64).repeat(T, 1)
        losses, window = [], []
        for j, (X, lS_i, Tt) in enumerate(batches):
            sl = slice(rid * lbs, (rid + 1) * lbs)
            if j % L == 0:
                batch_fifo, eviction_fifo = queue.Queue(), queue.Queue()
                if rid == 0:        # only rank 0's FIFO holds the window (the reference's rank 0 alone calls get())
                    win = torch.cat([b[1] for b in batches[j:j + L]], dim=1)
                    batch_fifo.put(Prefetcher.process_batch_slice(win, emb_tables))
                    torch.manual_seed(5000 + j)
                reqs = load_caches_and_broadcast(cache_group, batch_fifo, eviction_fifo, rid)
                wait_wrap(reqs)
                if rid == 0:
                    evq = queue.Queue()
                    evq.put(eviction_fifo.get_nowait())
                    Prefetcher.eviction_manager(emb_tables, evq, False, min(os.sched_getaffinity(0)), 1)
                else:
                    assert eviction_fifo.empty()
                dist.barrier()
            lookups, cgi = cache_group(lS_o, lS_i[:, sl], emb_tables, rid)
            Z = dlrm(X[sl].to(dev), lookups)
            E = loss_fn(Z, Tt[sl].to(dev))
            optimizer_mlps.zero_grad()
            optimizer_embeds.zero_grad()
            E.backward()
            reqs = aggregate_gradients(dlrm)
            optimizer_embeds.step()
            wait_wrap(reqs)
            optimizer_mlps.step()
            if j > 0 and j % agg_freq == 0:
                idxs = torch.cat(window + [torch.stack(cgi)], dim=1)
                broadcast_and_aggregate(cache_group, idxs, rid, agg_op)
                window = []
            else:
                window.append(torch.stack(cgi))
            losses.append(float(E.detach()))
        cache_group.ctx.check()
        lin = M._linears(dlrm.top_l)
        ret.put((rid, dict(losses=np.array(losses), occ=[o.cpu().numpy() for o in cache_group.occupancy_tables],
                           top_w=[l.weight.data.cpu().numpy() for l in lin],
                           top_b=[l.bias.data.cpu().numpy() for l in lin],
                           wsum=[float(cache_group.emb_l[k].weight[: int(g["ways"]) * cache_group.cache_sizes[k]].double().sum())
                                 for k in range(T)])))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException:
        import traceback
        ret.put((rid, {"error": traceback.format_exc()}))
        raise


@pytest.mark.parametrize("name,port", [("train_w2_mean", 29841), ("train_w2_max", 29842), ("train_w2_freq1", 29843),
                                       ("train_w2_sum", 29844)])
def test_reference_named_collectives_two_ranks(golden, name, port):
    """aggregate_gradients + wait_wrap, broadcast_and_aggregate (mean / max, merge every 3 / 2 / 1 iterations) and
    load_caches_and_broadcast in the reference's own loop shape, two processes on one GPU over gloo, against the
    reference's two-process run: per-rank losses, tags, per-rank top-MLP weights AND biases (bias gradients are not
    reduced, main_no_ddp.py:237-245), cache-row checksums, host tables."""
    from oracle import cdlrm_oracle as O
    g = golden(name)
    np.random.seed(int(g["seed"]))
    host = [h.share_memory_() for h in O.init_host_tables([int(x) for x in g["ln_emb"]], int(g["m_spa"]))]
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_w2_worker, args=(r, port, name, host, ret)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, payload = ret.get(timeout=300)
        assert "error" not in payload, payload["error"]
        got[r] = payload
    for p in procs:
        p.join(timeout=60)
    for r in range(2):
        np.testing.assert_allclose(got[r]["losses"], g[f"r{r}_losses"], rtol=1e-5)
        for k in range(len(g["ln_emb"])):
            assert np.array_equal(got[r]["occ"][k], g[f"occ_{k}"]), (r, k)
            np.testing.assert_allclose(got[r]["wsum"][k], float(g[f"r{r}_weight_sum_{k}"]), rtol=1e-5, atol=1e-4)
        for i in range(len(got[r]["top_w"])):
            np.testing.assert_allclose(got[r]["top_w"][i], g[f"r{r}_top_w{i}"], rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(got[r]["top_b"][i], g[f"r{r}_top_b{i}"], rtol=1e-4, atol=1e-6)
    for k in range(len(g["ln_emb"])):
        np.testing.assert_allclose(float(host[k].double().sum()), float(g[f"host_sum_{k}"]), rtol=1e-6)
