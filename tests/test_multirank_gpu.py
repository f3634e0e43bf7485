"""Two trainer processes with the REAL HIP kernels (both on cuda:0, collectives over gloo because one GPU cannot
host two RCCL ranks) against the reference's 2-process golden run: exercises the shared, registered host tables
(rank 0 writes evictions back, rank 1 reads them over PCIe), replicated deterministic inserts, sync-to-rank-0,
the touched-row merge and the flat weight-grad all-reduce."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, name, host_shared, ret, defer=False, chunk=0, long_batch=False):
    import faulthandler
    faulthandler.dump_traceback_later(150, exit=True)        # a stuck worker says where, instead of a silent time-out
    try:
        _worker_body(rank, world, port, name, host_shared, ret, defer, chunk, long_batch)
    except BaseException as e:      # a dead worker must fail the test, not hang it
        import traceback
        ret.put((rank, {"error": traceback.format_exc()}))
        raise


def _worker_body(rank, world, port, name, host_shared, ret, defer=False, chunk=0, long_batch=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from test_distributed_gloo import _batches
    import cdlrm_amd.engine as engine
    import cdlrm_amd.model_no_ddp as M
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = "cuda:0"
    torch.cuda.set_device(0)
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    ln_emb = np.array([int(x) for x in g["ln_emb"]])
    m_spa, seed, B, L = int(g["m_spa"]), int(g["seed"]), int(g["B"]), int(g["L"])
    nf = len(ln_emb) + 1
    ln_top = np.array([m_spa + nf * (nf - 1) // 2] + [int(x) for x in g["top"]])
    eg = M.Embedding_Table_Group(m_spa, ln_emb, init="empty_meta")
    for k in range(len(ln_emb)):
        eg.emb_l[k].weight.data = host_shared[k]
    eg.register_shared()
    np.random.seed(seed)
    torch.manual_seed(seed)
    cg = M.Embedding_Table_Cache_Group(m_spa, ln_emb, int(g["cache_size"]), B, int(g["ways"])).to(dev)
    dl = M.DLRM_Net(np.array(g["ln_bot"]), ln_top, "dot", False, True, -1, ln_top.size - 2, 0.0).to(dev)
    eng = engine.TrainEngine(cg, dl, eg, lr=float(g["lr"]), lr_embeds=float(g["lr_emb"]), world_size=world, rank=rank,
                             table_agg_freq=int(g["agg_freq"]), table_agg_op=str(g["agg_op"]), defer_top_update=defer)
    pipe = engine.WindowPipeline(cg, eg, L * B, parity_rng=True, rank=rank, world_size=world)
    if chunk:       # the touched-row merge in chunks of `chunk` rows: gather i+1 / reduce i / scatter i-1 pipelined
        eng.agg_chunk_rows = chunk
    if long_batch:  # the long-batch schedule (gather alone on the main stream, chained take on the window-resident probe)
        eng.gather_alone_min = 1
    lbs = -(-B // world)           # ceil (main_no_ddp.py:344): the last rank's slice is shorter when world does not divide B
    losses = []
    dev_idx = {}
    batches = _batches(g)
    for j, (X, lS_i, Tt) in enumerate(batches):
        if j % L == 0:
            eng.sync_touched_to_rank0()
            torch.manual_seed(5000 + j)
            pipe.plan_window(torch.cat([b[1] for b in batches[j:j + L]], dim=1).to(dev))
            pipe.commit()
            pipe.wait_writeback()
            rs = engine.WindowResolver(eng, torch.cat([b[1] for b in batches[j:j + L]], dim=1).to(dev), B, chunk=2) \
                if long_batch else None
        sl = slice(rank * lbs, (rank + 1) * lbs)
        if j not in dev_idx:
            dev_idx[j] = lS_i[:, sl].contiguous().to(dev)
        nxt = None
        if j + 1 < len(batches) and (j + 1) % L != 0:       # same window: pipeline the next batch's probe / aux fill
            dev_idx[j + 1] = batches[j + 1][1][:, sl].contiguous().to(dev)
            nxt = dev_idx[j + 1]
        loss = eng.step(X[sl].to(dev), dev_idx[j], Tt[sl].to(dev), j=j, next_idx=nxt,
                        res=rs.batch(j % L) if rs is not None else None,
                        next_res=rs.batch(j % L + 1) if (rs is not None and nxt is not None) else None)
        if rs is not None:
            rs.ensure(j % L + rs.CH + 2)
            assert nxt is None or eng._pref is not None       # the chained take was issued
        losses.append(float(loss[0]))
    eng.finish()
    cg.ctx.check()
    lin = M._linears(dl.top_l)
    ret.put((rank, dict(losses=np.array(losses), occ=[o.cpu().numpy() for o in cg.occupancy_tables],
                        top_w=[l.weight.data.cpu().numpy() for l in lin])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,port,defer,chunk,long_batch", [
    ("train_w2_mean", 29821, False, 0, False), ("train_w2_max", 29822, False, 0, False),
    ("train_w2_mean", 29823, True, 0, False), ("train_w2_freq1", 29824, True, 16, False),
    ("train_w2_max", 29825, False, 8, False), ("train_w2_sum", 29826, True, 0, False),
    # the long-batch schedule on two ranks: chained take + window-resident probe, a row merge every step / every other
    ("train_w2_freq1", 29827, True, 0, True), ("train_w2_mean", 29828, False, 0, True)])
def test_two_ranks_one_gpu_match_reference(golden, name, port, defer, chunk, long_batch):
    from oracle import cdlrm_oracle as O
    g = golden(name)
    np.random.seed(int(g["seed"]))
    host = [h.share_memory_() for h in O.init_host_tables([int(x) for x in g["ln_emb"]], int(g["m_spa"]))]
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, host, ret, defer, chunk, long_batch)) for r in range(2)]
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
        for i in range(len(got[r]["top_w"])):
            np.testing.assert_allclose(got[r]["top_w"][i], g[f"r{r}_top_w{i}"], rtol=1e-4, atol=1e-6)
    for k in range(len(g["ln_emb"])):
        np.testing.assert_allclose(float(host[k].double().sum()), float(g[f"host_sum_{k}"]), rtol=1e-6)


@pytest.mark.parametrize("name,port,long_batch", [("train_w2_mean", 29841, True), ("train_w2_freq1", 29842, False)])
def test_three_ranks_one_gpu_short_last_slice_vs_oracle(golden, name, port, long_batch):
    """Three trainer processes with the real kernels on a batch of 32: lbs = ceil(32 / 3) = 11, the last rank trains on 10
    samples (the reference raises there, main_no_ddp.py:388-391 -- oracle and engine define the natural extension).  With the
    long-batch schedule the window-resident probe numbers the misses per GLOBAL batch and rank slice
    (cdlrm_window_resolve(batch_len, seg_len)): a numbering that ran across batch boundaries would hand two misses of one
    rank-batch the same aux row.  Against the oracle's 3-rank emulation: per-rank losses, shared tags, replicated weights."""
    from oracle import cdlrm_oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_golden import make_batches as oracle_batches
    world = 3
    g = golden(name)
    ln_emb = [int(x) for x in g["ln_emb"]]
    L, m_spa = int(g["L"]), int(g["m_spa"])
    ln_top = np.array([m_spa + (len(ln_emb) + 1) * len(ln_emb) // 2] + list(g["top"]))
    tr = O.OracleTrainer(ln_emb, m_spa, g["ln_bot"], ln_top, cache_size=int(g["cache_size"]), num_ways=int(g["ways"]),
                         mini_batch_size=int(g["B"]), world_size=world, lr=float(g["lr"]), lr_embeds=float(g["lr_emb"]),
                         lookahead=L, table_agg_freq=int(g["agg_freq"]), table_agg_op=str(g["agg_op"]), seed=int(g["seed"]))
    ob = oracle_batches(g)
    for j, (X, lS_o, lS_i, Tt) in enumerate(ob):
        if j % L == 0:
            torch.manual_seed(5000 + j)
            tr.refill(torch.cat([b[2] for b in ob[j:j + L]], dim=1))
        tr.step(j, X, lS_o, lS_i, Tt)
    np.random.seed(int(g["seed"]))
    host = [h.share_memory_() for h in O.init_host_tables(ln_emb, m_spa)]
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, host, ret, True, 0, long_batch)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, payload = ret.get(timeout=300)
        assert "error" not in payload, payload["error"]
        got[r] = payload
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        np.testing.assert_allclose(got[r]["losses"], np.array([l[r] for l in tr.losses]), rtol=1e-5)
        for k in range(len(ln_emb)):
            assert np.array_equal(got[r]["occ"][k], tr.occ[k].numpy()), (r, k)
        for i in range(len(got[r]["top_w"])):
            np.testing.assert_allclose(got[r]["top_w"][i], tr.top[r][0][i].numpy(), rtol=1e-4, atol=1e-6)
            assert np.array_equal(got[r]["top_w"][i], got[0]["top_w"][i])
    for k in range(len(ln_emb)):
        np.testing.assert_allclose(float(host[k].double().sum()), float(tr.host[k].double().sum()), rtol=1e-6)


def _shard_worker(rank, world, port, host_shared, ret):
    import faulthandler
    faulthandler.dump_traceback_later(150, exit=True)
    try:
        sys.path.insert(0, ROOT)
        import torch.distributed as dist
        import cdlrm_amd.engine as engine
        import cdlrm_amd.model_no_ddp as M
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = "cuda:0"
        torch.cuda.set_device(0)
        ln_emb, m_spa, B, L, ways, cache = np.array([6000, 90, 11, 2500]), 16, 64, 6, 4, 60
        eg = M.Embedding_Table_Group(m_spa, ln_emb, init="empty_meta")
        for k in range(len(ln_emb)):
            eg.emb_l[k].weight.data = host_shared[k]
        eg.register_shared()
        rng = np.random.RandomState(9)
        wins = [torch.stack([torch.from_numpy((rng.zipf(1.15, size=L * B).astype(np.int64) * 2654435761 % n)) for n in ln_emb])
                for _ in range(4)]
        out = {}
        for shard in (True, False):
            torch.manual_seed(3)
            cg = M.Embedding_Table_Cache_Group(m_spa, ln_emb, cache, B, ways).to(dev)
            pipe = engine.WindowPipeline(cg, eg, L * B, parity_rng=False, seed=77, rank=rank, world_size=world,
                                         host_gather=True, gather_threads=2, shard_fetch=shard)
            assert pipe.shard == shard and pipe.host_gather
            snaps, exchanged = [], 0
            for w in wins:
                pipe.plan_window(w.to(dev))
                pipe._worker.join()
                exchanged += len(pipe._exchange)
                pipe.commit()
                pipe.wait_writeback()           # barrier: rank 0's evictions are in the host tables before the next plan
                torch.cuda.synchronize()
                vic = pipe.victims[pipe._vnext ^ 1]
                nv = int(vic.off.cpu()[-2])           # off[T]: entries listed
                snaps.append((cg.tags.cpu().clone(), cg.weight.cpu().clone(), nv, vic.idx[:nv].cpu().clone(),
                              vic.rows[:nv].cpu().clone()))
            cg.ctx.check()
            out[shard] = (snaps, exchanged)
        assert out[True][1] == 2 * len(wins) and out[False][1] == 0, "winners + victims must travel as slices"
        assert sum(s[2] for s in out[True][0]) > 0, "the fixture must produce victims"
        for a, b in zip(out[True][0], out[False][0]):
            assert torch.equal(a[0], b[0]), "tags"
            assert a[2] == b[2] and torch.equal(a[3], b[3]), "victim list"
            assert torch.equal(a[4], b[4]), "victim rows"
            # rows of occupied slots (never-filled slots keep the N(0,1) init of each cache group: same seed, equal too)
            assert torch.equal(a[1], b[1]), "cache rows"
        ret.put((rank, {"ok": True}))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException:
        import traceback
        ret.put((rank, {"error": traceback.format_exc()}))
        raise


def test_sharded_window_fetch_two_ranks():
    """WindowPipeline(shard_fetch=True): each rank fetches half of the winners' / victims' rows from the host tables, the
    halves meet in one all-gather per list at commit().  Same cache rows, tags and victim rows as the unsharded plan,
    window after window (evictions written back by rank 0 in between)."""
    from oracle import cdlrm_oracle as O
    np.random.seed(4)
    host = [h.share_memory_() for h in O.init_host_tables([6000, 90, 11, 2500], 16)]
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, 29831, host, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for _ in range(2):
        r, payload = ret.get(timeout=300)
        assert "error" not in payload, payload["error"]
    for p in procs:
        p.join(timeout=60)


def _overlap_worker(rank, world, port, host_shared, ret):
    import faulthandler
    faulthandler.dump_traceback_later(200, exit=True)
    try:
        sys.path.insert(0, ROOT)
        import torch.distributed as dist
        import cdlrm_amd.engine as engine
        import cdlrm_amd.model_no_ddp as M
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = "cuda:0"
        torch.cuda.set_device(0)
        ln_emb, m_spa, B, L, ways, cache, nwin = np.array([6000, 90, 11, 2500, 30000]), 16, 128, 6, 4, 60, 4
        lbs = B // world
        nf = len(ln_emb) + 1
        ln_bot, ln_top = np.array([13, 32, m_spa]), np.array([m_spa + nf * (nf - 1) // 2, 32, 1])
        eg = M.Embedding_Table_Group(m_spa, ln_emb, init="empty_meta")
        for k in range(len(ln_emb)):
            eg.emb_l[k].weight.data = host_shared[k]
        eg.register_shared()
        rng = np.random.RandomState(9)
        wins = [torch.stack([torch.from_numpy((rng.zipf(1.15, size=L * B).astype(np.int64) * 2654435761 % n)) for n in ln_emb])
                .to(dev) for _ in range(nwin)]
        Xs = [torch.from_numpy(rng.rand(B, 13).astype(np.float32)).to(dev) for _ in range(L)]
        Ts = [torch.from_numpy(np.round(rng.rand(B, 1)).astype(np.float32)).to(dev) for _ in range(L)]
        host0 = [h.clone() for h in host_shared] if rank == 0 else None
        out = {}
        for lookahead in (True, False):
            dist.barrier()
            if rank == 0:
                for h, h0 in zip(host_shared, host0):
                    h.copy_(h0)
            dist.barrier()
            np.random.seed(3)
            torch.manual_seed(3)
            cg = M.Embedding_Table_Cache_Group(m_spa, ln_emb, cache, B, ways).to(dev)
            dl = M.DLRM_Net(ln_bot, ln_top, "dot", False, True, -1, ln_top.size - 2, 0.0).to(dev)
            eng = engine.TrainEngine(cg, dl, eg, lr=0.1, lr_embeds=0.3, world_size=world, rank=rank, table_agg_freq=1,
                                     table_agg_op="mean", defer_top_update=True)
            eng.agg_chunk_rows = 32
            pipe = engine.WindowPipeline(cg, eg, L * B, parity_rng=False, seed=77, rank=rank, world_size=world,
                                         host_gather=True, gather_threads=2)
            losses, in_flight = [], 0
            planned = False
            for w in range(nwin):
                if not planned:
                    pipe.plan_window(wins[w])
                eng.sync_touched_to_rank0()
                pipe.commit()
                pipe.wait_writeback()
                planned = False
                if lookahead and w + 1 < nwin:
                    # the plan of window w+1 (scans + compactions on the plan stream, CPU row gather in a thread) runs
                    # WHILE the steps below merge touched rows every iteration (compactions on the main stream)
                    pipe.plan_window(wins[w + 1])
                    planned = True
                for jj in range(L):
                    sl = slice(jj * B + rank * lbs, jj * B + (rank + 1) * lbs)
                    nxt = wins[w][:, (jj + 1) * B + rank * lbs:(jj + 1) * B + (rank + 1) * lbs] if jj + 1 < L else None
                    in_flight += int(pipe.plan_in_flight())
                    lo = eng.step(Xs[jj][rank * lbs:(rank + 1) * lbs], wins[w][:, sl], Ts[jj][rank * lbs:(rank + 1) * lbs],
                                  j=jj, next_idx=nxt)
                    losses.append(lo[0:1].clone())
            eng.finish()
            cg.ctx.check()
            torch.cuda.synchronize()
            out[lookahead] = (torch.cat(losses).cpu(), cg.tags.cpu().clone(), cg.weight.data.cpu().clone(), in_flight)
            dist.barrier()
        a, b = out[True], out[False]
        assert torch.equal(a[0], b[0]), "losses differ between the in-flight and the boundary plan"
        assert torch.equal(a[1], b[1]), "tags"
        assert torch.equal(a[2], b[2]), "cache rows"
        ret.put((rank, {"ok": True, "in_flight": a[3]}))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException:
        import traceback
        ret.put((rank, {"error": traceback.format_exc()}))
        raise


def test_plan_in_flight_while_rows_merge_every_step():
    """Two ranks, --table-agg-freq=1 (a touched-row merge, i.e. a stream compaction on the MAIN stream, every step) while
    the NEXT window's plan (its own compactions, on the plan stream) is in flight: the two compactions use separate
    scratch, so the run equals -- bit for bit: losses, tags, every cache row -- the run that plans at the boundary with
    nothing else in flight.  (With one shared block-sum scratch the scans overwrite each other: wrong winner lists.)"""
    from oracle import cdlrm_oracle as O
    np.random.seed(4)
    host = [h.share_memory_() for h in O.init_host_tables([6000, 90, 11, 2500, 30000], 16)]
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, 29851, host, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for _ in range(2):
        r, payload = ret.get(timeout=400)
        assert "error" not in payload, payload["error"]
    for p in procs:
        p.join(timeout=60)


# ---- config c5 at world > 1: lbs = 8192 per rank, streamed windows, a resolver per chunk, the long-batch schedule ----------
C5 = dict(cap=30000, D=128, ways=16, cache=500, B=16384, L=4, C=2, nwin=2, agg=3, seed=31, alpha=1.05, lr=0.8, lr_emb=0.8,
          bot=[13, 512, 256, 128], top=[512, 512, 256, 1])


def _c5_data():
    """Indices / dense features / targets of the capped c5 shape, from the CPU restatement of the synthetic stream (the ranks
    and the oracle must see the same integers: the device generator may differ from it in a last place of a float64 pow)."""
    sys.path.insert(0, ROOT)
    from cdlrm_amd.synth import CriteoSynth, TERABYTE_COUNTS
    ln_emb = [min(n, C5["cap"]) for n in TERABYTE_COUNTS]
    syn = CriteoSynth(ln_emb, 13, C5["B"], seed=C5["seed"], alpha=C5["alpha"], device="cpu")
    wins = [syn.window(w, C5["L"]) for w in range(C5["nwin"])]
    dense = [syn.dense(j) for j in range(C5["L"] * C5["nwin"])]
    return ln_emb, wins, dense


def _c5_worker(rank, world, port, host_shared, ret, chained=True):
    import faulthandler
    # a stuck rank says where, well inside the runner's patience -- into a file the GPU runner brings back even if it has to
    # kill the whole run (the parent's captured stderr would be lost with it)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    trace = open(os.path.join(ROOT, "gpurun_out", "c5_flow_rank%d.trace" % rank), "w")
    faulthandler.dump_traceback_later(200, exit=True, file=trace)
    try:
        sys.path.insert(0, ROOT)
        import torch.distributed as dist
        import cdlrm_amd.engine as engine
        import cdlrm_amd.model_no_ddp as M
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = "cuda:0"
        torch.cuda.set_device(0)
        ln_emb_l, wins, dense = _c5_data()
        ln_emb = np.array(ln_emb_l)
        D, B, L, C = C5["D"], C5["B"], C5["L"], C5["C"]
        lbs = B // world
        nf = len(ln_emb) + 1
        ln_top = np.array([D + nf * (nf - 1) // 2] + C5["top"])
        eg = M.Embedding_Table_Group(D, ln_emb, init="empty_meta")
        for k in range(len(ln_emb)):
            eg.emb_l[k].weight.data = host_shared[k]
        eg.register_shared()
        np.random.seed(C5["seed"])
        torch.manual_seed(C5["seed"])
        cg = M.Embedding_Table_Cache_Group(D, ln_emb, C5["cache"], B, C5["ways"]).to(dev)
        dl = M.DLRM_Net(np.array(C5["bot"]), ln_top, "dot", False, True, -1, ln_top.size - 2, 0.0).to(dev)
        eng = engine.TrainEngine(cg, dl, eg, lr=C5["lr"], lr_embeds=C5["lr_emb"], world_size=world, rank=rank,
                                 table_agg_freq=C5["agg"], table_agg_op="mean", defer_top_update=True)
        eng.agg_chunk_rows = 4096           # the touched-row merge in several chunks on the exchange stream
        # the chained-take schedule (c5 on 1 - 4 ranks: local batches >= 16384; since round 5 a local batch of 8192 takes the
        # two-region schedule by default, which the other multi-rank cases run): forced here at the size this box can check
        if chained:
            eng.gather_alone_min = lbs
            assert not eng._side_gather(lbs), "the long-batch schedule (interaction forward alone on the training queue, chained take)"
        else:
            # the production choice at this local batch: two aux regions, the look-ahead resolve PLACED behind the interaction
            # forward (place_resolve_min) -- at world > 1 behind the row merge's deadline pass as well
            assert eng._side_gather(lbs) and lbs >= eng.place_resolve_min
        pipe = engine.WindowPipeline(cg, eg, L * B, parity_rng=True, rank=rank, world_size=world)
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
        losses, j = [], 0
        for w, win in enumerate(wins):
            win_d = win.to(dev)
            eng.sync_touched_to_rank0()
            torch.manual_seed(5000 + w * L)
            # the window reaches the plan as a STREAM of chunks of C batches (cdlrm_window_unique_add / _finish)
            pipe.plan_window(lambda: (win_d[:, c * C * B:(c + 1) * C * B].contiguous() for c in range(L // C)))
            pipe.commit()
            pipe.wait_writeback()
            rs = chunk = None
            for jj in range(L):
                c, jc = divmod(jj, C)
                if jc == 0:         # a resolver per chunk, as bench.py runs streamed windows
                    chunk = win_d[:, c * C * B:(c + 1) * C * B].contiguous()
                    rs = engine.WindowResolver(eng, chunk, B, chunk=1)
                col = jc * B + rank * lbs
                idx = chunk[:, col:col + lbs]
                nxt = chunk[:, col + B:col + B + lbs] if jc + 1 < C else None
                X, T = dense[j]
                loss = eng.step(X[rank * lbs:(rank + 1) * lbs].to(dev), idx, T[rank * lbs:(rank + 1) * lbs].to(dev), j=jj,
                                next_idx=nxt, res=rs.batch(jc), next_res=rs.batch(jc + 1) if nxt is not None else None)
                rs.ensure(jc + rs.CH + 2)
                losses.append(float(loss[0]))
                j += 1
        eng.finish()
        cg.ctx.check()
        lin = M._linears(dl.top_l)
        ret.put((rank, dict(losses=np.array(losses), occ=[o.cpu().numpy() for o in cg.occupancy_tables],
                            top_w=[l.weight.data.cpu().numpy() for l in lin])))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException:
        import traceback
        ret.put((rank, {"error": traceback.format_exc()}))
        raise


@pytest.mark.parametrize("chained,port", [(True, 29865), (False, 29866)])
def test_c5_flow_two_ranks_vs_oracle(chained, port):
    """BASELINE configs[4] at world > 1 on its own code path: a local batch of 8192 per rank (the long-batch schedule: gather
    alone on the training queue, chained take, resolve placed inside the step), the window streamed into the plan in chunks, a
    window-resident resolver per chunk, the touched-row merge in chunks on the exchange stream, deferred top-MLP update --
    26 capped Terabyte tables, D = 128, 16-way, the c3 / c5 MLPs -- against the oracle's 2-rank emulation: per-rank loss of every
    iteration 1e-5, shared tag state bit-exact, replicated weights."""
    from oracle import cdlrm_oracle as O
    world = 2
    ln_emb, wins, dense = _c5_data()
    D, B, L = C5["D"], C5["B"], C5["L"]
    nf = len(ln_emb) + 1
    ln_top = np.array([D + nf * (nf - 1) // 2] + C5["top"])
    np.random.seed(C5["seed"])
    host0 = O.init_host_tables(ln_emb, D)
    host = [h.clone().share_memory_() for h in host0]
    # The oracle first, the ranks afterwards (as the other multi-rank tests do).  Run WHILE the ranks trained, this process's
    # torch-CPU work crawled inside the full suite (> 200 s instead of 12: sixteen OpenMP threads against the ranks' host threads
    # on the box's CPU share), the finished ranks sat in their result queue's pipe until their watchdog ended them, and the
    # half-written results hung the parent.
    import faulthandler
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    ptrace = open(os.path.join(ROOT, "gpurun_out", "c5_flow_parent.trace"), "w")
    faulthandler.dump_traceback_later(330, exit=False, file=ptrace)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    tr = O.OracleTrainer(ln_emb, D, np.array(C5["bot"]), ln_top, cache_size=C5["cache"], num_ways=C5["ways"],
                         mini_batch_size=B, world_size=world, lr=C5["lr"], lr_embeds=C5["lr_emb"], lookahead=L,
                         table_agg_freq=C5["agg"], table_agg_op="mean", seed=C5["seed"], host_tables=host0)
    lS_o = torch.arange(B // world).repeat(len(ln_emb), 1)
    j = 0
    for w, win in enumerate(wins):
        torch.manual_seed(5000 + w * L)
        tr.refill(win)
        for jj in range(L):
            X, T = dense[j]
            tr.step(jj, X, lS_o, win[:, jj * B:(jj + 1) * B], T)
            j += 1
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_c5_worker, args=(r, world, port, host, ret, chained)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, payload = ret.get(timeout=300)
        assert "error" not in payload, payload["error"]
        got[r] = payload
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        np.testing.assert_allclose(got[r]["losses"], np.array([l[r] for l in tr.losses]), rtol=1e-5)
        for k in range(len(ln_emb)):
            assert np.array_equal(got[r]["occ"][k], tr.occ[k].numpy()), (r, k)
        for i in range(len(got[r]["top_w"])):
            # (eight steps at lr = 0.8 on 8192-sample gradients: weights of magnitude ~0.05 agree to ~1e-5 absolute)
            np.testing.assert_allclose(got[r]["top_w"][i], tr.top[r][0][i].numpy(), rtol=2e-4, atol=5e-5)
            assert np.array_equal(got[r]["top_w"][i], got[0]["top_w"][i])
    faulthandler.cancel_dump_traceback_later()
    for k in range(len(ln_emb)):        # (evicted rows carry eight steps of lr = 0.8 updates in another summation order)
        np.testing.assert_allclose(float(host[k].double().sum()), float(tr.host[k].double().sum()), rtol=2e-5)


# ---- the touched-row merge in deadline order (engine.MergePump) ----------------------------------------------------------
LZ = dict(ln_emb=[6000, 90, 11, 2500, 30000], m_spa=16, B=64, L=24, nwin=2, agg=7, ways=4, cache=60, seed=21, alpha=1.15,
          lr=0.1, lr_emb=0.3, bot=[13, 32, 16], top=[32, 1])


def _lz_batches():
    rng = np.random.RandomState(LZ["seed"] + 1)
    out = []
    for _ in range(LZ["L"] * LZ["nwin"]):
        X = torch.from_numpy(rng.rand(LZ["B"], 13).astype(np.float32))
        idx = torch.stack([torch.from_numpy((rng.zipf(LZ["alpha"], size=LZ["B"]).astype(np.int64) * 2654435761 % n))
                           for n in LZ["ln_emb"]])
        T = torch.from_numpy(np.round(rng.rand(LZ["B"], 1)).astype(np.float32))
        out.append((X, idx, T))
    return out


def _lz_worker(rank, world, port, host_shared, ret, long_batch, auto_budget=False):
    import faulthandler
    faulthandler.dump_traceback_later(300, exit=True)
    try:
        sys.path.insert(0, ROOT)
        import torch.distributed as dist
        import cdlrm_amd.engine as engine
        import cdlrm_amd.model_no_ddp as M
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = "cuda:0"
        torch.cuda.set_device(0)
        ln_emb, D, B, L = np.array(LZ["ln_emb"]), LZ["m_spa"], LZ["B"], LZ["L"]
        lbs = -(-B // world)            # ceil: the last rank's slice is shorter when world does not divide B
        r0, r1 = min(rank * lbs, B), min((rank + 1) * lbs, B)
        nf = len(ln_emb) + 1
        ln_top = np.array([D + nf * (nf - 1) // 2] + LZ["top"])
        batches = _lz_batches()
        host0 = [h.clone() for h in host_shared] if rank == 0 else None
        outs = {}
        eval_at, evals = -1, []
        for lazy in (True, False):
            dist.barrier()
            if rank == 0:
                for h, h0 in zip(host_shared, host0):
                    h.copy_(h0)
            dist.barrier()
            eg = M.Embedding_Table_Group(D, ln_emb, init="empty_meta")
            for k in range(len(ln_emb)):
                eg.emb_l[k].weight.data = host_shared[k]
            eg.register_shared()
            np.random.seed(LZ["seed"])
            torch.manual_seed(LZ["seed"])
            cg = M.Embedding_Table_Cache_Group(D, ln_emb, LZ["cache"], B, LZ["ways"]).to(dev)
            dl = M.DLRM_Net(np.array(LZ["bot"]), ln_top, "dot", False, True, -1, ln_top.size - 2, 0.0).to(dev)
            eng = engine.TrainEngine(cg, dl, eg, lr=LZ["lr"], lr_embeds=LZ["lr_emb"], world_size=world, rank=rank,
                                     table_agg_freq=LZ["agg"], table_agg_op="mean", defer_top_update=True)
            eng.lazy_merge = lazy
            eng.agg_chunk_rows, eng.merge_budget_rows, eng.merge_budget_auto = 8, 8, False     # small chunks: rows stay on their way
            if auto_budget:
                # the per-step budget follows the rank's lookups per step: it has to be the SAME number on a rank with a short
                # slice (it cuts the exchange into pieces; 22 * 5 rows on every rank, not 20 * 5 on the last)
                eng.merge_budget_auto = True
            if long_batch:
                eng.gather_alone_min = 1
            pipe = engine.WindowPipeline(cg, eg, L * B, parity_rng=True, rank=rank, world_size=world)
            losses, pending, deferred_rows = [], 0, 0
            for w in range(LZ["nwin"]):
                wb = batches[w * L:(w + 1) * L]
                win = torch.cat([b[1] for b in wb], dim=1).to(dev)
                eng.sync_touched_to_rank0()
                assert eng._pump is None
                torch.manual_seed(5000 + w * L)
                pipe.plan_window(win)
                pipe.commit()
                pipe.wait_writeback()
                rs = engine.WindowResolver(eng, win, B, chunk=4)
                for jj, (X, idx, T) in enumerate(wb):
                    col, w_ = jj * B + r0, r1 - r0
                    nxt = win[:, col + B:col + B + w_] if jj + 1 < L else None
                    loss = eng.step(X[r0:r1].to(dev), win[:, col:col + w_],
                                    T[r0:r1].to(dev), j=jj, next_idx=nxt, res=rs.batch(jj),
                                    next_res=rs.batch(jj + 1) if nxt is not None else None)
                    rs.ensure(jj + rs.CH + 2)
                    gstep = w * L + jj
                    if eng._pump is not None:
                        pending += 1
                        deferred_rows = max(deferred_rows, eng._pump["U"] - eng._pump["issued"])
                        if pending == 1 and rank == 0:
                            # a rank-local test loop (main_no_ddp.py:478-494 runs on rank 0 only) must not land the rows that
                            # are still travelling: that takes collectives, which this rank would issue alone
                            issued = eng._pump["issued"]
                            with pytest.raises(RuntimeError, match="drain_merge"):
                                eng.evaluate(batches[0][0].to(dev), batches[0][1].to(dev))
                            assert eng._pump is not None and eng._pump["issued"] == issued
                        if pending == 2:
                            eval_at = gstep
                    if gstep == eval_at:
                        # what Run does at a test step: every rank drains, then rank 0 alone evaluates
                        eng.drain_merge()
                        assert eng._pump is None
                        if rank == 0:
                            evals.append(eng.evaluate(batches[0][0].to(dev), batches[0][1].to(dev)).cpu().clone())
                    losses.append(loss[0:1].clone())
            eng.finish()
            cg.ctx.check()
            torch.cuda.synchronize()
            lin = M._linears(dl.top_l)
            outs[lazy] = dict(losses=torch.cat(losses).cpu(), tags=cg.tags.cpu().clone(),
                              rows=[cg.emb_l[k].weight.data[: LZ["ways"] * cg.cache_sizes[k]].cpu().clone() for k in range(len(ln_emb))],
                              top_w=[l.weight.data.cpu().clone() for l in lin], pending=pending, deferred=deferred_rows)
        a, b = outs[True], outs[False]
        assert a["pending"] > 0 and a["deferred"] > 0, "the fixture must leave merge rows on their way across steps"
        assert b["pending"] == 0
        # Two ranks: a + b is one rounding whatever the order, so the two schedules agree BIT FOR BIT.  Three and more: a ring
        # all-reduce sums an element's contributions in an order that depends on the element's place in the exchanged buffer
        # ((a + b) + c here, (b + c) + a there), so the same rows exchanged in other pieces agree to the last rounding only -- the
        # freedom the reference's own NCCL all-reduce has (main_no_ddp.py:279-284); the replicas still agree with each other.
        if world == 2:
            same = torch.equal
        else:
            same = lambda x, y: torch.allclose(x, y, rtol=2e-5, atol=2e-7)
        assert eval_at >= 0 and (rank != 0 or (len(evals) == 2 and same(evals[0], evals[1]))), \
            "rank 0's mid-training test batch behind drain_merge() must see the one-piece merge's rows"
        assert same(a["losses"], b["losses"]), "losses differ between the deadline-ordered and the one-piece merge"
        assert torch.equal(a["tags"], b["tags"])
        for x, y in zip(a["rows"], b["rows"]):
            assert same(x, y), "cache rows"
        for x, y in zip(a["top_w"], b["top_w"]):
            assert same(x, y)
        ret.put((rank, dict(losses=a["losses"].numpy(), occ=[o.cpu().numpy() for o in cg.occupancy_tables],
                            top_w=[w.numpy() for w in a["top_w"]], pending=a["pending"])))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException:
        import traceback
        ret.put((rank, {"error": traceback.format_exc()}))
        raise


@pytest.mark.parametrize("long_batch,port,world,auto_budget", [(False, 29871, 2, False), (True, 29872, 2, False),
                                                               (True, 29873, 3, True)])
def test_merge_in_deadline_order_two_ranks(long_batch, port, world, auto_budget):
    """engine.MergePump: the touched-row merge of step j (--table-agg-freq 7 inside windows of 24 batches) applied row by row
    before each row's next use -- the rows the next batch needs at once, the rest in deadline order over the following steps in
    small chunks -- against (a) the same run with the merge in one piece: bit for bit, losses, tags, cache rows, weights; (b)
    the oracle's 2-rank emulation of the reference's loop (main_no_ddp.py:387-423): per-rank losses 1e-5, tags exact."""
    from oracle import cdlrm_oracle as O
    ln_emb, D, B, L = LZ["ln_emb"], LZ["m_spa"], LZ["B"], LZ["L"]
    nf = len(ln_emb) + 1
    ln_top = np.array([D + nf * (nf - 1) // 2] + LZ["top"])
    np.random.seed(LZ["seed"])
    host0 = O.init_host_tables(ln_emb, D)
    host = [h.clone().share_memory_() for h in host0]
    torch.set_num_threads(1)
    tr = O.OracleTrainer(ln_emb, D, np.array(LZ["bot"]), ln_top, cache_size=LZ["cache"], num_ways=LZ["ways"], mini_batch_size=B,
                         world_size=world, lr=LZ["lr"], lr_embeds=LZ["lr_emb"], lookahead=L, table_agg_freq=LZ["agg"],
                         table_agg_op="mean", seed=LZ["seed"], host_tables=host0)
    batches = _lz_batches()
    lS_o = torch.arange(-(-B // world)).repeat(len(ln_emb), 1)
    for w in range(LZ["nwin"]):
        wb = batches[w * L:(w + 1) * L]
        torch.manual_seed(5000 + w * L)
        tr.refill(torch.cat([b[1] for b in wb], dim=1))
        for jj, (X, idx, T) in enumerate(wb):
            tr.step(jj, X, lS_o, idx, T)
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_lz_worker, args=(r, world, port, host, ret, long_batch, auto_budget)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, payload = ret.get(timeout=400)
        assert "error" not in payload, payload["error"]
        got[r] = payload
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        np.testing.assert_allclose(got[r]["losses"], np.array([l[r] for l in tr.losses]), rtol=1e-5)
        for k in range(len(ln_emb)):
            assert np.array_equal(got[r]["occ"][k], tr.occ[k].numpy()), (r, k)
        for i in range(len(got[r]["top_w"])):
            np.testing.assert_allclose(got[r]["top_w"][i], tr.top[r][0][i].numpy(), rtol=1e-4, atol=1e-6)
            assert np.array_equal(got[r]["top_w"][i], got[0]["top_w"][i]), "weight replicas agree bit for bit"
